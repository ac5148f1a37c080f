"""Hyper-parameters and paths, same names as the reference's main_code/utils/config.py.

Only the constants the hot path reads are kept (the other seven heads are out of scope).  Paths are
taken from the environment instead of the reference author's absolute home directories
(config.py:1-9 upstream)."""
import os

DATASET_PATH = os.environ.get("FR_DATASET_PATH", os.path.join(os.getcwd(), "dataset"))
WORKING_PATH = os.environ.get("FR_WORKING_PATH", os.path.join(os.getcwd(), "working"))

# upstream ships 'resnet18' (config.py:11) although its README and BASELINE.json name ResNet-50;
# only 'resnet50' has a native engine here
BACKBONE = os.environ.get("FR_BACKBONE", "resnet50")

FEATURE_DIM = 512      # config.py:13
LAMBDA_G = 0.0         # config.py:14

M_sphere, S_sphere = 2, 20.0                 # config.py:17-18 (S_sphere is unused upstream too)
M_cos, S_cos = 0.35, 64.0                    # config.py:21-22
M_arc, S_arc = 0.5, 64.0                     # config.py:25-26
M_curricular, S_curricular, MOMENTUM_curricular = 0.5, 64.0, 0.01   # config.py:35-37

# additive (not in the reference): arithmetic mode of the native engine.
#   'bf16' = speed mode (bf16 activations / MFMA, fp32 accumulate, fp32 master weights)
#   'f32'  = parity mode (fp32 everywhere; matches the reference CPU path to 1e-3)
COMPUTE_DTYPE = os.environ.get("FR_COMPUTE_DTYPE", "bf16")
