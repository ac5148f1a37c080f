"""Hyper-parameters and paths, same names as the reference's main_code/utils/config.py.

The constants of the heads with a native epilogue are kept (QAFace is out of scope).  Paths are
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
M_mv, WEIGHT_mv, S_mv, MARGIN_TYPE_mv = 0.35, 1.12, 32.0, 'am'       # config.py:29-32 ('arc' for MV-Arc; 'am' for MV-Cos)
M_curricular, S_curricular, MOMENTUM_curricular = 0.5, 64.0, 0.01   # config.py:35-37
S_vpl, M_vpl, EASY_MARGIN_vpl, LAMDA_vpl, DELTA_vpl = 64.0, 0.50, False, 0.15, 100   # config.py:40-44
S_ada, M_ada, H_ada, T_ALPHA_ada = 64.0, 0.4, 0.333, 0.99            # config.py:47-50
S_elastic_arc, M_elastic_arc, STD_elastic_arc, PLUS_elastic_arc = 64.0, 0.50, 0.0125, False   # config.py:53-56
S_elastic_cos, M_elastic_cos, STD_elastic_cos, PLUS_elastic_cos = 64.0, 0.35, 0.0125, False   # config.py:59-62
S_mag, EASY_MARGIN_mag, L_MARGIN_mag, U_MARGIN_mag, L_A_mag, U_A_mag = 64.0, False, 0.45, 0.8, 10.0, 110.0   # config.py:65-70

# additive (not in the reference): arithmetic mode of the native engine.
#   'bf16' = speed mode (bf16 activations / MFMA, fp32 accumulate, fp32 master weights)
#   'f32'  = parity mode (fp32 everywhere; matches the reference CPU path to 1e-3)
COMPUTE_DTYPE = os.environ.get("FR_COMPUTE_DTYPE", "bf16")
