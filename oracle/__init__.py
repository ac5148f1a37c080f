"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product (`face-recognition-models_amd/`) may import this package.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it,
and there only as the checker / the reported CPU baseline.

Pinning status (see DESIGN.md "Oracle"):
  * heads, CE, accuracy, CustomStepLR, train-loop ordering, threshold / accuracy /
    10-fold arithmetic: PINNED to the reference's own Python, imported in the build
    container by tests/golden/make_golden.py; vectors committed under tests/golden/.
  * ResNet-50 backbone: the arithmetic lives in torchvision (not vendored, version
    unpinned, absent from the image).  Restated from torch.nn primitives following
    torchvision's public ResNet v1.5 topology -- "parity unpinned" at that boundary.
"""
