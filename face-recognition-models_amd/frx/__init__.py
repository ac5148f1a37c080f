"""frx -- Python binding of libfrx.so (hand-written gfx950 HIP kernels behind a C ABI).

There is NO CPU fallback: importing works anywhere (so host logic can be tested),
but every compute entry point raises unless the native library is loaded and the
tensors live on a HIP device.
"""
from ._lib import lib, load_library, FrxError, library_path  # noqa: F401
