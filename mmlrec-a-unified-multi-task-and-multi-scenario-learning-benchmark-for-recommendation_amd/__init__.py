"""MI355X-native hot path of the MMLRec benchmark (alipay/MMLRec): multi-field embedding gather, expert/gate/tower
MLPs, heads + BCE, sparse row-scatter backward and optimizers as hand-written HIP kernels for gfx950 behind the
C ABI in include/mmlrec.h, with drop-in nn.Modules (model/*.py) above it.

Import name: `mmlrec_amd` (see ../mmlrec_amd.py; the directory name itself is not a Python identifier).
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (ctypes signatures; the library itself is loaded on first use)
