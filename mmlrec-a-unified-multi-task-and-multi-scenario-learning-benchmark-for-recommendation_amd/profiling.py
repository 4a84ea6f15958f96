"""roctx ranges for `rocprofv3 --marker-trace` (the reference has wall-clock per epoch only,
model/basemodel.py:255, :351-365).  Off by default; `--profile` on main.py / bench.py (or MMLREC_PROFILE=1) turns it
on: every eagerly issued C-ABI call is wrapped in a range named after its kernel, every replayed HIP-graph segment
and every epoch / step in a range of its own.  Host-side markers only: nothing changes on the device."""
import contextlib
import ctypes
import os

enabled = False
_lib = None


def _load():
    global _lib
    if _lib is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            for d in ("", "/opt/rocm/lib/"):
                try:
                    _lib = ctypes.CDLL(d + name)
                    _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    _lib.roctxRangePushA.restype = ctypes.c_int
                    _lib.roctxRangePop.restype = ctypes.c_int
                    return _lib
                except (OSError, AttributeError):
                    _lib = None
        _lib = False
    return _lib


def enable(on=True):
    """Returns True when ranges will actually be emitted (the roctx library was found)."""
    global enabled
    enabled = bool(on) and bool(_load())
    return enabled


def push(name):
    if enabled:
        _lib.roctxRangePushA(str(name).encode())


def pop():
    if enabled:
        _lib.roctxRangePop()


@contextlib.contextmanager
def range(name):  # noqa: A001 (the roctx vocabulary)
    push(name)
    try:
        yield
    finally:
        pop()


if os.environ.get("MMLREC_PROFILE", "") not in ("", "0"):
    enable()
