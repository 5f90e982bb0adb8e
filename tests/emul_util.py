"""TEST INFRASTRUCTURE: build + load the host-emulated copy of the kernel library and call the C
ABI on numpy arrays (host pointers).  See tests/emul/hip/hip_runtime.h."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_LIB = os.path.join(EMUL_DIR, "_build", "libdeepspeaker_emul.so")

_lib = None


def _stale():
    if not os.path.exists(EMUL_LIB):
        return True
    t = os.path.getmtime(EMUL_LIB)
    srcs = []
    for d in (os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc"), EMUL_DIR, os.path.join(EMUL_DIR, "hip"),
              os.path.join(ROOT, "include")):
        srcs += [os.path.join(d, f) for f in os.listdir(d) if os.path.isfile(os.path.join(d, f))]
    return any(os.path.getmtime(s) > t for s in srcs)


def emul_lib():
    global _lib
    if _lib is None:
        if _stale():
            # one builder at a time (pytest-xdist workers all arrive here with a stale library)
            import fcntl
            os.makedirs(os.path.join(EMUL_DIR, "_build"), exist_ok=True)
            with open(os.path.join(EMUL_DIR, "_build", ".lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if _stale():
                    subprocess.run([os.path.join(EMUL_DIR, "build_emul.sh")], check=True, capture_output=True)
        from deepspeaker_pytorch_amd._native import NativeLib
        _lib = NativeLib(EMUL_LIB, host_memory=True)
    return _lib


def ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"] and (a.ctypes.data % 16) == 0
    return a.ctypes.data


def aligned(shape, dtype=np.float32, fill=None):
    """64-byte aligned array (the kernels do 16-byte vector accesses)."""
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    itemsize = np.dtype(dtype).itemsize
    raw = np.empty(n * itemsize + 64, np.uint8)
    off = (-raw.ctypes.data) % 64
    a = raw[off:off + n * itemsize].view(dtype).reshape(shape)
    if fill is not None:
        a[...] = fill
    return a


def to_aligned(x, dtype=np.float32):
    a = aligned(x.shape, dtype)
    a[...] = x
    return a


def nhwc(x):
    return to_aligned(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))


def nchw(y):
    return np.ascontiguousarray(y.transpose(0, 3, 1, 2))
