"""Minimal ctypes binding of the HIP runtime (libamdhip64) for device buffers, copies and events.

The product path does not use PyTorch: device memory that outlives a call (benchmark-resident
inputs, multi-GPU shards) is managed through these few runtime entry points.
"""
import ctypes as C

import numpy as np

_hip = None


class HipError(RuntimeError):
    pass


def lib():
    global _hip
    if _hip is None:
        try:
            _hip = C.CDLL("libamdhip64.so")
        except OSError as e:  # pragma: no cover
            raise HipError("libamdhip64.so not loadable: %s" % e)
        _hip.hipGetErrorString.restype = C.c_char_p
    return _hip


def _ck(err, what):
    if err != 0:
        raise HipError("%s failed: %s" % (what, lib().hipGetErrorString(err).decode()))


def device_count():
    n = C.c_int(0)
    try:
        err = lib().hipGetDeviceCount(C.byref(n))
    except HipError:
        return 0
    return n.value if err == 0 else 0


def set_device(i):
    _ck(lib().hipSetDevice(C.c_int(i)), "hipSetDevice")


def synchronize():
    _ck(lib().hipDeviceSynchronize(), "hipDeviceSynchronize")


def pinned_buffer(nbytes):
    """nbytes of page-locked host memory (hipHostMalloc) as a ctypes byte array: copies to and from the device run at the link
    rate and asynchronously, and its pages exist (no first-touch faults).  Freed when the object and every array made from it
    (np.frombuffer) are gone."""
    import weakref
    nbytes = max(int(nbytes), 8)
    p = C.c_void_p()
    _ck(lib().hipHostMalloc(C.byref(p), C.c_size_t(nbytes), C.c_uint(0)), "hipHostMalloc")
    buf = (C.c_char * nbytes).from_address(p.value)
    weakref.finalize(buf, lib().hipHostFree, C.c_void_p(p.value))
    return buf


def pinned_empty(shape, dtype=np.float64):
    """A page-locked host array (pinned_buffer) of the given shape."""
    shape = tuple(int(s) for s in np.atleast_1d(shape))
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    return np.frombuffer(pinned_buffer(count * dt.itemsize), dtype=dt, count=count).reshape(shape)


class DeviceArray:
    """A device allocation holding a C-contiguous float64/int array of given shape."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _ck(lib().hipMalloc(C.byref(p), C.c_size_t(max(self.nbytes, 8))), "hipMalloc")
        self.ptr = p.value

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        d = cls(arr.shape, arr.dtype)
        d.upload(arr)
        return d

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        _ck(lib().hipMemcpy(C.c_void_p(self.ptr), C.c_void_p(arr.ctypes.data), C.c_size_t(self.nbytes), C.c_int(1)), "hipMemcpy H2D")

    def zero(self):
        _ck(lib().hipMemset(C.c_void_p(self.ptr), C.c_int(0), C.c_size_t(self.nbytes)), "hipMemset")

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _ck(lib().hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr), C.c_size_t(self.nbytes), C.c_int(2)), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            lib().hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """A non-blocking HIP stream (does not synchronise with the null stream)."""

    def __init__(self):
        s = C.c_void_p()
        _ck(lib().hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)), "hipStreamCreateWithFlags")   # hipStreamNonBlocking
        self.s = s.value

    def synchronize(self):
        _ck(lib().hipStreamSynchronize(C.c_void_p(self.s)), "hipStreamSynchronize")

    def __del__(self):
        try:
            if self.s:
                lib().hipStreamDestroy(C.c_void_p(self.s))
                self.s = None
        except Exception:
            pass


class Event:
    def __init__(self):
        e = C.c_void_p()
        _ck(lib().hipEventCreate(C.byref(e)), "hipEventCreate")
        self.e = e

    def record(self, stream=None):
        _ck(lib().hipEventRecord(self.e, C.c_void_p(stream)), "hipEventRecord")

    def synchronize(self):
        _ck(lib().hipEventSynchronize(self.e), "hipEventSynchronize")

    def elapsed_ms(self, end):
        ms = C.c_float(0)
        _ck(lib().hipEventElapsedTime(C.byref(ms), self.e, end.e), "hipEventElapsedTime")
        return ms.value
