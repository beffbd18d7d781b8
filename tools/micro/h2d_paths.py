"""H2D / D2H paths for the host-pointer API (development tool, GPU): pageable hipMemcpy, registered, pinned staging."""
import ctypes as C, time, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from climt_amd import _hip
h = _hip.lib()
n = 64 * 1024 * 1024 // 8      # 64 MB
a = np.random.rand(n)
d = _hip.DeviceArray((n,))
def t(f, reps=5):
    f(); _hip.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    _hip.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: h.hipMemcpy(C.c_void_p(d.ptr), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1))
print("pageable hipMemcpy H2D   %.1f GB/s" % (a.nbytes / dt / 1e9))
b = np.empty(n)
dt = t(lambda: h.hipMemcpy(C.c_void_p(b.ctypes.data), C.c_void_p(d.ptr), C.c_size_t(a.nbytes), 2))
print("pageable hipMemcpy D2H   %.1f GB/s" % (a.nbytes / dt / 1e9))
p = C.c_void_p(); h.hipHostMalloc(C.byref(p), C.c_size_t(a.nbytes), 0)
pin = np.ctypeslib.as_array((C.c_double * n).from_address(p.value))
dt = t(lambda: np.copyto(pin, a))
print("numpy copy into pinned   %.1f GB/s" % (a.nbytes / dt / 1e9))
dt = t(lambda: h.hipMemcpy(C.c_void_p(d.ptr), p, C.c_size_t(a.nbytes), 1))
print("pinned hipMemcpy H2D     %.1f GB/s" % (a.nbytes / dt / 1e9))
dt = t(lambda: h.hipMemcpy(p, C.c_void_p(d.ptr), C.c_size_t(a.nbytes), 2))
print("pinned hipMemcpy D2H     %.1f GB/s" % (a.nbytes / dt / 1e9))
t0 = time.perf_counter(); rc = h.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 0); t1 = time.perf_counter()
print("hipHostRegister 64 MB    %.2f ms (rc %d)" % ((t1 - t0) * 1e3, rc))
dt = t(lambda: h.hipMemcpy(C.c_void_p(d.ptr), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1))
print("registered hipMemcpy H2D %.1f GB/s" % (a.nbytes / dt / 1e9))
t0 = time.perf_counter(); h.hipHostUnregister(C.c_void_p(a.ctypes.data)); print("unregister %.2f ms" % ((time.perf_counter() - t0) * 1e3))
print("cpus", os.cpu_count())
