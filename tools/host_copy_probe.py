#!/usr/bin/env python3
"""Host <-> device copies of one 8192 x 8192 RGBA8 image (256 MiB) on the GPU box: pageable hipMemcpy, hipHostRegister + copy +
unregister, and copies through pinned staging buffers -- what kmg_reduce / kmg_find pay around the GPU work."""
import ctypes as C, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
n = 256 << 20
def chk(e):
    assert e == 0, e
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), C.c_size_t(n)))
h = np.random.default_rng(1).integers(0, 255, n, dtype=np.uint8)
out = np.empty_like(h)
H2D, D2H = 1, 2
def t(f, reps=3):
    f(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); hip.hipDeviceSynchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"pageable H2D {t(lambda: chk(hip.hipMemcpy(d, h.ctypes.data_as(C.c_void_p), C.c_size_t(n), H2D))):7.2f} ms")
print(f"pageable D2H {t(lambda: chk(hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d, C.c_size_t(n), D2H))):7.2f} ms")
def reg_copy():
    chk(hip.hipHostRegister(h.ctypes.data_as(C.c_void_p), C.c_size_t(n), 0))
    chk(hip.hipMemcpy(d, h.ctypes.data_as(C.c_void_p), C.c_size_t(n), H2D))
    chk(hip.hipHostUnregister(h.ctypes.data_as(C.c_void_p)))
print(f"register + H2D + unregister {t(reg_copy):7.2f} ms")
chk(hip.hipHostRegister(h.ctypes.data_as(C.c_void_p), C.c_size_t(n), 0))
chk(hip.hipHostRegister(out.ctypes.data_as(C.c_void_p), C.c_size_t(n), 0))
print(f"pinned H2D {t(lambda: chk(hip.hipMemcpy(d, h.ctypes.data_as(C.c_void_p), C.c_size_t(n), H2D))):7.2f} ms")
print(f"pinned D2H {t(lambda: chk(hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d, C.c_size_t(n), D2H))):7.2f} ms")
t0 = time.perf_counter(); tmp = h.copy(); print(f"numpy memcpy 256 MiB (one thread) {(time.perf_counter() - t0) * 1e3:7.2f} ms")
