import ctypes, time
hip = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
n = 1840 * 1024 * 1024
a = ctypes.c_void_p(); b = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(a), ctypes.c_size_t(n)) == 0
assert hip.hipMalloc(ctypes.byref(b), ctypes.c_size_t(n)) == 0
hip.hipMemset(a, 1, ctypes.c_size_t(n)); hip.hipDeviceSynchronize()
for rep in range(3):
    t = time.time()
    for i in range(10):
        hip.hipMemcpyAsync(b, a, ctypes.c_size_t(n), 3, None)
    hip.hipDeviceSynchronize()
    dt = (time.time() - t) / 10
    print('D2D copy %.2f GB: %.3f ms  -> %.2f TB/s (read+write)' % (n / 1e9, dt * 1e3, 2 * n / dt / 1e12))
t = time.time()
for i in range(10):
    hip.hipMemsetAsync(b, 0, ctypes.c_size_t(n), None)
hip.hipDeviceSynchronize()
dt = (time.time() - t) / 10
print('memset %.3f ms -> %.2f TB/s (write)' % (dt * 1e3, n / dt / 1e12))
