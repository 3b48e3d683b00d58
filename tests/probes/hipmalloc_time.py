import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def t_malloc(nbytes, reps=3):
    out = []
    for _ in range(reps):
        p = C.c_void_p()
        t = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)); dt = time.perf_counter() - t
        t = time.perf_counter(); hip.hipFree(p); df = time.perf_counter() - t
        out.append((dt * 1e3, df * 1e3))
    return out
hip.hipSetDevice(0)
p = C.c_void_p(); hip.hipMalloc(C.byref(p), C.c_size_t(1 << 20)); hip.hipFree(p)
for mb in (1, 16, 86, 256, 1024, 1280, 5000):
    print(mb, "MB:", ["malloc %.2f ms free %.2f ms" % x for x in t_malloc(mb << 20)])
# many small vs one big
t = time.perf_counter(); ps = []
for k in range(20):
    p = C.c_void_p(); hip.hipMalloc(C.byref(p), C.c_size_t(86 << 20)); ps.append(p)
print("20 x 86 MB: %.2f ms" % ((time.perf_counter() - t) * 1e3))
for p in ps: hip.hipFree(p)
t = time.perf_counter(); p = C.c_void_p(); hip.hipMalloc(C.byref(p), C.c_size_t(20 * 86 << 20)); print("1 x 1720 MB: %.2f ms" % ((time.perf_counter() - t) * 1e3)); hip.hipFree(p)
