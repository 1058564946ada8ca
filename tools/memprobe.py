import ctypes
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
def alloc(gb):
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb * 1e9)))
    return rc, p
def free(p): hip.hipFree(p)
fr = ctypes.c_size_t(); tot = ctypes.c_size_t()
hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)); print("free %.1f total %.1f GB" % (fr.value/1e9, tot.value/1e9))
for gb in (120, 140, 160, 174, 200, 240, 280):
    rc, p = alloc(gb); print("single", gb, "rc", rc)
    if rc == 0: free(p)
rc1, p1 = alloc(56); rc2, p2 = alloc(30); rc3, p3 = alloc(0.5)
free(p1)
hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)); print("after 56+30+0.5, free 56: free %.1f" % (fr.value/1e9))
for gb in (140, 174, 200, 230):
    rc, p = alloc(gb); print("with 30 GB held:", gb, "rc", rc)
    if rc == 0: free(p)
