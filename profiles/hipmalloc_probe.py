#!/usr/bin/env python3
"""What a large device allocation costs: hipMalloc / first touch (hipMemset) / hipFree of one block, three times over.
python3 profiles/hipmalloc_probe.py [GiB]"""
import ctypes as C, json, sys, time
hip = C.CDLL("libamdhip64.so")
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = C.c_size_t(gib << 30)
out = []
hip.hipFree(None)
for rep in range(3):
    p = C.c_void_p()
    t = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); hip.hipDeviceSynchronize(); ta = time.perf_counter() - t
    t = time.perf_counter(); hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); tm = time.perf_counter() - t
    t = time.perf_counter(); hip.hipMemset(p, 1, n); hip.hipDeviceSynchronize(); tm2 = time.perf_counter() - t
    t = time.perf_counter(); hip.hipFree(p); hip.hipDeviceSynchronize(); tf = time.perf_counter() - t
    out.append({"GiB": gib, "rc": rc, "malloc_s": round(ta, 4), "first_memset_s": round(tm, 4), "second_memset_s": round(tm2, 4), "free_s": round(tf, 4)})
print(json.dumps(out))
