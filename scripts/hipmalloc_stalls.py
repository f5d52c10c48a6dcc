#!/usr/bin/env python3
"""Diagnostic: how long do large hipMalloc / hipFree calls take when blocks of several GB are allocated,
used and freed over and over (the decode's scratch: an 8-GiB view, 7.5 GB of joined columns)?  Plain HIP
through ctypes, no library of this repo involved."""
import ctypes as C
import sys
import time

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize.argtypes = []


def alloc(nbytes, touch=True):
    p = C.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), nbytes)
    dt = time.perf_counter() - t
    assert rc == 0, rc
    if touch:
        hip.hipMemset(p, 1, nbytes)
        hip.hipDeviceSynchronize()
    return p, dt


def free(p):
    t = time.perf_counter()
    hip.hipFree(p)
    return time.perf_counter() - t


GB = 1 << 30
sizes = [int(8.6 * GB), 1 * GB, 1 * GB, int(7.5 * GB), 2 * GB, 2 * GB, 2 * GB]
for pattern in ("alloc all, free all", "alloc, free the two largest, alloc them again"):
    print(pattern, flush=True)
    for rep in range(10):
        ps, ta = [], []
        for s in sizes:
            p, dt = alloc(s)
            ps.append(p); ta.append(dt)
        if pattern.startswith("alloc,"):
            tf = [free(ps[0]), free(ps[3])]
            p0, d0 = alloc(sizes[0]); p3, d3 = alloc(sizes[3])
            ps[0], ps[3] = p0, p3
            ta += [d0, d3]
        tf = [free(p) for p in ps]
        print(f"  rep {rep}: malloc ms {[round(x * 1e3, 1) for x in ta]}  free ms {[round(x * 1e3, 1) for x in tf]}", flush=True)
