#!/usr/bin/env python3
"""Diagnostic, plain HIP through ctypes (no library of this repo involved): does a process's FIRST large hipMalloc
pay for memory that OTHER processes used and freed before it?  Process A allocates, touches and frees `--gb` GB (and
exits); processes B1, B2, B3 then each time four hipMalloc calls of 8 GB (touched) in a fresh process.  The cold
file-level call of a fresh session makes about 36 such calls for 30 GB; on some leases the first session after a
large tenant loses 0.2-0.7 s inside them (bench.py: end_to_end.cold_call_sessions_stages_s, alloc_total)."""
import ctypes as C
import subprocess
import sys
import time


def hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    h.hipFree.argtypes = [C.c_void_p]
    h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    return h


def child(kind, gb):
    h = hip()
    GB = 1 << 30
    t0 = time.perf_counter()
    p = C.c_void_p()
    assert h.hipMalloc(C.byref(p), 256) == 0
    t_first = time.perf_counter() - t0
    h.hipFree(p)
    if kind == "A":
        ps = []
        for _ in range(int(gb) // 8):
            p = C.c_void_p()
            assert h.hipMalloc(C.byref(p), 8 * GB) == 0
            h.hipMemset(p, 1, 8 * GB)
            ps.append(p)
        h.hipDeviceSynchronize()
        for p in ps:
            h.hipFree(p)
        print(f"A: allocated, touched and freed {int(gb) // 8 * 8} GB", flush=True)
        return
    ts, ps = [], []
    for _ in range(4):
        p = C.c_void_p()
        t = time.perf_counter()
        assert h.hipMalloc(C.byref(p), 8 * GB) == 0
        ts.append(time.perf_counter() - t)
        h.hipMemset(p, 1, 8 * GB)
        ps.append(p)
    h.hipDeviceSynchronize()
    tf = []
    for p in ps:
        t = time.perf_counter()
        h.hipFree(p)
        tf.append(time.perf_counter() - t)
    print(f"{kind}: first hipMalloc of the process {t_first * 1e3:.1f} ms; 4 x hipMalloc(8 GB) ms {[round(x * 1e3, 1) for x in ts]}; "
          f"hipFree ms {[round(x * 1e3, 1) for x in tf]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], float(sys.argv[3]))
    else:
        gb = sys.argv[1] if len(sys.argv) > 1 else "120"
        for rnd in range(2):
            for kind in ("A", "B1", "B2", "B3"):
                subprocess.run([sys.executable, __file__, "--child", kind, gb], check=True)
