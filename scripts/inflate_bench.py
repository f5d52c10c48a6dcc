#!/usr/bin/env python3
"""Diagnostic: k_inflate ALONE (bsig_debug_inflate_bench) on one round of blocks of a bare-record BAM (the north
star's shape) and of a real-shaped one, for every build listed in BSIG_VARIANTS (space-separated paths of
libbamsignals_hip*.so; default: the shipped build).  One child process per build.  Ablated builds
(-DBSIG_ABLATE=n) write garbage by design: their CRC status is reported, not required."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib_path, files, n_blocks, reps):
    lib = ctypes.CDLL(lib_path)
    fn = lib.bsig_debug_inflate_bench
    fn.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    for tag, path in files:
        for env in ({}, {"BAMSIGNALS_INFLATE_LANES": "64"}) if os.environ.get("BSIG_BENCH_LANES64") else ({},):
            os.environ.update(env)
            ms = (ctypes.c_double * 2)()
            by = (ctypes.c_int64 * 3)()
            st = ctypes.c_int(0)
            rc = fn(0, path.encode(), 1, n_blocks, reps, ms, by, ctypes.byref(st))
            print(json.dumps(dict(lib=os.path.basename(lib_path), file=tag, env=env, rc=rc, crc_status=st.value, blocks=by[2],
                                  ms_best=round(ms[0], 3), ms_mean=round(ms[1], 3), out_GBps=round(by[1] / ms[0] / 1e6, 1) if ms[0] else None,
                                  in_MB=round(by[0] / 1e6, 1), out_MB=round(by[1] / 1e6, 1))), flush=True)
            for k in env:
                os.environ.pop(k)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], json.loads(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    from bamsignals_amd.bamio import write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    tmp = os.environ.get("TMPDIR", "/tmp")
    files = []
    kinds = (os.environ.get("BSIG_BENCH_FILES") or "bare real real6").split()
    for tag, n, l_seq, level in (("bare", 100_000_000, 0, 1), ("real", 20_000_000, 100, 1), ("real6", 20_000_000, 100, 6)):
        if tag not in kinds:
            continue
        path = os.path.join(tmp, f"ib_{tag}.bam")
        if not os.path.exists(path):
            cols = synth_reads(n, [250_000_000] * (10 if tag == "bare" else 1), seed=9 if tag == "bare" else 12)
            write_columns_as_bam(path, ["c%d" % i for i in range(10 if tag == "bare" else 1)], cols, level=level, l_seq=l_seq, seed=3)
            del cols
        files.append((tag, path))
    n_blocks = int(os.environ.get("BSIG_BENCH_BLOCKS", "57344"))
    reps = int(os.environ.get("BSIG_BENCH_REPS", "5"))
    variants = (os.environ.get("BSIG_VARIANTS") or os.path.join(ROOT, "bamsignals_amd", "libbamsignals_hip.so")).split()
    for v in variants:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, json.dumps(files), str(n_blocks), str(reps)],
                           capture_output=True, text=True, timeout=600)
        sys.stdout.write(r.stdout)
        if r.returncode:
            sys.stdout.write(f"variant {v} failed: {r.stderr[-500:]}\n")
    if os.environ.get("BSIG_BENCH_KEEP"):            # (the files stay for a profiler run of the child: see pmc_inflate_bench.sh)
        print("FILES " + json.dumps(files))
        return
    for _, path in files:
        for p in (path, path + ".bai"):
            if os.path.exists(p):
                os.remove(p)


if __name__ == "__main__":
    main()
