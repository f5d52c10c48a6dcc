#!/usr/bin/env python3
"""Diagnostic: tests/test_gpu_parity.py::test_fuzz_small_inputs for many seeds, in the three launch forms a small input
can take -- the fused kernels (default), the forms for resolved windows that launches of 32,768 tiles and more run
(bsig_debug_set_resolve_min(1): k_resolve_tiles in front, k_profile_multi for narrow tiles), and with the packed class
switched off (BAMSIGNALS_PACK=0: everything through class 0's read-by-read form).  Every case is HIP against the C oracle,
bit for bit; prints one line per (seed, form).   usage: fuzz_campaign.py [first seed] [seeds] [cases per seed]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cases = sys.argv[3] if len(sys.argv) > 3 else "400"
    import test_gpu_parity as T
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context
    fn = _lib.load().bsig_debug_set_resolve_min
    fn.argtypes = [ctypes.c_longlong]
    ctx = Context(0)
    os.environ["BSIG_FUZZ_CASES"] = cases
    for seed in range(first, first + seeds):
        os.environ["BSIG_FUZZ_SEED"] = str(seed)
        for form in ("fused", "resolved", "class 0 alone"):
            fn(1 if form == "resolved" else -1)
            if form == "class 0 alone":
                os.environ["BAMSIGNALS_PACK"] = "0"
            t0 = time.time()
            try:
                T.test_fuzz_small_inputs(ctx)
                print(f"seed {seed}, {form}: {cases} cases identical to the oracle ({time.time() - t0:.1f} s)", flush=True)
            finally:
                os.environ.pop("BAMSIGNALS_PACK", None)
                fn(-1)
    ctx.close()


if __name__ == "__main__":
    main()
