#!/usr/bin/env python3
"""Diagnostic: bamCount on config 3's tiling under the tiles-per-wave / prefetch-depth variants of
k_count_multi (env BAMSIGNALS_COUNT_TILES x BAMSIGNALS_COUNT_PRE; 1 = the one-tile kernel).  One child
process per setting (the library reads the knobs once)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
settings = [(1, 4)] + [(t, p) for t in (2, 4, 8) for p in (2, 3, 4)]
for t, p in settings:
    env = dict(os.environ, BAMSIGNALS_COUNT_TILES=str(t), BAMSIGNALS_COUNT_PRE=str(p))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "profile_case.py"), "count", "60"], env=env,
                         capture_output=True, text=True)
    if out.returncode:
        print(json.dumps(dict(tiles=t, pre=p, error=out.stderr[-400:])), flush=True)
        continue
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(json.dumps(dict(tiles=t, pre=p, kernel_ms=r["kernel_ms"], frac=r["frac_of_8TBps"])), flush=True)
