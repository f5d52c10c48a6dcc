#!/usr/bin/env python3
"""Register / LDS / scratch use of the gfx950 kernels of one source file, one line per kernel
(hipcc -Rpass-analysis=kernel-resource-usage, device code only; nothing runs)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "kernels.hip"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=8", "-S",
       "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src]
err = subprocess.run(cmd, capture_output=True, text=True, cwd="bamsignals_amd/csrc").stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        full = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        full = full.replace("(anonymous namespace)::", "").replace("void ", "")
        cur = {"name": re.sub(r"\(.*$", "", full)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1).split(" [")[0]] = m.group(2)
for r in rows:
    if pat in r["name"]:
        print(f"{r['name'][:70]:70s} sgpr {r.get('TotalSGPRs'):>4} vgpr {r.get('VGPRs'):>4} spill s{r.get('SGPRs Spill')}/v{r.get('VGPRs Spill')} "
              f"scratch {r.get('ScratchSize'):>4} occ {r.get('Occupancy'):>2} lds {r.get('LDS Size')}")
