"""GPU, opt-in (BAMSIGNALS_FULLSIZE=1; about 6 minutes and 120 GB of host memory): BASELINE config 5 at FULL
size through the single-process multi-GPU route -- 1e9 reads on 24 references as a 6-GB BAM, 1,000,000 x 1 kb
ranges, four GPU slots on the box's one GPU -- every one of the 1e9 result cells against the C oracle, under
the in-HBM gather and the per-GPU PCIe gather.  The default `-m gpu` run covers the same route at reduced
size (tests/test_large_genome_gpu.py, tests/test_multi_gpu_route_gpu.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.fullsize,
              pytest.mark.skipif(os.environ.get("BAMSIGNALS_FULLSIZE") != "1", reason="opt-in: set BAMSIGNALS_FULLSIZE=1")]


@pytest.mark.timeout(1500)
def test_c5_full_size_in_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "c5_in_process.py")], capture_output=True, text=True,
                         timeout=1400)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("identical to the oracle (1000000000 cells") == 2, out.stdout[-3000:]
