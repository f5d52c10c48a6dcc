"""GPU: BASELINE config 5 -- 1,000,000 x 1 kb ranges over a 24-reference, 3.1-Gbp genome, bamProfile(binsize=1) --
through the single-process multi-GPU route (the box's one GPU listed several times), EVERY one of the 1e9 result
cells against the C oracle under each result route (gather on the first GPU, per-GPU PCIe with host reassembly,
per-GPU contiguous slices, and the library's own choice).

  * default `-m gpu` run: 1e8 reads (a 600-MB BAM), eight slots -- the full range set and the full route, a tenth of
    the reads (about 40 s);
  * opt-in (BAMSIGNALS_FULLSIZE=1; about 6 minutes and 120 GB of host memory): the full 1e9 reads as a 6-GB BAM, four
    slots (four resident copies of 12.5 GB).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(*args, timeout):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "c5_in_process.py"), *args], capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    # xgmi, pcie, blocks and the library's own choice: four comparisons of all 1e9 cells
    assert out.stdout.count("identical to the oracle (1000000000 cells") == 4, out.stdout[-3000:]
    assert "result: pcie/blocks" in out.stdout and "result: xgmi/" in out.stdout and "result: pcie (" in out.stdout
    return out.stdout


@pytest.mark.timeout(600)
def test_c5_all_ranges_eight_slots_a_tenth_of_the_reads():
    out = _run("100000000", "8", "1000000", timeout=550)
    assert "8 GPU slot(s); reads: sharded decode" in out


@pytest.mark.fullsize
@pytest.mark.skipif(os.environ.get("BAMSIGNALS_FULLSIZE") != "1", reason="opt-in: set BAMSIGNALS_FULLSIZE=1")
@pytest.mark.timeout(1500)
def test_c5_full_size_in_process():
    _run(timeout=1400)
