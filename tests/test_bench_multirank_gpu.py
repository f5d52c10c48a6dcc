"""GPU: bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), at a
small size and over gloo with two ranks sharing the box's one GPU: the line must carry the strong-scaling block
(config 5's shape at fixed total size: kernel + gather + reassembly timed, checked against the 1-GPU result
and the oracle) and the in-process block (one process, two GPU slots, every gather route)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_ranks_emit_strong_and_in_process_blocks():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BAMSIGNALS_DEVICES", "BAMSIGNALS_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "C5",
           "--reads", "3000000", "--ranges", "3000", "--strong-ranges", "6000", "--strong-steps", "3", "--steps", "6", "--warmup", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=560, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    st = res["strong"]
    assert st["n_gpus"] == 2 and st["ranges_total"] == 6000 and st["ranges_per_gpu"] == 3000
    assert "identical to the 1-GPU result" in st["checked"] and "identical to the oracle" in st["checked"]
    assert st["ms_per_step"] > 0 and st["one_gpu_ms"] > 0 and 0 < st["efficiency_vs_1gpu"]
    ip = res["in_process"]
    assert ip["devices"] == "0,0"
    assert [r["kind"] for r in ip["xgmi"]] == ["cold", "warm", "warm"]
    for g in ("xgmi", "direct", "pcie"):
        assert all(r["call_s"] > 0 for r in ip[g])
        assert all(r["route"].startswith("2 GPU slot(s)") for r in ip[g])
    assert "sharded decode" in ip["xgmi"][0]["route"] and "reads: resident" in ip["xgmi"][1]["route"]
    assert "result: pcie" in ip["pcie"][-1]["route"]
