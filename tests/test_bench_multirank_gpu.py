"""GPU: bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), at a
small size and over gloo with two ranks sharing the box's one GPU.  The headline must be the workload at FIXED
total size with the whole north-star step in the timed region -- kernels on the rank's shard, the gather of the
shards to rank 0, the reassembly in rank 0's HBM (ref: src/bamsignals.cpp:164,181,186 is why the shard is legal;
the reassembled result is the product) -- checked against the oracle on the ASSEMBLED result; beside it the
informational blocks: the replicas without a collective, config 5's shape at fixed total size, and the in-process
route (one process, two GPU slots, every gather route)."""
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
@pytest.mark.parametrize("wire", ["narrow", "int32"])
def test_two_ranks_emit_strong_and_in_process_blocks(wire):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BAMSIGNALS_DEVICES", "BAMSIGNALS_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "C5",
           "--reads", "3000000", "--ranges", "3000", "--strong-ranges", "6000", "--strong-steps", "3", "--steps", "6", "--warmup", "2", "--wire", wire]
    if wire == "int32":
        cmd += ["--no-in-process"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=560, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    # the headline: fixed total size, and `value` = the whole job's bases x steps / the region that holds the gather
    cfg = res["config"]
    assert cfg["workload"].startswith("C5:") and cfg["ranges_total"] == 3000 and cfg["ranges_per_gpu"] == 1500
    assert "gather to rank 0" in cfg["parallelism"] and "inside the timed region" in cfg["parallelism"]
    bases = cfg["ranges_total"] * cfg["range_width"]
    assert abs(res["value"] - bases * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"]) / 1e6) <= 1e-6 * res["value"]
    par = res["parity_checked"]
    assert par["ranges_per_batch"] == 3000 and "assembled on rank 0" in par["how"] and par["cells"] == par["batches"] * bases
    ph = res["step_phases"]
    cells = 1500 * cfg["range_width"]
    wr = res["wire"]
    assert wr["kind"] == wire and wr["int32_bytes"] == 4 * cells
    if wire == "narrow":
        # two bits a cell + 16 bytes in front + 8 bytes for every exception the longest list has room for
        assert wr["message_bytes"] == 16 + 4 * ((cells + 15) // 16) + 8 * wr["exceptions_room"] < cells
        assert ph["gather_bytes_into_rank0"] == wr["message_bytes"] and ph["rank0_ms"]["pack"] >= 0      # (the root packs nothing: its own shard crosses no link)
    else:
        assert ph["gather_bytes_into_rank0"] == 4 * cells
    assert len(ph["kernel_ms_by_rank"]) == 2
    # a step cannot be shorter than rank 0's kernel + what it waits for the shards + their placement
    assert res["ms_per_step"] > 0.5 * sum(ph["rank0_ms"].values())
    # the replicas (no collective) are informational and are NOT the headline
    rep = res["no_collective"]
    assert rep["value"] > 0 and "replicas" in rep["what"] and rep["value"] != res["value"]
    st = res["strong"]
    assert st["n_gpus"] == 2 and st["ranges_total"] == 6000 and st["ranges_per_gpu"] == 3000
    assert "identical to the 1-GPU result" in st["checked"] and "identical to the oracle" in st["checked"]
    assert st["ms_per_step"] > 0 and st["one_gpu_ms"] > 0 and 0 < st["efficiency_vs_1gpu"]
    if wire == "int32":
        return
    ip = res["in_process"]
    assert ip["devices"] == "0,0"
    assert [r["kind"] for r in ip["xgmi"]] == ["cold", "warm", "warm"]
    for g in ("xgmi", "direct", "pcie"):
        assert all(r["call_s"] > 0 for r in ip[g])
        assert all(r["route"].startswith("2 GPU slot(s)") for r in ip[g])
    assert "sharded decode" in ip["xgmi"][0]["route"] and "reads: resident" in ip["xgmi"][1]["route"]
    assert "result: pcie" in ip["pcie"][-1]["route"]
