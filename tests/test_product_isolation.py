"""CPU: the product (bamsignals_amd/, include/) never touches the oracle or the reference, and has
no CPU fallback for the compute path."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _product_files():
    for base in ("bamsignals_amd", "include"):
        for d, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c", ".R")) or f in ("Makefile", "Makevars"):
                    yield os.path.join(d, f)


def test_product_does_not_use_the_oracle_or_the_reference_tree():
    bad = []
    for path in _product_files():
        txt = open(path, errors="replace").read()
        for pat in (r"^\s*(from|import)\s+oracle\b", r"oracle_c\b", r"oracle_np\b", r"liboracle", r"bsor_\w+", r"/root/reference"):
            if re.search(pat, txt, flags=re.M):
                bad.append((os.path.relpath(path, ROOT), pat))
    assert not bad, bad


def test_only_tests_smoke_and_bench_baseline_import_the_oracle():
    users = []
    for d, _, fs in os.walk(ROOT):
        if any(part in d for part in (".git", "gpurun_out", "__pycache__")):
            continue
        for f in fs:
            if f.endswith(".py"):
                p = os.path.join(d, f)
                if re.search(r"^\s*from oracle import|^\s*import oracle", open(p, errors="replace").read(), flags=re.M):
                    users.append(os.path.relpath(p, ROOT))
    allowed = ("tests/", "oracle/", "__graft_entry__.py", "bench.py", "scripts/")
    assert all(u.startswith(allowed) for u in users), users
    # in bench.py the oracle is only reached inside the parity check and the cpu_baseline leg
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle import") == 1 and "if rank == 0:" in bench.split("from oracle import")[0][-400:]
