"""CPU: the C-ABI library loads and exports every symbol include/bamsignals_abi.h declares.
No compute call is made here (no GPU in the dev container)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(bsig_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_header_declares_something():
    assert len(_declared()) >= 15


def test_every_declared_symbol_is_exported():
    from bamsignals_amd import _lib
    lib = _lib.load()
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.bsig_abi_version() == 4


def test_layout_matches_oracle():
    from bamsignals_amd.device import layout
    from oracle import oracle_c
    rng = np.random.default_rng(3)
    ln = rng.integers(0, 5000, 200).astype(np.int32)
    for bs in (-1, 1, 2, 7, 50, 4999, 100000):
        for ss in (False, True):
            assert np.array_equal(layout(ln, bs, ss), oracle_c.layout(ln, bs, ss)), (bs, ss)


def test_magic_division_constants():
    """host_util.h magic_u31: n / d == umulhi(n, magic) >> shift for 0 <= n < 2^31."""
    rng = np.random.default_rng(5)
    ds = list(range(2, 70)) + [100, 127, 128, 129, 1000, 4096, 65535, 65536, 65537, 10**6, 2**30, 2**31 - 1]
    ns = np.concatenate([rng.integers(0, 2**31, 2000), [0, 1, 2**31 - 1, 2**31 - 2]]).astype(np.uint64)
    for d in ds:
        s = 0
        while (1 << s) < d:
            s += 1
        magic = ((1 << (31 + s)) + d - 1) // d
        assert magic < 2**32
        q = ((ns * np.uint64(magic)) >> np.uint64(32)) >> np.uint64(s - 1)
        assert np.array_equal(q, ns // np.uint64(d)), d


def test_compute_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context
    with pytest.raises(_lib.BsigError) as ei:
        Context(0)
    assert ei.value.code_name == "BSIG_ERR_DEVICE"
