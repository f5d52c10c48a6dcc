"""Test-only: drives bamsignals_amd/r_package/src/shim.c -- compiled together with the stand-in R
runtime tests/r_stub/r_mock.c and linked against the real libbamsignals_hip.so -- through ctypes,
so that the .Call routines an R session would reach are executed here, where R is absent."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NA_INTEGER = -2**31
TYPES = dict(NILSXP=0, LGLSXP=10, INTSXP=13, REALSXP=14, STRSXP=16, VECSXP=19, S4SXP=25)


class RError(RuntimeError):
    """Rf_error() in the shim: what the R user would see as the condition message."""


class RViolation(AssertionError):
    """The shim broke a rule of the R API (unprotected object across an allocation, protect-stack
    imbalance, wrong accessor)."""


def build(out_dir):
    so = os.path.join(str(out_dir), "libshim_mock.so")
    lib_dir = os.path.join(ROOT, "bamsignals_amd")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-cast-function-type",
                           "-Wno-clobbered", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tests", "r_stub"), "-I", os.path.join(ROOT, "include"),
                           "-o", so, os.path.join(lib_dir, "r_package", "src", "shim.c"), os.path.join(ROOT, "tests", "r_stub", "r_mock.c"),
                           "-L", lib_dir, "-lbamsignals_hip", "-Wl,-rpath," + lib_dir])
    return so


class MockR:
    def __init__(self, so_path):
        from bamsignals_amd import _lib
        _lib.load()                                   # libamdhip64 etc. resolved as the product does
        L = self.L = C.CDLL(so_path)
        P = C.c_void_p
        for name, res, args in (
            ("mock_int", P, [P, C.c_ssize_t]), ("mock_lgl", P, [P, C.c_ssize_t]), ("mock_real", P, [P, C.c_ssize_t]),
            ("mock_str", P, [C.POINTER(C.c_char_p), C.c_ssize_t]), ("mock_list", P, [C.c_ssize_t]),
            ("mock_list_set", None, [P, C.c_ssize_t, P]), ("mock_s4", P, []), ("mock_set_attr", None, [P, C.c_char_p, P]),
            ("mock_get_attr", P, [P, C.c_char_p]), ("mock_nil", P, []), ("mock_type", C.c_int, [P]), ("mock_len", C.c_longlong, [P]),
            ("mock_int_data", P, [P]), ("mock_list_get", P, [P, C.c_ssize_t]), ("mock_string", C.c_char_p, [P, C.c_ssize_t]),
            ("mock_error", C.c_char_p, []), ("mock_protect_depth", C.c_int, []), ("mock_poisoned", C.c_int, []),
            ("mock_call", P, [C.c_char_p, C.POINTER(P), C.c_int, C.POINTER(C.c_int)]), ("mock_init", None, []),
            ("mock_n_registered", C.c_int, []), ("mock_registered_name", C.c_char_p, [C.c_int]),
            ("mock_registered_arity", C.c_int, [C.c_int]), ("mock_dynamic_symbols", C.c_int, []),
            ("mock_last_call_bytes", None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        ):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.mock_init()
        self.nil = L.mock_nil()

    def last_call_bytes(self):
        """(payload bytes of the vectors the last .Call allocated, bytes it took from R_alloc)"""
        v, r = C.c_longlong(0), C.c_longlong(0)
        self.L.mock_last_call_bytes(C.byref(v), C.byref(r))
        return v.value, r.value

    # ---- building R objects -------------------------------------------------------------------
    def int(self, v):
        a = np.ascontiguousarray(v, dtype=np.int32)
        return self.L.mock_int(a.ctypes.data, len(a))

    def lgl(self, v):
        a = np.ascontiguousarray([NA_INTEGER if x is None else int(bool(x)) for x in np.atleast_1d(v)], dtype=np.int32)
        return self.L.mock_lgl(a.ctypes.data, len(a))

    def real(self, v):
        a = np.ascontiguousarray(v, dtype=np.float64)
        return self.L.mock_real(a.ctypes.data, len(a))

    def str(self, v):
        v = [v] if isinstance(v, str) else list(v)
        arr = (C.c_char_p * max(len(v), 1))(*[s.encode() for s in v])
        return self.L.mock_str(arr, len(v))

    def list(self, items):
        l = self.L.mock_list(len(items))
        for i, x in enumerate(items):
            self.L.mock_list_set(l, i, x)
        return l

    def attr(self, x, name, v):
        self.L.mock_set_attr(x, name.encode(), v)
        return x

    def matrix(self, v, nrow):
        a = np.ascontiguousarray(v, dtype=np.int32)
        x = self.int(a.T.reshape(-1) if a.ndim == 2 else a)        # column-major
        ncol = a.shape[1] if a.ndim == 2 else len(a) // nrow
        return self.attr(x, "dim", self.int([nrow, ncol]))

    def factor(self, codes1, levels):
        return self.attr(self.attr(self.int(codes1), "levels", self.str(levels)), "class", self.str("factor"))

    def s4(self, klass, **slots):
        o = self.L.mock_s4()
        self.attr(o, "class", self.str(klass))
        for k, v in slots.items():
            self.attr(o, k, v)
        return o

    def rle(self, values, lengths):
        return self.s4(["Rle"], values=values, lengths=self.int(lengths))

    def granges(self, gr, seq_levels=None, klass=("GRanges", "GenomicRanges")):
        """An S4 GRanges as GenomicRanges lays it out (the slots parseRegions reads, ref:
        src/bamsignals.cpp:97-104): seqnames / strand as factor-Rle with run lengths, ranges as IRanges.
        `seq_levels` fixes the level order (it need not be the BAM's order, nor all used)."""
        def runs(vals):
            v, l = [], []
            for x in vals:
                if v and v[-1] == x:
                    l[-1] += 1
                else:
                    v.append(x); l.append(1)
            return v, l
        seq_levels = list(seq_levels or dict.fromkeys(gr.seqnames))
        sv, sl = runs(gr.seqnames)
        tv, tl = runs(gr.strand)
        strand_levels = ["+", "-", "*"]
        return self.s4(list(klass),
                       seqnames=self.rle(self.factor([seq_levels.index(x) + 1 for x in sv], seq_levels), sl),
                       strand=self.rle(self.factor([strand_levels.index(x) + 1 for x in tv], strand_levels), tl),
                       ranges=self.s4(["IRanges"], start=self.int(gr.start), width=self.int(gr.width)))

    # ---- calling and reading back -----------------------------------------------------------------
    def call(self, name, *args):
        arr = (C.c_void_p * len(args))(*args)
        status = C.c_int(0)
        res = self.L.mock_call(name.encode(), arr, len(args), C.byref(status))
        msg = self.L.mock_error().decode("utf-8", "replace")
        assert self.L.mock_protect_depth() == 0
        if status.value == 1:
            raise RError(msg)
        if status.value:
            raise RViolation(msg)
        return res

    def to_py(self, x):
        t = self.L.mock_type(x)
        n = self.L.mock_len(x)
        if t in (TYPES["INTSXP"], TYPES["LGLSXP"]):
            a = np.ctypeslib.as_array(C.cast(self.L.mock_int_data(x), C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
            d = self.L.mock_get_attr(x, b"dim")
            if d != self.nil and d is not None:
                dims = self.to_py(d)
                a = a.reshape(tuple(int(q) for q in dims[::-1])).T          # column-major
            return a.astype(bool) if t == TYPES["LGLSXP"] else a
        if t == TYPES["STRSXP"]:
            return [self.L.mock_string(x, i).decode() for i in range(n)]
        if t == TYPES["VECSXP"]:
            return [self.to_py(self.L.mock_list_get(x, i)) for i in range(n)]
        if t == TYPES["NILSXP"]:
            return None
        raise TypeError(f"type {t} not modelled")

    def dimnames(self, x):
        d = self.L.mock_get_attr(x, b"dimnames")
        return None if d == self.nil or d is None else self.to_py(d)

    def registered(self):
        return {self.L.mock_registered_name(i).decode(): self.L.mock_registered_arity(i) for i in range(self.L.mock_n_registered())}
