"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/liboracle.so (the C restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


class _Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_ref", C.c_int32), ("ref_off", C.c_void_p),
                ("pos", C.c_void_p), ("end", C.c_void_p), ("flag", C.c_void_p),
                ("mapq", C.c_void_p), ("tlen", C.c_void_p), ("max_span", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.bsor_layout.restype = C.c_int64
        _lib.bsor_max_span.restype = C.c_int32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleReads:
    """Columns pinned in numpy arrays + the struct the C oracle reads."""

    def __init__(self, ref_off, pos, end, flag, mapq, tlen):
        self.ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.end = np.ascontiguousarray(end, dtype=np.int32)
        self.flag = np.ascontiguousarray(flag, dtype=np.uint16)
        self.mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        self.tlen = np.ascontiguousarray(tlen, dtype=np.int32)
        n = len(self.pos)
        ms = lib().bsor_max_span(C.c_int64(n), _p(self.pos), _p(self.end))
        self.c = _Reads(n, len(self.ref_off) - 1, _p(self.ref_off).value, _p(self.pos).value,
                        _p(self.end).value, _p(self.flag).value, _p(self.mapq).value,
                        _p(self.tlen).value, ms)


def cigar_end(pos, flag, cigar_off, cigar):
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    flag = np.ascontiguousarray(flag, dtype=np.uint16)
    cigar_off = np.ascontiguousarray(cigar_off, dtype=np.int64)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    out = np.empty(len(pos), dtype=np.int32)
    lib().bsor_cigar_end(C.c_int64(len(pos)), _p(pos), _p(flag), _p(cigar_off), _p(cigar), _p(out))
    return out


def layout(length, binsize, ss):
    length = np.ascontiguousarray(length, dtype=np.int32)
    off = np.empty(len(length) + 1, dtype=np.int64)
    lib().bsor_layout(C.c_int64(len(length)), _p(length), C.c_int(binsize), C.c_int(int(ss)), _p(off))
    return off


def _ranges(ranges):
    return [np.ascontiguousarray(ranges[k], dtype=np.int32) for k in ("rid", "loc", "len", "strand")]


def pileup_core(reads: OracleReads, ranges, tlen_filter=(), mapqual=0, binsize=1, shift=0,
                ss=False, requiredF=0, filteredF=-1, pe_mid=False, maxgap=16385):
    rid, loc, ln, strand = _ranges(ranges)
    off = layout(ln, binsize, ss)
    out = np.empty(int(off[-1]), dtype=np.int32)
    tf = np.ascontiguousarray(tlen_filter if tlen_filter is not None else (), dtype=np.int32)
    rc = lib().bsor_pileup_core(C.byref(reads.c), C.c_int64(len(loc)), _p(rid), _p(loc), _p(ln),
                                _p(strand), _p(tf), C.c_int(len(tf)), C.c_int(mapqual),
                                C.c_int(binsize), C.c_int(shift), C.c_int(int(ss)),
                                C.c_int(requiredF), C.c_int(filteredF), C.c_int(int(pe_mid)),
                                C.c_int(maxgap), _p(out), _p(off))
    if rc != 0:
        raise ValueError("oracle pileup_core rejected its arguments")
    return out, off


def coverage_core(reads: OracleReads, ranges, tlen_filter=(), mapqual=0, requiredF=0,
                  filteredF=-1, tspan=False, maxgap=16385):
    rid, loc, ln, strand = _ranges(ranges)
    off = layout(ln, 1, False)
    out = np.empty(int(off[-1]), dtype=np.int32)
    tf = np.ascontiguousarray(tlen_filter if tlen_filter is not None else (), dtype=np.int32)
    rc = lib().bsor_coverage_core(C.byref(reads.c), C.c_int64(len(loc)), _p(rid), _p(loc), _p(ln),
                                  _p(strand), _p(tf), C.c_int(len(tf)), C.c_int(mapqual),
                                  C.c_int(requiredF), C.c_int(filteredF), C.c_int(int(tspan)),
                                  C.c_int(maxgap), _p(out), _p(off))
    if rc != 0:
        raise ValueError("oracle coverage_core rejected its arguments")
    return out, off
