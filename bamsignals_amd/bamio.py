"""Host-side BAM access through the C ABI's decode stage (csrc/bamio.cpp): header + BAI,
BAM -> columnar arrays, and the BAM/BAI writers."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class BamFile:
    """An indexed BAM: header, BAI, and decoding of regions (or everything) into columns."""

    def __init__(self, path):
        self._lib = _lib.load()
        self.path = os.path.expanduser(str(path))
        h = C.c_void_p()
        _lib.check(self._lib.bsig_bam_open(self.path.encode(), C.byref(h)))
        self._h = h
        n = self._lib.bsig_bam_n_ref(h)
        self.ref_names = [self._lib.bsig_bam_ref_name(h, i).decode() for i in range(n)]
        self.ref_len = np.asarray([self._lib.bsig_bam_ref_len(h, i) for i in range(n)], dtype=np.int32)

    def name2id(self, name):
        return int(self._lib.bsig_bam_name2id(self._h, str(name).encode()))

    def decode_timing(self):
        """Stage seconds of this thread's last whole-file decode (scan, inflate wait, boundary scan,
        column extraction, total)."""
        t = (C.c_double * 6)()
        self._lib.bsig_bam_decode_timing(t)
        return dict(zip(("block_scan", "inflate_wait", "boundary_scan", "extract", "total", "inflate_busy"), list(t)))

    def decode(self, rid=None, beg=None, end=None, threads=0):
        """Columns of the records the index lists for regions [beg, end) (0-based); all if rid is None."""
        cols = _lib.Columns()
        if rid is None:
            _lib.check(self._lib.bsig_bam_decode(self._h, -1, None, None, None, int(threads), C.byref(cols)))
        else:
            rid = np.ascontiguousarray(rid, dtype=np.int32)
            beg = np.ascontiguousarray(beg, dtype=np.int64)
            end = np.ascontiguousarray(end, dtype=np.int64)
            _lib.check(self._lib.bsig_bam_decode(self._h, len(rid), rid.ctypes.data, beg.ctypes.data,
                                                 end.ctypes.data, int(threads), C.byref(cols)))
        n = cols.n_reads
        nref = cols.n_ref
        out = dict(ref_len=_view(cols.ref_len, nref, np.int32), ref_off=_view(cols.ref_off, nref + 1, np.int64),
                   pos=_view(cols.pos, n, np.int32), flag=_view(cols.flag, n, np.uint16),
                   mapq=_view(cols.mapq, n, np.uint8), tlen=_view(cols.tlen, n, np.int32),
                   cigar_off=_view(cols.cigar_off, n + 1, np.int64))
        out["cigar"] = _view(cols.cigar, int(out["cigar_off"][-1]) if n else 0, np.uint32)
        out["rid"] = np.repeat(np.arange(nref, dtype=np.int32), np.diff(out["ref_off"]))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bsig_bam_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


def write_columns_as_bam(path, ref_names, cols, level=1, l_seq=0, seed=0):
    """Coordinate-sorted columns (ref_len, ref_off, pos, flag, mapq, tlen, cigar_off, cigar) -> BAM + BAI.
    ``l_seq > 0``: real-shaped records (read name, ``l_seq`` random bases + qualities, an NM tag) instead
    of bare ones, for decode benchmarks on data shaped like real BAMs."""
    lib = _lib.load()
    keep = dict(ref_len=np.ascontiguousarray(cols["ref_len"], dtype=np.int32),
                ref_off=np.ascontiguousarray(cols["ref_off"], dtype=np.int64),
                pos=np.ascontiguousarray(cols["pos"], dtype=np.int32),
                flag=np.ascontiguousarray(cols["flag"], dtype=np.uint16),
                mapq=np.ascontiguousarray(cols["mapq"], dtype=np.uint8),
                tlen=np.ascontiguousarray(cols["tlen"], dtype=np.int32),
                cigar_off=np.ascontiguousarray(cols["cigar_off"], dtype=np.int64),
                cigar=np.ascontiguousarray(cols["cigar"], dtype=np.uint32))
    c = _lib.Columns()
    c.n_reads = len(keep["pos"])
    c.n_ref = len(keep["ref_len"])
    for k, a in keep.items():
        setattr(c, k, a.ctypes.data)
    names = (C.c_char_p * len(ref_names))(*[str(s).encode() for s in ref_names])
    if l_seq:
        _lib.check(lib.bsig_write_columns_as_bam_with_seq(os.path.expanduser(str(path)).encode(), len(ref_names), names,
                                                          C.byref(c), int(level), int(l_seq), int(seed)))
    else:
        _lib.check(lib.bsig_write_columns_as_bam(os.path.expanduser(str(path)).encode(), len(ref_names), names,
                                                 C.byref(c), int(level)))


def writeSamAsBamAndIndex(sampath, bampath):
    """Text SAM -> BAM + BAI (ref: writeSamAsBamAndIndex, src/bamsignals.cpp:496-534)."""
    lib = _lib.load()
    _lib.check(lib.bsig_write_sam_as_bam_and_index(os.path.expanduser(str(sampath)).encode(),
                                                   os.path.expanduser(str(bampath)).encode()))
    return True
