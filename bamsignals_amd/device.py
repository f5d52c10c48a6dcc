"""Handle-level host API over the C ABI: a device context, reads resident in HBM, plans.

This is plumbing for the reference-shaped entry points in ``wrappers.py`` and for ``bench.py``;
all arithmetic happens in the HIP kernels (csrc/kernels.hip).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class _PinnedBlock:
    """Owner of one bsig_host_alloc() block; freed when the last numpy view of it is collected."""

    def __init__(self, nbytes):
        self._lib = _lib.load()
        p = C.c_void_p()
        _lib.check(self._lib.bsig_host_alloc(int(nbytes), C.byref(p)))
        self.ptr = p.value
        self.buf = (C.c_char * max(int(nbytes), 1)).from_address(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            self._lib.bsig_host_free(C.c_void_p(self.ptr))
            self.ptr = None


def pinned_empty(n, dtype):
    """numpy array of ``n`` elements in page-locked host memory."""
    dtype = np.dtype(dtype)
    block = _PinnedBlock(n * dtype.itemsize)
    arr = np.frombuffer(block.buf, dtype=dtype, count=n)
    # the ctypes buffer keeps `block` alive through arr.base; tie them explicitly as well
    block.buf._owner = block
    return arr


class Context:
    """One GPU + the HIP stream the kernels are launched on."""

    def __init__(self, device=0, stream=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.bsig_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def stream(self):
        return self._lib.bsig_ctx_stream(self._h)

    def sync(self):
        _lib.check(self._lib.bsig_ctx_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bsig_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Reads:
    """Read columns resident in HBM (span classes + bucket index, see csrc/bsig_types.h).

    ``end`` may be omitted when ``cigar_off``/``cigar`` (packed ``len<<4|op``) are given: the GPU
    then derives ``bam_endpos - 1`` itself.
    """

    def __init__(self, ctx, ref_len, ref_off, pos, flag, mapq, tlen, end=None, cigar_off=None, cigar=None):
        self._lib = _lib.load()
        self.ctx = ctx
        ref_len = _i32(ref_len)
        ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
        pos = _i32(pos)
        flag = np.ascontiguousarray(flag, dtype=np.uint16)
        mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        tlen = _i32(tlen)
        n = len(pos)
        if not (len(flag) == len(mapq) == len(tlen) == n):
            raise ValueError("read columns differ in length")
        cols = _lib.Columns()
        cols.n_reads = n
        cols.n_ref = len(ref_len)
        cols.ref_len = _ptr(ref_len).value
        cols.ref_off = _ptr(ref_off).value
        cols.pos, cols.flag, cols.mapq, cols.tlen = (_ptr(a).value for a in (pos, flag, mapq, tlen))
        keep = [ref_len, ref_off, pos, flag, mapq, tlen]
        if end is not None:
            end = _i32(end)
            if len(end) != n:
                raise ValueError("end column differs in length")
            cols.end = _ptr(end).value
            keep.append(end)
        else:
            if cigar_off is None or cigar is None:
                raise ValueError("need either end or cigar_off + cigar")
            cigar_off = np.ascontiguousarray(cigar_off, dtype=np.int64)
            cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
            if len(cigar_off) != n + 1:
                raise ValueError("cigar_off must have n_reads + 1 entries")
            cols.cigar_off = _ptr(cigar_off).value
            cols.cigar = _ptr(cigar).value
            keep += [cigar_off, cigar]
        h = C.c_void_p()
        _lib.check(self._lib.bsig_reads_upload(ctx._h, C.byref(cols), C.byref(h)))
        self._h = h
        self.n_reads = n
        self.n_ref = len(ref_len)

    @classmethod
    def from_bam(cls, ctx, bam, threads=0):
        """The whole of an open ``bamio.BamFile`` decoded to HBM: BGZF inflate on the CPU thread pool,
        records -> columns on the GPU (csrc/devdecode.hip; CPU decode where the file needs it)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.ctx = ctx
        h = C.c_void_p()
        _lib.check(self._lib.bsig_reads_from_bam(ctx._h, bam._h, int(threads), C.byref(h)))
        self._h = h
        self.n_ref = len(bam.ref_len)
        self.n_reads = self.info()["n_reads"]
        return self

    @classmethod
    def from_bam_multi(cls, ctxs, bam, threads=0):
        """The whole BAM resident on every context's GPU: each GPU decodes one share of the BGZF blocks,
        the column shares are exchanged over xGMI.  Returns (list of Reads, sharded?)."""
        lib = _lib.load()
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c._h for c in ctxs])
        hs = (C.c_void_p * n)()
        sharded = C.c_int32(0)
        _lib.check(lib.bsig_reads_from_bam_multi(arr, n, bam._h, int(threads), hs, C.byref(sharded)))
        out = []
        for k in range(n):
            self = cls.__new__(cls)
            self._lib = lib
            self.ctx = ctxs[k]
            self._h = C.c_void_p(hs[k])
            self.n_ref = len(bam.ref_len)
            self.n_reads = self.info()["n_reads"]
            out.append(self)
        return out, bool(sharded.value)

    @classmethod
    def from_bam_regions(cls, ctx, bam, rid, beg, end, threads=0):
        """The records the BAI lists for the regions [beg, end) (0-based) decoded to HBM -- a superset
        of the overlapping records, each once, in file order (what ``BamFile.decode(rid, beg, end)``
        returns on the host)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.ctx = ctx
        rid = np.ascontiguousarray(rid, dtype=np.int32)
        beg = np.ascontiguousarray(beg, dtype=np.int64)
        end = np.ascontiguousarray(end, dtype=np.int64)
        if not (len(rid) == len(beg) == len(end)):
            raise ValueError("region arrays differ in length")
        h = C.c_void_p()
        _lib.check(self._lib.bsig_reads_from_bam_regions(ctx._h, bam._h, len(rid), _ptr(rid), _ptr(beg), _ptr(end),
                                                         int(threads), C.byref(h)))
        self._h = h
        self.n_ref = len(bam.ref_len)
        self.n_reads = self.info()["n_reads"]
        return self

    def save(self, path, stamp=""):
        """Write the resident layout to ``path`` (bsig_reads_save); ``stamp`` ties it to its source."""
        _lib.check(self._lib.bsig_reads_save(self._h, str(path).encode(), str(stamp).encode()))

    @classmethod
    def load(cls, ctx, path, stamp=""):
        """Resident reads from a file written by ``save`` (fails if ``stamp`` differs)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.ctx = ctx
        h = C.c_void_p()
        _lib.check(self._lib.bsig_reads_load(ctx._h, str(path).encode(), str(stamp).encode(), C.byref(h)))
        self._h = h
        self.n_reads = self.info()["n_reads"]
        self.n_ref = None
        return self

    def clone(self, ctx):
        """A copy of these resident reads on ``ctx``'s GPU (device-to-device)."""
        other = type(self).__new__(type(self))
        other._lib = self._lib
        other.ctx = ctx
        h = C.c_void_p()
        _lib.check(self._lib.bsig_reads_clone(self._h, ctx._h, C.byref(h)))
        other._h = h
        other.n_ref, other.n_reads = self.n_ref, self.n_reads
        return other

    @staticmethod
    def device_decode_timing():
        """Stage seconds of this thread's last ``from_bam`` (all 0 when it took the CPU decode)."""
        t = (C.c_double * 6)()
        _lib.load().bsig_device_decode_timing(t)
        return dict(zip(("block_scan", "inflate", "copy_wait", "gpu_parse", "total", "layout"), list(t)))

    def info(self):
        inf = _lib.ReadsInfo()
        _lib.check(self._lib.bsig_reads_get_info(self._h, C.byref(inf)))
        # class_*: [span <= 256 with a rare flag/mapq pair, <= 4096, <= 65536, longer, packed (span <= 256, one
        # word per read)]; n_codes = pairs in the packed class's table
        return dict(n_reads=inf.n_reads, hbm_bytes=inf.hbm_bytes, n_classes=inf.n_classes, n_codes=inf.n_codes,
                    class_n=list(inf.class_n), class_maxspan=list(inf.class_maxspan),
                    class_bucket_shift=list(inf.class_bucket_shift))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bsig_reads_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def make_params(mode, tlen_filter=(), mapqual=0, binsize=1, shift=0, ss=False, requiredF=0,
                filteredF=-1, pe_mid=False, tspan=False, tile_cells=0, threads=0):
    p = _lib.Params()
    p.mode = mode
    p.mapqual = int(mapqual)
    p.binsize = int(binsize)
    p.shift = int(shift)
    p.ss = int(bool(ss))
    p.requiredF = int(requiredF)
    p.filteredF = int(filteredF)
    p.pe_mid = int(bool(pe_mid))
    p.tspan = int(bool(tspan))
    tf = [] if tlen_filter is None else [int(x) for x in tlen_filter]
    p.n_tlen_filter = len(tf)
    if len(tf) not in (0, 2):
        raise ValueError("tlen_filter must have 0 or 2 elements")
    for i, v in enumerate(tf):
        p.tlen_filter[i] = v
    p.tile_cells = int(tile_cells)
    p.threads = int(threads)
    return p


class Plan:
    """Ranges + call parameters resident in HBM; run it any number of times."""

    def __init__(self, ctx, reads, rid, loc, length, strand, params):
        self._lib = _lib.load()
        self.ctx, self.reads = ctx, reads
        rid, loc, length, strand = _i32(rid), _i32(loc), _i32(length), _i32(strand)
        n = len(rid)
        if not (len(loc) == len(length) == len(strand) == n):
            raise ValueError("range arrays differ in length")
        h = C.c_void_p()
        _lib.check(self._lib.bsig_plan_create(ctx._h, reads._h, n, _ptr(rid), _ptr(loc), _ptr(length),
                                              _ptr(strand), C.byref(params), C.byref(h)))
        self._h = h
        self.n_ranges = n
        self.cells = int(self._lib.bsig_plan_cells(h))
        self.offsets = np.ctypeslib.as_array(self._lib.bsig_plan_offsets(h), shape=(n + 1,)).copy()

    def run_host(self, out=None, pinned=False):
        """Run and return the flat int32 result in host memory.  ``out``: a reusable int32 array of
        ``cells`` elements (e.g. from ``pinned_empty``); ``pinned=True`` allocates a page-locked one
        (bsig_host_alloc), into which the D2H copy runs at PCIe DMA rate."""
        if out is None:
            out = pinned_empty(self.cells, np.int32) if pinned else np.empty(self.cells, dtype=np.int32)
        elif out.dtype != np.int32 or out.size != self.cells or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous int32 array of plan.cells elements")
        _lib.check(self._lib.bsig_plan_run_host(self._h, _ptr(out)))
        return out

    def run_device(self, out_ptr):
        """Asynchronous launch on the context's stream; ``out_ptr`` is a device address."""
        _lib.check(self._lib.bsig_plan_run(self._h, C.c_void_p(out_ptr)))

    def stats(self):
        s = _lib.PlanStats()
        _lib.check(self._lib.bsig_plan_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bsig_plan_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class SegmentMap:
    """Device-side reassembly of sharded results: segment k of a source buffer goes to the destination at
    ``dst_off[which[k]]`` (bsig_segmap_*; the host-side twin is bsig_scatter_segments).  The tables are
    uploaded once; ``run`` is one asynchronous kernel launch on the context's stream."""

    def __init__(self, ctx, src_off, dst_off, which):
        self._lib = _lib.load()
        self.ctx = ctx
        src_off = np.ascontiguousarray(src_off, dtype=np.int64)
        dst_off = np.ascontiguousarray(dst_off, dtype=np.int64)
        which = np.ascontiguousarray(which, dtype=np.int64)
        if len(src_off) != len(which) + 1:
            raise ValueError("src_off must have one more entry than which")
        h = C.c_void_p()
        _lib.check(self._lib.bsig_segmap_create(ctx._h, len(which), _ptr(src_off), len(dst_off) - 1, _ptr(dst_off), _ptr(which),
                                                C.byref(h)))
        self._h = h

    def run(self, src_ptr, dst_ptr):
        _lib.check(self._lib.bsig_segmap_run(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr)))

    def run_narrow(self, msg_ptr, n_cells, cap, dst_ptr):
        """``run`` from a shard that travelled as a narrow message (``narrow_pack``): two bits a cell + exceptions."""
        _lib.check(self._lib.bsig_segmap_run_narrow(self._h, C.c_void_p(msg_ptr), int(n_cells), int(cap), C.c_void_p(dst_ptr)))

    def narrow_overflowed(self):
        """Did any narrow message so far hold more exceptions than its list had room for?  (Synchronises.)"""
        v = C.c_int(0)
        _lib.check(self._lib.bsig_segmap_narrow_overflowed(self._h, C.byref(v)))
        return bool(v.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bsig_segmap_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def narrow_bytes(n_cells, cap):
    """Bytes of the narrow message of a shard of ``n_cells`` cells with room for ``cap`` exceptions (bsig_narrow_bytes)."""
    return int(_lib.load().bsig_narrow_bytes(int(n_cells), int(cap)))


def narrow_pack(ctx, src_ptr, n_cells, msg_ptr, cap):
    """Pack the int32 shard at ``src_ptr`` (device, 16-B aligned) into the narrow message at ``msg_ptr``: asynchronous, on
    the context's stream."""
    _lib.check(_lib.load().bsig_narrow_pack(ctx._h, C.c_void_p(src_ptr), int(n_cells), C.c_void_p(msg_ptr), int(cap)))


def narrow_count(ctx, msg_ptr):
    """Exceptions of the packed shard at ``msg_ptr`` (synchronises): what ``cap`` has to hold."""
    v = C.c_int64(0)
    _lib.check(_lib.load().bsig_narrow_count(ctx._h, C.c_void_p(msg_ptr), C.byref(v)))
    return int(v.value)


def layout(length, binsize, ss):
    """Flat offsets of the result (allocateList's shapes, ref: src/bamsignals.cpp:139-192)."""
    lib = _lib.load()
    length = _i32(length)
    off = np.empty(len(length) + 1, dtype=np.int64)
    lib.bsig_layout(len(length), _ptr(length), int(binsize), int(bool(ss)), _ptr(off))
    return off
