"""bamCount / bamProfile / bamCoverage: the reference's user API (R/wrappers.R:75-184), with the
same argument names, defaults, normalisation, warnings, errors and return shapes, over the C ABI.

R is not available in the build image, so this Python module is the host side that can be run
and tested here.  Under R nothing of this is needed: the reference's own ``R/`` files stay as they
are and only ``src/`` is swapped (bamsignals_amd/r_package/graft_into_reference.sh).
``paired.end`` is spelled ``paired_end``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import warnings

import numpy as np

from . import _lib
from .countsignals import CountSignals
from .granges import GRanges


def _match_arg(value, choices, name):
    """R's match.arg: a vector of choices means the first one; unique prefixes are accepted."""
    if isinstance(value, (list, tuple)):
        if list(value) == list(choices):
            return choices[0]
        if len(value) != 1:
            raise ValueError(f"'{name}' must be of length 1")
        value = value[0]
    hits = [c for c in choices if c.startswith(str(value))] if value != "" else []
    if value in choices:
        return value
    if len(hits) != 1:
        raise ValueError(f"'{name}' should be one of " + ", ".join(f"‘{c}’" for c in choices))
    return hits[0]


def flagMask(paired_end):
    """R/wrappers.R:76-81: only the first read of a properly mapped pair (0x2 | 0x40) unless ignored."""
    return 66 if paired_end != "ignore" else 0


def tlenFilter(tlenFilter, paired_end):  # noqa: N802,N803 - reference names
    """R/wrappers.R:84-98."""
    if paired_end == "ignore":
        return ()
    if tlenFilter is None:
        return (0, 1000)
    tf = list(np.atleast_1d(tlenFilter))
    if len(tf) != 2 or tf[0] < 0 or tf[1] < 0:
        raise ValueError("tlenFilter must be NULL or vector of 2 positive integers")
    if tf[0] > tf[1]:
        raise ValueError("tlenFilter[1] must be smaller or equal to tlenFilter[2]")
    return (int(tf[0]), int(tf[1]))


def _print_sentence(path):
    """The reference prints a progress line per call when ``verbose`` (R/wrappers.R:175-184); only its
    presence matters, so the text here is a plain one."""
    print(f"Processing {path}", file=sys.stderr)


# GPU the file-level calls run on when the caller passes no ``device``: -1 = the library's own choice
# (env BAMSIGNALS_DEVICES, else BAMSIGNALS_DEVICE, else GPU 0).  One-process-per-GPU hosts
# (bamsignals_amd.dist) set it to their rank's GPU.
_default_device = -1


def set_default_device(device):
    """Make ``device`` (a GPU ordinal, or -1 for the library's choice) the default of bamCount /
    bamProfile / bamCoverage in this process."""
    global _default_device
    _default_device = int(device)


def _dev(device):
    return _default_device if device is None else int(device)


def _check_gr(gr):
    if not isinstance(gr, GRanges):
        raise TypeError("must provide a GRanges object")        # ref: src/bamsignals.cpp:93-94


def _split(out, off, ss):
    out.setflags(write=False)          # the signals are views of this buffer: read-only container
    n = len(off) - 1
    if n > 64:
        # ranges of one width (tilings, fixed windows around features): the rows of ONE reshaped view -- a million
        # per-range views cost 0.16 s this way, 0.35 s sliced one by one
        w = int(off[1] - off[0])
        if w > 0 and int(off[-1]) == n * w and bool(np.all(np.diff(off) == w)):
            if ss:
                return list(out[:n * w].reshape(n, w // 2, 2).transpose(0, 2, 1))
            return list(out[:n * w].reshape(n, w))
    off = off.tolist()
    if ss:
        return [out[a:b].reshape(-1, 2).T for a, b in zip(off[:-1], off[1:])]
    return [out[a:b] for a, b in zip(off[:-1], off[1:])]


def pileup_core(bampath, gr, tlen_filter, mapqual=0, binsize=1, shift=0, ss=False, requiredF=0,
                filteredF=-1, pe_mid=False, maxgap=16385, device=None):
    """The native entry point behind bamCount/bamProfile (ref: R/RcppExports.R:12-14).  Returns the
    R list as a Python list: per-range vectors / 2 x w matrices, or, for binsize <= 0, a list of
    length one holding the count vector / 2 x n matrix."""
    _check_gr(gr)
    lib = _lib.load()
    levels, codes, start, width, strand = gr.flatten()
    n = len(gr)
    off = np.empty(n + 1, dtype=np.int64)
    cells = lib.bsig_layout(n, width.ctypes.data, int(binsize), int(bool(ss)), off.ctypes.data)
    out = np.zeros(cells, dtype=np.int32)
    tf = np.asarray([int(x) for x in tlen_filter], dtype=np.int32)
    names = (C.c_char_p * max(len(levels), 1))(*[s.encode() for s in levels])
    _lib.check(lib.bsig_pileup_core(os.path.expanduser(str(bampath)).encode(), n, codes.ctypes.data, len(levels),
                                    names, start.ctypes.data, width.ctypes.data, strand.ctypes.data,
                                    tf.ctypes.data, len(tf), int(mapqual), int(binsize), int(shift),
                                    int(bool(ss)), int(requiredF), int(filteredF), int(bool(pe_mid)),
                                    int(maxgap), _dev(device), out.ctypes.data, off.ctypes.data))
    if binsize <= 0:
        return [out.reshape(-1, 2).T if ss else out]
    return _split(out, off, ss)


def coverage_core(bampath, gr, tlen_filter, mapqual=0, requiredF=0, filteredF=-1, tspan=False,
                  maxgap=16385, device=None):
    """The native entry point behind bamCoverage (ref: R/RcppExports.R:16-18)."""
    _check_gr(gr)
    lib = _lib.load()
    levels, codes, start, width, strand = gr.flatten()
    n = len(gr)
    off = np.empty(n + 1, dtype=np.int64)
    cells = lib.bsig_layout(n, width.ctypes.data, 1, 0, off.ctypes.data)
    out = np.zeros(cells, dtype=np.int32)
    tf = np.asarray([int(x) for x in tlen_filter], dtype=np.int32)
    names = (C.c_char_p * max(len(levels), 1))(*[s.encode() for s in levels])
    _lib.check(lib.bsig_coverage_core(os.path.expanduser(str(bampath)).encode(), n, codes.ctypes.data,
                                      len(levels), names, start.ctypes.data, width.ctypes.data,
                                      strand.ctypes.data, tf.ctypes.data, len(tf), int(mapqual),
                                      int(requiredF), int(filteredF), int(bool(tspan)), int(maxgap),
                                      _dev(device), out.ctypes.data, off.ctypes.data))
    return _split(out, off, False)


def _alloc_signals(width, binsize, ss):
    """allocateList (ref: src/bamsignals.cpp:139-192): the per-range vectors / 2 x w matrices, made before the
    counting; returns (list as the caller sees it, the arrays the native side writes into)."""
    mult = 2 if ss else 1
    if binsize <= 0:
        v = np.empty(len(width) * mult, dtype=np.int32)
        return [v.reshape(-1, 2).T if ss else v], [v]
    cells = [0 if w <= 0 else mult * ((int(w) + binsize - 1) // binsize) for w in width]
    vs = [np.empty(c, dtype=np.int32) for c in cells]
    return [v.reshape(-1, 2).T if ss else v for v in vs], vs


def _dest_pointers(vs):
    return (C.c_void_p * max(len(vs), 1))(*[v.ctypes.data if v.size else None for v in vs])


def pileup_core_into(bampath, gr, tlen_filter, mapqual=0, binsize=1, shift=0, ss=False, requiredF=0,
                     filteredF=-1, pe_mid=False, maxgap=16385, device=None):
    """pileup_core with the result delivered in place (bsig_pileup_core_into, what the R shim binds): the
    per-range arrays are allocated first, as allocateList does, and the native side writes straight into them."""
    _check_gr(gr)
    lib = _lib.load()
    levels, codes, start, width, strand = gr.flatten()
    out, vs = _alloc_signals(width, int(binsize), bool(ss))
    tf = np.asarray([int(x) for x in tlen_filter], dtype=np.int32)
    names = (C.c_char_p * max(len(levels), 1))(*[s.encode() for s in levels])
    _lib.check(lib.bsig_pileup_core_into(os.path.expanduser(str(bampath)).encode(), len(gr), codes.ctypes.data, len(levels),
                                         names, start.ctypes.data, width.ctypes.data, strand.ctypes.data,
                                         tf.ctypes.data, len(tf), int(mapqual), int(binsize), int(shift),
                                         int(bool(ss)), int(requiredF), int(filteredF), int(bool(pe_mid)),
                                         int(maxgap), _dev(device), _dest_pointers(vs)))
    return out


def coverage_core_into(bampath, gr, tlen_filter, mapqual=0, requiredF=0, filteredF=-1, tspan=False,
                       maxgap=16385, device=None):
    """coverage_core with the result delivered in place (bsig_coverage_core_into)."""
    _check_gr(gr)
    lib = _lib.load()
    levels, codes, start, width, strand = gr.flatten()
    out, vs = _alloc_signals(width, 1, False)
    tf = np.asarray([int(x) for x in tlen_filter], dtype=np.int32)
    names = (C.c_char_p * max(len(levels), 1))(*[s.encode() for s in levels])
    _lib.check(lib.bsig_coverage_core_into(os.path.expanduser(str(bampath)).encode(), len(gr), codes.ctypes.data,
                                           len(levels), names, start.ctypes.data, width.ctypes.data,
                                           strand.ctypes.data, tf.ctypes.data, len(tf), int(mapqual),
                                           int(requiredF), int(filteredF), int(bool(tspan)), int(maxgap),
                                           _dev(device), _dest_pointers(vs)))
    return out


def last_call_timing():
    """Stage seconds of this thread's last bamCount/bamProfile/bamCoverage call, and where the stages' time went
    (``alloc_*``: seconds inside the driver's allocator -- hipMalloc, hipFree, hipHostMalloc -- metered per
    process; ``plan`` / ``kernels`` / ``download``: the parts of ``plan_run_download`` on one GPU)."""
    t = (C.c_double * 16)()
    _lib.load().bsig_last_call_timing_ex(t, 16)
    return dict(open=t[0], decode=t[1], upload_and_layout=t[2], plan_run_download=t[3], total=t[4],
                bam_was_resident=bool(t[5]), alloc_in_decode_and_layout=t[6], alloc_in_plan_run_download=t[7],
                alloc_total=t[8], plan=t[9], kernels=t[10], download=t[11], alloc_calls=int(t[12]),
                reserved_bytes=int(t[14]), reservation_wait=t[15])


def last_call_route():
    """How this thread's last file-level call was carried out (GPUs, decode route, result route)."""
    return _lib.load().bsig_last_call_route().decode()


def bamCount(bampath, gr, mapqual=0, shift=0, ss=False, paired_end=("ignore", "filter", "midpoint"),  # noqa: N802
             tlenFilter=None, filteredFlag=-1, verbose=True):  # noqa: N803
    """For each range, count the reads whose 5' end maps in it (R/wrappers.R:101-120).
    Returns an int32 vector, or a 2 x n matrix (rows sense, antisense) with ``ss=True``."""
    if verbose:
        _print_sentence(bampath)
    pe = _match_arg(paired_end, ("ignore", "filter", "midpoint"), "paired.end")
    pu = pileup_core(os.path.expanduser(str(bampath)), gr, globals()["tlenFilter"](tlenFilter, pe), mapqual, -1,
                     shift, ss, flagMask(pe), filteredFlag, pe == "midpoint")
    return pu[0]


def bamProfile(bampath, gr, binsize=1, mapqual=0, shift=0, ss=False,  # noqa: N802
               paired_end=("ignore", "filter", "midpoint"), tlenFilter=None, filteredFlag=-1, verbose=True):  # noqa: N803
    """For each base pair (or bin) of the ranges, the number of reads whose 5' end maps there
    (R/wrappers.R:124-151).  Returns a CountSignals."""
    if verbose:
        _print_sentence(bampath)
    if binsize < 1:
        raise ValueError("provide a binsize greater or equal to 1")
    _check_gr(gr)
    if binsize > 1 and np.any(gr.width % binsize != 0):
        warnings.warn("some ranges' widths are not a multiple of the selected\n"
                      "             binsize, some bins will correspond to less than binsize basepairs")
    pe = _match_arg(paired_end, ("ignore", "filter", "midpoint"), "paired.end")
    pu = pileup_core(os.path.expanduser(str(bampath)), gr, globals()["tlenFilter"](tlenFilter, pe), mapqual,
                     int(binsize), shift, ss, flagMask(pe), filteredFlag, pe == "midpoint")
    return CountSignals(pu, bool(ss), _trusted=True)


def bamCoverage(bampath, gr, mapqual=0, paired_end=("ignore", "extend"), tlenFilter=None,  # noqa: N802,N803
                filteredFlag=-1, verbose=True):
    """For each base pair of the ranges, the number of reads covering it (R/wrappers.R:154-173)."""
    if verbose:
        _print_sentence(bampath)
    pe = _match_arg(paired_end, ("ignore", "extend"), "paired.end")
    pu = coverage_core(os.path.expanduser(str(bampath)), gr, globals()["tlenFilter"](tlenFilter, pe), mapqual,
                       flagMask(pe), filteredFlag, pe == "extend")
    return CountSignals(pu, False, _trusted=True)
