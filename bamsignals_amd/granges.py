"""A minimal GRanges: exactly the slots the reference's native code reads
(``seqnames``, ``ranges@start``, ``ranges@width``, ``strand``; ref: parseRegions,
src/bamsignals.cpp:92-135).  Coordinates are 1-based and closed, as in GenomicRanges.

A GRanges is immutable: ``seqnames`` and ``strand`` are tuples behind read-only properties and ``start`` /
``width`` are read-only arrays, because the factor encodings the C ABI takes are made once at construction
(an edit afterwards would silently be computed on the old values).  Subset with ``gr[i]`` or build a new one."""
from __future__ import annotations

import numpy as np


class GRanges:
    def __init__(self, seqnames, start, width=None, end=None, strand="*"):
        self._start = np.atleast_1d(np.asarray(start, dtype=np.int64)).astype(np.int32)
        n = len(self._start)
        if width is None:
            if end is None:
                raise ValueError("give width or end")
            width = np.atleast_1d(np.asarray(end, dtype=np.int64)) - self._start + 1
        self._width = np.broadcast_to(np.asarray(width, dtype=np.int64), (n,)).astype(np.int32)
        self._start.setflags(write=False)
        self._width.setflags(write=False)
        if np.any(self._width < 0):
            raise ValueError("negative widths are not allowed")
        if isinstance(seqnames, str):
            seqnames = [seqnames] * n
        self._seqnames = tuple(str(s) for s in seqnames)
        if isinstance(strand, str):
            strand = [strand] * n
        self._strand = tuple(str(s) for s in strand)
        if len(self.seqnames) != n or len(self.strand) != n:
            raise ValueError("seqnames, start, width and strand differ in length")
        bad = set(self.strand) - {"+", "-", "*"}
        if bad:
            raise ValueError(f"invalid strand values {sorted(bad)}")
        # seqnames and strand as GenomicRanges holds them -- factor codes + levels (ref: src/bamsignals.cpp:97-104 reads
        # exactly those): encoded once here, not on every call that flattens the ranges for the C ABI
        self._levels = list(dict.fromkeys(self.seqnames))
        lut = {s: k for k, s in enumerate(self._levels)}
        self._codes = np.fromiter((lut[s] for s in self.seqnames), dtype=np.int32, count=n)
        self._codes.setflags(write=False)
        smap = {"+": 1, "-": -1, "*": 0}
        self._strand_int = np.fromiter((smap[s] for s in self.strand), dtype=np.int32, count=n)
        self._strand_int.setflags(write=False)

    seqnames = property(lambda self: self._seqnames)
    strand = property(lambda self: self._strand)
    start = property(lambda self: self._start)
    width = property(lambda self: self._width)

    def __len__(self):
        return len(self.start)

    @property
    def end(self):
        return self.start + self.width - 1

    def __getitem__(self, i):
        idx = np.arange(len(self))[i]
        idx = np.atleast_1d(idx)
        return GRanges([self.seqnames[k] for k in idx], self.start[idx], width=self.width[idx],
                       strand=[self.strand[k] for k in idx])

    def flatten(self):
        """(levels, codes, start, width, strand_int) as the C ABI's file-level entry points take them."""
        return self._levels, self._codes, self.start, self.width, self._strand_int

    def __repr__(self):
        head = ", ".join(f"{s}:{a}-{b}:{t}" for s, a, b, t in
                         list(zip(self.seqnames, self.start, self.end, self.strand))[:4])
        return f"GRanges with {len(self)} ranges [{head}{', ...' if len(self) > 4 else ''}]"
