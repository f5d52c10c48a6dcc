"""CountSignals: the container bamProfile / bamCoverage return (ref: R/zzzCountSignals.R:27-143,
src/CountSignals.cpp:4-29).  A read-only list of int32 vectors, or of 2 x width int32 matrices
(row 0 = sense, row 1 = antisense) when strand-specific.  Python indices are 0-based."""
from __future__ import annotations

import numpy as np

ROWNAMES = ("sense", "antisense")


def _check_list(signals, ss):
    """checkList (ref: src/CountSignals.cpp:4-16) through the native routine the R shim calls too:
    per element "is an int32 array", the number of dimensions (a plain vector has no dim attribute
    in R: 0) and the first dimension."""
    from . import _lib
    n = len(signals)
    is_int = np.asarray([isinstance(s, np.ndarray) and s.dtype == np.int32 for s in signals], dtype=np.int32)
    n_dim = np.asarray([s.ndim if isinstance(s, np.ndarray) and s.ndim > 1 else 0 for s in signals], dtype=np.int32)
    dim0 = np.asarray([s.shape[0] if isinstance(s, np.ndarray) and s.ndim > 1 else 0 for s in signals], dtype=np.int32)
    return bool(_lib.load().bsig_check_list(n, is_int.ctypes.data, n_dim.ctypes.data, dim0.ctypes.data, int(bool(ss))))


class CountSignals:
    def __init__(self, signals, ss, _trusted=False):
        if not isinstance(ss, (bool, np.bool_)):
            raise ValueError("invalid ss slot")                      # R/zzzCountSignals.R:36
        signals = list(signals)
        # _trusted: lists produced by the native call itself (views of one read-only int32 buffer)
        if not _trusted and not _check_list(signals, bool(ss)):
            raise ValueError("invalid list")                         # R/zzzCountSignals.R:38
        self._signals = signals
        self.ss = bool(ss)
        if not _trusted:
            for s in self._signals:
                s.setflags(write=False)

    def __len__(self):                                               # length(), :46
        return len(self._signals)

    def width(self):                                                 # width() -> fastWidth, :54-56
        from . import _lib
        length = np.asarray([s.size for s in self._signals], dtype=np.int64)
        out = np.empty(len(length), dtype=np.int32)
        _lib.load().bsig_fast_width(len(length), length.ctypes.data, int(self.ss), out.ctypes.data)
        return out

    def __getitem__(self, i):                                        # "[", :68-77
        if isinstance(i, (int, np.integer)):
            return self._signals[i]
        idx = np.arange(len(self))[i]
        return CountSignals([self._signals[k] for k in np.atleast_1d(idx)], self.ss)

    def as_list(self):                                               # as.list, :83-96
        return list(self._signals)

    def alignSignals(self):                                          # :99-113 (simplify2array)
        ws = self.width()
        if len(ws) and np.any(ws != ws[0]):
            raise ValueError("all signals must have the same length")
        if not len(ws):
            return np.zeros((0, 0), dtype=np.int32)
        return np.stack(self._signals, axis=-1)                      # [w, n] or [2, w, n]

    def __iter__(self):
        return iter(self._signals)

    def __repr__(self):                                              # show, :116-143
        n = len(self)
        lines = [f"CountSignals object with {n}{' strand-specific' if self.ss else ''} signal{'s' if n != 1 else ''}"]

        def counts(v):
            txt = " ".join(str(int(x)) for x in v[:10])
            return txt + (" ..." if len(v) > 10 else "")

        for i in range(min(5, n)):
            el = self._signals[i]
            npos = el.shape[1] if self.ss else len(el)
            lines.append(f"[{i + 1}] signal of width {npos}")
            if self.ss:
                lines.append("sense      " + counts(el[0]))
                lines.append("antisense  " + counts(el[1]))
            else:
                lines.append(counts(el))
        if n > 5:
            lines.append("....")
        return "\n".join(lines)
