"""Minimal reader for R's XDR serialisation (``save()`` files: gzip + RDX2/RDX3).

Used only by ``make_golden.py`` (in the dev container, where /root/reference is
mounted) to turn the reference's ``.RData`` fixtures into plain arrays.  Written
from the R Internals description of the serialisation format.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np


class RObj:
    """A decoded R value: ``value`` plus ``attr`` (dict, insertion-ordered)."""

    def __init__(self, kind, value=None, attr=None):
        self.kind = kind
        self.value = value
        self.attr = attr or {}

    def __repr__(self):
        return f"RObj({self.kind}, attr={list(self.attr)})"


class _Reader:
    def __init__(self, data):
        self.d = data
        self.o = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.d, self.o)[0]
        self.o += 4
        return v

    def take(self, n):
        b = self.d[self.o:self.o + n]
        self.o += n
        return b

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) | (lo & 0xFFFFFFFF)
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)
        if t == 254:      # NILVALUE
            return None
        if t in (253, 252, 251, 250, 249, 248):
            return RObj("special%d" % t)
        if t == 255:      # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:        # SYMSXP
            name = self.item()
            sym = RObj("sym", name.value)
            self.refs.append(sym)
            return sym
        if t in (247, 246, 245):   # NAMESPACESXP / PACKAGESXP / PERSISTSXP
            self.i32()
            n = self.i32()
            vals = [self.item() for _ in range(n)]
            o = RObj("namespace", [v.value for v in vals])
            self.refs.append(o)
            return o
        if t == 2 or t == 6:        # LISTSXP / LANGSXP: pairlist, iterate over cdr
            items = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                items.append((tag.value if tag is not None else None, car))
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6):
                    raise ValueError("unexpected pairlist tail type %d" % t2)
                has_attr = bool(flags & 0x200)
                has_tag = bool(flags & 0x400)
            return RObj("pairlist", items)
        if t == 9:        # CHARSXP
            n = self.i32()
            if n == -1:
                return RObj("char", None)
            return RObj("char", self.take(n).decode("utf-8", "replace"))
        if t == 10 or t == 13:
            n = self.length()
            v = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
            o = RObj("lgl" if t == 10 else "int", v)
        elif t == 14:
            n = self.length()
            v = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
            o = RObj("real", v)
        elif t == 16:
            n = self.length()
            o = RObj("str", [self.item().value for _ in range(n)])
        elif t == 19 or t == 20:
            n = self.length()
            o = RObj("list", [self.item() for _ in range(n)])
        elif t == 25:     # S4SXP: only attributes
            o = RObj("S4")
        elif t == 238:    # ALTREP: (info, state, attr)
            info = self.item()
            state = self.item()
            attr = self.item()
            cls = info.value[0][1].value if info is not None else ""
            if cls == "compact_intseq":
                n, start, step = (int(x) for x in state.value)
                o = RObj("int", (start + step * np.arange(n)).astype(np.int32))
            elif cls == "compact_realseq":
                n, start, step = state.value
                o = RObj("real", start + step * np.arange(int(n)))
            elif cls.startswith("wrap_"):
                o = state.value[0][1] if state.kind == "pairlist" else state.value[0]
            else:
                raise ValueError("unsupported ALTREP class " + cls)
            if attr is not None:
                o.attr.update({k: v for k, v in attr.value})
            return o
        else:
            raise ValueError("unsupported SEXP type %d at offset %d" % (t, self.o))
        if has_attr:
            a = self.item()
            if a is not None:
                o.attr.update({k: v for k, v in a.value})
        return o


def load_rdata(path):
    """Returns {name: RObj} for every object saved in the file."""
    data = gzip.open(path, "rb").read()
    if data[:5] not in (b"RDX2\n", b"RDX3\n"):
        raise ValueError("not an RDX2/RDX3 file")
    if data[5:7] != b"X\n":
        raise ValueError("not XDR")
    r = _Reader(data)
    r.o = 7
    version = r.i32()
    r.i32(); r.i32()
    if version == 3:
        n = r.i32()
        r.take(n)
    top = r.item()
    return {k: v for k, v in top.value}


def factor_codes(o):
    """(codes int32 1-based, levels list[str]) of an R factor."""
    return o.value, o.attr["levels"].value
