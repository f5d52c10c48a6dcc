#!/usr/bin/env python3
"""Diagnostic: what a new session pays before its first call -- loading the library (the ROCm runtime starts
with it), the first context of a device (first allocation, first launches out of every code object, first
copies), a second context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from bamsignals_amd import _lib
lib = _lib.load()
t1 = time.perf_counter()
from bamsignals_amd.device import Context
t2 = time.perf_counter()
c = Context(0)
t3 = time.perf_counter()
c2 = Context(0)
t4 = time.perf_counter()
print("load lib %.3f s, import device %.3f s, first context %.3f s, second context %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
c2.close(); c.close()
