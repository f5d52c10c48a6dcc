"""bamsignals_amd — MI355X-native interval counting with the interface of Bioconductor's bamsignals.

User API (mirrors R/wrappers.R and R/zzzCountSignals.R of the reference):
    bamCount, bamProfile, bamCoverage, CountSignals, GRanges, writeSamAsBamAndIndex
Handle-level API for resident data and benchmarking: ``bamsignals_amd.device``.
All compute runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/bamsignals_abi.h; there is no CPU fallback.
"""
from .bamio import BamFile, write_columns_as_bam, writeSamAsBamAndIndex  # noqa: F401
from .countsignals import CountSignals  # noqa: F401
from .granges import GRanges  # noqa: F401
from .wrappers import bamCount, bamCoverage, bamProfile, coverage_core, pileup_core  # noqa: F401

__all__ = ["bamCount", "bamProfile", "bamCoverage", "CountSignals", "GRanges", "BamFile",
           "writeSamAsBamAndIndex", "write_columns_as_bam", "pileup_core", "coverage_core"]
