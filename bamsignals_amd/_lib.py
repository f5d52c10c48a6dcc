"""ctypes binding of libbamsignals_hip.so (the C ABI declared in include/bamsignals_abi.h).

The product path has no CPU fallback: if the shared object is missing this module raises at
import of the first symbol, and every compute call needs a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSIG_LIB_PATH selects another build of the same ABI (the diagnostic stamps build); never a fallback
SO_PATH = os.environ.get("BSIG_LIB_PATH") or os.path.join(_HERE, "libbamsignals_hip.so")

BSIG_OK = 0
MODE_PROFILE, MODE_COUNT, MODE_COVERAGE = 0, 1, 2

ERR_NAMES = {-1: "BSIG_ERR_ARG", -2: "BSIG_ERR_IO", -3: "BSIG_ERR_NOINDEX", -4: "BSIG_ERR_CHROM",
             -5: "BSIG_ERR_EXT", -6: "BSIG_ERR_DEVICE", -7: "BSIG_ERR_NOMEM", -8: "BSIG_ERR_FORMAT"}


class BsigError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code
        self.code_name = ERR_NAMES.get(code, str(code))


class Columns(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_ref", C.c_int32), ("ref_len", C.c_void_p),
                ("ref_off", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p),
                ("mapq", C.c_void_p), ("tlen", C.c_void_p), ("end", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p)]


class ReadsInfo(C.Structure):
    # classes 0..3 by span, 4 = the packed class (include/bamsignals_abi.h: BSIG_N_CLASSES)
    _fields_ = [("n_reads", C.c_int64), ("hbm_bytes", C.c_int64), ("n_classes", C.c_int32), ("n_codes", C.c_int32),
                ("class_n", C.c_int64 * 5), ("class_maxspan", C.c_int32 * 5),
                ("class_bucket_shift", C.c_int32 * 5)]


class Params(C.Structure):
    _fields_ = [("mode", C.c_int32), ("mapqual", C.c_int32), ("binsize", C.c_int32),
                ("shift", C.c_int32), ("ss", C.c_int32), ("requiredF", C.c_int32),
                ("filteredF", C.c_int32), ("pe_mid", C.c_int32), ("tspan", C.c_int32),
                ("n_tlen_filter", C.c_int32), ("tlen_filter", C.c_int32 * 2),
                ("tile_cells", C.c_int32), ("threads", C.c_int32)]


class PlanStats(C.Structure):
    _fields_ = [("n_ranges", C.c_int64), ("n_items", C.c_int64), ("cells", C.c_int64),
                ("visits", C.c_int64), ("visits_short", C.c_int64), ("streamed", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("bytes_per_visit_short", C.c_int32),
                ("bytes_per_visit_long", C.c_int32), ("visits_packed", C.c_int64),
                ("bytes_per_visit_packed", C.c_int32), ("heavy_tiles", C.c_int32)]


_lib = None


def load():
    """Load the HIP shared object.  torch (when importable) is imported first so that both use
    the one libamdhip64 the process already holds."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bamsignals_amd/csrc`.  bamsignals_amd has no CPU fallback.")
    try:  # noqa: SIM105
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement
        pass
    lib = C.CDLL(SO_PATH)
    lib.bsig_last_error.restype = C.c_char_p
    lib.bsig_layout.restype = C.c_int64
    lib.bsig_layout.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.bsig_ctx_create.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.bsig_ctx_destroy.argtypes = [C.c_void_p]
    lib.bsig_ctx_destroy.restype = None
    lib.bsig_ctx_sync.argtypes = [C.c_void_p]
    lib.bsig_ctx_stream.argtypes = [C.c_void_p]
    lib.bsig_ctx_stream.restype = C.c_void_p
    lib.bsig_host_alloc.argtypes = [C.c_int64, C.POINTER(C.c_void_p)]
    lib.bsig_host_free.argtypes = [C.c_void_p]
    lib.bsig_host_free.restype = None
    lib.bsig_reads_upload.argtypes = [C.c_void_p, C.POINTER(Columns), C.POINTER(C.c_void_p)]
    lib.bsig_reads_get_info.argtypes = [C.c_void_p, C.POINTER(ReadsInfo)]
    lib.bsig_reads_clone.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.bsig_reads_free.argtypes = [C.c_void_p]
    lib.bsig_reads_free.restype = None
    lib.bsig_reads_save.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.bsig_reads_load.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.bsig_last_call_route.restype = C.c_char_p
    lib.bsig_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.POINTER(Params), C.POINTER(C.c_void_p)]
    lib.bsig_plan_offsets.argtypes = [C.c_void_p]
    lib.bsig_plan_offsets.restype = C.POINTER(C.c_int64)
    lib.bsig_plan_cells.argtypes = [C.c_void_p]
    lib.bsig_plan_cells.restype = C.c_int64
    lib.bsig_plan_get_stats.argtypes = [C.c_void_p, C.POINTER(PlanStats)]
    lib.bsig_plan_run.argtypes = [C.c_void_p, C.c_void_p]
    lib.bsig_plan_run_host.argtypes = [C.c_void_p, C.c_void_p]
    lib.bsig_plan_run_host_async.argtypes = [C.c_void_p, C.c_void_p]
    lib.bsig_plan_free.argtypes = [C.c_void_p]
    lib.bsig_plan_free.restype = None
    lib.bsig_pileup_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]
    lib.bsig_bam_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.bsig_bam_close.argtypes = [C.c_void_p]
    lib.bsig_bam_close.restype = None
    lib.bsig_bam_path.argtypes = [C.c_void_p]
    lib.bsig_bam_path.restype = C.c_char_p
    lib.bsig_reads_from_bam.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.bsig_reads_from_bam_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_int32)]
    lib.bsig_reads_from_bam_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.POINTER(C.c_void_p)]
    lib.bsig_device_decode_timing.argtypes = [C.POINTER(C.c_double)]
    lib.bsig_device_decode_timing.restype = None
    lib.bsig_bam_n_ref.argtypes = [C.c_void_p]
    lib.bsig_bam_ref_name.argtypes = [C.c_void_p, C.c_int32]
    lib.bsig_bam_ref_name.restype = C.c_char_p
    lib.bsig_bam_ref_len.argtypes = [C.c_void_p, C.c_int32]
    lib.bsig_bam_name2id.argtypes = [C.c_void_p, C.c_char_p]
    lib.bsig_bam_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.POINTER(Columns)]
    core_head = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p,
                 C.c_void_p, C.c_void_p, C.c_int32]
    lib.bsig_bam_decode_timing.argtypes = [C.POINTER(C.c_double)]
    lib.bsig_bam_decode_timing.restype = None
    lib.bsig_pileup_core.argtypes = core_head + [C.c_int32] * 9 + [C.c_void_p, C.c_void_p]
    lib.bsig_coverage_core.argtypes = core_head + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]
    lib.bsig_pileup_core_into.argtypes = core_head + [C.c_int32] * 9 + [C.c_void_p]
    lib.bsig_coverage_core_into.argtypes = core_head + [C.c_int32] * 6 + [C.c_void_p]
    lib.bsig_write_sam_as_bam_and_index.argtypes = [C.c_char_p, C.c_char_p]
    lib.bsig_write_columns_as_bam.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(Columns),
                                              C.c_int32]
    lib.bsig_write_columns_as_bam_with_seq.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(Columns),
                                                       C.c_int32, C.c_int32, C.c_uint64]
    lib.bsig_cache_clear.restype = None
    lib.bsig_debug_scratch_allocs.restype = C.c_int64
    lib.bsig_effective_cpus.restype = C.c_int32
    lib.bsig_debug_block_table.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    lib.bsig_debug_block_table_progressive.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    lib.bsig_segmap_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.bsig_segmap_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bsig_segmap_free.argtypes = [C.c_void_p]
    lib.bsig_segmap_free.restype = None
    lib.bsig_narrow_bytes.argtypes = [C.c_int64, C.c_int64]
    lib.bsig_narrow_bytes.restype = C.c_int64
    lib.bsig_narrow_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    lib.bsig_narrow_count.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.bsig_segmap_run_narrow.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.bsig_segmap_narrow_overflowed.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.bsig_last_call_timing.argtypes = [C.POINTER(C.c_double)]
    lib.bsig_last_call_timing.restype = None
    lib.bsig_last_call_timing_ex.argtypes = [C.POINTER(C.c_double), C.c_int32]
    lib.bsig_last_call_timing_ex.restype = None
    lib.bsig_check_list.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.bsig_check_list.restype = C.c_int32
    lib.bsig_fast_width.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
    lib.bsig_fast_width.restype = None
    lib.bsig_scatter_segments.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def check(rc):
    if rc != BSIG_OK:
        raise BsigError(rc, load().bsig_last_error().decode("utf-8", "replace"))
