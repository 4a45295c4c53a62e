"""ctypes binding of libaindex_hip.so (the C ABI in include/aindex_hip.h).

The HIP library is the only compute path: if it cannot be loaded, importing this module raises —
there is no CPU fallback. torch (if importable) is imported first so that the process-wide HIP
runtime is the one PyTorch ships and device pointers of torch tensors are valid inside our kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "lib", "libaindex_hip.so")
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(ROOT, "include", "aindex_hip.h")

AIX_OK = 0
AIX_ERR_CONFLICT = -12
AIX_ERR_ARG = -1
AIX_ERR_FORMAT = -3
FMT_AUTO, FMT_PLAIN, FMT_FASTA, FMT_FASTQ = -1, 0, 1, 2
CANON_NONE, CANON_REF_X86, CANON_TRUE_RC = 0, 1, 2
TOTAL_13MERS = 4 ** 13


class AixError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().aix_strerror(status).decode() if _LIB is not None else str(status)
        super().__init__(f"{what}: {msg} (status {status})" if what else f"{msg} (status {status})")


class Info(C.Structure):
    _fields_ = [("k", C.c_uint32), ("device", C.c_uint32), ("n", C.c_uint64), ("mphf_n", C.c_uint64),
                ("hash_domain", C.c_uint64), ("seed", C.c_uint64), ("bitpairs", C.c_uint64),
                ("device_bytes", C.c_uint64), ("canonical_only", C.c_uint32), ("bucket_table", C.c_uint32),
                ("buckets", C.c_uint64), ("bucket_unfiled_keys", C.c_uint64), ("bucket_lanes", C.c_uint32), ("absence_filter_words", C.c_uint32),
                ("minimizer_lines", C.c_uint64), ("minimizer_unfiled_keys", C.c_uint64), ("count23_backend", C.c_uint32), ("count23_passes", C.c_uint32), ("positions_backend", C.c_uint32), ("reserved0", C.c_uint32)]


class IngestStats(C.Structure):
    _fields_ = [("bytes_in", C.c_uint64), ("plain_bytes", C.c_uint64), ("parts", C.c_uint64), ("part_bytes", C.c_uint64), ("pieces", C.c_uint64),
                ("pinned_bytes", C.c_uint64), ("device_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64), ("seconds_total", C.c_double), ("seconds_read", C.c_double),
                ("seconds_wait", C.c_double), ("seconds_h2d", C.c_double), ("seconds_normalise", C.c_double), ("seconds_compute", C.c_double), ("seconds_output", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_LIB = None
vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int

# name -> (restype, argtypes); every symbol declared in include/aindex_hip.h
SIGNATURES = {
    "aix_version": (C.c_char_p, []),
    "aix_strerror": (C.c_char_p, [i32]),
    "aix_device_count": (i32, [C.POINTER(i32)]),
    "aix_pf_check": (i32, [vp, u64, C.POINTER(u64 * 4)]),
    "aix_index_open_23": (i32, [C.c_char_p, C.c_char_p, C.c_char_p, i32, C.POINTER(vp)]),
    "aix_index_open_13": (i32, [C.c_char_p, C.c_char_p, i32, C.POINTER(vp)]),
    "aix_index_create_23": (i32, [vp, u64, vp, vp, u64, i32, C.POINTER(vp)]),
    "aix_index_create_13": (i32, [vp, u64, vp, i32, C.POINTER(vp)]),
    "aix_index_close": (i32, [vp]),
    "aix_index_info": (i32, [vp, C.POINTER(Info)]),
    "aix_index_set_canonical_fastpath": (i32, [vp, i32]),
    "aix_index_set_fingerprint_filter": (i32, [vp, i32]),
    "aix_index_set_early_exit": (i32, [vp, i32]),
    "aix_index_set_bucket_table": (i32, [vp, i32, i32]),
    "aix_index_set_absence_filter": (i32, [vp, i32]),
    "aix_index_set_minimizer_table": (i32, [vp, i32]),
    "aix_index_set_tf_13": (i32, [vp, vp]),
    "aix_index_get_tf": (i32, [vp, vp, u64]),
    "aix_index_get_checker": (i32, [vp, vp, u64]),
    "aix_tf_batch_ascii": (i32, [vp, vp, u64, vp]),
    "aix_tf_batch_ascii_dev": (i32, [vp, vp, u64, vp, vp]),
    "aix_tf_batch_codes": (i32, [vp, vp, u64, vp]),
    "aix_tf_batch_codes_dev": (i32, [vp, vp, u64, vp, vp]),
    "aix_tf_batch_ragged": (i32, [vp, vp, vp, u64, vp]),
    "aix_tf_batch_ragged_dev": (i32, [vp, vp, vp, u64, vp, vp]),
    "aix_lines_batch_ascii_dev": (i32, [vp, vp, u64, vp, vp]),
    "aix_hash_batch_ascii": (i32, [vp, vp, u64, vp]),
    "aix_hash_batch_ascii_dev": (i32, [vp, vp, u64, vp, vp]),
    "aix_kid_strand_batch_ascii": (i32, [vp, vp, u64, vp, vp]),
    "aix_kid_strand_batch_ascii_dev": (i32, [vp, vp, u64, vp, vp, vp]),
    "aix_tf_both_batch_ascii": (i32, [vp, vp, u64, vp, vp]),
    "aix_tf_both_batch_ascii_dev": (i32, [vp, vp, u64, vp, vp, vp]),
    "aix_tf_total_batch_ascii": (i32, [vp, vp, u64, vp]),
    "aix_tf_total_batch_ascii_dev": (i32, [vp, vp, u64, vp, vp]),
    "aix_coverage_batch": (i32, [vp, vp, vp, u64, u32, vp, vp]),
    "aix_coverage_batch_dev": (i32, [vp, vp, vp, u64, u64, u32, vp, vp, vp]),
    "aix_count13": (i32, [vp, vp, u64, i32, vp]),
    "aix_count13_dev": (i32, [vp, vp, u64, vp, vp]),
    "aix_count23_fixed": (i32, [vp, vp, u64, i32, i32, vp]),
    "aix_count23_fixed_dev": (i32, [vp, vp, u64, i32, vp, vp]),
    "aix_ingest_warm": (i32, [i32]),
    "aix_count13_file": (i32, [vp, C.c_char_p, i32, C.c_char_p, vp, C.POINTER(IngestStats)]),
    "aix_count23_fixed_file": (i32, [vp, C.c_char_p, i32, i32, vp, C.POINTER(IngestStats)]),
    "aix_count_distinct_file": (i32, [C.c_char_p, i32, i32, i32, u64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(IngestStats)]),
    "aix_merge_runs_dev": (i32, [vp, vp, vp, u32, u64, i32, vp, C.POINTER(vp)]),
    "aix_count_distinct": (i32, [vp, u64, i32, i32, i32, u64, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]),
    "aix_positions_fill": (i32, [vp, vp, u64, vp, vp, u64, C.POINTER(u64)]),
    "aix_positions_total": (i32, [vp, C.POINTER(u64)]),
    "aix_positions_fill_dev": (i32, [vp, vp, u64, u64, vp, vp, u64, vp]),
    "aix_positions_bucket_counts": (i32, [vp, vp, u64, i32, vp]),
    "aix_positions_start": (i32, [vp, u64, C.POINTER(u64)]),
    "aix_positions_start_k": (i32, [vp, u64, i32, C.POINTER(u64)]),
    "aix_index_scatter_shard_codes_dev": (i32, [vp, u64, vp, vp, u64, u64, i32, vp, vp, vp, vp]),
    "aix_positions_indices_dev": (i32, [vp, vp, vp]),
    "aix_positions_bucket_counts_dev": (i32, [vp, vp, u64, u64, vp, vp]),
    "aix_positions_fill_shard_dev": (i32, [vp, vp, u64, u64, u64, vp, vp, vp, vp]),
    "aix_positions_fill_shard": (i32, [vp, vp, u64, i32, u64, vp, vp, u64]),
    "aix_window_codes_dev": (i32, [vp, u64, i32, i32, vp, vp]),
    "aix_normalize_reads": (i32, [vp, u64, i32, i32, vp, C.POINTER(u64)]),
    "aix_compute_reads": (i32, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "aix_dat_load": (i32, [C.c_char_p, i32, C.POINTER(u64), C.POINTER(vp), C.POINTER(vp)]),
    "aix_pf_build_file": (i32, [C.c_char_p, C.POINTER(vp), C.POINTER(u64)]),
    "aix_kmers_write_text": (i32, [C.c_char_p, vp, vp, u64, i32]),
    "aix_file_write": (i32, [C.c_char_p, vp, u64]),
    "aix_ridx_load": (i32, [C.c_char_p, C.POINTER(u64), C.POINTER(vp)]),
    "aix_normalize_reads_dev": (i32, [vp, u64, i32, i32, vp, C.POINTER(u64), vp]),
    "aix_detect_format": (i32, [vp, u64]),
    "aix_synth_genome_dev": (i32, [u64, u64, vp, vp]),
    "aix_synth_kmers_dev": (i32, [u64, u64, u64, i32, vp, vp]),
    "aix_synth_mix23_dev": (i32, [u64, vp, u64, u64, u64, vp, vp]),
    "aix_synth_reads_dev": (i32, [u64, vp, u64, u64, u64, u32, i32, u32, vp, vp]),
    "aix_bench_gather_dev": (i32, [vp, u64, i32, i32, u64, u64, vp, vp]),
    "aix_pf_build": (i32, [vp, u64, u32, C.POINTER(vp), C.POINTER(u64)]),
    "aix_pf_build_ragged": (i32, [vp, vp, u64, C.POINTER(vp), C.POINTER(u64)]),
    "aix_pf_build_codes": (i32, [vp, u64, i32, C.POINTER(vp), C.POINTER(u64)]),
    "aix_index_scatter": (i32, [vp, u64, vp, vp, u64, i32, vp, vp]),
    "aix_count_distinct_dev": (i32, [vp, u64, i32, i32, u64, i32, vp, C.POINTER(vp)]),
    "aix_merge_counts_dev": (i32, [vp, vp, u64, u64, i32, vp, C.POINTER(vp)]),
    "aix_distinct_size": (i32, [vp, C.POINTER(u64)]),
    "aix_distinct_copy_dev": (i32, [vp, vp, vp, vp]),
    "aix_distinct_free": (None, [vp]),
    "aix_index_scatter_shard": (i32, [vp, u64, vp, vp, u64, u64, i32, vp, vp, vp]),
    "aix_index_build_23_codes_dev": (i32, [vp, u64, vp, vp, u64, i32, vp, C.POINTER(vp)]),
    "aix_pf_build_codes_dev": (i32, [vp, u64, i32, i32, vp, C.POINTER(vp), C.POINTER(u64)]),
    "aix_pf_build_all_13mers": (i32, [C.POINTER(vp), C.POINTER(u64)]),
    "aix_free": (None, [vp]),
    "aix_scratch_trim": (None, []),
    "aix_host_alloc": (i32, [u64, C.POINTER(vp)]),
    "aix_host_free": (i32, [vp]),
    "aix_debug_relocate_table": (i32, [vp, vp]),
    "aix_debug_relocate_bloom": (i32, [vp, u64]),
    "aix_selftest_lower_bound_dev": (i32, [vp, C.c_uint32, vp, C.c_uint32, vp, vp]),
    "aix_debug_rehome": (i32, [vp, C.c_uint32]),
    "aix_debug_pointers": (i32, [vp, C.POINTER(u64)]),
    "aix_selftest_mod": (u64, [u64, u64]),
    "aix_selftest_revcomp": (u64, [u64, i32]),
}


def header_symbols():
    """Every function name declared in include/aindex_hip.h (used by the CPU export test)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aix_[a-z0-9_]+)\s*\(", text)))


def build(force: bool = False) -> str:
    """Compile libaindex_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + [HEADER]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.environ.get("AIX_NO_TORCH"):                  # the command-line tools set it: they never touch a tensor, and importing torch costs seconds
            try:
                import torch  # noqa: F401  (loads PyTorch's HIP runtime first; see module docstring)
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = ABI drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(status: int, what: str = ""):
    if status != AIX_OK:
        raise AixError(status, what)


def device_count() -> int:
    n = i32(0)
    check(lib().aix_device_count(C.byref(n)), "aix_device_count")
    return n.value
