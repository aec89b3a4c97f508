"""ctypes bindings of include/sigax.h.  Fails loudly when the native library is missing."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIGAX_LIB", os.path.join(HERE, "lib", "libsigax.so"))

SIGAX_IRREDUCIBLE = 1
SIGAX_RC = 2
SIGAX_EDGES = 4
SIGAX_DUPLICATE = 8
SIGAX_E_ARG = -1
SIGAX_E_STATE = -6

BLOCK_DTYPE = np.dtype([
    ("capped0_lo", "<u8"), ("capped0_hi", "<u8"), ("capped1_lo", "<u8"), ("capped1_hi", "<u8"),
    ("raw0_lo", "<u8"), ("raw0_hi", "<u8"), ("raw1_lo", "<u8"), ("raw1_hi", "<u8"),
    ("length", "<u4"), ("af", "<u4"), ("reserved", "<u8")])
EDGE_DTYPE = np.dtype([("query", "<u4"), ("target", "<u4"), ("length", "<u4"), ("af", "<u4")])
assert BLOCK_DTYPE.itemsize == 80 and EDGE_DTYPE.itemsize == 16


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_reads", "n_candidate_blocks", "n_blocks", "n_edges", "n_occ_find", "n_occ_extract", "n_substring",
        "n_slow_reads", "n_extract_errors", "n_sectors_find", "n_sectors_extract")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Result(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("block_offs", C.POINTER(C.c_uint64)), ("blocks", C.c_void_p),
                ("substring", C.POINTER(C.c_uint8)), ("n_edges", C.c_uint64), ("edges", C.c_void_p), ("stats", Stats)]


class IndexInfo(C.Structure):
    _fields_ = [("n_symbols", C.c_uint64), ("n_strings", C.c_uint64), ("device_bytes", C.c_uint64),
                ("pred", C.c_uint64 * 5), ("device", C.c_int), ("wide", C.c_int)]


class RunInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_sub", "find_per_sub", "two_step", "coop", "read_order", "cap", "worst_cap", "row_bits",
                                          "row_syms", "row_text", "row_direct", "deep_k")] + \
               [("arena_bytes", C.c_uint64), ("workspace_bytes", C.c_uint64), ("reruns", C.c_uint64), ("order_ms", C.c_float)]

    def as_dict(self):
        return {n: (float(getattr(self, n)) if n == "order_ms" else int(getattr(self, n))) for n, _ in self._fields_}


# every symbol include/sigax.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "sigax_last_error", "sigax_device_count", "sigax_stream_create", "sigax_stream_destroy", "sigax_index_open", "sigax_index_open_mem", "sigax_index_prepare", "sigax_index_prepare_overlap", "sigax_index_clone", "sigax_index_close",
    "sigax_index_info_get", "sigax_index_set_reads", "sigax_index_check_order", "sigax_occ_batch", "sigax_kmer_count_batch",
    "sigax_correct_batch", "sigax_correct_device", "sigax_overlap_batch", "sigax_result_free", "sigax_batch_create", "sigax_batch_destroy", "sigax_batch_upload",
    "sigax_batch_set_device_reads", "sigax_batch_upload_read_ids", "sigax_batch_set_device_read_ids", "sigax_batch_set_subbatches", "sigax_batch_run", "sigax_batch_finish", "sigax_batch_device_outputs",
    "sigax_batch_download", "sigax_batch_download_edges", "sigax_batch_size_hint", "sigax_batch_kernel_ms", "sigax_batch_run_info", "sigax_build_strand", "sigax_build_session", "sigax_free",
    "sigax_comm_unique_id", "sigax_comm_create", "sigax_comm_destroy", "sigax_gather_counts", "sigax_gather_edges",
    "sigax_locality_keys",
]

_lib = None


def lib():
    """Load libsigax.so.  No fallback: a missing library is an error, not a slow path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "siga_amd native library missing: %s (build it with `python -m siga_amd.build` or "
            "__graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, cp, ci = C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_int
    pvp = C.POINTER(C.c_void_p)
    L.sigax_last_error.restype = cp
    L.sigax_device_count.argtypes = [C.POINTER(ci)]
    L.sigax_stream_create.argtypes = [ci, C.POINTER(C.c_void_p)]
    L.sigax_stream_destroy.argtypes = [ci, C.c_void_p]
    L.sigax_stream_destroy.restype = None
    L.sigax_index_open.argtypes = [cp, cp, cp, cp, ci, pvp]
    L.sigax_index_open_mem.argtypes = [vp, u64, vp, u64, u64, u64, vp, vp, ci, pvp]
    L.sigax_index_prepare.argtypes = [vp]
    L.sigax_index_prepare_overlap.argtypes = [vp, u32]
    L.sigax_index_clone.argtypes = [vp, ci, pvp]
    L.sigax_index_close.argtypes = [vp]
    L.sigax_index_close.restype = None
    L.sigax_index_info_get.argtypes = [vp, C.POINTER(IndexInfo)]
    L.sigax_index_set_reads.argtypes = [vp, vp, vp, u64]
    L.sigax_index_check_order.argtypes = [vp, ci, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.sigax_occ_batch.argtypes = [vp, ci, vp, u64, vp]
    L.sigax_kmer_count_batch.argtypes = [vp, cp, u32, u64, vp]
    L.sigax_correct_batch.argtypes = [vp, cp, cp, vp, u32, u32, C.c_int32, u32, u32, vp, vp]
    L.sigax_correct_device.argtypes = [vp, vp, vp, vp, u64, u32, C.c_int32, u32, u32, vp, vp, vp, vp]
    L.sigax_overlap_batch.argtypes = [vp, cp, vp, u32, u32, u32, u32, C.POINTER(Result)]
    L.sigax_result_free.argtypes = [C.POINTER(Result)]
    L.sigax_result_free.restype = None
    L.sigax_batch_create.argtypes = [vp, u32, u64, u32, pvp]
    L.sigax_batch_destroy.argtypes = [vp]
    L.sigax_batch_destroy.restype = None
    L.sigax_batch_upload.argtypes = [vp, cp, vp, u32, vp]
    L.sigax_batch_set_device_reads.argtypes = [vp, vp, vp, u32, u64, u32]
    L.sigax_batch_upload_read_ids.argtypes = [vp, vp, u32, vp]
    L.sigax_locality_keys.argtypes = [C.c_int, vp, vp, u32, vp, vp]
    L.sigax_batch_set_device_read_ids.argtypes = [vp, vp, u32]
    L.sigax_batch_run.argtypes = [vp, u32, u32, u32, vp]
    L.sigax_batch_finish.argtypes = [vp, vp, C.POINTER(Stats)]
    L.sigax_batch_device_outputs.argtypes = [vp, pvp, pvp, pvp, pvp]
    L.sigax_batch_download.argtypes = [vp, C.POINTER(Result)]
    L.sigax_batch_download_edges.argtypes = [vp, vp, pvp, C.POINTER(u64)]
    L.sigax_batch_size_hint.argtypes = [vp, u32, u32, u32, u32, C.POINTER(u32)]
    L.sigax_batch_kernel_ms.argtypes = [vp, C.POINTER(C.c_float * 5), C.POINTER(C.c_uint32)]
    L.sigax_batch_set_subbatches.argtypes = [vp, u32]
    L.sigax_batch_run_info.argtypes = [vp, C.POINTER(RunInfo)]
    L.sigax_build_strand.argtypes = [vp, vp, u64, ci, ci, pvp, C.POINTER(u64), pvp, C.POINTER(u64)]
    L.sigax_free.argtypes = [vp]
    L.sigax_free.restype = None
    L.sigax_build_session.argtypes = [ci]
    L.sigax_build_session.restype = None
    L.sigax_comm_unique_id.argtypes = [vp]
    L.sigax_comm_create.argtypes = [ci, ci, ci, vp, pvp]
    L.sigax_comm_destroy.argtypes = [vp]
    L.sigax_comm_destroy.restype = None
    L.sigax_gather_counts.argtypes = [vp, u64, vp, vp]
    L.sigax_gather_edges.argtypes = [vp, vp, vp, ci, vp, vp]
    _lib = L
    return L


def last_error():
    return lib().sigax_last_error().decode(errors="replace")
