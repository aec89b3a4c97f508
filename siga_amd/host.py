"""ctypes bindings of libsiga_host.so (host C++ mirror of the reference classes: `siga index`, OverlapBuilder::build)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SIGA_HOST_LIB / SIGA_CLI: another build of the host side (tools/sanitize_host.sh runs the CPU tests on ASan/UBSan and TSan builds)
LIB_PATH = os.environ.get("SIGA_HOST_LIB", os.path.join(HERE, "lib", "libsiga_host.so"))
CLI_PATH = os.environ.get("SIGA_CLI", os.path.join(HERE, "lib", "siga"))
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("siga_amd host library missing: %s (python -m siga_amd.build)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.sigah_index_build.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_index_build_dev.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_char_p, C.c_uint64]
        L.sigah_index_file_dev.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_index_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_index_file_sais.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_overlap_file.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_uint64,
                                         C.c_uint64, C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_overlap_file_gpus.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_uint64,
                                              C.c_uint64, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_parse_file.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.sigah_parse_file.restype = C.c_int64
        L.sigah_stem.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        L.sigah_write_file.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64]
        L.sigah_correct_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_int, C.c_char_p, C.c_uint64]
        L.sigah_format_asqg.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
        L.sigah_format_asqg.restype = C.c_int64
        L.sigah_rmdup_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_uint64]
        _lib = L
    return _lib


def index_build(seq_bytes, offs, prefix, threads=2):
    """`siga index` for in-memory reads: writes <prefix>.{bwt,sai,rbwt,rsai}."""
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    err = C.create_string_buffer(512)
    buf = seq_bytes if isinstance(seq_bytes, (bytes, bytearray)) else np.ascontiguousarray(seq_bytes).tobytes()
    if lib().sigah_index_build(buf, offs.ctypes.data, len(offs) - 1, prefix.encode(), threads, err, 512) != 0:
        raise RuntimeError("siga index failed: " + err.value.decode())


def index_build_gpu(seq_bytes, offs, prefix, device=0, threads=2):
    """`siga index` for in-memory reads on the GPU (sigax_build_strand): writes <prefix>.{bwt,sai,rbwt,rsai}."""
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    err = C.create_string_buffer(512)
    if isinstance(seq_bytes, np.ndarray):
        arr = np.ascontiguousarray(seq_bytes)
        ptr = C.c_char_p(arr.ctypes.data)  # no copy: BASELINE-sized sets are gigabytes
    else:
        ptr = seq_bytes
    if lib().sigah_index_build_dev(ptr, offs.ctypes.data, len(offs) - 1, prefix.encode(), device, threads, 1, 1, err, 512) != 0:
        raise RuntimeError("siga index failed: " + err.value.decode())


def index_file_gpu(reads_path, prefix, device=0, threads=2):
    err = C.create_string_buffer(512)
    if lib().sigah_index_file_dev(reads_path.encode(), prefix.encode(), device, threads, 1, 1, err, 512) != 0:
        raise RuntimeError("siga index failed: " + err.value.decode())


def index_file(reads_path, prefix, threads=2):
    err = C.create_string_buffer(512)
    if lib().sigah_index_file(reads_path.encode(), prefix.encode(), threads, err, 512) != 0:
        raise RuntimeError("siga index failed: " + err.value.decode())


def index_file_sais(reads_path, prefix, threads=2):
    """`siga index -a sais`: SAISBuilder's suffix order (every read's own sentinel, ordered by read index), host sorter"""
    err = C.create_string_buffer(512)
    if lib().sigah_index_file_sais(reads_path.encode(), prefix.encode(), threads, 1, 1, err, 512) != 0:
        raise RuntimeError("siga index -a sais failed: " + err.value.decode())


def overlap_file(reads_path, prefix, min_overlap, output, irreducible=True, rc=True, threads=1, batch=10000, device=0, gpus=1):
    """FMIndex::load + OverlapBuilder::build in the host C++ library (GPU compute), reads sharded over `gpus` GPUs."""
    err = C.create_string_buffer(512)
    r = lib().sigah_overlap_file_gpus(reads_path.encode(), prefix.encode(), min_overlap, output.encode(), int(irreducible), int(rc),
                                      threads, batch, device, gpus, err, 512)
    if r != 0:
        raise RuntimeError("siga overlap failed: " + err.value.decode())


def rmdup_file(reads_path, prefix, output, duplicates, device=0):
    """FMIndex::load + OverlapBuilder::rmdup in the host C++ library (GPU compute)."""
    err = C.create_string_buffer(512)
    if lib().sigah_rmdup_file(reads_path.encode(), prefix.encode(), output.encode(), duplicates.encode(), device, err, 512) != 0:
        raise RuntimeError("siga rmdup failed: " + err.value.decode())


def correct_file(reads_path, prefix, output, k=31, threshold=3, rounds=10, offset=1, device=0):
    """FMIndex::load + CorrectProcessor::process (k-mer algorithm) in the host C++ library (GPU compute)."""
    err = C.create_string_buffer(512)
    if lib().sigah_correct_file(reads_path.encode(), prefix.encode(), output.encode(), k, threshold, rounds, offset, device,
                                err, 512) != 0:
        raise RuntimeError("siga correct failed: " + err.value.decode())


def parse_file(path, out_path, parallel=True, threads=4):
    """records of a FASTA/FASTQ file as the host library reads them (parallel loader or record-at-a-time reader)"""
    return lib().sigah_parse_file(path.encode(), 0 if parallel else 1, out_path.encode(), threads)


def read_table(path, out_path, threads=4):
    """the edge converter's read table as the host builds it: one "rank<TAB>length" line per read (name ranks under
    std::string's order, equal names equal rank)"""
    return lib().sigah_parse_file(path.encode(), 2, out_path.encode(), threads)


def write_file(path, data, pieces=1):
    """The host library's output stream (multi-threaded single-member gzip when the name ends with .gz)."""
    if lib().sigah_write_file(path.encode(), data, len(data), pieces) != 0:
        raise IOError("cannot write " + path)


def stem(path):
    out = C.create_string_buffer(1024)
    lib().sigah_stem(path.encode(), out, 1024)
    return out.value.decode()


def format_asqg(reads_path, substring, edges, min_overlap, out_path, threads=4):
    """Test hook: the text side of OverlapBuilder::build without a GPU (loader, VT lines, raw-pointer ED formatter, output
    stream) for given substring flags (uint8[n]) and edge records (EDGE_DTYPE[k]).  Returns the number of reads."""
    sub = np.ascontiguousarray(substring, dtype=np.uint8)
    ed = np.ascontiguousarray(edges)
    assert ed.dtype.itemsize == 16
    n = lib().sigah_format_asqg(reads_path.encode(), sub.ctypes.data, ed.ctypes.data, len(ed), min_overlap, out_path.encode(), threads)
    if n < 0:
        raise RuntimeError("sigah_format_asqg failed")
    return int(n)
