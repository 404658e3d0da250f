"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the conga_amd package.  PARITY UNPINNED (see conga_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class OracleSV(C.Structure):
    """Mirror of `oracle_sv` (conga_oracle.h); 72 bytes."""
    _fields_ = [
        ("start", C.c_int32), ("end", C.c_int32),
        ("observed_rd_sv", C.c_int32), ("expected_rd_sv", C.c_float),
        ("lhomo", C.c_double), ("lhetero", C.c_double), ("lnone", C.c_double),
        ("likelihood_score", C.c_double),
        ("copy_number", C.c_int32), ("rp", C.c_int32), ("border_rp", C.c_int32), ("pad_", C.c_int32),
        ("mappability", C.c_double),
    ]


SV_DTYPE = np.dtype([
    ("start", "<i4"), ("end", "<i4"), ("observed", "<i4"), ("expected", "<f4"),
    ("lhomo", "<f8"), ("lhetero", "<f8"), ("lnone", "<f8"), ("score", "<f8"),
    ("cn", "<i4"), ("rp", "<i4"), ("border_rp", "<i4"), ("pad", "<i4"), ("mappability", "<f8"),
])
assert SV_DTYPE.itemsize == C.sizeof(OracleSV) == 72

SPLIT_ROW_DTYPE = np.dtype([("left_end", "<i4"), ("right_start", "<i4"), ("sv_type", "S1")], align=True)


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("conga_oracle.c", "conga_oracle_sr.c", "conga_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        p = C.c_void_p
        L.oracle_gc.restype = C.c_int
        L.oracle_gc.argtypes = [p, C.c_int64, C.c_int, C.c_int64]
        L.oracle_count_reads.restype = C.c_int64
        L.oracle_count_reads.argtypes = [p, C.c_int64, p, p, C.c_int64, C.c_int]
        L.oracle_calc_mu_per_chr.restype = C.c_float
        L.oracle_calc_mu_per_chr.argtypes = [p, C.c_int64, p]
        L.oracle_calc_mean_per_chr.restype = None
        L.oracle_calc_mean_per_chr.argtypes = [p, C.c_int64, p, C.c_int64, C.c_int, p, p, p]
        L.oracle_paint_mappability.restype = None
        L.oracle_paint_mappability.argtypes = [p, C.c_int64, p, p, p, C.c_int64]
        L.oracle_lpoisson.restype = C.c_double
        L.oracle_lpoisson.argtypes = [C.c_int, C.c_double]
        L.oracle_score.restype = None
        L.oracle_score.argtypes = [C.c_int, C.c_float, C.c_char, p]
        L.oracle_find_depths.restype = None
        L.oracle_find_depths.argtypes = [p, p, C.c_int64, p, C.c_int64, C.c_int, p, C.c_char, p, C.c_int64]
        L.oracle_sort_svs.restype = None
        L.oracle_sort_svs.argtypes = [p, C.c_int64]
        L.oracle_load_known_SVs.restype = C.c_int64
        L.oracle_load_known_SVs.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(p)]
        L.oracle_load_mappability_regions.restype = C.c_int64
        L.oracle_load_mappability_regions.argtypes = [C.c_char_p, C.c_char_p, p, C.c_int64]
        L.oracle_split_read_rows.restype = C.c_int64
        L.oracle_split_read_rows.argtypes = [C.c_char_p, C.c_int64, p, p, C.c_int64, C.c_int64, p, p, p, p, p, p, p,
                                             C.c_int, C.c_int, C.POINTER(p), p]
        L.oracle_count_ReadPairs.restype = None
        L.oracle_count_ReadPairs.argtypes = [p, C.c_int64, p, C.c_int64, p, C.c_int64]
        L.oracle_output_SVs_paths.restype = C.c_int
        L.oracle_output_SVs_paths.argtypes = [C.c_char_p, p, C.c_int64, C.c_int, p, C.c_int64, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_float, C.c_char_p, C.c_char_p,
                                              C.c_char_p, C.c_int, p, p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def count_reads(L, pos, mapq, mq_threshold=-1):
    """-> (read_depth int16[L], counted)"""
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
    rd = np.empty(L, dtype=np.int16)
    n = lib().oracle_count_reads(_ptr(rd), L, _ptr(pos), _ptr(mapq), len(pos), mq_threshold)
    return rd, int(n)


def calc_mu_per_chr(rd):
    cnt = C.c_int64(0)
    mean = lib().oracle_calc_mu_per_chr(_ptr(rd), len(rd), C.byref(cnt))
    return np.float32(mean), int(cnt.value)


def calc_mean_per_chr(rd, gc_hist_w, step=100):
    """-> (E float32[101], S int64[101], W int32[101])"""
    gc = np.ascontiguousarray(gc_hist_w, dtype=np.uint8)
    E = np.zeros(101, dtype=np.float32)
    S = np.zeros(101, dtype=np.int64)
    W = np.zeros(101, dtype=np.int32)
    lib().oracle_calc_mean_per_chr(_ptr(rd), len(rd), _ptr(gc), len(gc), step, _ptr(E), _ptr(S), _ptr(W))
    return E, S, W


def paint_mappability(L, start, end, val):
    start = np.ascontiguousarray(start, dtype=np.int32)
    end = np.ascontiguousarray(end, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    m = np.empty(L, dtype=np.float32)
    lib().oracle_paint_mappability(_ptr(m), L, _ptr(start), _ptr(end), _ptr(val), len(start))
    return m


def lpoisson(observed, lam):
    return float(lib().oracle_lpoisson(int(observed), float(lam)))


def score(observed, expected, sv_type):
    """Scoring only (likelihood.c:131-168) -> numpy record."""
    out = np.zeros(1, dtype=SV_DTYPE)
    lib().oracle_score(int(observed), C.c_float(float(expected)), sv_type.encode()[:1], _ptr(out))
    return out[0]


def make_svs(start, end):
    svs = np.zeros(len(start), dtype=SV_DTYPE)
    svs["start"] = start
    svs["end"] = end
    return svs


def find_depths(rd, mappability, gc_like_w, E, sv_type, svs, step=100):
    """Runs calculate_likelihood_CNV over `svs` (SV_DTYPE array, modified in place and returned)."""
    gc = np.ascontiguousarray(gc_like_w, dtype=np.uint8)
    E = np.ascontiguousarray(E, dtype=np.float32)
    assert svs.dtype == SV_DTYPE and svs.flags.c_contiguous
    lib().oracle_find_depths(_ptr(rd), _ptr(mappability), len(rd), _ptr(gc), len(gc), step, _ptr(E),
                             sv_type.encode()[:1], _ptr(svs), len(svs))
    return svs


def sort_svs(svs):
    lib().oracle_sort_svs(_ptr(svs), len(svs))
    return svs


def load_known_SVs(bed_path, chrom, min_sv_size=1000):
    out = C.c_void_p()
    n = lib().oracle_load_known_SVs(os.fsencode(bed_path), chrom.encode(), min_sv_size, C.byref(out))
    if n < 0:
        raise FileNotFoundError(bed_path)
    if n == 0:
        arr = np.zeros(0, dtype=SV_DTYPE)
    else:
        buf = (C.c_char * (n * SV_DTYPE.itemsize)).from_address(out.value)
        arr = np.frombuffer(buf, dtype=SV_DTYPE).copy()
    C.CDLL(None).free(out)
    return arr


def load_mappability_regions(bed_path, chrom, L):
    m = np.empty(L, dtype=np.float32)
    n = lib().oracle_load_mappability_regions(os.fsencode(bed_path), chrom.encode(), _ptr(m), L)
    if n < 0:
        raise FileNotFoundError(bed_path)
    return m, int(n)


def split_read_rows(ref, sat_start, sat_end, pos, mapq, flag, l_qseq, data_off, seq, qual, mq_threshold=-1,
                    min_read_length=60):
    """One chromosome's split-read rows. ref: bytes (upper-case); seq: uint8 base codes (one per byte);
    -> (rows SPLIT_ROW_DTYPE[], counts int64[4] = elements, mappings, DEL rows, DUP rows)"""
    ss = np.ascontiguousarray(sat_start, dtype=np.int32)
    se = np.ascontiguousarray(sat_end, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
    flag = np.ascontiguousarray(flag, dtype=np.uint16)
    lq = np.ascontiguousarray(l_qseq, dtype=np.int32)
    off = np.ascontiguousarray(data_off, dtype=np.uint64)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    out = C.c_void_p()
    counts = np.zeros(4, dtype=np.int64)
    n = lib().oracle_split_read_rows(ref, len(ref), _ptr(ss), _ptr(se), len(ss), len(pos), _ptr(pos), _ptr(mapq),
                                     _ptr(flag), _ptr(lq), _ptr(off), _ptr(seq), _ptr(qual), mq_threshold,
                                     min_read_length, C.byref(out), _ptr(counts))
    if n > 0:
        buf = (C.c_char * (n * SPLIT_ROW_DTYPE.itemsize)).from_address(out.value)
        rows = np.frombuffer(buf, dtype=SPLIT_ROW_DTYPE).copy()
    else:
        rows = np.zeros(0, dtype=SPLIT_ROW_DTYPE)
    if out.value:
        C.CDLL(None).free(out)
    return rows, counts


def count_read_pairs(rows, dels, dups):
    rows = np.ascontiguousarray(rows, dtype=SPLIT_ROW_DTYPE)
    lib().oracle_count_ReadPairs(_ptr(rows), len(rows), _ptr(dels), len(dels), _ptr(dups), len(dups))


def output_svs(chrom, dels, dups, path_svs, path_del, path_dup, *, have_mappability, no_sr=1,
               rp_support=10, c_score=0.5, write_headers=False):
    """dels / dups: SV_DTYPE arrays or None when the BED was not given. -> (sv_cnt_del, sv_cnt_dup)"""
    nd, nu = C.c_int(0), C.c_int(0)
    rc = lib().oracle_output_SVs_paths(
        chrom.encode(), _ptr(dels), 0 if dels is None else len(dels), int(dels is not None),
        _ptr(dups), 0 if dups is None else len(dups), int(dups is not None),
        int(have_mappability), int(no_sr), int(rp_support), C.c_float(c_score),
        os.fsencode(path_svs), os.fsencode(path_del) if path_del else None,
        os.fsencode(path_dup) if path_dup else None, int(write_headers), C.byref(nd), C.byref(nu))
    if rc != 0:
        raise OSError("oracle_output_SVs_paths failed")
    return nd.value, nu.value
