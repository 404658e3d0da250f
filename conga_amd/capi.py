"""ctypes binding of the C-ABI in include/conga_hip.h (conga_amd/libconga_hip.so).

The shared library is the product; this module only marshals numpy arrays through its plain-C
entry points.  There is no CPU fallback: if the library is missing, or no HIP device is present,
the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONGA_LIB_PATH") or os.path.join(_HERE, "libconga_hip.so")  # override: A/B runs of two builds

CONGA_OK = 0
CONGA_ERR_INVALID = -1
CONGA_ERR_NO_DEVICE = -2
CONGA_ERR_HIP = -3
CONGA_ERR_NOMEM = -4
CONGA_ERR_UNSORTED = -5
CONGA_ERR_RANGE = -6
CONGA_ERR_DATA = -7

FLAG_READS_UNSORTED = 0x1
FLAG_PROFILE = 0x2
FLAG_BATCH = 0x4
FLAG_MATERIALIZE_DEPTH = 0x8
FLAG_RESULTS_ON_DEVICE = 0x10
FLAG_EXPECT_BGZF = 0x20

DELETION = "D"
DUPLICATION = "E"

KERNEL_NAMES = ("ingest", "depth_tile", "expected_table", "paint", "interval_reduce", "interval_score",
                "interval_chain", "interval_count", "split_map", "delta_expand")

# every symbol include/conga_hip.h declares
EXPORTS = (
    "conga_create", "conga_destroy", "conga_strerror", "conga_last_error", "conga_abi_version",
    "conga_device_count", "conga_reset", "conga_chrom_count", "conga_chrom_select", "conga_chrom_begin", "conga_reads_staging", "conga_reads_commit",
    "conga_reads_bgzf", "conga_reads_bgzf_fd", "conga_reads_bgzf_next_fd", "conga_reads_bgzf_next_table", "conga_reads_bgzf_next_blocks", "conga_reads_bgzf_next_go", "conga_reads_bgzf_forget", "conga_release_staging", "conga_inflate_blocks", "conga_host_alloc", "conga_host_free", "conga_sample_reads", "conga_sample_reads_d16", "conga_sample_reads_packed", "conga_packer_create", "conga_packer_destroy", "conga_packer_threads", "conga_pack_bound", "conga_packer_start", "conga_packer_start_v", "conga_packer_finish", "conga_sample_begin",
    "conga_sample_chrom", "conga_sample_fetch", "conga_chrom_compute_ahead", "conga_sample_fetch_previous", "conga_sync_previous",
    "conga_results_copy_previous",
    "conga_mappability", "conga_intervals", "conga_reference", "conga_satellites", "conga_split_reads_staging",
    "conga_split_reads_commit", "conga_split_support", "conga_chrom_compute",
    "conga_chrom_fetch", "conga_chrom_finish", "conga_results_device", "conga_results_copy",
    "conga_set_profile", "conga_stream", "conga_sync",
    "conga_copy_read_depth", "conga_copy_mappability", "conga_host_repeat_add_f32", "conga_host_window_add_f32",
)


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mq_threshold", C.c_int32), ("gc_step", C.c_int32),
                ("flags", C.c_uint32), ("min_read_length", C.c_int32), ("reserved", C.c_int32 * 3)]


class SplitStaging(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_int32)), ("mapq", C.POINTER(C.c_uint8)), ("flag", C.POINTER(C.c_uint16)),
                ("l_qseq", C.POINTER(C.c_int32)), ("data_off", C.POINTER(C.c_uint64)), ("data", C.POINTER(C.c_uint8)),
                ("capacity_reads", C.c_size_t), ("capacity_bytes", C.c_size_t)]


class ReadStaging(C.Structure):
    _fields_ = [("pos", C.POINTER(C.c_int32)), ("mapq", C.POINTER(C.c_uint8)), ("capacity", C.c_size_t)]


class BgzfBlock(C.Structure):
    """conga_bgzf_block (conga_reads_bgzf)."""
    _fields_ = [("data_off", C.c_uint64), ("data_len", C.c_uint32), ("inflated_len", C.c_uint32), ("crc32", C.c_uint32),
                ("reserved", C.c_uint32)]


class BamSegment(C.Structure):
    """conga_bam_segment (conga_reads_bgzf)."""
    _fields_ = [("start", C.c_uint64), ("pos_lo", C.c_int32), ("pos_hi", C.c_int32), ("ref_id", C.c_int32), ("chrom", C.c_int32)]


class ChromStats(C.Structure):
    _fields_ = [("reads_committed", C.c_int64), ("reads_counted", C.c_int64),
                ("reads_out_of_range", C.c_int64), ("rd_sum", C.c_int64), ("mean", C.c_float),
                ("n_kernels", C.c_int32), ("rd_per_gc", C.c_int64 * 101), ("window_per_gc", C.c_int64 * 101),
                ("kernel_ms", C.c_double * 12), ("split_elements", C.c_int64), ("split_mappings", C.c_int64),
                ("split_del_rows", C.c_int64), ("split_dup_rows", C.c_int64),
                ("depth_materialized", C.c_int32), ("reserved", C.c_int32)]


RESULT_DTYPE = np.dtype([
    ("observed", "<i4"), ("expected", "<f4"), ("lhomo", "<f8"), ("lhetero", "<f8"), ("lnone", "<f8"),
    ("score", "<f8"), ("cn", "<i4"), ("rp", "<i4"), ("border_rp", "<i4"), ("reserved", "<i4"),
    ("mappability", "<f8"),
])
assert RESULT_DTYPE.itemsize == 64


class CongaError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("conga status %d: %s" % (status, msg))
        self.status = status


_lib = None
# OR-ed into the flags of every Context (tests run each parity case under both formulations with it)
EXTRA_FLAGS = 0


def encode_d16(pos, chrom_off):
    """What a producer that subtracts sends (conga_sample_reads_d16): pos int32[n] with chromosome c in [chrom_off[c], chrom_off[c + 1])
    -> (delta uint16[n], esc_index uint32[k], esc_pos int32[k]).  A difference outside [0, 0xFFFE] and the first read of every
    chromosome are exceptions (0xFFFF in the stream, index and position in the list)."""
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    n = len(pos)
    d = np.empty(n, np.int64)
    if n:
        d[0] = -1
        d[1:] = pos[1:].astype(np.int64) - pos[:-1].astype(np.int64)
        firsts = np.asarray(chrom_off[:-1], np.int64)
        firsts = firsts[(firsts < n) & (np.asarray(chrom_off[1:], np.int64) > firsts)]
        d[firsts] = -1
    esc = (d < 0) | (d > 0xFFFE)
    delta = np.where(esc, 0xFFFF, d).astype(np.uint16)
    idx = np.flatnonzero(esc).astype(np.uint32)
    return delta, idx, pos[idx].astype(np.int32)


def encode_packed(pos, chrom_off, width=None):
    """conga_sample_reads_packed's input: pos int32[n] -> (bits uint8[], width, esc_index uint32[k], esc_pos int32[k]).  width: 4 .. 16;
    None picks the one that sends the fewest bytes (differences + 8 bytes per exception) among those with at most one exception in
    a thousand reads: an exception costs the expansion a search of the list (at 1x, 9 bits are 5 % fewer bytes than 10 and a
    slower step -- 0.6 % of the reads are exceptions and the expansion no longer hides under the copy; profiles/README.md, r03h)."""
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    n = len(pos)
    d = np.full(n, -1, np.int64)
    if n:
        d[1:] = pos[1:].astype(np.int64) - pos[:-1].astype(np.int64)
        firsts = np.asarray(chrom_off[:-1], np.int64)
        firsts = firsts[(firsts < n) & (np.asarray(chrom_off[1:], np.int64) > firsts)]
        d[firsts] = -1
    if width is None:
        n_exc = {w: int(np.count_nonzero((d < 0) | (d >= (1 << w) - 1))) for w in range(4, 17)}
        cost = {w: n * w / 8 + 8 * n_exc[w] for w in range(4, 17) if n_exc[w] <= max(n // 1000, 64) or w == 16}
        width = min(cost, key=cost.get)
    top = (1 << width) - 1
    esc = (d < 0) | (d >= top)
    v = np.where(esc, top, d).astype("<u2")
    idx = np.flatnonzero(esc).astype(np.uint32)
    if width == 16:
        bits = v.view(np.uint8).copy()
    else:
        pad = (-n) % 8
        v = np.concatenate([v, np.zeros(pad, "<u2")])
        b = np.unpackbits(v.view(np.uint8).reshape(-1, 2), axis=1, bitorder="little")[:, :width]
        bits = np.packbits(b.reshape(-1), bitorder="little")
    return bits, width, idx, pos[idx].astype(np.int32)


def pack_inline(bits, esc_index, esc_pos):
    """[differences | pad to 16 bytes | esc_index | esc_pos] as one uint8 array (conga_sample_reads_packed with esc_index == NULL)."""
    at = (len(bits) + 15) & ~15
    out = np.zeros(at + 8 * len(esc_index) + 64, np.uint8)
    out[:len(bits)] = bits
    out[at:at + 4 * len(esc_index)] = esc_index.view(np.uint8)
    out[at + 4 * len(esc_index):at + 8 * len(esc_index)] = esc_pos.view(np.uint8)
    return out


class Packer:
    """conga_packer_*: the library's own producer of the packed hand-over (host threads, no device).  start() returns at once;
    finish() -> (width, n_esc, out_bytes); the bytes are in the `out` array start() was given (the one-copy layout)."""

    def __init__(self, n_threads=0):
        self._lib = load()
        self._h = self._lib.conga_packer_create(int(n_threads))
        if not self._h:
            raise MemoryError("conga_packer_create")
        self._keep = None

    def start(self, pos, chrom_off, out, width=0):
        if pos.dtype != np.int32 or chrom_off.dtype != np.uint64 or out.dtype != np.uint8:
            raise TypeError("Packer.start takes int32 pos, uint64 chrom_off, uint8 out")
        self._keep = (pos, chrom_off, out)
        rc = self._lib.conga_packer_start(self._h, pos.ctypes.data, chrom_off.ctypes.data, len(chrom_off) - 1, int(width), out.ctypes.data, out.nbytes)
        if rc != CONGA_OK:
            raise CongaError(rc, "conga_packer_start")

    def start_v(self, chrom_pos, out, width=0):
        """conga_packer_start_v: one int32 array per chromosome (what a decoder that works chromosome by chromosome leaves)."""
        if any(a.dtype != np.int32 or not a.flags.c_contiguous for a in chrom_pos) or out.dtype != np.uint8:
            raise TypeError("Packer.start_v takes contiguous int32 arrays and a uint8 out")
        off = np.zeros(len(chrom_pos) + 1, np.uint64)
        off[1:] = np.cumsum([len(a) for a in chrom_pos], dtype=np.uint64)
        ptrs = (C.c_void_p * max(len(chrom_pos), 1))(*[a.ctypes.data if len(a) else None for a in chrom_pos])
        self._keep = (chrom_pos, off, ptrs, out)
        rc = self._lib.conga_packer_start_v(self._h, C.cast(ptrs, C.c_void_p), off.ctypes.data, len(chrom_pos), int(width), out.ctypes.data, out.nbytes)
        if rc != CONGA_OK:
            raise CongaError(rc, "conga_packer_start_v")
        return off

    def finish(self):
        w, ne, nb = C.c_int(0), C.c_size_t(0), C.c_size_t(0)
        rc = self._lib.conga_packer_finish(self._h, C.byref(w), C.byref(ne), C.byref(nb))
        self._keep = None
        if rc != CONGA_OK:
            raise CongaError(rc, "conga_packer_finish")
        return w.value, ne.value, nb.value

    def threads(self):
        return int(self._lib.conga_packer_threads(self._h))

    def bound(self, n_reads, max_esc):
        return int(self._lib.conga_pack_bound(int(n_reads), int(max_esc)))

    def close(self):
        if self._h:
            self._lib.conga_packer_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def load():
    """dlopen the HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(conga_amd has no CPU path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, sz = C.c_void_p, C.c_int64, C.c_size_t
    L.conga_create.restype = vp
    L.conga_create.argtypes = [C.c_int, C.POINTER(Opts), C.POINTER(C.c_int)]
    L.conga_destroy.restype = None
    L.conga_destroy.argtypes = [vp]
    L.conga_strerror.restype = C.c_char_p
    L.conga_strerror.argtypes = [C.c_int]
    L.conga_last_error.restype = C.c_char_p
    L.conga_last_error.argtypes = [vp]
    L.conga_abi_version.restype = C.c_int
    L.conga_device_count.restype = C.c_int
    L.conga_reset.restype = C.c_int
    L.conga_reset.argtypes = [vp]
    L.conga_chrom_count.restype = C.c_int
    L.conga_chrom_count.argtypes = [vp]
    L.conga_chrom_select.restype = C.c_int
    L.conga_chrom_select.argtypes = [vp, C.c_int]
    L.conga_chrom_begin.restype = C.c_int
    L.conga_chrom_begin.argtypes = [vp, i64, vp, vp, i64]
    L.conga_reads_staging.restype = C.c_int
    L.conga_reads_staging.argtypes = [vp, C.POINTER(ReadStaging)]
    L.conga_reads_commit.restype = C.c_int
    L.conga_reads_commit.argtypes = [vp, sz]
    L.conga_reads_bgzf.restype = C.c_int
    L.conga_reads_bgzf.argtypes = [vp, vp, sz, C.POINTER(BgzfBlock), sz, C.POINTER(BamSegment), sz, C.POINTER(C.c_uint64)]
    L.conga_reads_bgzf_fd.restype = C.c_int
    L.conga_reads_bgzf_fd.argtypes = [vp, C.c_int, C.c_uint64, sz, C.POINTER(BgzfBlock), sz, C.POINTER(BamSegment), sz, C.POINTER(C.c_uint64)]
    L.conga_reads_bgzf_next_fd.restype = C.c_int
    L.conga_reads_bgzf_next_fd.argtypes = [vp, C.c_int, C.c_uint64, sz, C.POINTER(C.c_uint64), sz, C.c_uint64, C.POINTER(C.c_uint64)]
    L.conga_reads_bgzf_next_table.restype = C.c_int
    L.conga_reads_bgzf_next_table.argtypes = [vp, C.c_uint64, C.POINTER(C.POINTER(BgzfBlock)), C.POINTER(sz)]
    L.conga_reads_bgzf_next_blocks.restype = C.c_int
    L.conga_reads_bgzf_next_blocks.argtypes = [vp, C.c_uint64, C.POINTER(BgzfBlock), sz]
    L.conga_reads_bgzf_next_go.restype = C.c_int
    L.conga_reads_bgzf_next_go.argtypes = [vp, C.c_uint64]
    L.conga_reads_bgzf_forget.restype = C.c_int
    L.conga_reads_bgzf_forget.argtypes = [vp, C.c_uint64]
    L.conga_release_staging.restype = C.c_int
    L.conga_release_staging.argtypes = [vp]
    L.conga_inflate_blocks.restype = C.c_int
    L.conga_inflate_blocks.argtypes = [vp, vp, sz, C.POINTER(BgzfBlock), sz, vp, sz, vp, C.POINTER(C.c_double)]
    L.conga_host_alloc.restype = vp
    L.conga_host_alloc.argtypes = [vp, sz]
    L.conga_host_free.restype = None
    L.conga_host_free.argtypes = [vp, vp]
    L.conga_sample_reads.restype = C.c_int
    L.conga_sample_reads.argtypes = [vp, vp, vp, vp, C.c_int]
    L.conga_sample_reads_d16.restype = C.c_int
    L.conga_sample_reads_d16.argtypes = [vp, vp, vp, vp, sz, vp, vp, C.c_int]
    L.conga_sample_reads_packed.restype = C.c_int
    L.conga_sample_reads_packed.argtypes = [vp, vp, C.c_int, vp, vp, sz, vp, vp, C.c_int]
    L.conga_packer_create.restype = vp
    L.conga_packer_create.argtypes = [C.c_int]
    L.conga_packer_destroy.restype = None
    L.conga_packer_destroy.argtypes = [vp]
    L.conga_packer_threads.restype = C.c_int
    L.conga_packer_threads.argtypes = [vp]
    L.conga_pack_bound.restype = sz
    L.conga_pack_bound.argtypes = [C.c_uint64, sz]
    L.conga_packer_start.restype = C.c_int
    L.conga_packer_start.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, sz]
    L.conga_packer_start_v.restype = C.c_int
    L.conga_packer_start_v.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, sz]
    L.conga_packer_finish.restype = C.c_int
    L.conga_packer_finish.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(sz), C.POINTER(sz)]
    L.conga_sample_begin.restype = C.c_int
    L.conga_sample_begin.argtypes = [vp]
    L.conga_sample_chrom.restype = C.c_int
    L.conga_sample_chrom.argtypes = [vp, C.c_int]
    L.conga_sample_fetch.restype = C.c_int
    L.conga_sample_fetch.argtypes = [vp, vp, sz, vp, C.POINTER(ChromStats)]
    L.conga_sample_fetch_previous.restype = C.c_int
    L.conga_sample_fetch_previous.argtypes = [vp, vp, sz, vp, C.POINTER(ChromStats)]
    L.conga_chrom_compute_ahead.restype = C.c_int
    L.conga_chrom_compute_ahead.argtypes = [vp]
    L.conga_sync_previous.restype = C.c_int
    L.conga_sync_previous.argtypes = [vp]
    L.conga_results_copy_previous.restype = C.c_int
    L.conga_results_copy_previous.argtypes = [vp, vp, sz]
    L.conga_mappability.restype = C.c_int
    L.conga_mappability.argtypes = [vp, vp, vp, vp, sz]
    L.conga_intervals.restype = C.c_int
    L.conga_intervals.argtypes = [vp, C.c_char, vp, vp, sz]
    L.conga_reference.restype = C.c_int
    L.conga_reference.argtypes = [vp, C.c_char_p, i64]
    L.conga_satellites.restype = C.c_int
    L.conga_satellites.argtypes = [vp, vp, vp, sz]
    L.conga_split_reads_staging.restype = C.c_int
    L.conga_split_reads_staging.argtypes = [vp, C.POINTER(SplitStaging)]
    L.conga_split_reads_commit.restype = C.c_int
    L.conga_split_reads_commit.argtypes = [vp, sz, sz]
    L.conga_split_support.restype = C.c_int
    L.conga_split_support.argtypes = [vp, C.c_char, vp, sz]
    L.conga_chrom_compute.restype = C.c_int
    L.conga_chrom_compute.argtypes = [vp]
    L.conga_chrom_fetch.restype = C.c_int
    L.conga_chrom_fetch.argtypes = [vp, vp, vp, vp, C.POINTER(ChromStats)]
    L.conga_chrom_finish.restype = C.c_int
    L.conga_chrom_finish.argtypes = [vp, vp, vp, vp, C.POINTER(ChromStats)]
    L.conga_results_device.restype = C.c_int
    L.conga_results_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.conga_results_copy.restype = C.c_int
    L.conga_results_copy.argtypes = [vp, vp, sz]
    L.conga_set_profile.restype = C.c_int
    L.conga_set_profile.argtypes = [vp, C.c_int]
    L.conga_stream.restype = vp
    L.conga_stream.argtypes = [vp]
    L.conga_sync.restype = C.c_int
    L.conga_sync.argtypes = [vp]
    L.conga_copy_read_depth.restype = C.c_int
    L.conga_copy_read_depth.argtypes = [vp, vp, i64]
    L.conga_copy_mappability.restype = C.c_int
    L.conga_copy_mappability.argtypes = [vp, vp, i64]
    L.conga_host_repeat_add_f32.restype = C.c_float
    L.conga_host_repeat_add_f32.argtypes = [C.c_float, C.c_float, C.c_uint32]
    L.conga_host_window_add_f32.restype = C.c_float
    L.conga_host_window_add_f32.argtypes = [C.c_float, C.c_float, C.c_uint32]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One conga_ctx: one GPU, one stream, one chromosome in flight."""

    def __init__(self, device=0, mq_threshold=-1, gc_step=100, flags=0, min_read_length=0):
        self._lib = load()
        flags |= EXTRA_FLAGS
        opts = Opts(C.sizeof(Opts), mq_threshold, gc_step, flags, min_read_length)
        st = C.c_int(0)
        self._h = self._lib.conga_create(device, C.byref(opts), C.byref(st))
        if not self._h:
            raise CongaError(st.value, self._lib.conga_strerror(st.value).decode())
        self.gc_step = gc_step
        self.batch = bool(flags & FLAG_BATCH)
        self._meta = []   # per chromosome: [length, n_dels, n_dups]
        self._cur = -1
        self._pinned = []

    def close(self):
        if getattr(self, "_h", None):
            self.host_free_all()
            self._lib.conga_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != CONGA_OK:
            raise CongaError(rc, "%s: %s" % (self._lib.conga_strerror(rc).decode(),
                                             self._lib.conga_last_error(self._h).decode()))

    # -- one chromosome ------------------------------------------------------------------
    def chrom_begin(self, chrom_len, gc_hist_w, gc_like_w=None):
        gh = np.ascontiguousarray(gc_hist_w, dtype=np.uint8)
        gl = gh if gc_like_w is None else np.ascontiguousarray(gc_like_w, dtype=np.uint8)
        if len(gl) != len(gh):
            raise ValueError("gc_hist_w and gc_like_w differ in length")
        self._check(self._lib.conga_chrom_begin(self._h, chrom_len, _p(gh), _p(gl), len(gh)))
        if not self.batch:
            self._meta = []
        self._meta.append([chrom_len, 0, 0])
        self._cur = len(self._meta) - 1
        return self._cur

    def reset(self):
        self._check(self._lib.conga_reset(self._h))
        self._meta, self._cur = [], -1

    def select(self, index):
        self._check(self._lib.conga_chrom_select(self._h, index))
        self._cur = index

    def chrom_count(self):
        return self._lib.conga_chrom_count(self._h)

    @property
    def chrom_len(self):
        return self._meta[self._cur][0]

    @property
    def n_dels(self):
        return self._meta[self._cur][1]

    @property
    def n_dups(self):
        return self._meta[self._cur][2]

    @property
    def n_records(self):
        return sum(m[1] + m[2] for m in self._meta)

    def reads(self, pos, mapq):
        """Streams (pos, mapq) through the pinned staging ring, like the BAM loop would."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        if len(pos) != len(mapq):
            raise ValueError("pos and mapq differ in length")
        n, off = len(pos), 0
        stg = ReadStaging()
        while off < n or (n == 0 and off == 0):
            self._check(self._lib.conga_reads_staging(self._h, C.byref(stg)))
            k = min(stg.capacity, n - off)
            if k:
                C.memmove(stg.pos, pos[off:].ctypes.data, k * 4)
                C.memmove(stg.mapq, mapq[off:].ctypes.data, k)
            self._check(self._lib.conga_reads_commit(self._h, k))
            off += k
            if n == 0:
                break

    # -- cohort mode: the next sample's reads behind the same layout ----------------------------
    def host_alloc(self, n, dtype):
        """A pinned host array (conga_host_alloc) as numpy; freed with the context (or by host_free)."""
        dt = np.dtype(dtype)
        nbytes = max(int(n) * dt.itemsize, 1)
        p = self._lib.conga_host_alloc(self._h, nbytes)
        if not p:
            raise CongaError(CONGA_ERR_NOMEM, self._lib.conga_last_error(self._h).decode())
        self._pinned.append(p)
        buf = (C.c_uint8 * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dt, count=int(n))

    def host_free_all(self):
        for p in self._pinned:
            self._lib.conga_host_free(self._h, p)
        self._pinned = []

    def sample_reads(self, pos, mapq, chrom_off):
        """Replaces the reads of every chromosome; pos / mapq / chrom_off are passed as they are (no copy: the caller
        keeps them alive and unchanged until the next fetch / sync).  Pinned arrays (host_alloc) go at PCIe rate."""
        if pos.dtype != np.int32 or (mapq is not None and mapq.dtype != np.uint8) or chrom_off.dtype != np.uint64:
            raise TypeError("sample_reads takes int32 pos, uint8 mapq (or None when mq_threshold < 0), uint64 chrom_off")
        self._check(self._lib.conga_sample_reads(self._h, pos.ctypes.data, None if mapq is None else mapq.ctypes.data,
                                                 chrom_off.ctypes.data, len(chrom_off) - 1))

    def sample_reads_d16(self, delta, esc_index, esc_pos, mapq, chrom_off):
        """conga_sample_reads_d16: the positions as 16-bit differences + exceptions (encode_d16); arrays passed as they are."""
        if delta.dtype != np.uint16 or esc_index.dtype != np.uint32 or esc_pos.dtype != np.int32 or chrom_off.dtype != np.uint64 \
                or (mapq is not None and mapq.dtype != np.uint8):
            raise TypeError("sample_reads_d16 takes uint16 delta, uint32 esc_index, int32 esc_pos, uint8 mapq (or None), uint64 chrom_off")
        self._check(self._lib.conga_sample_reads_d16(self._h, delta.ctypes.data, esc_index.ctypes.data, esc_pos.ctypes.data, len(esc_index),
                                                     None if mapq is None else mapq.ctypes.data, chrom_off.ctypes.data, len(chrom_off) - 1))

    def sample_reads_packed(self, bits, width, esc_index, esc_pos, mapq, chrom_off):
        """conga_sample_reads_packed: the positions as `width`-bit differences + exceptions (encode_packed); arrays passed as they are."""
        if isinstance(esc_index, int):   # the exceptions lie behind the differences in `bits` (pack_inline): one copy per sample
            self._check(self._lib.conga_sample_reads_packed(self._h, bits.ctypes.data, width, None, None, esc_index,
                                                            None if mapq is None else mapq.ctypes.data, chrom_off.ctypes.data, len(chrom_off) - 1))
            return
        if bits.dtype != np.uint8 or esc_index.dtype != np.uint32 or esc_pos.dtype != np.int32 or chrom_off.dtype != np.uint64 \
                or (mapq is not None and mapq.dtype != np.uint8):
            raise TypeError("sample_reads_packed takes uint8 bits, uint32 esc_index, int32 esc_pos, uint8 mapq (or None), uint64 chrom_off")
        self._check(self._lib.conga_sample_reads_packed(self._h, bits.ctypes.data, width, esc_index.ctypes.data, esc_pos.ctypes.data, len(esc_index),
                                                        None if mapq is None else mapq.ctypes.data, chrom_off.ctypes.data, len(chrom_off) - 1))

    def sample_begin(self):
        self._check(self._lib.conga_sample_begin(self._h))

    def sample_chrom(self, index):
        self._check(self._lib.conga_sample_chrom(self._h, index))

    def sample_fetch(self, records=None, expected=None, want_stats=False):
        """Every chromosome's records in one call -> (records RESULT_DTYPE[n_records], E float32[n_chrom, 101], stats|None)."""
        n, nc = self.n_records, len(self._meta)
        if records is None:
            records = np.zeros(n, dtype=RESULT_DTYPE)
        if expected is None:
            expected = np.zeros((nc, 101), dtype=np.float32)
        st = (ChromStats * nc)() if want_stats else None
        self._check(self._lib.conga_sample_fetch(self._h, records.ctypes.data if n else None, n, expected.ctypes.data, st))
        return records, expected, st

    def sample_fetch_previous(self, records=None, expected=None, want_stats=False):
        """conga_sample_fetch_previous: the same for the compute BEFORE the latest one (compute_ahead)."""
        n, nc = self.n_records, len(self._meta)
        if records is None:
            records = np.zeros(n, dtype=RESULT_DTYPE)
        if expected is None:
            expected = np.zeros((nc, 101), dtype=np.float32)
        st = (ChromStats * nc)() if want_stats else None
        self._check(self._lib.conga_sample_fetch_previous(self._h, records.ctypes.data if n else None, n, expected.ctypes.data, st))
        return records, expected, st

    def release_staging(self):
        """conga_release_staging: the pinned ring of conga_reads_bgzf back to the system (made again when needed)."""
        self._check(self._lib.conga_release_staging(self._h))

    def reads_bgzf(self, data, blocks, segments):
        """conga_reads_bgzf: a stretch of a BAM file as it is on disk + its block table + start points.
        blocks: [(data_off, data_len, inflated_len, crc32)], segments: [(start, pos_lo, pos_hi, ref_id, chrom)].
        -> reads per chromosome"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        bl = (BgzfBlock * len(blocks))(*[BgzfBlock(*b, 0) for b in blocks])
        sg = (BamSegment * len(segments))(*[BamSegment(*x) for x in segments])
        per = (C.c_uint64 * max(self.chrom_count(), 1))()
        self._check(self._lib.conga_reads_bgzf(self._h, data.ctypes.data, len(data), bl, len(blocks), sg, len(segments), per))
        return list(per)[:self.chrom_count()]

    def reads_bgzf_fd(self, fd, file_off, n_bytes, blocks, segments):
        """conga_reads_bgzf_fd: the same with the bytes still in the file (read with pread into the pinned pieces that go up)."""
        bl = (BgzfBlock * len(blocks))(*[BgzfBlock(*b, 0) for b in blocks])
        sg = (BamSegment * len(segments))(*[BamSegment(*x) for x in segments])
        per = (C.c_uint64 * max(self.chrom_count(), 1))()
        self._check(self._lib.conga_reads_bgzf_fd(self._h, int(fd), int(file_off), int(n_bytes), bl, len(blocks), sg, len(segments), per))
        return list(per)[:self.chrom_count()]

    def reads_bgzf_next_fd(self, fd, file_off, n_bytes, known_starts=(), stop_at=0):
        """conga_reads_bgzf_next_fd -> ticket (0: nothing was started)"""
        ks = (C.c_uint64 * max(len(known_starts), 1))(*known_starts)
        ticket = C.c_uint64(0)
        self._check(self._lib.conga_reads_bgzf_next_fd(self._h, int(fd), int(file_off), int(n_bytes), ks if len(known_starts) else None,
                                                       len(known_starts), int(stop_at), C.byref(ticket)))
        return ticket.value

    def reads_bgzf_next_blocks(self, ticket, blocks):
        bl = (BgzfBlock * len(blocks))(*[BgzfBlock(*b, 0) for b in blocks])
        self._check(self._lib.conga_reads_bgzf_next_blocks(self._h, int(ticket), bl, len(blocks)))

    def reads_bgzf_next_table(self, ticket):
        """conga_reads_bgzf_next_table -> [(data_off, data_len, inflated_len, crc32)] (empty: no table)"""
        b, n = C.POINTER(BgzfBlock)(), C.c_size_t(0)
        self._check(self._lib.conga_reads_bgzf_next_table(self._h, int(ticket), C.byref(b), C.byref(n)))
        return [(b[i].data_off, b[i].data_len, b[i].inflated_len, b[i].crc32) for i in range(n.value)]

    def reads_bgzf_next_go(self, ticket):
        self._check(self._lib.conga_reads_bgzf_next_go(self._h, int(ticket)))

    def reads_bgzf_forget(self, ticket):
        self._check(self._lib.conga_reads_bgzf_forget(self._h, int(ticket)))

    def inflate_blocks(self, data, blocks, want_out=True):
        """conga_inflate_blocks -> (payloads uint8[], status uint8[n_blocks], kernel_ms)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        bl = (BgzfBlock * len(blocks))(*[BgzfBlock(*b, 0) for b in blocks])
        total = sum(b[2] for b in blocks)
        out = np.zeros(total if want_out else 0, np.uint8)
        status = np.zeros(len(blocks), np.uint8)
        ms = C.c_double(0)
        self._check(self._lib.conga_inflate_blocks(self._h, data.ctypes.data, len(data), bl, len(blocks),
                                                   out.ctypes.data if want_out else None, len(out), status.ctypes.data, C.byref(ms)))
        return out, status, ms.value

    def mappability(self, start, end, val):
        s = np.ascontiguousarray(start, dtype=np.int32)
        e = np.ascontiguousarray(end, dtype=np.int32)
        v = np.ascontiguousarray(val, dtype=np.float32)
        self._check(self._lib.conga_mappability(self._h, _p(s), _p(e), _p(v), len(s)))

    def intervals(self, sv_type, start, end):
        s = np.ascontiguousarray(start, dtype=np.int32)
        e = np.ascontiguousarray(end, dtype=np.int32)
        self._check(self._lib.conga_intervals(self._h, sv_type.encode()[:1], _p(s), _p(e), len(s)))
        self._meta[self._cur][1 if sv_type == DELETION else 2] = len(s)

    def reference(self, seq):
        """Chromosome sequence (bytes) for the split-read path."""
        self._check(self._lib.conga_reference(self._h, seq, len(seq)))

    def satellites(self, start, end):
        s = np.ascontiguousarray(start, dtype=np.int32)
        e = np.ascontiguousarray(end, dtype=np.int32)
        self._check(self._lib.conga_satellites(self._h, _p(s), _p(e), len(s)))

    def split_reads(self, pos, mapq, flag, l_qseq, seq_codes, qual, code_off):
        """Every record of the BAM loop: seq_codes holds one 4-bit base code per byte and qual the Phred
        bytes, both starting at code_off[i]; they are packed here into BAM's nibble layout."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        flag = np.ascontiguousarray(flag, dtype=np.uint16)
        lq = np.ascontiguousarray(l_qseq, dtype=np.int32)
        stg = SplitStaging()
        i, n = 0, len(pos)
        while i < n:
            self._check(self._lib.conga_split_reads_staging(self._h, C.byref(stg)))
            data = np.ctypeslib.as_array(stg.data, shape=(stg.capacity_bytes,))
            k, nb = 0, 0
            while i + k < n and k < stg.capacity_reads:
                l = int(lq[i + k])
                need = (l + 1) // 2 + l
                if nb + need > stg.capacity_bytes:
                    break
                o = int(code_off[i + k])
                codes = np.zeros((l + 1) // 2 * 2, np.uint8)
                codes[:l] = seq_codes[o:o + l]
                data[nb:nb + (l + 1) // 2] = (codes[0::2] << 4) | codes[1::2]
                data[nb + (l + 1) // 2:nb + need] = qual[o:o + l]
                stg.data_off[k] = nb
                stg.pos[k], stg.mapq[k], stg.flag[k], stg.l_qseq[k] = int(pos[i + k]), int(mapq[i + k]), int(flag[i + k]), l
                nb += need
                k += 1
            self._check(self._lib.conga_split_reads_commit(self._h, k, nb))
            i += k

    def split_support(self, sv_type, support):
        s = np.ascontiguousarray(support, dtype=np.int32)
        self._check(self._lib.conga_split_support(self._h, sv_type.encode()[:1], _p(s), len(s)))

    def compute(self):
        self._check(self._lib.conga_chrom_compute(self._h))

    def compute_ahead(self):
        """conga_chrom_compute_ahead: behind the last compute, whose results stay fetchable (sample_fetch_previous, sync_previous,
        results_copy_previous)."""
        self._check(self._lib.conga_chrom_compute_ahead(self._h))

    def sync_previous(self):
        self._check(self._lib.conga_sync_previous(self._h))

    def results_copy_previous(self, dst_ptr, dst_bytes):
        self._check(self._lib.conga_results_copy_previous(self._h, C.c_void_p(dst_ptr), dst_bytes))

    def fetch(self):
        """-> (dels RESULT_DTYPE[n_dels], dups RESULT_DTYPE[n_dups], E float32[101], ChromStats)"""
        dels = np.zeros(self.n_dels, dtype=RESULT_DTYPE)
        dups = np.zeros(self.n_dups, dtype=RESULT_DTYPE)
        E = np.zeros(101, dtype=np.float32)
        st = ChromStats()
        self._check(self._lib.conga_chrom_fetch(self._h, _p(dels) if self.n_dels else None,
                                                _p(dups) if self.n_dups else None, _p(E), C.byref(st)))
        return dels, dups, E, st

    def finish(self):
        self.compute()
        return self.fetch()

    def sync(self):
        self._check(self._lib.conga_sync(self._h))

    def stream(self):
        return self._lib.conga_stream(self._h)

    def results_device(self):
        """-> (device pointer (int), n_records)"""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.conga_results_device(self._h, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def fetch_all(self):
        """Batch mode: [(dels, dups, E, stats)] for every chromosome, in begin order."""
        out = []
        for i in range(len(self._meta)):
            self.select(i)
            out.append(self.fetch())
        return out

    def results_copy(self, dst_ptr, dst_bytes):
        """Enqueue a D2D copy of the result records to a device pointer (e.g. tensor.data_ptr())."""
        self._check(self._lib.conga_results_copy(self._h, C.c_void_p(dst_ptr), dst_bytes))

    def set_profile(self, on):
        self._check(self._lib.conga_set_profile(self._h, int(bool(on))))

    def read_depth(self):
        out = np.empty(self.chrom_len, dtype=np.int16)
        self._check(self._lib.conga_copy_read_depth(self._h, _p(out), len(out)))
        return out

    def mappability_track(self):
        out = np.empty(self.chrom_len, dtype=np.float32)
        self._check(self._lib.conga_copy_mappability(self._h, _p(out), len(out)))
        return out


def host_repeat_add_f32(s, c, k):
    """Both host builds of the fast-forward (the general routine and the per-window one the chain kernels call)
    must agree; returns their common value."""
    a = np.float32(load().conga_host_repeat_add_f32(C.c_float(float(s)), C.c_float(float(c)), int(k)))
    b = np.float32(load().conga_host_window_add_f32(C.c_float(float(s)), C.c_float(float(c)), int(k)))
    if a.view(np.uint32) != b.view(np.uint32) and not (np.isnan(a) and np.isnan(b)):
        raise AssertionError("conga_window_add_f32 %r != conga_repeat_add_f32 %r for s=%r c=%r k=%d" % (b, a, s, c, k))
    return a
