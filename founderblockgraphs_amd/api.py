"""Host-side mirror of the reference's in-process seam for the segmentation path.

Names and argument meaning follow founderblockgraph.cpp ("fbg.cpp"):
  segment_elastic_minmaxlength  fbg.cpp:1836-2040   (elastic: f[] scan + min-max-length DP)
  segment                       fbg.cpp:526-664     (non-elastic: v[] scan + s/prev DP)
Every call goes through the C ABI of libfbg_hip.so (include/fbg_hip.h); nothing here computes.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FBG_ERR_NO_SEGMENTATION, FBG_OK, STAGES


class FbgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfbg_hip error {code}: {msg}")
        self.code = code


class NoSegmentation(FbgError):
    """'No valid segmentation found!' (fbg.cpp:1934) / 'No proper segmentation exists.' (fbg.cpp:650)."""


def _u8(a):
    return a.ctypes.data_as(_lib.u8p)


def _u64(a):
    return a.ctypes.data_as(_lib.u64p)


def _ignore(ignorechars):
    if isinstance(ignorechars, str):
        ignorechars = ignorechars.encode()
    ig = np.frombuffer(bytes(ignorechars or b"") + b"\0", dtype=np.uint8).copy()
    return ig, len(ig) - 1


def as_msa(rows):
    """(m, n) uint8 array from an array or a list of equal-length str/bytes rows."""
    if isinstance(rows, np.ndarray):
        a = np.ascontiguousarray(rows, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("MSA must be 2-D (rows x columns)")
        return a
    rows = [r.encode() if isinstance(r, str) else bytes(r) for r in rows]
    if not rows or any(len(r) != len(rows[0]) for r in rows):
        raise ValueError("MSA rows must be non-empty and of equal length")
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0])).copy()


class Engine:
    """One fbg_ctx (one GPU, one stream).  Calls on one engine must be serialised by the caller."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.fbg_ctx_create(int(device), C.byref(h))
        if rc != FBG_OK:
            raise FbgError(rc, self._L.fbg_last_error(None).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):      # a group's member belongs to the group
                self._L.fbg_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc == FBG_OK:
            return
        msg = self._L.fbg_last_error(self._h).decode()
        raise (NoSegmentation if rc == FBG_ERR_NO_SEGMENTATION else FbgError)(rc, msg)

    # ---- host-buffer entry points --------------------------------------------------------
    def elastic_f(self, msa, ignorechars="", disable_efg_tricks=False, f=None):
        """compute_f: f is max-merged into (zeros if omitted), as fbg.cpp:1681 / 3388."""
        msa = as_msa(msa)
        m, n = msa.shape
        self._msa_rows = m
        ig, il = _ignore(ignorechars)
        f = np.zeros(n, dtype=np.uint64) if f is None else np.ascontiguousarray(f, dtype=np.uint64).copy()
        rc = self._L.fbg_elastic_f(self._h, _u8(msa), m, n, _u8(ig), il, int(disable_efg_tricks), _u64(f))
        if rc == FBG_ERR_NO_SEGMENTATION:
            raise NoSegmentation(rc, self._L.fbg_last_error(self._h).decode())
        self._chk(rc)
        return f

    def minmax_dp(self, f, full=False):
        f = np.ascontiguousarray(f, dtype=np.uint64)
        n = len(f)
        b = np.empty(n + 1, dtype=np.uint64)
        mml = np.empty(n + 1, dtype=np.uint64)
        bt = np.empty(n + 1, dtype=np.uint64)
        cnt = C.c_uint64(0)
        self._chk(self._L.fbg_minmax_dp(self._h, _u64(f), n, _u64(b), C.byref(cnt), _u64(mml), _u64(bt)))
        b = b[:cnt.value].copy()
        return (b, mml, bt) if full else b

    def repeatfree_v(self, msa):
        msa = as_msa(msa)
        m, n = msa.shape
        v = np.empty(n, dtype=np.uint64)
        self._chk(self._L.fbg_repeatfree_v(self._h, _u8(msa), m, n, _u64(v)))
        return v

    def repeatfree_dp(self, v):
        """-> (s, prev, boundaries); boundaries is None when no proper segmentation exists."""
        v = np.ascontiguousarray(v, dtype=np.uint64)
        n = len(v)
        s = np.empty(n, dtype=np.uint64)
        prev = np.empty(n, dtype=np.uint64)
        b = np.empty(n, dtype=np.uint64)
        cnt = C.c_uint64(0)
        rc = self._L.fbg_repeatfree_dp(self._h, _u64(v), n, _u64(s), _u64(prev), _u64(b), C.byref(cnt))
        if rc == FBG_ERR_NO_SEGMENTATION:
            return s, prev, None
        self._chk(rc)
        return s, prev, b[:cnt.value].copy()

    def gapped_v(self, msa):
        """v[] of segment2elasticValid (fbg.cpp:763-822): non-elastic mode, rows may hold gaps."""
        msa = as_msa(msa)
        m, n = msa.shape
        v = np.empty(n, dtype=np.uint64)
        self._chk(self._L.fbg_gapped_v(self._h, _u8(msa), m, n, _u64(v)))
        return v

    def gapped_dp(self, v):
        """fbg.cpp:827-866 -> (s, prev, boundaries); boundaries is None for 'No valid segmentation found!'."""
        v = np.ascontiguousarray(v, dtype=np.uint64)
        n = len(v)
        s = np.empty(n, dtype=np.uint64)
        prev = np.empty(n, dtype=np.uint64)
        b = np.empty(n, dtype=np.uint64)
        cnt = C.c_uint64(0)
        rc = self._L.fbg_gapped_dp(self._h, _u64(v), n, _u64(s), _u64(prev), _u64(b), C.byref(cnt))
        if rc == FBG_ERR_NO_SEGMENTATION:
            return s, prev, None
        self._chk(rc)
        return s, prev, b[:cnt.value].copy()

    def block_graph(self, boundaries):
        """Nodes / edges of the elastic founder graph for a segmentation of the current MSA (fbg_block_graph):
        (node_of[nb, m], first_node[nb + 1], rep_row[nb, m], edge_count[nb], edges[nb, m])."""
        b = np.ascontiguousarray(boundaries, dtype=np.uint64)
        nb = len(b)
        m = self._msa_rows
        node_of = np.empty((nb, m), dtype=np.uint32)
        rep_row = np.empty((nb, m), dtype=np.uint32)
        first = np.empty(nb + 1, dtype=np.uint64)
        ecount = np.empty(nb, dtype=np.uint64)
        edges = np.empty((nb, m), dtype=np.uint64)
        self._chk(self._L.fbg_block_graph(self._h, _u64(b), nb, node_of.ctypes.data_as(_lib.u32p), _u64(first),
                                          rep_row.ctypes.data_as(_lib.u32p), _u64(ecount), _u64(edges)))
        return node_of, first, rep_row, ecount, edges

    # ---- device-resident staged API ------------------------------------------------------
    def msa_load_host(self, msa):
        msa = as_msa(msa)
        self._msa_rows = msa.shape[0]
        self._chk(self._L.fbg_msa_load_host(self._h, _u8(msa), msa.shape[0], msa.shape[1]))

    def msa_set_device(self, ptr, m, n):
        self._msa_rows = m
        self._chk(self._L.fbg_msa_set_device(self._h, C.c_void_p(ptr), m, n))

    def msa_synthetic(self, ptr, m, n, seed=0x5EED0001, seed2=0x5EED0002, gap_fraction=0.0, gap_run=0,
                      seed3=0x5EED0003, n_fraction=0.0):
        gthr = int((1 << 64) * (gap_fraction / gap_run)) if gap_run else 0
        nthr = int((1 << 64) * n_fraction)
        self._chk(self._L.fbg_msa_synthetic(self._h, C.c_void_p(ptr), m, n, seed, seed2, gthr, gap_run, seed3, nthr))

    def index_build(self, reversed=False, ignorechars=""):
        ig, il = _ignore(ignorechars)
        self._chk(self._L.fbg_index_build(self._h, int(reversed), _u8(ig), il))

    # partitioned index (multi-GPU, include/fbg_hip.h): each returns ok; False = use index_build on every rank
    def part_index_build(self, part, nparts, d_blob_ptr, reversed=False, ignorechars="", disable_efg_tricks=False):
        """One key-range partition of the index (fbg_part_index_build_ignore).  MSAs with gaps / ignore characters are
        scanned for ONE setting of the elastic tricks (option part_tricks_off), given here."""
        ok = C.c_int(0)
        ig, il = _ignore(ignorechars)
        self.set_option("part_tricks_off", int(bool(disable_efg_tricks)))
        self._chk(self._L.fbg_part_index_build_ignore(self._h, int(reversed), part, nparts, _u8(ig), il, C.c_void_p(d_blob_ptr), C.byref(ok)))
        return bool(ok.value)

    def part_scan(self, d_blobs_ptr, d_gmax_ptr):
        ok = C.c_int(0)
        self._chk(self._L.fbg_part_scan(self._h, C.c_void_p(d_blobs_ptr), C.c_void_p(d_gmax_ptr), C.byref(ok)))
        return bool(ok.value)

    def part_finish(self, d_gmax_ptr):
        """0: declined, 1: index ready, 2: call part_rescan, reduce again, then part_finish again"""
        ok = C.c_int(0)
        self._chk(self._L.fbg_part_finish(self._h, C.c_void_p(d_gmax_ptr), C.byref(ok)))
        return int(ok.value)

    def part_rescan(self, d_gmax_ptr):
        self._chk(self._L.fbg_part_rescan(self._h, C.c_void_p(d_gmax_ptr)))

    def scan_f(self, x0, x1, d_f_ptr, disable_efg_tricks=False):
        self._chk(self._L.fbg_scan_f(self._h, x0, x1, int(disable_efg_tricks), C.c_void_p(d_f_ptr)))

    def scan_v(self, x0, x1, d_v_ptr):
        self._chk(self._L.fbg_scan_v(self._h, x0, x1, C.c_void_p(d_v_ptr)))

    def minmax_dp_device(self, d_f_ptr, n, d_boundaries_ptr, d_mml_ptr=None, d_bt_ptr=None):
        cnt = C.c_uint64(0)
        self._chk(self._L.fbg_minmax_dp_device(self._h, C.c_void_p(d_f_ptr), n, C.c_void_p(d_boundaries_ptr),
                                               C.byref(cnt), C.c_void_p(d_mml_ptr), C.c_void_p(d_bt_ptr)))
        return cnt.value

    def repeatfree_dp_device(self, d_v_ptr, n, d_boundaries_ptr, d_s_ptr=None, d_prev_ptr=None):
        cnt = C.c_uint64(0)
        self._chk(self._L.fbg_repeatfree_dp_device(self._h, C.c_void_p(d_v_ptr), n, C.c_void_p(d_s_ptr),
                                                   C.c_void_p(d_prev_ptr), C.c_void_p(d_boundaries_ptr),
                                                   C.byref(cnt)))
        return cnt.value

    def scan_gapped_v(self, d_v_ptr):
        self._chk(self._L.fbg_scan_gapped_v(self._h, C.c_void_p(d_v_ptr)))

    def gapped_dp_device(self, d_v_ptr, n, d_boundaries_ptr, d_s_ptr=None, d_prev_ptr=None):
        cnt = C.c_uint64(0)
        rc = self._L.fbg_gapped_dp_device(self._h, C.c_void_p(d_v_ptr), n, C.c_void_p(d_s_ptr),
                                          C.c_void_p(d_prev_ptr), C.c_void_p(d_boundaries_ptr), C.byref(cnt))
        if rc == FBG_ERR_NO_SEGMENTATION:
            return None
        self._chk(rc)
        return cnt.value

    def set_option(self, key, value=1):
        """fbg_set_option: behaviour switch of this context (include/fbg_hip.h lists the keys)."""
        self._chk(self._L.fbg_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        self._chk(self._L.fbg_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def options(self, **kv):
        """Context manager: set the given options, restore the previous values on exit."""
        eng = self

        class _Scope:
            def __enter__(self_):
                self_.old = {k: eng.get_option(k) for k in kv}
                for k, v in kv.items():
                    eng.set_option(k, v)
                return eng

            def __exit__(self_, *exc):
                for k, v in self_.old.items():
                    eng.set_option(k, v)
        return _Scope()

    def set_stream(self, stream_handle):
        self._chk(self._L.fbg_set_stream(self._h, C.c_void_p(stream_handle)))

    def sync(self):
        self._chk(self._L.fbg_sync(self._h))

    def stage_ms(self):
        out = {}
        for k, name in enumerate(STAGES):
            ms, ln = C.c_float(0), C.c_int(0)
            self._chk(self._L.fbg_stage_ms(self._h, k, C.byref(ms), C.byref(ln)))
            out[name] = (ms.value, ln.value)
        return out

    def release_scratch(self):
        self._chk(self._L.fbg_release_scratch(self._h))

    def device_bytes(self):
        return int(self._L.fbg_device_bytes(self._h))

    def text_length(self):
        return int(self._L.fbg_text_length(self._h))

    def index_download(self):
        N = self.text_length()
        T = np.empty(N, dtype=np.uint8)
        arrs = [np.empty(N, dtype=np.uint32) for _ in range(4)]
        self._chk(self._L.fbg_index_download(self._h, _u8(T), *[a.ctypes.data_as(_lib.u32p) for a in arrs]))
        return (T, *arrs)


class _DeviceArray:
    """A device pointer dressed up for torch.as_tensor (zero copy): __cuda_array_interface__, version 2."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, count, typestr="<i8"):
    """torch view (no copy) of `count` elements of engine-owned device memory; valid as long as the engine keeps it."""
    import torch
    return torch.as_tensor(_DeviceArray(ptr, count, typestr), device="cuda")


class Group:
    """fbg_group: several contexts (one per entry of `devices`; an id may repeat) driven as one engine --
    include/fbg_hip.h, 'several GPUs as one engine'.  The sweep and block_graph run on member(0)."""

    def __init__(self, devices=None):
        self._L = _lib.lib()
        h = C.c_void_p()
        if devices is None:
            rc = self._L.fbg_group_create(0, None, C.byref(h))
        else:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._L.fbg_group_create(len(devices), ids, C.byref(h))
        if rc != FBG_OK:
            raise FbgError(rc, self._L.fbg_group_last_error(None).decode())
        self._h = h
        self._members = {}

    def close(self):
        if getattr(self, "_h", None):
            for e in self._members.values():
                e._h = None                        # owned by the group
            self._L.fbg_group_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc == FBG_OK:
            return
        msg = self._L.fbg_group_last_error(self._h).decode()
        raise (NoSegmentation if rc == FBG_ERR_NO_SEGMENTATION else FbgError)(rc, msg)

    def size(self):
        return int(self._L.fbg_group_size(self._h))

    def member(self, i=0):
        """Engine view of member i's context (borrowed: closing it is a no-op)."""
        if i not in self._members:
            e = Engine.__new__(Engine)
            e._L = self._L
            e._h = C.c_void_p(self._L.fbg_group_member(self._h, i))
            e.device = None
            e._borrowed = True
            self._members[i] = e
        return self._members[i]

    def set_option(self, key, value=1):
        self._chk(self._L.fbg_group_set_option(self._h, key.encode(), int(value)))

    def plan_used(self):
        parts = C.c_int(0)
        plan = self._L.fbg_group_plan_used(self._h, C.byref(parts))
        return _lib.PLANS[plan], parts.value

    def elastic_f(self, msa, ignorechars="", disable_efg_tricks=False, f=None):
        msa = as_msa(msa)
        m, n = msa.shape
        self.member(0)._msa_rows = m
        ig, il = _ignore(ignorechars)
        f = np.zeros(n, dtype=np.uint64) if f is None else np.ascontiguousarray(f, dtype=np.uint64).copy()
        self._chk(self._L.fbg_group_elastic_f(self._h, _u8(msa), m, n, _u8(ig), il, int(disable_efg_tricks), _u64(f)))
        return f

    def repeatfree_v(self, msa):
        msa = as_msa(msa)
        v = np.empty(msa.shape[1], dtype=np.uint64)
        self._chk(self._L.fbg_group_repeatfree_v(self._h, _u8(msa), msa.shape[0], msa.shape[1], _u64(v)))
        return v

    def gapped_v(self, msa):
        msa = as_msa(msa)
        v = np.empty(msa.shape[1], dtype=np.uint64)
        self._chk(self._L.fbg_group_gapped_v(self._h, _u8(msa), msa.shape[0], msa.shape[1], _u64(v)))
        return v

    def msa_load_host(self, msa):
        msa = as_msa(msa)
        self.member(0)._msa_rows = msa.shape[0]
        self._chk(self._L.fbg_group_msa_load_host(self._h, _u8(msa), msa.shape[0], msa.shape[1]))

    def msa_synthetic(self, m, n, seed=0x5EED0001, seed2=0x5EED0002, gap_fraction=0.0, gap_run=0, seed3=0x5EED0003,
                      n_fraction=0.0):
        gthr = int((1 << 64) * (gap_fraction / gap_run)) if gap_run else 0
        nthr = int((1 << 64) * n_fraction)
        self.member(0)._msa_rows = m
        self._chk(self._L.fbg_group_msa_synthetic(self._h, m, n, seed, seed2, gthr, gap_run, seed3, nthr))

    def scan_f(self, ignorechars="", disable_efg_tricks=False):
        """f of the current MSA -> device pointer (member 0's memory, n uint64 values)."""
        ig, il = _ignore(ignorechars)
        p = C.c_void_p()
        self._chk(self._L.fbg_group_scan_f(self._h, _u8(ig), il, int(disable_efg_tricks), C.byref(p)))
        return p.value


# ---- reference-shaped free functions ---------------------------------------------------------

def segment_elastic_minmaxlength(MSA, ignorechars="", disable_efg_tricks=False, f=None, segment=True,
                                 engine=None):
    """fbg.cpp:1836-2040.  Returns (out_indices, f); out_indices is None when segment=False."""
    eng = engine or Engine()
    try:
        f = eng.elastic_f(MSA, ignorechars, disable_efg_tricks, f)
        if not segment:
            return None, f
        return eng.minmax_dp(f), f
    finally:
        if engine is None:
            eng.close()


def segment2elasticValid(MSA, engine=None):
    """fbg.cpp:738-866 up to the boundaries (non-elastic mode with --gap-limit != 1).  Returns
    (status, v, s, prev, boundaries): status 1 = EXIT_FAILURE ('No valid segmentation found!')."""
    eng = engine or Engine()
    try:
        v = eng.gapped_v(MSA)
        s, prev, b = eng.gapped_dp(v)
        return (0 if b is not None else 1), v, s, prev, b
    finally:
        if engine is None:
            eng.close()


def segment(MSA, engine=None):
    """fbg.cpp:526-664 up to the boundaries.  Returns (status, v, s, prev, boundaries):
    status 0 = EXIT_SUCCESS, 1 = EXIT_FAILURE ('No proper segmentation exists.')."""
    eng = engine or Engine()
    try:
        v = eng.repeatfree_v(MSA)
        s, prev, b = eng.repeatfree_dp(v)
        return (0 if b is not None else 1), v, s, prev, b
    finally:
        if engine is None:
            eng.close()
