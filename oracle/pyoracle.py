"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see the header of fbg_oracle.c).  PARITY UNPINNED: nothing here was produced or
checked by the reference binary, which cannot be built in this pipeline.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_OK, ORC_ERR_ALLOC, ORC_ERR_TOO_LARGE, ORC_ERR_NO_SEGMENTATION, ORC_ERR_IO = range(5)


def build(march=None, out=None):
    """Compile liboracle.so with gcc (seconds).  march='native' mirrors the reference Makefile:3."""
    cmd = ["make", "-C", _HERE, "-B"]
    if march:
        cmd.append(f"MARCH={march}")
    if out:
        cmd.append(f"OUT={out}")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return out or os.path.join(_HERE, "liboracle.so")


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(p):
        build()
    L = C.CDLL(p)
    u8p, u64p, i32p, i64p, dp = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double))
    L.orc_suffix_array.argtypes = [u8p, C.c_int64, i32p]
    L.orc_msa_index.argtypes = [u8p, C.c_int64, C.c_int64, i64p, u8p, i32p, i32p, i32p]
    L.orc_compute_f.argtypes = [u8p, C.c_int64, C.c_int64, u8p, C.c_int64, C.c_int, C.c_int, u64p, dp, dp]
    L.orc_compute_f_literal.argtypes = [u8p, C.c_int64, C.c_int64, u8p, C.c_int64, C.c_int, u64p]
    L.orc_minmax_dp.argtypes = [u64p, C.c_int64, u64p, u64p, u64p, i64p]
    L.orc_segment_v_literal.argtypes = [u8p, C.c_int64, C.c_int64, u64p]
    L.orc_segment_v.argtypes = [u8p, C.c_int64, C.c_int64, u64p, dp, dp]
    L.orc_segment_dp.argtypes = [u64p, C.c_int64, u64p, u64p, u64p, i64p]
    L.orc_gapped_v.argtypes = [u8p, C.c_int64, C.c_int64, u64p]
    L.orc_segment2_dp.argtypes = [u64p, C.c_int64, u64p, u64p, u64p, i64p]
    L.orc_write_xgfa.argtypes = [u8p, C.c_int64, C.c_int64, u64p, C.c_int64, C.c_int, u8p, i64p, C.c_char_p]
    L.orc_segment_stats.argtypes = [u8p, C.c_int64, C.c_int64, u64p, C.c_int64, u64p]
    for f in ("orc_suffix_array", "orc_msa_index", "orc_compute_f", "orc_compute_f_literal",
              "orc_minmax_dp", "orc_segment_v_literal", "orc_segment_v", "orc_segment_dp",
              "orc_gapped_v", "orc_segment2_dp",
              "orc_write_xgfa", "orc_segment_stats"):
        getattr(L, f).restype = C.c_int
    if path is None:
        _LIB = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def msa_array(rows):
    """list of equal-length str/bytes rows -> (m, n) uint8 array."""
    rows = [r.encode() if isinstance(r, str) else bytes(r) for r in rows]
    n = len(rows[0])
    assert all(len(r) == n for r in rows)
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), n).copy()


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def suffix_array(text: bytes):
    t = np.frombuffer(text, dtype=np.uint8).copy()
    sa = np.empty(len(t), dtype=np.int32)
    _chk(lib().orc_suffix_array(_p(t, C.c_uint8), len(t), _p(sa, C.c_int32)))
    return sa


def msa_index(msa):
    msa = np.ascontiguousarray(msa, dtype=np.uint8)
    m, n = msa.shape
    cap = m * n + m + 1
    T = np.empty(cap, dtype=np.uint8)
    SA = np.empty(cap, dtype=np.int32)
    ISA = np.empty(cap, dtype=np.int32)
    LCP = np.empty(cap, dtype=np.int32)
    N = C.c_int64(0)
    _chk(lib().orc_msa_index(_p(msa, C.c_uint8), m, n, C.byref(N), _p(T, C.c_uint8),
                             _p(SA, C.c_int32), _p(ISA, C.c_int32), _p(LCP, C.c_int32)))
    N = N.value
    return T[:N].copy(), SA[:N].copy(), ISA[:N].copy(), LCP[:N].copy()


def _ign(ignore):
    ig = np.frombuffer((ignore or "").encode() if isinstance(ignore, str) else (ignore or b""),
                       dtype=np.uint8).copy()
    if len(ig) == 0:
        ig = np.zeros(1, dtype=np.uint8)
        return ig, 0
    return ig, len(ig)


def compute_f(msa, ignore="", disable_tricks=False, threads=1, f_init=None, literal=False,
              timings=None, library=None):
    msa = np.ascontiguousarray(msa, dtype=np.uint8)
    m, n = msa.shape
    ig, il = _ign(ignore)
    f = np.zeros(n, dtype=np.uint64) if f_init is None else np.array(f_init, dtype=np.uint64)
    L = library or lib()
    if literal:
        _chk(L.orc_compute_f_literal(_p(msa, C.c_uint8), m, n, _p(ig, C.c_uint8), il,
                                     int(disable_tricks), _p(f, C.c_uint64)))
    else:
        ti, ts = C.c_double(0), C.c_double(0)
        _chk(L.orc_compute_f(_p(msa, C.c_uint8), m, n, _p(ig, C.c_uint8), il, int(disable_tricks),
                             int(threads), _p(f, C.c_uint64), C.byref(ti), C.byref(ts)))
        if timings is not None:
            timings["index_s"] = ti.value
            timings["scan_s"] = ts.value
    return f


def minmax_dp(f, library=None):
    """-> (mml[0..n], bt[0..n], boundaries).  Raises OracleError(3) if backtrack leaves the array."""
    f = np.ascontiguousarray(f, dtype=np.uint64)
    n = len(f)
    mml = np.empty(n + 1, dtype=np.uint64)
    bt = np.empty(n + 1, dtype=np.uint64)
    b = np.empty(n + 1, dtype=np.uint64)
    cnt = C.c_int64(0)
    rc = (library or lib()).orc_minmax_dp(_p(f, C.c_uint64), n, _p(mml, C.c_uint64), _p(bt, C.c_uint64),
                                          _p(b, C.c_uint64), C.byref(cnt))
    if rc != 0:
        raise OracleError(rc)
    return mml, bt, b[:cnt.value].copy()


def segment_v(msa, literal=False, timings=None):
    msa = np.ascontiguousarray(msa, dtype=np.uint8)
    m, n = msa.shape
    v = np.empty(n, dtype=np.uint64)
    if literal:
        _chk(lib().orc_segment_v_literal(_p(msa, C.c_uint8), m, n, _p(v, C.c_uint64)))
    else:
        ti, ts = C.c_double(0), C.c_double(0)
        _chk(lib().orc_segment_v(_p(msa, C.c_uint8), m, n, _p(v, C.c_uint64), C.byref(ti), C.byref(ts)))
        if timings is not None:
            timings["index_s"] = ti.value
            timings["scan_s"] = ts.value
    return v


def segment_dp(v):
    """-> (s, prev, boundaries or None when no proper segmentation exists)."""
    v = np.ascontiguousarray(v, dtype=np.uint64)
    n = len(v)
    s = np.empty(n, dtype=np.uint64)
    prev = np.empty(n, dtype=np.uint64)
    b = np.empty(n, dtype=np.uint64)
    cnt = C.c_int64(0)
    rc = lib().orc_segment_dp(_p(v, C.c_uint64), n, _p(s, C.c_uint64), _p(prev, C.c_uint64),
                              _p(b, C.c_uint64), C.byref(cnt))
    if rc == ORC_ERR_NO_SEGMENTATION:
        return s, prev, None
    _chk(rc)
    return s, prev, b[:cnt.value].copy()


def gapped_v(msa, literal=False):
    """v[] of segment2elasticValid (fbg.cpp:763-822); literal = the two-pointer over BWT intervals."""
    msa = np.ascontiguousarray(msa, dtype=np.uint8)
    m, n = msa.shape
    v = np.empty(n, dtype=np.uint64)
    if literal:
        _chk(lib().orc_segment_v_literal(_p(msa, C.c_uint8), m, n, _p(v, C.c_uint64)))
    else:
        _chk(lib().orc_gapped_v(_p(msa, C.c_uint8), m, n, _p(v, C.c_uint64)))
    return v


def segment2_dp(v):
    """segment2elasticValid's DP (fbg.cpp:827-866) -> (s, prev, boundaries or None)."""
    v = np.ascontiguousarray(v, dtype=np.uint64)
    n = len(v)
    s = np.empty(n, dtype=np.uint64)
    prev = np.empty(n, dtype=np.uint64)
    b = np.empty(n, dtype=np.uint64)
    cnt = C.c_int64(0)
    rc = lib().orc_segment2_dp(_p(v, C.c_uint64), n, _p(s, C.c_uint64), _p(prev, C.c_uint64),
                               _p(b, C.c_uint64), C.byref(cnt))
    if rc == ORC_ERR_NO_SEGMENTATION:
        return s, prev, None
    _chk(rc)
    return s, prev, b[:cnt.value].copy()


def write_xgfa(msa, boundaries, path, ids=None):
    msa = np.ascontiguousarray(msa, dtype=np.uint8)
    m, n = msa.shape
    b = np.ascontiguousarray(boundaries, dtype=np.uint64)
    if ids is not None:
        enc = [i.encode() if isinstance(i, str) else bytes(i) for i in ids]
        blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
        off = np.zeros(m + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(e) for e in enc])
        _chk(lib().orc_write_xgfa(_p(msa, C.c_uint8), m, n, _p(b, C.c_uint64), len(b), 1,
                                  _p(blob, C.c_uint8), _p(off, C.c_int64), os.fsencode(path)))
    else:
        _chk(lib().orc_write_xgfa(_p(msa, C.c_uint8), m, n, _p(b, C.c_uint64), len(b), 0,
                                  None, None, os.fsencode(path)))
    with open(path, "rb") as fh:
        return fh.read()


def segment_stats(msa, boundaries):
    msa = np.ascontiguousarray(msa, dtype=np.uint8)
    m, n = msa.shape
    b = np.ascontiguousarray(boundaries, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    _chk(lib().orc_segment_stats(_p(msa, C.c_uint8), m, n, _p(b, C.c_uint64), len(b), _p(out, C.c_uint64)))
    return dict(nodes=int(out[0]), total_label_length=int(out[1]), founders=int(out[2]), edges=int(out[3]))
