"""ctypes loader of libfbg_hip.so (C ABI declared in include/fbg_hip.h).

The library is the product; there is no Python or CPU fallback.  Importing this module
without the built .so raises, and every compute call on a box without a HIP device
returns FBG_ERR_NO_DEVICE (raised as FbgError).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfbg_hip.so")

FBG_OK, FBG_ERR_INVALID, FBG_ERR_NO_SEGMENTATION, FBG_ERR_OOM, FBG_ERR_HIP, FBG_ERR_TOO_LARGE, \
    FBG_ERR_NO_DEVICE, FBG_ERR_HASH_COLLISION = range(8)
PART_HALO = 64                               # FBG_PART_HALO (include/fbg_hip.h)
PART_HALO_BYTES = 2 * PART_HALO * 12 + 16    # FBG_PART_HALO_BYTES
STAGES = ("text", "suffix_sort", "lcp", "rank_scan", "scan", "dp", "rank_kernel", "sort_pass1", "sort_pass2", "sort_pass3")

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/fbg_hip.h one to one
SIGNATURES = {
    "fbg_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "fbg_ctx_destroy": (None, [vp]),
    "fbg_last_error": (C.c_char_p, [vp]),
    "fbg_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "fbg_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "fbg_set_stream": (C.c_int, [vp, vp]),
    "fbg_stage_ms": (C.c_int, [vp, C.c_int, fp, ip]),
    "fbg_device_bytes": (C.c_uint64, [vp]),
    "fbg_release_scratch": (C.c_int, [vp]),
    "fbg_elastic_f": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, u8p, C.c_uint64, C.c_int, u64p]),
    "fbg_minmax_dp": (C.c_int, [vp, u64p, C.c_uint64, u64p, u64p, u64p, u64p]),
    "fbg_repeatfree_v": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, u64p]),
    "fbg_repeatfree_dp": (C.c_int, [vp, u64p, C.c_uint64, u64p, u64p, u64p, u64p]),
    "fbg_gapped_v": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, u64p]),
    "fbg_gapped_dp": (C.c_int, [vp, u64p, C.c_uint64, u64p, u64p, u64p, u64p]),
    "fbg_msa_set_device": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64]),
    "fbg_msa_load_host": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64]),
    "fbg_msa_synthetic": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                    C.c_uint32, C.c_uint64, C.c_uint64]),
    "fbg_index_build": (C.c_int, [vp, C.c_int, u8p, C.c_uint64]),
    "fbg_part_index_build": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_int)]),
    "fbg_part_index_build_ignore": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_uint64, vp, C.POINTER(C.c_int)]),
    "fbg_part_scan": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int)]),
    "fbg_part_finish": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
    "fbg_part_rescan": (C.c_int, [vp, vp]),
    "fbg_block_graph": (C.c_int, [vp, u64p, C.c_uint64, u32p, u64p, u32p, u64p, u64p]),
    "fbg_scan_f": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, vp]),
    "fbg_scan_v": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp]),
    "fbg_minmax_dp_device": (C.c_int, [vp, vp, C.c_uint64, vp, u64p, vp, vp]),
    "fbg_repeatfree_dp_device": (C.c_int, [vp, vp, C.c_uint64, vp, vp, vp, u64p]),
    "fbg_scan_gapped_v": (C.c_int, [vp, vp]),
    "fbg_gapped_dp_device": (C.c_int, [vp, vp, C.c_uint64, vp, vp, vp, u64p]),
    "fbg_text_length": (C.c_uint64, [vp]),
    "fbg_index_download": (C.c_int, [vp, u8p, u32p, u32p, u32p, u32p]),
    "fbg_sync": (C.c_int, [vp]),
    "fbg_group_create": (C.c_int, [C.c_int, ip, C.POINTER(vp)]),
    "fbg_group_destroy": (None, [vp]),
    "fbg_group_last_error": (C.c_char_p, [vp]),
    "fbg_group_size": (C.c_int, [vp]),
    "fbg_group_member": (vp, [vp, C.c_int]),
    "fbg_group_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "fbg_group_plan_used": (C.c_int, [vp, ip]),
    "fbg_group_elastic_f": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, u8p, C.c_uint64, C.c_int, u64p]),
    "fbg_group_repeatfree_v": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, u64p]),
    "fbg_group_gapped_v": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64, u64p]),
    "fbg_group_msa_load_host": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint64]),
    "fbg_group_msa_synthetic": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                          C.c_uint64, C.c_uint64]),
    "fbg_group_scan_f": (C.c_int, [vp, u8p, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "fbg_host_alloc": (vp, [C.c_uint64]),
    "fbg_host_free": (None, [vp]),
}
PLANS = ("auto", "partitioned", "columns", "row_pairs")   # FBG_PLAN_*

_LIB = None


def build():
    """Compile the HIP sources in csrc/ for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C founderblockgraphs_amd/csrc` "
                              "(or __graft_entry__.build()); there is no fallback path")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 and a
        # process that loads both that copy and /opt/rocm's ends up with torch seeing no GPU.
        # Loading torch first makes the dynamic loader hand the same (already loaded) runtime to
        # libfbg_hip.so.  The C++ host program never loads torch and uses /opt/rocm's runtime.
        if os.environ.get("FBG_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = header / library mismatch
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB
