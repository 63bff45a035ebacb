"""ctypes binding of csrc/libaoc_hip.so (the C ABI declared in include/aoc_hip.h).

There is deliberately no fallback: if the shared library is missing the import of an operator
fails with a clear error instead of silently running something else.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
RELEASE_SO = os.path.join(CSRC, "libaoc_hip.so")
DEV_SO = os.path.join(CSRC, "libaoc_hip_dev.so")
# AOC_LIB_VARIANT=dev (read HERE, in Python, never by the library): load the development build, the only one that knows the
# developer switches of tools/README.md (timing experiments, alternative kernels).  The release library ignores the environment.
VARIANT = os.environ.get("AOC_LIB_VARIANT", "release")
if VARIANT not in ("release", "dev"):
    raise ImportError(f"AOC_LIB_VARIANT={VARIANT!r}: 'release' or 'dev'")
SO_PATH = DEV_SO if VARIANT == "dev" else RELEASE_SO
if os.environ.get("AOC_LIB_FILE"):          # developer override (python side): an experimental build of the library, by file name in csrc/
    SO_PATH = os.path.join(CSRC, os.environ["AOC_LIB_FILE"])

_vp, _i, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); mirrors include/aoc_hip.h one to one
SIGNATURES = {
    "aoc_version": (ctypes.c_char_p, []),
    "aoc_label_prep_workspace_bytes": (_sz, [_i64, _i]),
    "aoc_label_prep": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_label_bits": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "aoc_kmeans_plan": (_i, [_vp, _i, _i, _vp, _vp]),
    "aoc_kmeans_init_rows_draw": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "aoc_kmeans_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "aoc_kmeans_segmented": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_kmeans_segmented_ex": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_kmeans_segmented_rep": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_prehead_workspace_bytes": (_sz, [_i, _i, _i, _i64]),
    "aoc_prehead": (_i, [_vp, _i, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _sz, _vp]),
    "aoc_cond_codes": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "aoc_plane_reduce": (_i, [_vp, _i64, _i64, _i, _vp, _vp]),
    "aoc_gct_gate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp]),
    "aoc_object_logit": (_i, [_vp, _i, _i, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "aoc_confident_labels": (_i, [_vp, _i, _i64, ctypes.c_uint32, _vp, ctypes.c_float, _vp, _vp, _vp, _vp]),
    "aoc_label_onehot_nearest": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "aoc_mask_jf_workspace_bytes": (_sz, [_i, _i]),
    "aoc_mask_jf_accumulate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _i, _vp, _vp]),
    "aoc_kmeans_replicate": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp]),
    "aoc_kmeans_replicate_levels": (_i, [_vp, _vp, _i, _i, _vp, _i, _i64, _vp, _vp, _vp, _vp]),
    "aoc_build_proxies_workspace_bytes": (_sz, [_i64, _i, _i]),
    "aoc_build_proxies": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "aoc_proxy_corr_min": (_i, [_vp, _i64, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "aoc_proxy_corr_min_batched_workspace_bytes": (_sz, []),
    "aoc_proxy_corr_min_batched": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "aoc_proxy_corr_min_records": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "aoc_proxy_corr_min_records_cached_workspace_bytes": (_sz, []),
    "aoc_proxy_corr_min_records_cached": (_i, [_vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "aoc_dense_match_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "aoc_dense_match_min": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp, _i64, _i64, _i, _vp, _sz, _vp]),
    "aoc_dense_match_min_f16": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp, _i64, _i64, _i, _vp, _sz, _vp]),
    "aoc_proxy_corr_min_f16": (_i, [_vp, _i64, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "aoc_dense_match_set_probe": (_i, [_vp, _vp]),
    "aoc_set_stream_cus": (_i, [_i]),
    "aoc_split_record_bytes": (_sz, [_i]),
    "aoc_split_rows": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "aoc_split_rows_tiled_bytes": (_sz, [_i64, _i]),
    "aoc_split_rows_tiled": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "aoc_dense_match_split_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "aoc_dense_prune_stats": (_i, [_vp, _i]),
    "aoc_dense_prune_stats_ex": (_i, [_vp, _i]),
    "aoc_dense_match_min_split": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i64, _i,
                                       _vp, _sz, _vp]),
    "aoc_dense_match_min_split_cached": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i64, _i,
                                              _vp, _sz, _i, _vp]),
    "aoc_frame_channels": (_i, [_i, _i, _i]),
    "aoc_frame_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "aoc_frame_enqueue": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "aoc_cluster_chain_workspace_bytes": (_sz, [_vp]),
    "aoc_cluster_chain_layout": (_i, [_vp, _vp]),
    "aoc_cluster_chain_enqueue": (_i, [_vp, _vp, _sz, _vp]),
    "aoc_gates_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "aoc_gates_enqueue": (_i, [_vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "aoc_local_window_match": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "aoc_resize_bilinear_hwc": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "aoc_resize_bilinear_hwc_ex": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "aoc_local_window_match_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "aoc_resize_bilinear_planes": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i64, _i64, _i64, _vp]),
    "aoc_resize_nearest_bits": (_i, [_vp, _i, _i, _vp, _i, _i, _vp]),
    "aoc_atrous_subsample": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "aoc_fg2bg_min": (_i, [_vp, _i, _i, _i64, _i64, _vp, _i64, _vp]),
    "aoc_resize_bilinear_planes_grouped": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _vp]),
    "aoc_local_window_match_pair": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "aoc_local_prep": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "aoc_proto_finish": (_i, [_vp, _i, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "aoc_masked_mean_pool_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "aoc_masked_mean_pool": (_i, [_vp, _vp, _i, _i64, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_film_gain": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "aoc_channel_scale": (_i, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "aoc_film_scale": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp, _vp]),
    "aoc_cond_gate_pool_workspace_bytes": (_sz, [_i, _i, _i64]),
    "aoc_cond_gate_pool": (_i, [_vp, _i, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_cond_gate_pool_ex": (_i, [_vp, _i, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "aoc_groupnorm_relu_workspace_bytes": (_sz, [_i, _i]),
    "aoc_groupnorm_relu": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _sz, _vp]),
    "aoc_linear": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "aoc_label_mix": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "aoc_plane_mean": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "aoc_head_delta": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
}

STATUS = {0: "AOC_OK", -1: "AOC_ERR_INVALID_ARG", -2: "AOC_ERR_WORKSPACE", -3: "AOC_ERR_LAUNCH", -4: "AOC_ERR_UNSUPPORTED"}

_lib = None


class AocHipError(RuntimeError):
    pass


def build(force=False, dev=True):
    """hipcc --offload-arch=gfx950 build of csrc/*.hip -> csrc/libaoc_hip.so (in-tree) and, with dev=True, the development
    build csrc/libaoc_hip_dev.so (-DAOC_DEV: the same sources with the developer switches compiled in)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    if dev:
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "DEV=1"])
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise AocHipError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the aoc_amd operators.")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise AocHipError(f"{what} failed: {STATUS.get(rc, rc)}")
