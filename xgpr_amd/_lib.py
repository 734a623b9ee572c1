"""ctypes binding of libxgpr_hip.so (the C ABI declared in include/xgpr_hip.h).

There is no CPU fallback: if the library is missing the import of any product module
fails loudly, and every compute entry point needs a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XGPR_HIP_LIB: development aid (ablation builds of the same library); never a fallback
LIB_PATH = os.environ.get("XGPR_HIP_LIB", os.path.join(_HERE, "libxgpr_hip.so"))

_vp, _l, _i, _d, _sz = C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_size_t

# name -> argtypes; mirrors include/xgpr_hip.h one to one (tests/test_cabi.py checks that
# every function declared in the header is listed here and exported by the library).
SIGNATURES = {
    "xgpr_fht_f32": [_vp, _l, _l, _l, _vp],
    "xgpr_fht_f64": [_vp, _l, _l, _l, _vp],
    "xgpr_srht_f32": [_vp, _vp, _l, _l, _l, _vp],
    "xgpr_srht_f64": [_vp, _vp, _l, _l, _l, _vp],
    "xgpr_srht_sample_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _vp, _sz, _vp],
    "xgpr_srht_sample_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _vp, _sz, _vp],
    "xgpr_rbf_feature_gen_f32": [_vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_rbf_feature_gen_f64": [_vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_rbf_grad_f32": [_vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _d, _i, _vp, _sz, _vp],
    "xgpr_rbf_grad_f64": [_vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _d, _i, _vp, _sz, _vp],
    "xgpr_mini_ard_grad_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _l, _l, _l, _i, _vp],
    "xgpr_mini_ard_grad_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _l, _l, _l, _i, _vp],
    "xgpr_conv1d_fgen_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _i, _i, _vp, _sz, _vp],
    "xgpr_conv1d_fgen_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _i, _i, _vp, _sz, _vp],
    "xgpr_conv_grad_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _l, _l, _d, _i, _i,
                           _vp, _sz, _vp],
    "xgpr_conv_grad_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _l, _l, _d, _i, _i,
                           _vp, _sz, _vp],
    "xgpr_conv1d_maxpool_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_conv1d_maxpool_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_ztz_matvec_f32": [_vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_zty_f32": [_vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_cg_step1_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _l, _d, _vp, _vp],
    "xgpr_cg_step2_f64": [_vp, _vp, _vp, _vp, _vp, _l, _d, _vp],
    "xgpr_softmax_residual_f64": [_vp, _vp, _l, _l, _vp, _vp],
    "xgpr_precond_utr_block_f64": [_vp, _vp, _vp, _l, _l, _l, _vp, _sz, _vp],
    "xgpr_precond_apply_block_f64": [_vp, _vp, _d, _vp, _vp, _l, _l, _l, _vp, _sz, _vp],
    "xgpr_cg_step1_block_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _l, _l, _vp, _sz, _vp],
    "xgpr_cg_step2_block_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _vp, _sz, _vp],
    "xgpr_precond_apply_f64": [_vp, _vp, _d, _vp, _vp, _l, _l, _vp, _sz, _vp],
    "xgpr_rbf_feature_cache_f32": [_vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _vp, _sz, _vp],
    "xgpr_zcache_matvec_f32": [_vp, _vp, _vp, _l, _l, _i, _vp, _sz, _vp],
    "xgpr_zcache_matvec_scaled_f32": [_vp, _vp, _vp, _l, _l, _d, _vp, _sz, _vp],
    "xgpr_zcache_block_matvec_f32": [_vp, _vp, _vp, _l, _l, _l, _i, _d, _i, _vp, _sz, _vp],
    "xgpr_zcache_block_project_f32": [_vp, _vp, _vp, _l, _l, _l, _i, _d, _vp, _sz, _vp],
    "xgpr_zcache_block_backproject_f32": [_vp, _vp, _vp, _l, _l, _l, _i, _d, _i, _vp, _sz, _vp],
    "xgpr_srht_sample_rows_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _l, _l, _l, _l, _d, _i, _vp, _sz, _vp],
    "xgpr_sketch_gemm_f64": [_vp, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _d, _i, _i, _vp, _sz, _vp],
    "xgpr_ztz_gram_f64": [_vp, _l, _l, _vp, _l, _l, _d, _i, _i, _vp, _sz, _vp],
    "xgpr_selftest_lane_xor": [_vp, _vp],
    "xgpr_ztz_matvec_plan": [_l, _l],
    "xgpr_rccl_load": [C.c_char_p],
    "xgpr_rccl_unique_id": [_vp],
    "xgpr_rccl_comm_init": [_vp, _i, _vp, _i],
    "xgpr_allreduce_sum_f64": [_vp, _vp, _l, _vp],
    "xgpr_rccl_comm_destroy": [_vp],
}
SIZE_FUNCS = {
    "xgpr_rbf_workspace_bytes": [_l],
    "xgpr_sorf_workspace_bytes": [_l, _l, _i],
    "xgpr_conv_workspace_bytes": [_l, _l, _i, _l],
    "xgpr_precond_apply_workspace_bytes": [_l],
    "xgpr_precond_utr_block_workspace_bytes": [_l, _l, _l],
    "xgpr_precond_apply_block_workspace_bytes": [_l, _l, _l],
    "xgpr_cg_block_workspace_bytes": [_l, _l],
    "xgpr_ztz_matvec_workspace_bytes": [_l, _l],
    "xgpr_zcache_block_workspace_bytes": [_l, _l, _l],
    "xgpr_zcache_block_project_workspace_bytes": [_l, _l, _l],
    "xgpr_srht_sample_workspace_bytes": [_l],
    "xgpr_sketch_gemm_workspace_bytes": [_l, _l, _l, _l, _i],
    "xgpr_ztz_gram_workspace_bytes": [_l, _l],
}
STRING_FUNCS = ["xgpr_last_error", "xgpr_build_arch", "xgpr_build_id"]

_lib = None


def load():
    """Load libxgpr_hip.so.  Raises ImportError (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64 and must be the HIP runtime of the process; loading this
    # library before torch would bind it to the system copy, and the second runtime then sees no device
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python xgpr_amd/build.py`); xgpr_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in SIZE_FUNCS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_size_t
    for name in STRING_FUNCS:
        fn = getattr(lib, name)
        fn.argtypes = []
        fn.restype = C.c_char_p
    _lib = lib
    return lib


def build_id():
    """sha256 of the sources and flags the loaded library was compiled from (xgpr_amd/build.py source_id())."""
    return load().xgpr_build_id().decode()


def last_error():
    return load().xgpr_last_error().decode()


def check(rc):
    """Map a negative return code to the RuntimeError the reference would have thrown."""
    if rc != 0:
        raise RuntimeError(last_error() or f"xgpr_hip error {rc}")
    return 0
