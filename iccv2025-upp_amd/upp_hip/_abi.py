"""ctypes binding of the C ABI declared in include/upp_hip.h.

There is NO fallback: if libupp_hip.so is missing or an entry point is absent,
importing an operator raises.  The operators only accept tensors that live on
a HIP device; CPU tensors are rejected loudly (the CPU restatement under
oracle/ is test infrastructure and is never reachable from here).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libupp_hip.so")

_c_f = ctypes.c_void_p   # device pointers travel as void*
_c_i = ctypes.c_int

# name -> (restype, argtypes); must list every symbol of include/upp_hip.h
SIGNATURES = {
    "upp_abi_version": (_c_i, []),
    "upp_error_string": (ctypes.c_char_p, [_c_i]),
    "upp_set_option": (_c_i, [_c_i, _c_i]),
    "upp_get_option": (_c_i, [_c_i]),
    "upp_fps": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f]),
    "upp_fps_ex": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_gather_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_gather_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_knn": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_knn_ex": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_group_fwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_group_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_fps_gather_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f]),
    "upp_chamfer_fwd": (_c_i, [_c_f] * 6 + [_c_i] * 3 + [_c_f]),
    "upp_chamfer_bwd": (_c_i, [_c_f] * 8 + [_c_i] * 3 + [_c_f]),
    "upp_chamfer_loss_work_floats": (ctypes.c_longlong, []),
    "upp_chamfer_loss": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_f]),
    "upp_emd_work_floats": (ctypes.c_longlong, [_c_i, _c_i, _c_i]),
    "upp_emd_approxmatch": (_c_i, [_c_f] * 4 + [_c_i] * 3 + [_c_f]),
    "upp_emd_matchcost": (_c_i, [_c_f] * 4 + [_c_i] * 3 + [_c_f]),
    "upp_emd_matchcost_bwd": (_c_i, [_c_f] * 6 + [_c_i] * 3 + [_c_f]),
    "upp_patch_embed_work_floats": (ctypes.c_longlong, [_c_i, _c_i]),
    "upp_patch_embed_fwd": (_c_i, [_c_f, _c_i, _c_i] + [_c_f] * 6 + [_c_f] * 2 + [_c_f] * 6 + [_c_f] * 2 + [_c_i]
                            + [ctypes.c_float, ctypes.c_float, _c_i] + [_c_f, _c_f, _c_f]),
    "upp_rowln_fwd": (_c_i, [_c_f] * 3 + [_c_i] * 2 + [_c_f] * 3 + [ctypes.c_float] + [_c_f] * 2 + [ctypes.c_float] + [_c_f] * 4
                      + [_c_i] * 4 + [_c_f]),
    "upp_bias_gelu_fwd": (_c_i, [_c_f] * 3 + [ctypes.c_longlong, _c_i, _c_f]),
    "upp_bias_gelu_fwd_d": (_c_i, [_c_f] * 4 + [ctypes.c_longlong, _c_i, _c_f]),
    "upp_bias_gelu_bwd": (_c_i, [_c_f] * 4 + [ctypes.c_longlong, _c_i, _c_f]),
    "upp_rowln_part_floats": (ctypes.c_longlong, [_c_i] * 5),
    "upp_rowln_bwd": (_c_i, [_c_f] * 6 + [_c_i] + [_c_f] + [ctypes.c_float] + [_c_f] * 4 + [_c_i] * 5 + [_c_f]),
    "upp_ln_param_grad": (_c_i, [_c_f] * 5 + [_c_i] * 3 + [_c_f]),
    "upp_attn_fwd": (_c_i, [_c_f] * 3 + [_c_i] * 4 + [ctypes.c_float, _c_f]),
    "upp_attn_bwd": (_c_i, [_c_f] * 5 + [_c_i] * 4 + [ctypes.c_float, _c_f]),
    "upp_prop_pool_fwd": (_c_i, [_c_f] * 3 + [ctypes.c_float] + [_c_f] * 2 + [_c_i] * 2 + [_c_f]),
    "upp_prop_pool_bwd": (_c_i, [_c_f] * 4 + [ctypes.c_float] + [_c_f] + [_c_i] * 3 + [_c_f]),
    "upp_prop_interp_fwd": (_c_i, [_c_f] * 6 + [_c_i] * 5 + [_c_f]),
    "upp_prop_interp_bwd": (_c_i, [_c_f] * 6 + [_c_i] * 5 + [_c_f]),
    "upp_batched_sum": (_c_i, [ctypes.POINTER(ctypes.c_void_p)] * 2 + [ctypes.POINTER(ctypes.c_int)] * 6 + [_c_i, _c_f]),
    "upp_cls_pool_fwd": (_c_i, [_c_f] * 3 + [ctypes.c_float] + [_c_f] * 4 + [_c_i] * 3 + [_c_f]),
    "upp_cls_pool_bwd": (_c_i, [_c_f] * 7 + [_c_i] * 3 + [_c_f]),
    "upp_ce_acc": (_c_i, [_c_f] * 4 + [_c_i] * 2 + [_c_f]),
    "upp_logsoftmax_rows_fwd": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_i, _c_f, _c_f]),
    "upp_logsoftmax_rows_bwd": (_c_i, [_c_f, _c_f, ctypes.c_longlong, _c_i, _c_f, ctypes.c_longlong, _c_i, _c_f]),
    "upp_nll_mean_part_floats": (ctypes.c_longlong, [ctypes.c_longlong]),
    "upp_nll_mean_fwd": (_c_i, [_c_f, _c_f, ctypes.c_longlong, _c_i, _c_f, _c_f, _c_f]),
    "upp_nll_mean_bwd": (_c_i, [_c_f, _c_f, ctypes.c_longlong, _c_i, _c_f, _c_f]),
    "upp_noise_loss_part_floats": (ctypes.c_longlong, [_c_i, _c_i]),
    "upp_noise_loss_fwd": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f]),
    "upp_noise_loss_bwd": (_c_i, [_c_f, _c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "upp_bn_relu_drop_fwd": (_c_i, [_c_f] * 5 + [ctypes.c_float] * 2 + [_c_i] + [_c_f] + [ctypes.c_float] + [_c_f] * 3 + [_c_i] * 2 + [_c_f]),
    "upp_bn_relu_drop_bwd": (_c_i, [_c_f] * 6 + [_c_i] + [_c_f] + [ctypes.c_float] + [_c_f] * 3 + [_c_i] * 2 + [_c_f]),
    "upp_csr_build": (_c_i, [_c_f] + [_c_i] * 4 + [_c_f] * 3),
    "upp_prop_index": (_c_i, [_c_f] * 4 + [_c_i] * 6 + [ctypes.c_float] + [_c_f] * 5),
    "upp_bn_rows_part_floats": (ctypes.c_longlong, [_c_i] * 2),
    "upp_bn_rows_fwd": (_c_i, [_c_f] * 5 + [ctypes.c_float] * 2 + [_c_i] * 2 + [_c_f] * 4 + [_c_i] * 2 + [_c_f]),
    "upp_bn_rows_bwd": (_c_i, [_c_f] * 6 + [_c_i] + [_c_f] * 4 + [_c_i] * 2 + [_c_f]),
    "upp_bn_rows_drop_fwd": (_c_i, [_c_f] * 5 + [ctypes.c_float] * 2 + [_c_i] + [ctypes.c_float, _c_f, ctypes.c_longlong, ctypes.c_uint] + [_c_f] * 4 + [_c_i] * 2 + [_c_f]),
    "upp_bn_rows_drop_bwd": (_c_i, [_c_f] * 6 + [_c_i] + [ctypes.c_float, _c_f, ctypes.c_longlong, ctypes.c_uint] + [_c_f] * 4 + [_c_i] * 2 + [_c_f]),
    "upp_sqdist_topk": (_c_i, [_c_f] * 4 + [_c_i] * 4 + [_c_f]),
    "upp_interp_fwd": (_c_i, [_c_f] * 2 + [_c_i] + [_c_f] * 2 + [_c_i] * 7 + [ctypes.c_float] + [_c_f]),
    "upp_interp_affine_fwd": (_c_i, [_c_f] * 2 + [_c_i] + [_c_f] * 4 + [_c_i] * 5 + [ctypes.c_float] + [_c_f]),
    "upp_interp_bwd": (_c_i, [_c_f] * 2 + [_c_i] + [_c_f] + [_c_i] * 2 + [_c_f] + [_c_i] * 5 + [ctypes.c_float] + [_c_f]),
    "upp_interp_geo_bwd": (_c_i, [_c_f, _c_f, _c_i, _c_f, _c_f, _c_i, _c_i, _c_f, _c_f] + [_c_i] * 5 + [ctypes.c_float] + [_c_f] * 4),
    "upp_posenc_fwd": (_c_i, [_c_f, ctypes.POINTER(ctypes.c_float), _c_i, _c_f, _c_i, _c_i, ctypes.c_longlong, _c_f]),
    "upp_prop_part_floats": (ctypes.c_longlong, [_c_i] * 2),
    "upp_prop_fwd": (_c_i, [_c_f] * 3 + [ctypes.c_float] + [_c_f] * 7 + [ctypes.c_float] * 2 + [_c_i] + [_c_f] * 6 + [_c_i] * 5 + [_c_f]),
    "upp_prop_w8_grad": (_c_i, [_c_f] * 10 + [_c_i] * 5 + [_c_f]),
    "upp_prop_weights_bwd": (_c_i, [_c_f] * 4 + [ctypes.c_float] + [_c_f] * 2 + [_c_i] * 3 + [_c_f]),
    "upp_prop_bwd": (_c_i, [_c_f] * 7 + [ctypes.c_float] + [_c_f] * 7 + [_c_i] + [_c_f] * 5 + [_c_i] * 5 + [_c_f]),
    "upp_adapter_part_floats": (ctypes.c_longlong, [_c_i, _c_i]),
    "upp_adapter_fwd": (_c_i, [_c_f] * 7 + [ctypes.c_float] * 2 + [_c_f] * 2 + [_c_i] * 3 + [_c_f]),
    "upp_adapter_bwd": (_c_i, [_c_f] * 6 + [ctypes.c_float] * 2 + [_c_f] * 2 + [_c_i] * 3 + [_c_f]),
    "upp_ln_adapter_fwd": (_c_i, [_c_f] * 4 + [ctypes.c_float, _c_i, _c_i, _c_f, _c_f, ctypes.c_float] + [_c_f] * 5 + [ctypes.c_float] * 2
                           + [_c_f] * 5 + [_c_i] * 5 + [_c_f]),
    "upp_ln_adapter_bwd": (_c_i, [_c_f] * 10 + [ctypes.c_float] * 2 + [_c_f] * 2 + [_c_i] * 3 + [_c_f]),
    "upp_ln_adapter_part_floats": (ctypes.c_longlong, [_c_i, _c_i]),
    "upp_ln_adapter_bwd_fused": (_c_i, [_c_f] * 10 + [ctypes.c_float] * 2 + [_c_f, ctypes.c_float, _c_i, _c_i] + [_c_f] * 4 + [_c_i] * 5 + [_c_f]),
    "upp_ln_adapter_bwd_factors": (_c_i, [_c_f] * 10 + [ctypes.c_float] * 2 + [_c_f, ctypes.c_float, _c_i, _c_i] + [_c_f] * 4 + [_c_i] * 5 + [_c_f]),
    "upp_adapter_wgrad_splits": (_c_i, [_c_i]),
    "upp_adapter_wgrad_batched": (_c_i, [ctypes.POINTER(ctypes.c_void_p)] * 7 + [ctypes.POINTER(_c_i), ctypes.POINTER(ctypes.c_float)] +
                                  [ctypes.POINTER(ctypes.c_void_p)] + [_c_i] * 4 + [_c_f]),
    "upp_adamw_scratch_floats": (ctypes.c_longlong, []),
    "upp_colsum_partials": (_c_i, [_c_f, ctypes.c_longlong, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "upp_copy_batched": (_c_i, [ctypes.POINTER(ctypes.c_void_p)] * 2 + [ctypes.POINTER(ctypes.c_longlong), _c_i, _c_f]),
    "upp_group_max_fwd": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f]),
    "upp_group_max_bwd": (_c_i, [_c_f, _c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "upp_argsort_rows": (_c_i, [_c_f, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "upp_rectify_select": (_c_i, [_c_f] * 6 + [ctypes.c_float, ctypes.c_float, _c_f, ctypes.c_float, _c_i, _c_i, _c_i] + [_c_f] * 6),
    "upp_wcolsum_partials": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "upp_linear_tile": (_c_i, [_c_i, _c_i, _c_i]),
    "upp_linear_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_f, _c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong]
                       + [_c_i] * 5 + [_c_f]),
    "upp_linear_sb_tile": (_c_i, [_c_i, _c_i, _c_i]),
    "upp_linear_sb_planes_bytes": (ctypes.c_longlong, [_c_i, _c_i]),
    "upp_linear_sb_prep": (_c_i, [_c_f, ctypes.c_longlong, _c_i, _c_i, _c_i, _c_f, _c_f]),
    "upp_linear_sb_prep_batched": (_c_i, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_longlong)] + [ctypes.POINTER(ctypes.c_int)] * 3
                                   + [ctypes.POINTER(ctypes.c_void_p), _c_i, _c_f]),
    "upp_linear_sb_group_bias_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, _c_f, _c_i, _c_f, ctypes.c_longlong, _c_i, _c_i, _c_i, _c_f]),
    "upp_linear_sb_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, _c_f, _c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong] + [_c_i] * 5 + [_c_f]),
    "upp_linear_group_bias_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_f, _c_i, _c_f, ctypes.c_longlong, _c_i, _c_i, _c_i, _c_f]),
    "upp_linear_smallk_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_f, _c_f, ctypes.c_longlong] + [_c_i] * 4 + [_c_f]),
    "upp_linear_smallk_gelu_d_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_f, _c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_i, _c_i, _c_i, _c_f]),
    "upp_linear_smallk_wgrad_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_f, _c_i, _c_i, _c_i, _c_i, _c_f]),
    "upp_transpose_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_i, _c_i, _c_f]),
    "upp_transpose_batched_f32": (_c_i, [_c_f, _c_f, _c_f, _c_f, _c_i, _c_f]),
    "upp_linear_wgrad_grouped_rows": (_c_i, [_c_i] + [ctypes.POINTER(ctypes.c_int)] * 4),
    "upp_linear_wgrad_grouped_sb_rows": (_c_i, [_c_i] + [ctypes.POINTER(ctypes.c_int)] * 4),
    "upp_linear_wgrad_grouped_f32": (_c_i, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p)] + [ctypes.POINTER(ctypes.c_int)] * 4
                                     + [_c_i, _c_f]),
    "upp_linear_wgrad_grouped_sb": (_c_i, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_void_p)] + [ctypes.POINTER(ctypes.c_int)] * 4
                                     + [_c_i, _c_f]),
    "upp_linear_wgrad_splits": (_c_i, [_c_i, _c_i, _c_i]),
    "upp_linear_wgrad_f32": (_c_i, [_c_f, ctypes.c_longlong, _c_f, ctypes.c_longlong, _c_f, _c_i, _c_i, _c_i, _c_f]),
    "upp_adamw_flat": (_c_i, [_c_f] * 4 + [ctypes.c_longlong] * 2 + [_c_f] * 2 + [ctypes.c_float] * 6 + [_c_f]),
}

_lib = None


OPTIONS = {"SB_TUNED": 0, "SB_XCD2D": 1, "STORE_WT": 2, "EMBED_SPLIT_BF16": 3}      # include/upp_hip.h UPP_OPT_*
ABI_VERSION = 5            # include/upp_hip.h UPP_ABI_VERSION


def load():
    """Load libupp_hip.so once; raise if it (or any declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libupp_hip.so is not built (%s). Run `python __graft_entry__.py` or "
            "`python iccv2025-upp_amd/upp_hip/build.py`; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> loud failure on a stale build
        fn.restype = res
        fn.argtypes = args
    if lib.upp_abi_version() != ABI_VERSION:
        raise RuntimeError("libupp_hip.so ABI version mismatch")
    # The library never reads the environment (include/upp_hip.h "options"); the A/B scripts under tools/ set these variables, and the
    # host forwards them ONCE, here.  Inside a process use ops.option(name, value).
    for name, key in OPTIONS.items():
        v = os.environ.get("UPP_" + name)
        if v is not None and v.strip().lstrip("-").isdigit():
            rc = lib.upp_set_option(key, int(v))
            if rc != 0:
                raise RuntimeError("UPP_%s=%s: %s" % (name, v, lib.upp_error_string(rc).decode()))
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise RuntimeError(load().upp_error_string(code).decode())


def ptr(t):
    """Device pointer of a tensor, or NULL for None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
