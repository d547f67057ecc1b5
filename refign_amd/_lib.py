"""ctypes binding of librefign_hip.so (C ABI: include/refign_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C refign_amd/csrc` into refign_amd/lib/.
Loading is lazy and LOUD: if the .so is missing or does not export a declared symbol we raise -- the product path
never falls back to a CPU or PyTorch implementation.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RFN_LIB: another build of the same library, for A/B runs of kernel variants -- tools/ab_build.sh)
_LIB_PATH = os.environ.get("RFN_LIB") or os.path.join(_HERE, "lib", "librefign_hip.so")
_lock = threading.Lock()
_lib = None

c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p

_CORR12 = [c_int] * 12
# name -> (restype, argtypes); must list every entry point declared in include/refign_hip.h
SIGNATURES = {
    "rfn_abi_version": (c_int, []),
    "rfn_last_error": (ctypes.c_char_p, []),
    "rfn_corr_fwd_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + _CORR12 + [c_void_p]),
    "rfn_corr_fwd_f64": (c_int, [c_void_p] * 3 + [c_int] * 4 + _CORR12 + [c_void_p]),
    "rfn_corr_bwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + _CORR12 + [c_void_p]),
    "rfn_corr_bwd_f64": (c_int, [c_void_p] * 5 + [c_int] * 4 + _CORR12 + [c_void_p]),
    "rfn_corr_fwd_f16": (c_int, [c_void_p] * 3 + [c_int] * 4 + _CORR12 + [c_void_p]),
    "rfn_corr_bwd_f16": (c_int, [c_void_p] * 5 + [c_int] * 4 + _CORR12 + [c_void_p]),
    "rfn_local_corr_layer_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "rfn_local_corr_layer_split_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "rfn_local_corr_layer_split_workspace_bytes": (ctypes.c_long, [c_int] * 4),
    "rfn_global_corr_layer_f32": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "rfn_warp_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "rfn_warp_bwd_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "rfn_l2norm_channels_f32": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "rfn_l2norm_channels_nhwc16_f32": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "rfn_maxpool2x2_nhwc16": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "rfn_retile_copy": (c_int, [c_void_p] * 2 + [c_int] * 7 + [c_void_p]),
    "rfn_area_resize_f32": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "rfn_refine_workspace_bytes": (ctypes.c_ulong, [c_int]),
    "rfn_label_majority": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_float, c_void_p]),
    "rfn_refine_f32": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "rfn_align_tail_f32": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "rfn_dwconv3x3_nhwc_stats": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "rfn_dwconv3x3_tri_usable": (c_int, [c_int] * 5),
    "rfn_dwconv3x3_tri_stats": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "rfn_dwconv3x3_tri_bn_act_fwd": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p, c_void_p, c_int, c_void_p]),
    "rfn_dwconv3x3_bn_act_nhwc_fwd": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_float, c_float, c_int, c_int, c_void_p]),
    "rfn_dwconv3x3_nhwc_fwd_stats": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "rfn_dwconv3x3_nhwc_fwd": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "rfn_dwconv3x3_gelu_nhwc_fwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "rfn_layernorm_fwd": (c_int, [c_void_p] * 6 + [ctypes.c_long, c_int, c_float, c_int, c_int, c_void_p]),
    "rfn_layernorm_bwd_workspace_bytes": (ctypes.c_ulong, [c_int]),
    "rfn_layernorm_bwd": (c_int, [c_void_p] * 9 + [ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "rfn_layernorm_bwd_add": (c_int, [c_void_p] * 10 + [ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "rfn_layernorm_bwd_add2": (c_int, [c_void_p] * 11 + [ctypes.c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "rfn_dwconv3x3_bwd_weight_workspace_bytes": (ctypes.c_ulong, [c_int]),
    "rfn_dwconv3x3_nhwc_bwd_weight": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "rfn_sum_rows_workspace_bytes": (ctypes.c_ulong, [ctypes.c_long, ctypes.c_long]),
    "rfn_sum_rows": (c_int, [c_void_p] * 3 + [ctypes.c_long, ctypes.c_long, c_int, c_int, c_void_p]),
    "rfn_linear_param_grads": (c_int, [c_void_p] * 3 + [ctypes.c_long, ctypes.c_long, c_int, c_void_p, c_void_p, c_int,
                                       ctypes.c_long, c_int, c_int, c_void_p]),
    "rfn_upsample_concat_nhwc": (c_int, [c_void_p] * 4 + [ctypes.POINTER(c_int)] * 3 + [c_int, c_void_p] + [c_int] * 4
                                 + [c_void_p]),
    "rfn_upsample_concat_nhwc_bwd": (c_int, [c_void_p] * 5 + [ctypes.POINTER(c_int)] * 3 + [c_int] * 5 + [c_void_p]),
    "rfn_patchify_tokens": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "rfn_patchify_tokens_cmajor": (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "rfn_multi_cast_chunk_elems": (c_int, []),
    "rfn_multi_transpose_tile": (c_int, []),
    "rfn_multi_permute_chunk_elems": (c_int, []),
    "rfn_multi_permute_cast_f32": (c_int, [c_void_p, c_int, c_void_p]),
    "rfn_multi_cast_f32_bf16": (c_int, [c_void_p, c_int, c_void_p]),
    "rfn_multi_ema_f32": (c_int, [c_void_p, c_int, c_float, c_void_p]),
    "rfn_multi_transpose_cast_f32_bf16": (c_int, [c_void_p, c_int, c_void_p]),
    "rfn_multi_adamw_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "rfn_gemm_nt": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p] + [ctypes.c_long] * 6 + [c_int, c_void_p]),
    "rfn_conv2d_nhwc": (c_int, [c_void_p] * 4 + [c_int, c_void_p] + [c_int] * 10 + [ctypes.c_long, ctypes.c_long, c_int,
                                                                                       c_void_p]),
    "rfn_gemm_nt_o32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p] + [ctypes.c_long] * 6 + [c_void_p]),
    "rfn_conv2d_nhwc_o32": (c_int, [c_void_p] * 3 + [c_int, c_void_p] + [c_int] * 10 + [ctypes.c_long, ctypes.c_long, c_int,
                                                                                           c_void_p]),
    "rfn_conv2d_nhwc_dgrad": (c_int, [c_void_p] * 3 + [c_int] * 10 + [ctypes.c_long, ctypes.c_long, c_int, c_void_p]),
    "rfn_conv2d_nhwc_wgrad": (c_int, [c_void_p] * 4 + [c_int] * 10 + [ctypes.c_long, ctypes.c_long, c_int, c_int, c_int,
                                      c_void_p]),
    "rfn_gemm_tn": (c_int, [c_void_p] * 3 + [ctypes.c_long] * 5 + [c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                            c_void_p]),
    "rfn_gemm_tn_grouped": (c_int, [c_int] + [c_void_p] * 12 + [c_int, c_void_p]),
    "rfn_attn_pack": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long] + [c_int] * 4 + [c_void_p] * 6),
    "rfn_attn_fwd": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long, c_void_p, c_void_p, c_void_p, ctypes.c_long,
                             ctypes.c_long, c_void_p] + [c_int] * 6 + [c_float, c_int, c_int, c_void_p]),
    "rfn_attn_bwd_dq": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long, c_void_p, c_void_p, ctypes.c_long, ctypes.c_long]
                        + [c_void_p] * 6 + [ctypes.c_long, ctypes.c_long] + [c_int] * 6 + [c_float, c_int, c_int,
                                                                                         c_void_p]),
    "rfn_attn_bwd_dkv": (c_int, [c_void_p, c_void_p, ctypes.c_long, ctypes.c_long] + [c_void_p] * 8 + [c_int] * 8
                         + [c_float, c_int, c_void_p]),
    "rfn_attn32_fwd": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long] * 3 + [c_void_p] + [c_int] * 6 + [c_float, c_void_p]),
    "rfn_attn32_bwd": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long] * 2 + [c_void_p, c_void_p, ctypes.c_long, ctypes.c_long,
                                                                                c_void_p, c_void_p]
                       + [c_void_p, ctypes.c_long, ctypes.c_long] * 2 + [c_int] * 7 + [c_float, c_void_p]),
    "rfn_split3_bf16": (c_int, [c_void_p, ctypes.c_long, c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, c_int, c_int,
                                c_int, c_void_p]),
    "rfn_ffn_fc1_dw_gelu_bf16": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    "rfn_split3_cat_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rfn_upsample_ce": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    "rfn_bn_stats_fwd": (c_int, [c_void_p] * 2 + [ctypes.c_long, c_int, c_int, c_void_p]),
    "rfn_bn_apply_fwd": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "rfn_bn_apply_fwd_ld": (c_int, [c_void_p] * 4 + [ctypes.c_long] + [c_void_p] * 3 + [ctypes.c_long, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "rfn_bn_stats_bwd": (c_int, [c_void_p] * 6 + [ctypes.c_long, c_int, c_float, c_int, c_int, c_void_p]),
    "rfn_bn_apply_bwd": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_float, c_int, c_int, c_void_p]),
    "rfn_bn_train_fwd": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_float, c_float, c_int, c_int, c_void_p]),
    "rfn_bn_train_bwd": (c_int, [c_void_p] * 7 + [ctypes.c_long, c_int, c_float, c_int, c_int, c_void_p]),
    "rfn_dacs_mix_jitter": (c_int, [c_void_p] * 9 + [c_int] * 3 + [c_void_p] * 7 + [c_void_p]),
    "rfn_crop_label_hist_u8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "rfn_crop_flip_norm_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p] * 4 + [c_void_p]),
    "rfn_dacs_blur": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p] * 3 + [c_void_p]),
    "rfn_gemm_nt_f8": (c_int, [c_void_p] * 3 + [c_float] + [c_void_p] * 3 + [c_int, c_int, c_void_p, c_int, c_float]
                       + [ctypes.c_long] * 6 + [c_void_p]),
    "rfn_quant_rows_f8": (c_int, [c_void_p, c_int, c_void_p]),
    "rfn_quant_f8": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_float, c_void_p]),
    "rfn_layernorm_fwd_f8": (c_int, [c_void_p] * 4 + [ctypes.c_long, c_int, c_float, c_float, c_void_p]),
    "rfn_dwconv3x3_gelu_nhwc_fwd_f8": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_float, c_float, c_void_p]),
    "rfn_attn_pack_f8": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long] + [c_int] * 4 + [c_void_p, c_void_p]),
    "rfn_attn_fwd_f8": (c_int, [c_void_p, ctypes.c_long, ctypes.c_long, c_void_p, c_void_p, ctypes.c_long, ctypes.c_long]
                        + [c_int] * 5 + [c_float] * 5 + [c_void_p]),
    "rfn_uncertainty9_weights_len": (c_int, []),
    "rfn_uncertainty9_frontend_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "rfn_uncertainty9_frontend_f16mm": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
}


# RFN_ABI_VERSION of include/refign_hip.h this table was written against (4: round 6 added rfn_attn32_fwd / _bwd, rfn_split3_bf16,
# rfn_split3_cat_bf16, rfn_ffn_fc1_dw_gelu_bf16; 2: rfn_global_corr_layer_f32 takes a workspace,
# rfn_dacs_mix_jitter accepts one half of the mix; 3: the transpose-cast table holds 64 x 64 tiles, rfn_multi_transpose_tile)
ABI_VERSION = 4


def library_path():
    return _LIB_PATH


def load_library():
    """Load librefign_hip.so and bind every declared entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"refign_amd: HIP library not built: {_LIB_PATH} is missing. Run `python -c 'import "
                f"__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch owns device memory and streams; import it FIRST so that its bundled libamdhip64.so.7 is the one HIP
        # runtime in the process (our .so NEEDs the same SONAME and binds to the already-loaded copy).  Loading ours
        # first would pull /opt/rocm's runtime in beside torch's: two runtimes, "no ROCm-capable device" at launch.
        import torch  # noqa: F401
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RuntimeError(f"refign_amd: {_LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if lib.rfn_abi_version() != ABI_VERSION:
            raise RuntimeError(f"refign_amd: {_LIB_PATH} speaks ABI {lib.rfn_abi_version()}, this package binds ABI {ABI_VERSION} "
                               f"(include/refign_hip.h: RFN_ABI_VERSION): rebuild the library")
        _lib = lib
    return _lib


def abi_version():
    return load_library().rfn_abi_version()


def check(rc, what):
    """Turn a non-zero ABI return code into RuntimeError (what TORCH_CHECK raises in the reference)."""
    if rc != 0:
        msg = load_library().rfn_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
