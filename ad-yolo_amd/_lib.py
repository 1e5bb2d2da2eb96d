"""ctypes binding of libadyolo_hip.so (C ABI declared in include/adyolo_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent, importing
the ops raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc, gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADYOLO_LIB") or os.path.join(_HERE, "libadyolo_hip.so")   # ADYOLO_LIB: an alternative build (A/B runs)

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_long
F = ctypes.c_float
U64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/adyolo_hip.h one to one
SIGNATURES = {
    "adyolo_abi_version": (I, []),
    "adyolo_last_error": (ctypes.c_char_p, []),
    "adyolo_feat_stft_mel": (I, [P] * 8 + [I, I] + [P] * 4 + [I, I, I, P]),
    "adyolo_feat_finish": (I, [P] * 4 + [I, I, I, P]),
    "adyolo_feat_gcc_phat": (I, [P] * 6 + [I, I, I, I, P]),
    "adyolo_nchw_to_nhwc8": (I, [P, P, I, I, I, I, P]),
    "adyolo_pack_w3x3": (I, [P, P, P, I, I, I, P]),
    "adyolo_conv3x3_tiles": (I, [I] * 3),
    "adyolo_conv3x3_fwd": (I, [P] * 13 + [I] * 7 + [P]),
    "adyolo_wino_pack_w": (I, [P, P, P, I, I, I, P]),
    "adyolo_wino_pack_many": (I, [P, I, I, I, P]),
    "adyolo_wino_tiles": (I, [I] * 3),
    "adyolo_wino_fwd": (I, [P] * 13 + [I] * 7 + [P]),
    "adyolo_wino4_pack_w": (I, [P, P, P, I, I, I, P]),
    "adyolo_wino4_pack_many": (I, [P, I, I, I, P]),
    "adyolo_wino4_tiles": (I, [I] * 3),
    "adyolo_wino4_fwd": (I, [P] * 13 + [I] * 7 + [P]),
    "adyolo_wino4_last_form": (I, []),
    "adyolo_reload_switches": (I, []),
    "adyolo_wino4_wgrad_slabs": (I, [I] * 5),
    "adyolo_wino4_wgrad": (I, [P] * 7 + [I] * 6 + [P]),
    "adyolo_wino_wgrad_slabs": (I, [I] * 5),
    "adyolo_wino_wgrad": (I, [P] * 7 + [I] * 6 + [P]),
    "adyolo_conv3x3_wgrad_slabs": (I, [I] * 5),
    "adyolo_conv3x3_wgrad": (I, [P] * 6 + [I] * 6 + [P]),
    "adyolo_gemm": (I, [P] * 5 + [I] * 10 + [P]),
    "adyolo_gemm_batched": (I, [P] * 3 + [I] * 10 + [L] * 6 + [F, I, P]),
    "adyolo_colsum": (I, [P, P, P, I, I, I, I, P]),
    "adyolo_bn_stats": (I, [P] * 7 + [I, I, I, F, F, P]),
    "adyolo_bn_stats_tiles": (I, [P] * 11 + [I, I, I, I, F, F, P]),
    "adyolo_bn_eval_stats": (I, [P] * 4 + [I, F, P]),
    "adyolo_bn_scale_shift": (I, [P] * 6 + [I, P]),
    "adyolo_affine_nhwc": (I, [P] * 4 + [L, I, P]),
    "adyolo_bn_bwd_reduce": (I, [P] * 7 + [L, I, P]),
    "adyolo_bn_bwd_tiles": (I, [P, P, P, P, I, I, P]),
    "adyolo_bn_bwd_apply": (I, [P] * 12 + [L, I, I, F, P]),
    "adyolo_bn_persample": (I, [P, P, P, I, I, I, P]),
    "adyolo_bn_finish": (I, [P] * 10 + [I, I, I, F, F, P]),
    "adyolo_se_fc_fwd": (I, [P] * 10 + [I, I, I, I, P]),
    "adyolo_relu_mask_words": (L, [I, I, I]),
    "adyolo_se_tail_fwd": (I, [P] * 9 + [I, I, I, P]),
    "adyolo_se_tail_fwd_pool_ok": (I, [I, I, I]),
    "adyolo_se_tail_fwd_pool": (I, [P] * 9 + [I, I, I, I, P]),
    "adyolo_se_tail_bwd_reduce": (I, [P] * 9 + [I, I, I, P]),
    "adyolo_se_tail_bwd_tiles": (I, [P, P, P, I, I, I, P]),
    "adyolo_se_fc_bwd_words": (L, [I, I]),
    "adyolo_se_fc_bwd": (I, [P] * 16 + [I, I, I, I, P]),
    "adyolo_se_tail_bwd_apply": (I, [P] * 13 + [I, I, I, F, P]),
    "adyolo_se_tail_bwd_reduce_pooled": (I, [P] * 8 + [I, I, I, I, P]),
    "adyolo_se_tail_bwd_apply_pooled": (I, [P] * 13 + [I, I, I, I, F, P]),
    "adyolo_avgpool2_fwd": (I, [P, P, I, I, I, I, P]),
    "adyolo_avgpool2_bwd": (I, [P, P, I, I, I, I, P]),
    "adyolo_add": (I, [P, P, P, L, P]),
    "adyolo_mul": (I, [P, P, P, L, P]),
    "adyolo_scale_dev": (I, [P, P, P, L, P]),
    "adyolo_sap_fwd": (I, [P] * 5 + [I, I, I, P]),
    "adyolo_sap_bwd": (I, [P] * 8 + [I, I, I, P]),
    "adyolo_gru_fwd": (I, [P] * 6 + [I, I, P]),
    "adyolo_gru_bwd": (I, [P] * 6 + [I, I, P]),
    "adyolo_ln_tanh_fwd": (I, [P] * 4 + [L, I, F, P]),
    "adyolo_ln_tanh_bwd": (I, [P] * 8 + [L, I, F, P]),
    "adyolo_dropout_mask": (I, [P, L, F, U64, U64, P]),
    "adyolo_dropout_apply": (I, [P, P, L, F, U64, U64, P]),
    "adyolo_dropout_apply_dev": (I, [P, P, L, F, U64, U64, P, P]),
    "adyolo_counter_add": (I, [P, U64, P]),
    "adyolo_loss_workspace_words": (L, [I, I, I, I]),
    "adyolo_loss_fwd_bwd": (I, [P] * 6 + [I] * 7 + [P, P, F, F, F, F, P]),
    "adyolo_loss_phase": (I, [P] * 6 + [I] * 7 + [P, P, F, F, F, F, I, L, P]),
    "adyolo_yolo_decode": (I, [P, P, L, I, I, I, I, F, F, F, P]),
    "adyolo_act_fwd": (I, [P, P, L, I, I, P]),
    "adyolo_act_bwd": (I, [P, P, P, L, I, I, P]),
    "adyolo_seddoa_loss": (I, [P] * 5 + [L, I, I, I, F, F, P]),
    "adyolo_adpit_loss": (I, [P] * 5 + [L, I, P]),
    "adyolo_conv_gemm": (I, [P, P, P, P] + [I] * 13 + [P]),
    "adyolo_pack_wk": (I, [P, P, I, I, I, I, I, P]),
    "adyolo_maxpool3_fwd": (I, [P, P, P, I, I, I, I, P]),
    "adyolo_maxpool3_bwd": (I, [P, P, P, I, I, I, I, P]),
    "adyolo_affine_relu_nhwc": (I, [P] * 4 + [L, I, P]),
    "adyolo_relu_bwd": (I, [P, P, P, L, P]),
    "adyolo_axpby": (I, [P, P, P, F, F, L, P]),
    "adyolo_dropout_axpby": (I, [P, P, P, L, F, U64, U64, P, F, F, P]),
    "adyolo_swish_fwd": (I, [P, P, L, P]),
    "adyolo_swish_bwd": (I, [P, P, P, L, P]),
    "adyolo_glu_fwd": (I, [P, P, L, I, P]),
    "adyolo_glu_bwd": (I, [P, P, P, L, I, P]),
    "adyolo_dwconv3_fwd": (I, [P] * 4 + [I, I, I, I, I, P]),
    "adyolo_dwconv3_wgrad": (I, [P] * 6 + [I, I, I, I, P]),
    "adyolo_softmax_fwd": (I, [P, P, L, I, F, P]),
    "adyolo_softmax_bwd": (I, [P, P, P, L, I, F, P]),
    "adyolo_avgpool1d_fwd": (I, [P, P, I, I, I, I, F, P]),
    "adyolo_avgpool1d_bwd": (I, [P, P, I, I, I, I, F, P]),
    "adyolo_wino1d_in": (I, [P, P, I, I, I, P]),
    "adyolo_wino1d_out": (I, [P, P, I, I, I, P]),
    "adyolo_wino1d_dy": (I, [P, P, I, I, I, P]),
    "adyolo_wino1d_filter": (I, [P, P, I, I, I, P]),
    "adyolo_ln_fwd": (I, [P] * 4 + [L, I, F, P]),
    "adyolo_ln_bwd": (I, [P] * 7 + [L, I, F, P]),
    "adyolo_attn_fwd": (I, [P] * 5 + [I, I, I, I, F, F, ctypes.c_uint32, P, P]),
    "adyolo_attn_bwd": (I, [P] * 10 + [I, I, I, I, F, F, ctypes.c_uint32, P, P]),
    "adyolo_seed32_dev": (I, [ctypes.c_uint64, ctypes.c_uint64, P, P, P]),
    "adyolo_attn_dropout_mask": (I, [P, I, I, I, F, ctypes.c_uint32, P]),
    "adyolo_foa_rotate": (I, [P, P, P, I, L, P]),
    "adyolo_pcm16_to_f32": (I, [P, P, L, P]),
    "adyolo_mask_ranges": (I, [P, P, I, I, I, I, P]),
    "adyolo_colstats": (I, [P, P, P, L, I, P]),
    "adyolo_adam_step": (I, [P] * 4 + [L, F, F, F, F, F, I, F, P]),
    "adyolo_adam_step_dev": (I, [P] * 4 + [L, F, F, F, F, F, P, P, F, P]),
}

_lib = None


class AdyoloHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly if it is not built (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own libamdhip64; it must be the HIP runtime already mapped when this library (linked against the
    # same SONAME) is loaded, otherwise the process ends up with two runtimes and this one sees no device
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AdyoloHipError(
            "libadyolo_hip.so is not built (%s). Run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise AdyoloHipError("libadyolo_hip.so does not export %s" % name)
        fn.restype = res
        fn.argtypes = args
    ver = lib.adyolo_abi_version()
    if ver != 1:
        raise AdyoloHipError("libadyolo_hip.so ABI version %d != 1" % ver)
    _lib = lib
    return lib


def loaded():
    return _lib is not None


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.adyolo_last_error()
        raise AdyoloHipError("%s failed (rc=%d): %s" % (name, rc, msg.decode() if msg else "?"))
    return rc
