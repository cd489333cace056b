"""ctypes binding of libsincformer_hip.so (C ABI: include/sincformer_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
is absent, importing/using the ops raises.  (The oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes
import os

# torch must be imported BEFORE the shared library is dlopen'ed: torch ships its own
# libamdhip64 (same SONAME as the system one).  Loading ours first would pull a second HIP
# runtime into the process and every launch on a torch stream would then fail.
import torch  # noqa: F401  (device memory / stream plumbing; see ops.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFM_LIB_PATH") or os.path.join(_HERE, "libsincformer_hip.so")   # override: A/B of two builds

c_vp = ctypes.c_void_p
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float

# name -> argtypes, in the order of include/sincformer_hip.h
SIGNATURES = {
    "sfm_abi_version": [],
    "sfm_gemm16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i,
                   c_i, c_ll, c_i, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_gemm16_ex": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i,
                   c_i, c_ll, c_i, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_gemm16_train": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i,
                   c_i, c_ll, c_i, c_ll, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.c_uint, c_vp],
    "sfm_conv16p": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i,
                    c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_framed_gemm_f32": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_i, c_i,
                            c_i, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_i, c_vp],
    "sfm_attention_fwd": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_ll, c_f, c_i, c_vp],
    "sfm_attention_fwd_ex": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_ll, c_f, c_i, c_i, c_i, c_vp],
    "sfm_layernorm": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_vp],
    "sfm_gn_finalize": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_ll, c_f, c_vp],
    "sfm_gn_apply": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_ll, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_dwconv_bn_swish": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_i, c_vp],
    "sfm_dwconv_folded": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_convert_rows": [c_vp, c_vp, c_ll, c_i, c_i, c_ll, c_ll, c_i, c_vp],
    "sfm_transpose": [c_vp, c_vp, c_i, c_i, c_i, c_ll, c_ll, c_ll, c_ll, c_i, c_i, c_i, c_vp],
    "sfm_pool_time": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_ll, c_ll, c_i, c_vp],
    "sfm_mean_time_scratch_floats": [c_i, c_i, c_i],
    "sfm_mean_time": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_ll, c_vp],
    "sfm_sum_time": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_ll, c_vp],
    "sfm_pool_time_bwd": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_ll, c_ll, c_vp],
    "sfm_pool_time_affine": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_ll, c_ll, c_i, c_vp],
    "sfm_pool_time_affine16": [c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_ll, c_ll, c_i, c_vp],
    "sfm_stft_lognorm_pack": [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_ll, c_i, c_vp],
    "sfm_stft_lognorm_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_ll, c_vp],
    "sfm_polar_mask": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_ll, c_i, c_f, c_ll, c_ll,
                       c_vp],
    "sfm_complex_mul": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_vp],
    "sfm_istft_ola": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_vp],
    "sfm_pack_spec": [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_ll, c_vp],
    "sfm_sinc_filters": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_f, c_f, c_f, c_vp],
    "sfm_ffn_fused": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_f, c_f, c_i, c_vp],
    "sfm_ffn_fused_ln": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_f, c_f, c_vp, c_vp, c_vp, c_i, c_i,
                         c_vp],
    "sfm_sinc_fir16_tiles": [c_i],
    "sfm_sinc_fir16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_sinc_fir16_ex": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_framed_gemm_split16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll,
                                c_ll, c_i, c_vp],
    "sfm_wave_moments": [c_vp, c_vp, c_vp, c_i, c_i, c_vp, c_vp],
    "sfm_spec_sums": [c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_vp],
    "sfm_enhancer_loss_finalize": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_ll, c_i, c_vp, c_vp],
    "sfm_sisnr_bwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_f, c_vp],
    "sfm_spec_loss_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_ll, c_i, c_i, c_f, c_vp],
    "sfm_stft_adjoint_ola": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_polar_mask_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_ll, c_i, c_f, c_ll, c_ll, c_vp],
    "sfm_attention_bwd_generic": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.c_uint, c_i,
                                  c_vp],
    "sfm_ssnr_frames": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_f, c_vp],
    "sfm_stoi_frames": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "sfm_sinc_wgrad_scratch_floats": [c_i, c_i, c_i, c_i],
    "sfm_sinc_wgrad": [c_vp, c_vp, c_i, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_sinc_shift_len": [c_i],
    "sfm_sinc_shift_pack": [c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "sfm_sinc_wgrad16": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_ll, c_vp],
    "sfm_gemm16_swish": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, ctypes.c_uint, c_i, c_vp],
    "sfm_gn_bwd_reduce": [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i,
                          c_i, c_i, c_i, c_vp, c_vp],
    "sfm_gn_bwd_coefs": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "sfm_gn_bwd_apply": [c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_vp, c_i, c_vp, c_vp, c_vp, c_vp,
                         c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_maa_update_stats": [c_vp, c_ll, c_vp, c_vp, c_vp, c_f, c_vp],
    "sfm_maa_forward": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_vp],
    "sfm_maa_backward": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_i, c_vp],
    "sfm_vq_forward": [c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_ll, c_vp],
    "sfm_vq_backward": [c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_f, c_vp, c_vp, c_ll, c_vp],
    "sfm_sumsq": [c_vp, c_ll, c_vp, c_vp, c_vp],
    "sfm_adamw_step": [c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_vp],
    "sfm_adamw_step_masked": [c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_vp, c_vp, c_i, c_vp],
    "sfm_adamw_step_scaled": [c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_vp, c_vp, c_i, c_vp,
                              c_f, c_f, c_i, c_vp],
    "sfm_tn_ws_floats": [c_i, c_i, c_i],
    "sfm_colsum_ws_floats": [c_i, c_i],
    "sfm_layernorm_bwd_ws_floats": [c_i, c_i],
    "sfm_col_stats_ws_floats": [c_i, c_i],
    "sfm_gn_bwd_reduce_ws_floats": [c_i, c_i, c_i],
    "sfm_gemm16_tn": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_ll, c_vp],
    "sfm_conv_wgrad16": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_i, c_i, c_i, c_vp, c_ll, c_vp],
    "sfm_colsum": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "sfm_layernorm_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp, c_vp],
    "sfm_layernorm_bwd_ex": [c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_vp, c_vp],
    "sfm_layernorm_bwd_next": [c_vp, c_vp, c_vp, c_i, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_vp, c_f, c_f,
                               ctypes.c_uint, c_vp, c_vp],
    "sfm_ew_train": [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_i, c_i, c_f, c_f, ctypes.c_uint, c_i, c_vp],
    "sfm_col_stats": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp, c_vp],
    "sfm_add_cols": [c_vp, c_vp, c_vp, c_ll, c_i, c_i, c_ll, c_ll, c_ll, c_vp],
    "sfm_lstm_hprev16": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "sfm_bn_finalize": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_ll, c_f, c_f, c_i, c_vp],
    "sfm_bn_swish_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "sfm_dwconv_wgrad": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_dwconv_wgrad_scratch_floats": [c_i, c_i, c_i, c_i],
    "sfm_attention_fwd_train": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_ll, c_ll, c_f, c_f,
                                ctypes.c_uint, c_i, c_vp],
    "sfm_attention_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f,
                          ctypes.c_uint, c_i, c_vp],
    "sfm_lin256": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_headpool_frames_per_tile": [c_i, c_i],
    "sfm_headpool": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_ln_lin256": [c_vp, c_i, c_vp, c_vp, c_f, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "sfm_bilstm_layer": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "sfm_bilstm_layer_ex": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "sfm_bilstm_layer_train": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "sfm_bilstm_layer_train_ex": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "sfm_bilstm_layer_bwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "sfm_memory_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp],
    "sfm_memory_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_f, c_vp, c_vp],
    "sfm_memory_param_floats": [c_i, c_i, c_i],
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises HipExtensionMissing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            "libsincformer_hip.so not built (%s). Run `python -m sincformer_metacog_speech_enhancement_amd.build` "
            "or __graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionMissing("symbol %s missing from %s" % (name, LIB_PATH)) from e
        fn.argtypes = args
        fn.restype = c_ll if name.endswith("_scratch_floats") or name.endswith("_ws_floats") or name in ("sfm_sinc_shift_len", "sfm_memory_param_floats") else c_i
    _lib = lib
    return lib


_ERR = {-1: "bad argument", -2: "unsupported shape", -3: "kernel launch failed"}


def check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (name, _ERR.get(rc, "error"), rc))
