"""ctypes binding of libevtpretrain.so (include/evtpretrain.h).

There is NO fallback: if the shared library is missing, or a call is made without a HIP device, this module raises.
PyTorch only supplies device memory (`tensor.data_ptr()`), the current stream and `torch.distributed`.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVP_LIB") or os.path.join(_HERE, "libevtpretrain.so")     # EVP_LIB: another build of the same ABI (A/B runs)
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 5            # include/evtpretrain.h EVP_ABI_VERSION: checked when the library is loaded (EVP_LIB overrides included)
EVP_F32, EVP_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_RELU, ACT_DRELU = 0, 1, 2, 3, 4

_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


class GemmDesc(C.Structure):
    _fields_ = [("dtype", _i), ("transA", _i), ("transB", _i), ("M", _i), ("N", _i), ("K", _i),
                ("A", _vp), ("lda", _i64), ("strideA0", _i64), ("strideA1", _i64),
                ("B", _vp), ("ldb", _i64), ("strideB0", _i64), ("strideB1", _i64),
                ("C", _vp), ("c_dtype", _i), ("ldc", _i64), ("strideC0", _i64), ("strideC1", _i64),
                ("batch0", _i), ("batch1", _i), ("alpha", _f), ("bias", _vp), ("act", _i),
                ("aux", _vp), ("ldaux", _i64), ("residual", _vp), ("ldres", _i64),
                ("accumulate", _i), ("tile", _i), ("splitk", _i)]


# name -> argtypes (all return int status unless listed in _OTHER_RESTYPE)
SIGNATURES = {
    "evp_voxel_scatter_f32": [_vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "evp_voxel_scatter_scaled_f32": [_vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp],
    "evp_events_build_added_f64": [_vp, _vp, _i, _vp, _vp, _vp, _i, _d, _d, _vp, _vp],
    "evp_voxel_scatter_fused_f32": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _d, _d, _vp, _i, _i, _i, _vp, _vp, _vp],
    "evp_events_plan_batch": [_vp, _i, _i64, C.c_uint64, _vp, _i, _vp, _i, _i, _i, _i, C.c_double, _vp, _vp, _vp, _vp],
    "evp_events_sorted_check": [_vp, _vp, _i, _i, _vp, _vp],
    "evp_events_erase_add_f64": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp],
    "evp_events_draw_erase_add": [_vp, _vp, _i, _vp, _vp, C.c_uint64, C.c_uint64, _i64, _vp, _i, _vp, _vp, _vp, _vp],
    "evp_events_erase_add_win_f64": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _d, _d, _vp, _vp, _vp, _vp],
    "evp_mask_from_noise": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "evp_density_noise": [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "evp_gemm": [C.POINTER(GemmDesc), _vp],
    "evp_gemm_grouped_tn_bf16": [_vp, _vp, _i, _vp],
    "evp_gemm_grouped_tn_g4_bf16": [_vp, _vp, _i, _vp],
    "evp_sum_slices_f32": [_vp, _vp, _i, _i64, _i, _vp],
    "evp_gemm_set_variant": [_i],
    "evp_voxel_set_debug": [_i],
    "evp_gemm_set_stamp_buffer": [_vp, C.c_longlong],
    "evp_gemm_stamp_count": [],
    "evp_attention_set_debug_buffer": [_vp],
    "evp_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp, _i, _vp, _vp, _vp],
    "evp_layernorm_bwd_nblk": [_i64],
    "evp_layernorm_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "evp_layernorm_bwd_cs": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp],
    "evp_colsum_nblk": [_i64],
    "evp_colsum_grouped": [_vp, _vp, _i, _vp],
    "evp_colsum": [_vp, _i, _i64, _i, _i64, _vp, _vp, _vp],
    "evp_attention_fwd": [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i64, _vp, _vp],
    "evp_attention_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i64, _vp, _vp, _vp, _vp],
    "evp_attention_fused_supported": [_i, _i, _i],
    "evp_attention_fused_fwd": [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i64, _vp],
    "evp_attention_fused_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp],
    "evp_softmax_rows": [_vp, _vp, _i, _i64, _i, _i64, _vp],
    "evp_softmax_rows_bwd": [_vp, _vp, _vp, _i, _i64, _i, _i64, _vp],
    "evp_patchify": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "evp_embed_post_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "evp_embed_post_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "evp_add_rows_gather_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "evp_patchify_nhwc": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "evp_unpatchify_nhwc": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "evp_dwconv5x5_fwd": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "evp_dwconv5x5_bwd_nslab": [_i, _i, _i],
    "evp_dwconv_set_band": [_i],
    "evp_dwconv5x5_bwd": [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "evp_unshuffle_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "evp_unshuffle_bwd": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "evp_rec_loss": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "evp_add_f32": [_vp, _vp, _vp, _i64, _vp, _vp],
    "evp_cast": [_vp, _i, _vp, _i, _i64, _vp],
    "evp_scale_f32": [_vp, _vp, _i64, _vp],
    "evp_transpose": [_vp, _vp, _i, _i64, _i64, _vp],
    "evp_adamw_multi": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _i, _f, _vp, _vp],
    "evp_grad_norm_multi": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "evp_batchnorm_fwd": [_vp, _i, _i64, _i, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evp_batchnorm_bwd": [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "evp_batchnorm_nblk": [_i64],
    "evp_l2norm_rows_fwd": [_vp, _i64, _i, _vp, _vp, _vp],
    "evp_l2norm_rows_bwd": [_vp, _vp, _vp, _i64, _i, _vp, _vp],
    "evp_cross_entropy": [_vp, _vp, _i64, _i, _i64, _vp, _vp, _vp, _vp],
    "evp_cross_entropy_smooth": [_vp, _vp, _i64, _i, _i64, _f, _vp, _vp, _vp, _vp],
    "evp_rowdot_f32": [_vp, _vp, _i64, _i, _vp, _vp],
    "evp_scale_rows_f32": [_vp, _vp, _vp, _i64, _i, _vp, _vp],
    "evp_infonce_queue": [_vp, _vp, _i64, _i, _i64, _f, _vp, _vp, _vp, _vp, _vp],
    "evp_enqueue_keys": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "evp_enqueue_keys_dev": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "evp_window_attention_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _f, _vp],
    "evp_window_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _f, _vp],
    "evp_window_attention_fused_np": [_i],
    "evp_window_bias_build": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "evp_window_attention_fused_fwd": [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "evp_window_attention_fused_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "evp_window_attention_fused_nchunk": [_i, _i, _i],
    "evp_window_bias_reduce": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "evp_gather_rows_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "evp_swin_fuse_gather_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "evp_swin_fuse_gather_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "evp_swin_group_windows": [_vp, _i, _i, _vp, _vp, _vp],
    "evp_view_augment_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "evp_frame_augment_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "evp_token_mean_fwd": [_vp, _i, _i, _i, _vp, _vp],
    "evp_token_mean_bwd": [_vp, _i, _i, _i, _vp, _vp],
    "evp_rows_scale_f32": [_vp, _vp, _f, _vp, _i64, _i, _i, _vp, _vp, _vp],
    "evp_dropout_fwd": [_vp, _i, _vp, _vp, _i64, _f, C.c_uint64, _vp, C.c_uint64, _vp],
    "evp_dropout_apply": [_vp, _i, _vp, _vp, _i64, _f, _vp],
    "evp_abi_version": [],
}
_OTHER_RESTYPE = {"evp_last_error": C.c_char_p, "evp_target_arch": C.c_char_p}
_INT_RESTYPE = {"evp_gemm_stamp_count": C.c_longlong}
_NO_STATUS = {"evp_dwconv5x5_bwd_nslab", "evp_gemm_set_variant", "evp_voxel_set_debug", "evp_gemm_stamp_count", "evp_attention_fused_supported", "evp_layernorm_bwd_nblk", "evp_colsum_nblk", "evp_batchnorm_nblk", "evp_abi_version",
              "evp_window_attention_fused_np", "evp_window_attention_fused_nchunk"}

_lib = None


class EvpError(RuntimeError):
    pass


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    out = subprocess.run(args, capture_output=True, text=True)
    if out.returncode != 0:
        raise EvpError("building libevtpretrain.so failed:\n" + out.stdout[-4000:] + out.stderr[-4000:])
    if verbose:
        print(out.stdout[-2000:])
    return LIB_PATH


def load():
    """Load the shared library; raises (no fallback) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvpError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(eventpretrain_amd has no CPU or PyTorch fallback for its kernels)")
    lib = C.CDLL(LIB_PATH)
    lib.evp_abi_version.restype, lib.evp_abi_version.argtypes = C.c_int, []
    have = lib.evp_abi_version()
    if have != ABI_VERSION:
        raise EvpError(f"{LIB_PATH} has ABI version {have}, this binding expects {ABI_VERSION} (include/evtpretrain.h): rebuild it "
                       "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _INT_RESTYPE.get(name, C.c_int)
    for name, rt in _OTHER_RESTYPE.items():
        getattr(lib, name).restype = rt
        getattr(lib, name).argtypes = []
    _lib = lib
    return lib


def exported_symbols():
    return list(SIGNATURES) + list(_OTHER_RESTYPE)


def require_device():
    if not torch.cuda.is_available():
        raise EvpError("no HIP device visible: eventpretrain_amd kernels only run on an AMD GPU (gfx950); "
                       "there is no CPU fallback")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw-stream accessor is ~5x cheaper than building a
    torch.cuda.Stream object; an eager step calls this ~900 times)."""
    if _raw_stream is None or _get_device is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_get_device())


def call(name, *args):
    """Invoke a status-returning entry point on the current torch stream; raise EvpError with evp_last_error()."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _NO_STATUS:
        return rc
    if rc != 0:
        raise EvpError(f"{name} failed ({rc}): {lib.evp_last_error().decode()}")
    return rc


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise EvpError("expected a tensor in device memory (got a CPU tensor): eventpretrain_amd has no CPU path")
    return t.data_ptr()


def dt(t):
    if t.dtype == torch.float32:
        return EVP_F32
    if t.dtype == torch.bfloat16:
        return EVP_BF16
    raise EvpError(f"unsupported dtype {t.dtype}")
