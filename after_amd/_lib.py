"""ctypes binding of libafter_hip.so (C ABI declared in include/after_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to
load, every product entry point raises."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libafter_hip.so")

AFTER_OK = 0
CFG_API, CFG_EXPORT, CFG_MIDI = 0, 1, 2


class DenoiserCfg(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("n_channels", "embed_dim", "cond_dim", "tcond_dim",
                                     "noise_embed_dims", "n_layers", "mlp_multiplier", "causal",
                                     "local_attention_size", "attention_chunk_size")]


class AECfg(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("pqmf_bands", "channels", "z_channels", "n_stages",
                                     "n_dilations", "kernel_size", "use_norm", "use_loudness",
                                     "causal")] + [("multipliers", c_int * 9),
                                                   ("dec_multipliers", c_int * 9),
                                                   ("factors", c_int * 8), ("dilations", c_int * 8),
                                                   ("encoder_out_channels", c_int), ("use_noise", c_int)]


class Encoder1dCfg(ctypes.Structure):
    _fields_ = [("in_size", c_int), ("n_blocks", c_int), ("channels", c_int * 8),
                ("ratios", c_int * 8), ("kernel_size", c_int), ("causal", c_int),
                ("use_tanh", c_int)]


class Unet1dCfg(ctypes.Structure):
    _fields_ = [("in_size", c_int), ("out_size", c_int), ("n_blocks", c_int), ("channels", c_int * 8),
                ("ratios", c_int * 8), ("kernel_size", c_int), ("time_channels", c_int),
                ("time_cond_in_channels", c_int), ("time_cond_channels", c_int),
                ("cond_channels", c_int), ("use_res_last", c_int), ("n_attn_layers", c_int)]


class EcapaCfg(ctypes.Structure):
    _fields_ = [("in_size", c_int), ("out_dim", c_int), ("n_blocks", c_int),
                ("channels", c_int * 8), ("kernel_sizes", c_int * 8), ("dilations", c_int * 8),
                ("res2net_scale", c_int), ("se_channels", c_int), ("attention_channels", c_int),
                ("use_tanh", c_int)]


class AFTERHipError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); also the list the symbol-export test checks against the header
SIGNATURES = {
    "after_last_error": (c_char_p, []),
    "after_version": (c_char_p, []),
    "after_denoiser_create": (c_int, [POINTER(DenoiserCfg), POINTER(c_void_p), c_int, c_int, c_int,
                                      c_int, POINTER(c_void_p)]),
    "after_denoiser_destroy": (None, [c_void_p]),
    "after_denoiser_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_void_p]),
    "after_model_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_float, c_float, c_float, c_int, c_int,
                                    c_void_p]),
    "after_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_float, c_float, c_float, c_int, c_void_p]),
    "after_denoiser_set_graph": (c_int, [c_void_p, c_int]),
    "after_denoiser_set_gemm_path": (c_int, [c_void_p, c_int, c_int]),
    "after_denoiser_gemm_path": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "after_denoiser_set_stream_persist": (c_int, [c_void_p, c_int]),
    "after_denoiser_stream_persist": (c_int, [c_void_p, POINTER(c_int)]),
    "after_denoiser_set_sample_persist": (c_int, [c_void_p, c_int]),
    "after_denoiser_sample_persist": (c_int, [c_void_p, POINTER(c_int)]),
    "after_denoiser_sample_arith": (c_int, [c_void_p, POINTER(c_int)]),
    "after_denoiser_sample_launches": (c_int, [c_void_p, POINTER(c_int)]),
    "after_denoiser_step_trace": (c_int, [c_void_p, c_void_p, c_int]),
    "after_denoiser_set_step_trace": (c_int, [c_void_p, c_int]),
    "after_denoiser_set_persist_check": (c_int, [c_void_p, c_int]),
    "after_denoiser_check": (c_int, [c_void_p, c_void_p]),
    "after_denoiser_enable_cache": (c_int, [c_void_p, c_int, c_int, c_int]),
    "after_denoiser_reset_cache": (c_int, [c_void_p, c_void_p]),
    "after_denoiser_roll_cache": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "after_denoiser_profile": (c_int, [c_void_p, c_int]),
    "after_denoiser_gemm_time_ms": (c_int, [c_void_p, POINTER(c_double), POINTER(c_longlong),
                                            POINTER(c_double)]),
    "after_denoiser_profile_min_flops": (c_int, [c_void_p, c_double]),
    "after_denoiser_profile_kernel": (c_int, [c_void_p, c_int]),
    "after_denoiser_gemm_time2": (c_int, [c_void_p, POINTER(c_double), POINTER(c_longlong), POINTER(c_double),
                                          POINTER(c_double)]),
    "after_ae_create": (c_int, [POINTER(AECfg), POINTER(c_void_p), c_int, c_int, c_int,
                                POINTER(c_void_p)]),
    "after_ae_destroy": (None, [c_void_p]),
    "after_ae_ratio": (c_int, [c_void_p]),
    "after_ae_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_ae_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_ae_set_noise": (c_int, [c_void_p, c_void_p]),
    "after_ae_set_stream_lanes": (c_int, [c_void_p, c_int]),
    "after_ae_encode_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "after_ae_decode_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_latent_reg": (c_int, [c_void_p, ctypes.c_longlong, ctypes.c_float, c_void_p, c_void_p]),
    "after_bottleneck_tanh": (c_int, [c_void_p, c_longlong, c_float, c_void_p]),
    "after_bottleneck_vae": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "after_ae_pqmf_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_ae_pqmf_inverse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_ae_enable_streaming": (c_int, [c_void_p, c_int]),
    "after_ae_reset_state": (c_int, [c_void_p, c_void_p]),
    "after_ae_enable_encoder_streaming": (c_int, [c_void_p, c_int, c_int]),
    "after_ae_encoder_delay": (c_int, [c_void_p]),
    "after_ae_set_decoder_gn_window": (c_int, [c_void_p, c_int]),
    "after_encoder1d_create": (c_int, [POINTER(Encoder1dCfg), POINTER(c_void_p), c_int, c_int, c_int,
                                       POINTER(c_void_p)]),
    "after_encoder1d_destroy": (None, [c_void_p]),
    "after_encoder1d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_encoder1d_enable_streaming": (c_int, [c_void_p, c_int]),
    "after_encoder1d_reset_state": (c_int, [c_void_p, c_void_p]),
    "after_ecapa_create": (c_int, [POINTER(EcapaCfg), POINTER(c_void_p), c_int, c_int, c_int,
                                   POINTER(c_void_p)]),
    "after_ecapa_destroy": (None, [c_void_p]),
    "after_ecapa_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "after_unet1d_create": (c_int, [POINTER(Unet1dCfg), POINTER(c_void_p), c_int, c_int, c_int,
                                    POINTER(c_void_p)]),
    "after_unet1d_destroy": (None, [c_void_p]),
    "after_unet1d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_int, c_void_p]),
    "after_unet1d_model_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int, c_float, c_float, c_float, c_int, c_void_p]),
    "after_unet1d_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_float, c_float, c_float, c_int, c_void_p]),
    "after_gemm_set_debug": (None, [c_void_p]),
    "after_convtm_create": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, POINTER(c_void_p)]),
    "after_convtm_run": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "after_convtm_destroy": (None, [c_void_p]),
    "after_convtm_set_tile": (None, [c_int]),
    "after_convtm_set_x6_tile": (None, [c_int]),
    "after_conv_x6_launches": (ctypes.c_longlong, []),
    "after_conv_h3_launches": (ctypes.c_longlong, []),
    "after_conv1_act_launches": (ctypes.c_longlong, []),
    "after_gemm_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "after_gemm_x6_set_debug": (None, [c_void_p]),
    "after_gemm_x6_split": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "after_gemm_x6_pick_tile": (c_int, [c_int, c_int, c_int]),
    "after_diag_split_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "after_gemm_x6_offset": (ctypes.c_longlong, [c_int, c_int, c_int, c_int]),
    "after_gemm_x6": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                              c_int, c_int, c_int, c_int, c_void_p]),
}


def lib():
    """Loads the shared library (once).  torch must be imported first so that the
    HIP runtime already in the process (torch's libamdhip64) is the one the
    library binds to -- device pointers and streams are shared with torch."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (load order matters, see docstring)
    if not os.path.exists(LIB_PATH):
        raise AFTERHipError(
            f"{LIB_PATH} is missing: build it with `python -m after_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback in the product path")
    try:
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise AFTERHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise AFTERHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != AFTER_OK:
        msg = lib().after_last_error()
        raise AFTERHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a contiguous fp32 CUDA tensor as c_void_p (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu_tensor(t, name, allow_row_stride=False):
    """A dense fp32 CUDA tensor.  The C ABI takes bare pointers and assumes row-major contiguous
    data everywhere except after_gemm_f32, whose lda/ldw arguments carry a row stride:
    only that caller passes allow_row_stride=True."""
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise AFTERHipError(
            f"{name} is on {t.device}: after_amd runs on MI355X only (HIP kernels); "
            "the CPU restatement lives in oracle/ and is test infrastructure")
    if t.dtype != torch.float32:
        t = t.float()
    if allow_row_stride and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()
