"""Diagnostic wrappers (unit tests / roofline measurements), not part of the
reference's surface."""
import torch

from . import _lib


def gemm(a, w, bias=None, residual=None, epilogue=0, tile=(0, 0), out=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias) through after_gemm_f32."""
    a = _lib.require_gpu_tensor(a, "a", allow_row_stride=True)
    w = _lib.require_gpu_tensor(w, "w", allow_row_stride=True)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    rc = _lib.lib().after_gemm_f32(_lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0),
                                   _lib.ptr(bias), _lib.ptr(residual),
                                   residual.stride(0) if residual is not None else 0,
                                   _lib.ptr(out), out.stride(0), M, N, K, int(epilogue),
                                   int(tile[0]), int(tile[1]), _lib.current_stream(a.device))
    _lib.check(rc, "after_gemm_f32")
    return out
