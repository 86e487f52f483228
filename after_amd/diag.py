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


def split_x6(w):
    """The three bf16 planes [N, 3, K] (int16 storage) of an fp32 weight, for gemm_x6."""
    w = _lib.require_gpu_tensor(w, "w")
    N, K = w.shape
    w3 = torch.empty(N, 3, K, device=w.device, dtype=torch.int16)
    _lib.check(_lib.lib().after_gemm_x6_split(_lib.ptr(w), w.stride(0), _lib.ptr(w3), N, K,
                                              _lib.current_stream(w.device)), "after_gemm_x6_split")
    return w3


def gemm_x6(a, w3, bias=None, residual=None, epilogue=0, tile=0, out=None):
    """EXPERIMENTAL: out[M,N] = epi(a @ w^T + bias) with fp32 products formed on the bf16 matrix pipe."""
    a = _lib.require_gpu_tensor(a, "a", allow_row_stride=True)
    M, K = a.shape
    N = w3.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    rc = _lib.lib().after_gemm_x6(_lib.ptr(a), a.stride(0), _lib.ptr(w3), _lib.ptr(bias), _lib.ptr(residual),
                                  residual.stride(0) if residual is not None else 0, _lib.ptr(out), out.stride(0),
                                  M, N, K, int(epilogue), int(tile), _lib.current_stream(a.device))
    _lib.check(rc, "after_gemm_x6")
    return out


class ConvTm:
    """One Conv1d layer on the time-major conv path (after_convtm_*): parity tests / tile sweeps."""

    def __init__(self, w, bias, B, T, dil=1, stride=1, left_pad=0, right_pad=0, act=0):
        import ctypes
        w = _lib.require_gpu_tensor(w, "w")
        self.Cout, self.Cin, self.k = w.shape
        self.B, self.T = B, T
        self.Tout = (T + left_pad + right_pad - (self.k - 1) * dil - 1) // stride + 1
        bias = _lib.require_gpu_tensor(bias, "bias") if bias is not None else None
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().after_convtm_create(_lib.ptr(w), _lib.ptr(bias), B, self.Cin, self.Cout, T,
                                                  self.Tout, self.k, dil, stride, left_pad, act,
                                                  ctypes.byref(h)), "after_convtm_create")
        self._h = h
        self.flops = 2.0 * B * self.Tout * self.Cout * self.Cin * self.k

    def run(self, x=None, y=None, mode=3):
        dev = torch.cuda.current_device()
        _lib.check(_lib.lib().after_convtm_run(self._h, _lib.ptr(x), _lib.ptr(y), int(mode),
                                               _lib.current_stream(dev)), "after_convtm_run")

    def __call__(self, x, stats=False, residual=False):
        x = _lib.require_gpu_tensor(x, "x")
        y = torch.empty(self.B, self.Cout, self.Tout, device=x.device, dtype=torch.float32)
        self.run(x, y, 3 | (4 if stats else 0) | (8 if residual else 0))
        return y

    def close(self):
        if self._h is not None:
            _lib.lib().after_convtm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_conv_tile(tile_id: int):
    _lib.lib().after_convtm_set_tile(int(tile_id))
