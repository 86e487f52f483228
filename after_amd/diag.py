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


class X6Planes:
    """The three bf16 planes of an fp32 [rows, K] matrix in gemm_x6's block layout (include/after_hip.h:
    1-KB blocks [ceil(rows / 16)][K / 32][plane]; int16 storage).  `data` is the flat device buffer."""

    def __init__(self, data, rows, K):
        self.data, self.rows, self.K = data, rows, K

    @staticmethod
    def empty(rows, K, device):
        n = ((rows + 15) // 16) * 16 * 3 * K
        return X6Planes(torch.zeros(n, device=device, dtype=torch.int16), rows, K)

    def index(self):
        """int64 [rows, 3, K]: flat element offset of (row, plane, column) -- the layout restated in torch
        (checked against after_gemm_x6_offset by tests/test_gemm_gpu.py)."""
        r = torch.arange(self.rows).view(-1, 1, 1)
        p = torch.arange(3).view(1, -1, 1)
        k = torch.arange(self.K).view(1, 1, -1)
        rr, c = r & 15, (k & 31) >> 3
        f = torch.tensor([0, 2, 3, 1])[(rr >> 2) & 3]
        slot = c ^ f
        return (((r >> 4) * (self.K >> 5) + (k >> 5)) * 3 + p) * 512 + rr * 32 + slot * 8 + (k & 7)

    def join(self):
        """fp32 [rows, K] value of the planes (h + m + l, exact), on the CPU."""
        d = self.data.cpu()[self.index()]
        f = (d.to(torch.int32) << 16).view(torch.float32)
        return (f[:, 0] + f[:, 1]) + f[:, 2]


def split_x6(w):
    """fp32 [rows, K] device matrix -> X6Planes (both operands of gemm_x6)."""
    w = _lib.require_gpu_tensor(w, "w", allow_row_stride=True)
    N, K = w.shape
    out = X6Planes.empty(N, K, w.device)
    _lib.check(_lib.lib().after_gemm_x6_split(_lib.ptr(w), w.stride(0), _lib.ptr(out.data), N, K,
                                              _lib.current_stream(w.device)), "after_gemm_x6_split")
    return out


def split_gemm(a, w, mode):
    """a [M, K] @ w [N, K]^T in one of the Linears' arithmetics (include/after_hip.h: after_diag_split_gemm) -- 0: the fp32 MFMA
    chain, 1: three bf16 planes x six products, 2: two fp16 pieces x three products (scales from the operands' maxima)."""
    a = _lib.require_gpu_tensor(a, "a")
    w = _lib.require_gpu_tensor(w, "w")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device)
    _lib.check(_lib.lib().after_diag_split_gemm(_lib.ptr(a), _lib.ptr(w), _lib.ptr(out), M, N, K, int(mode), float(a.abs().max()),
                                                float(w.abs().max()), _lib.current_stream(a.device)), "after_diag_split_gemm")
    return out


def gemm_x6(a3, w3, bias=None, residual=None, epilogue=0, tile=0, out=None, planes=False):
    """out[M,N] = epi(a @ w^T + bias) on the bf16-split path (gemm_x6.hip): a3 / w3 = X6Planes; the result is
    fp32 [M, N], or (planes=True) X6Planes of it."""
    M, K, N = a3.rows, a3.K, w3.rows
    assert w3.K == K
    if out is None:
        out = X6Planes.empty(M, N, a3.data.device) if planes else torch.empty(M, N, device=a3.data.device)
    rc = _lib.lib().after_gemm_x6(_lib.ptr(a3.data), _lib.ptr(w3.data), _lib.ptr(bias), _lib.ptr(residual),
                                  residual.stride(0) if residual is not None else 0,
                                  None if planes else _lib.ptr(out), _lib.ptr(out.data) if planes else None,
                                  0 if planes else out.stride(0), M, N, K, int(epilogue), int(tile),
                                  _lib.current_stream(a3.data.device))
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

    def __call__(self, x, stats=False, residual=False, x6=False):
        """x6: the bf16-pipe form of the layer (conv_x6.hip; stride-1 layers of <= 3 taps)."""
        x = _lib.require_gpu_tensor(x, "x")
        y = torch.empty(self.B, self.Cout, self.Tout, device=x.device, dtype=torch.float32)
        self.run(x, y, 3 | (4 if stats else 0) | (8 if residual else 0) | (16 if x6 else 0))
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


def conv_h3_launches() -> int:
    """... of them on two-piece fp16 operands (include/after_hip.h: after_conv_h3_launches)."""
    return int(_lib.lib().after_conv_h3_launches())


def conv_x6_launches() -> int:
    """Conv launches this process has sent down the bf16-pipe path (conv_x6.hip) so far."""
    return int(_lib.lib().after_conv_x6_launches())


def conv1_act_launches() -> int:
    """GroupNorm -> Snake -> Conv1d(k = 1) blocks this process has run as one launch (conv_tm.hip: conv1_act_kernel)."""
    return int(_lib.lib().after_conv1_act_launches())


def set_conv_x6_tile(tile_id: int):
    """Tile of the bf16-pipe convs: 0 = by shape, 1..8 pin one (conv_x6.hip: launch_conv_x6)."""
    _lib.lib().after_convtm_set_x6_tile(int(tile_id))
