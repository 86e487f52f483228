"""PQMF filter-bank design (host side, init time only).

Restates the design procedure of the reference (after/autoencoder/networks/pqmf.py:
35-92 `get_qmf_bank`, `kaiser_filter`, `loss_wc`, `get_prototype`; :203-217 and
:255-280 for the padded bank and the two conv kernels) so that a randomly
initialised AutoEncoder carries the same filters as the reference.  A checkpoint
overrides these buffers through load_state_dict.  Pinned against the reference's
bank in tests/test_boundary_cpu.py (fixture pqmf_bank.npz)."""
import math

import numpy as np


def _kaiser_filter(wc, atten, N=None):
    from scipy.signal import firwin, kaiserord
    N_, beta = kaiserord(atten, float(np.asarray(wc).reshape(-1)[0]) / np.pi)
    N_ = 2 * (N_ // 2) + 1
    N = N if N is not None else N_
    # scipy >= 1.13 dropped firwin(nyq=...): nyq = pi  <=>  fs = 2 pi
    return firwin(N, wc, window=("kaiser", beta), scale=False, fs=2 * np.pi)


def _loss_wc(wc, atten, M, N):
    h = _kaiser_filter(wc, atten, N)
    g = np.convolve(h, h[::-1], "full")
    g = abs(g[g.shape[-1] // 2::2 * M][1:])
    return np.max(g)


def prototype(atten, M, N=None):
    from scipy.optimize import fmin
    wc = fmin(lambda w: _loss_wc(w, atten, M, N), 1 / M, disp=0)[0]
    return _kaiser_filter(wc, atten, N)


def design(attenuation=100, n_band=16):
    """Returns (h, hk, forward_weight [M,1,K+1], inverse_weight [M,M,K/M+1]) as float32."""
    h = prototype(attenuation, n_band).astype(np.float32)
    k = np.arange(n_band).reshape(-1, 1)
    N = h.shape[-1]
    t = np.arange(-(N // 2), N // 2 + 1)
    p = (-1.0)**k * math.pi / 4
    # float32 arithmetic like the reference's torch ops (get_qmf_bank)
    mod = np.cos(((2 * k + 1) * math.pi / (2 * n_band) * t).astype(np.float32) + p.astype(np.float32),
                 dtype=np.float32)
    hk = (2 * h * mod).astype(np.float32)
    nxt = 2**math.ceil(math.log2(hk.shape[-1]))
    pad = nxt - hk.shape[-1]
    hk = np.pad(hk, ((0, 0), (pad // 2, pad // 2 + pad % 2)))
    hkf = np.pad(hk, ((0, 0), (0, 1)))[:, None, :] if hk.shape[-1] % 2 == 0 else hk[:, None, :]
    hki = hk[:, ::-1]
    hki = hki.reshape(n_band, -1, n_band).transpose(2, 0, 1)  # "c (t m) -> m c t"
    if hki.shape[-1] % 2 == 0:
        hki = np.pad(hki, ((0, 0), (0, 0), (0, 1)))
    return h, hk, np.ascontiguousarray(hkf, dtype=np.float32), np.ascontiguousarray(hki, dtype=np.float32)
