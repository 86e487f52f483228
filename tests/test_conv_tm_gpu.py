"""The time-major conv path (conv_tm.hip: activate + halo, then the conv as a balanced LDS-DMA GEMM)
against a plain fp64 torch conv1d of the same layer, through the C ABI (after_convtm_*).  -m gpu.

Covers what the codec / encoders ask of it: k = 1 / 3 / 5 / 8, dilation 1 / 3 / 9, stride 1 / 2 / 4,
centred and causal padding, ragged channel counts (K padding to the 32-deep slab, N tails), ragged
lengths, every tile configuration (row-split, 2- and 4-way split-K), fused statistics and residual."""
import pytest
import torch
import torch.nn.functional as F

from after_amd import diag

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def ref_conv(x, w, b, dil, stride, lp, rp, act):
    x = x.double()
    if act == 2:
        x = F.silu(x)
    elif act == 3:
        x = F.relu(x)
    y = F.conv1d(F.pad(x, (lp, rp)), w.double(), b.double() if b is not None else None, stride=stride, dilation=dil)
    return y


CASES = [
    # B, Cin, Cout, T, k, dil, stride, lp, rp, act
    (1, 64, 64, 256, 3, 1, 1, 1, 1, 0),
    (2, 96, 192, 300, 3, 3, 1, 3, 3, 2),
    (1, 128, 96, 1000, 3, 9, 1, 9, 9, 0),
    (2, 64, 32, 515, 3, 9, 1, 18, 0, 2),     # causal dilated
    (1, 16, 64, 2048, 3, 1, 1, 1, 1, 0),     # Cin below one K slab (PQMF bands -> stem)
    (3, 64, 12, 256, 5, 1, 1, 4, 0, 2),      # Encoder1D: causal k = 5, narrow output
    (1, 128, 256, 1024, 4, 1, 2, 1, 2, 0),   # Downsample1d f = 2: pad (1, 2)
    (2, 64, 128, 2048, 8, 1, 4, 3, 4, 0),    # Downsample1d f = 4: pad (3, 4)
    (1, 384, 384, 512, 1, 1, 1, 0, 0, 0),    # 1 x 1
    (1, 768, 768, 256, 3, 1, 1, 1, 1, 0),    # small T, long K -> split-K
    (1, 40, 72, 130, 3, 1, 1, 1, 1, 3),      # ragged everything
]


@pytest.mark.parametrize("case", CASES)
def test_conv_tm_matches_torch(case, hip_device):
    B, Cin, Cout, T, k, dil, stride, lp, rp, act = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    want = ref_conv(x, w, b, dil, stride, lp, rp, act)
    c = diag.ConvTm(w.to(hip_device), b.to(hip_device), B, T, dil, stride, lp, rp, act)
    assert c.Tout == want.shape[-1]
    got = c(x.to(hip_device)).cpu().double()
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    got2 = c(x.to(hip_device), stats=Cout % min(Cout, 8) == 0, residual=True).cpu().double()
    assert (got2 - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 9, 15, 16])
def test_conv_tm_every_tile_configuration(tile, hip_device):
    """All tile configurations give the same result on a shape every one of them accepts
    (K = 3 * 128 = 384: multiple of 128), with a dilated and an undilated conv."""
    g = torch.Generator().manual_seed(tile)
    B, Cin, Cout, T = 2, 128, 200, 777
    x = torch.randn(B, Cin, T, generator=g)
    try:
        diag.set_conv_tile(tile)
        for dil in (1, 3):
            w = torch.randn(Cout, Cin, 3, generator=g) / (Cin * 3) ** 0.5
            b = torch.randn(Cout, generator=g)
            want = ref_conv(x, w, b, dil, 1, dil, dil, 0)
            c = diag.ConvTm(w.to(hip_device), b.to(hip_device), B, T, dil, 1, dil, dil, 0)
            got = c(x.to(hip_device), stats=True).cpu().double()
            assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), (tile, dil)
    finally:
        diag.set_conv_tile(0)


X6_CASES = [
    # B, Cin, Cout, T, k, dil, lp, rp, act      (stride 1, <= 3 taps: the layers conv_x6.hip takes)
    (1, 384, 384, 2048, 3, 1, 1, 1, 0),
    (2, 384, 384, 1000, 3, 3, 3, 3, 0),      # ragged length: rows past the clip in the last tile
    (1, 384, 384, 1024, 3, 9, 9, 9, 2),
    (2, 192, 192, 777, 3, 9, 18, 0, 2),      # causal dilated, 2 clips
    (1, 384, 384, 512, 1, 1, 0, 0, 0),       # 1 x 1
    (1, 768, 384, 640, 2, 1, 1, 0, 0),       # two taps (a transposed-conv phase)
    (3, 96, 64, 300, 3, 3, 3, 3, 3),         # 64-column tile, ragged K blocks (Cin = 96 = 3 blocks)
    (1, 40, 72, 130, 3, 1, 1, 1, 3),         # ragged everything
]


@pytest.mark.parametrize("case", X6_CASES)
def test_conv_x6_matches_torch(case, hip_device):
    """The same layers through the bf16 matrix pipe (six exact bf16 MFMAs per fp32 product on three-way bf16 splits,
    conv_x6.hip) against the fp64 conv: same bar as the fp32 MFMA path, and at least as close to fp64 as it on the
    MFMA-bound shapes."""
    B, Cin, Cout, T, k, dil, lp, rp, act = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    want = ref_conv(x, w, b, dil, 1, lp, rp, act)
    c = diag.ConvTm(w.to(hip_device), b.to(hip_device), B, T, dil, 1, lp, rp, act)
    assert c.Tout == want.shape[-1]
    scale = max(1.0, want.abs().max().item())
    got = c(x.to(hip_device), x6=True).cpu().double()
    assert (got - want).abs().max().item() < 2e-5 * scale
    got2 = c(x.to(hip_device), stats=Cout % min(Cout, 8) == 0 and (Cout // min(Cout, 8)) % 4 == 0, residual=True, x6=True).cpu().double()
    assert (got2 - want).abs().max().item() < 2e-5 * scale
    ref32 = c(x.to(hip_device)).cpu().double()
    e6, e32 = (got - want).abs().max().item(), (ref32 - want).abs().max().item()
    assert e6 <= 3 * e32 + 1e-6 * scale, (e6, e32)  # (one accumulation chain here, two k-parts there on some shapes)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8])
def test_conv_x6_every_tile(tile, hip_device, monkeypatch):
    g = torch.Generator().manual_seed(tile)
    B, Cin, Cout, T = 2, 128, 192, 700
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, 3, generator=g) / (Cin * 3) ** 0.5
    b = torch.randn(Cout, generator=g)
    want = ref_conv(x, w, b, 3, 1, 3, 3, 0)
    c = diag.ConvTm(w.to(hip_device), b.to(hip_device), B, T, 3, 1, 3, 3, 0)
    try:
        diag.set_conv_x6_tile(tile)
        got = c(x.to(hip_device), stats=True, x6=True).cpu().double()
    finally:
        diag.set_conv_x6_tile(0)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_conv_x6_refuses_layers_without_a_bf16_pipe_form(hip_device):
    """Strided convs (their rows are not contiguous in the plane layout) and k > 3 stay on conv_tm: the diagnostic entry
    refuses mode bit 4 for them instead of silently running the fp32 kernel."""
    from after_amd import _lib
    g = torch.Generator().manual_seed(3)
    for (cin, cout, k, stride) in ((64, 128, 8, 4), (64, 64, 5, 1)):
        w = torch.randn(cout, cin, k, generator=g).to(hip_device)
        c = diag.ConvTm(w, None, 1, 256, 1, stride, k - 1, 0, 0)
        x = torch.randn(1, cin, 256, generator=g).to(hip_device)
        c(x)  # the fp32 form runs
        with pytest.raises(_lib.AFTERHipError):
            c(x, x6=True)


def test_conv_x6_random_shapes(hip_device):
    """Forty seeded random stride-1 layers (taps 1-3, dilation 1-9, centred / causal / lopsided padding, channel counts
    that are not multiples of the 32-deep K slab or of the column tiles, lengths that are not multiples of the 16-row
    plane blocks or of the row tiles, 1-3 clips, every tile configuration in turn) through the bf16-pipe form against
    the fp64 conv -- the row windows shifted by (tap x dilation) across 16-row plane blocks, the clamped blocks past a
    clip's end and the masked tile edges are where this kernel could go wrong."""
    import random
    rnd = random.Random(1234)
    g = torch.Generator().manual_seed(99)
    try:
        for it in range(40):
            k = rnd.choice((1, 2, 3, 3))
            dil = rnd.choice((1, 1, 2, 3, 5, 9)) if k > 1 else 1
            span = (k - 1) * dil
            lp = rnd.choice((span // 2, span, 0)) if span else 0
            rp = span - lp
            B = rnd.choice((1, 2, 3))
            Cin = rnd.choice((24, 32, 40, 64, 96, 136, 192))
            Cout = 4 * rnd.randint(3, 60)
            T = rnd.choice((17, 63, 64, 100, 129, 255, 400, 513))
            act = rnd.choice((0, 2, 3))
            x = torch.randn(B, Cin, T, generator=g)
            w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
            b = torch.randn(Cout, generator=g)
            want = ref_conv(x, w, b, dil, 1, lp, rp, act)
            c = diag.ConvTm(w.to(hip_device), b.to(hip_device), B, T, dil, 1, lp, rp, act)
            tile = it % 9  # 0 = by shape, 1..8
            if tile == 6 and (k * ((Cin + 31) // 32 * 32)) % 64:
                tile = 0  # (two k-parts need an even slab count)
            diag.set_conv_x6_tile(tile)
            stats = Cout % min(Cout, 8) == 0 and (Cout // min(Cout, 8)) % 4 == 0
            got = c(x.to(hip_device), stats=stats, residual=bool(it & 1), x6=True).cpu().double()
            err = (got - want).abs().max().item()
            assert err < 2e-5 * max(1.0, want.abs().max().item()), (it, tile, (B, Cin, Cout, T, k, dil, lp, rp, act), err)
            c.close()
    finally:
        diag.set_conv_x6_tile(0)
