"""fp32 MFMA GEMM (after_gemm_f32) against torch fp64 on the GPU box: every tile
configuration, ragged M/N/K, all epilogues.  v_mfma_f32_16x16x4_f32 is an exact
fp32 fma chain, so the error bound is fp32 round-off: |err| <= 2e-6 * sum|a||w|."""
import pytest
import torch

from after_amd import diag

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile", [(0, 0), (1, 1), (1, 2), (2, 2), (4, 2), (4, 4)])
@pytest.mark.parametrize("M,N,K", [(768, 1536, 512), (150, 512, 72), (33, 17, 12), (257, 130, 100),
                                   (1, 64, 512), (768, 64, 512)])
def test_gemm_shapes(tile, M, N, K, hip_device):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)  # asymmetric: catches transposed C/D maps
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    bound = 2e-6 * (a.abs().double() @ w.abs().double().t()).max().item() + 1e-6
    ad, wd, bd, rd = (t.to(hip_device) for t in (a, w, b, r))
    out = diag.gemm(ad, wd, bd, tile=tile).cpu().double()
    assert (out - ref).abs().max().item() < bound
    out = diag.gemm(ad, wd, bd, epilogue=1, tile=tile).cpu().double()
    assert (out - torch.nn.functional.gelu(ref)).abs().max().item() < bound
    out = diag.gemm(ad, wd, None, residual=rd, epilogue=2, tile=tile).cpu().double()
    assert (out - (ref - b.double() + r.double())).abs().max().item() < bound


# balanced split-K kernels: tile = (100 * KS / 2 + MB, 10 * NS + NB); K must be a multiple of 32 * KS
@pytest.mark.parametrize("tile", [(103, 23), (103, 33), (103, 21), (103, 41), (203, 23), (203, 21),
                                  (203, 31), (203, 41), (203, 22), (202, 22), (202, 32), (104, 22), (104, 23), (102, 22),
                                  (304, 23), (304, 22), (306, 23), (306, 22)])
@pytest.mark.parametrize("M,N,K", [(768, 1536, 512), (768, 512, 1536), (150, 520, 128), (33, 17, 256),
                                   (257, 130, 384), (1, 64, 512), (3072, 96, 128)])
def test_gemm_split_k_shapes(tile, M, N, K, hip_device):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    bound = 2e-6 * (a.abs().double() @ w.abs().double().t()).max().item() + 1e-6
    ad, wd, bd, rd = (t.to(hip_device) for t in (a, w, b, r))
    out = diag.gemm(ad, wd, bd, tile=tile)
    assert (out.cpu().double() - ref).abs().max().item() < bound
    # fixed summation order of the k-parts: bit-identical on every run
    assert torch.equal(out, diag.gemm(ad, wd, bd, tile=tile))
    out = diag.gemm(ad, wd, bd, epilogue=1, tile=tile).cpu().double()
    assert (out - torch.nn.functional.gelu(ref)).abs().max().item() < bound
    out = diag.gemm(ad, wd, None, residual=rd, epilogue=2, tile=tile).cpu().double()
    assert (out - (ref - b.double() + r.double())).abs().max().item() < bound


@pytest.mark.parametrize("tile,M", [((401, 0), 12), ((401, 0), 16), ((403, 0), 33), ((403, 0), 48), ((406, 0), 96),
                                    ((406, 0), 70), ((0, 0), 12), ((0, 0), 96)])
@pytest.mark.parametrize("N,K", [(1536, 512), (512, 1536), (264, 128), (72, 1024)])
def test_gemm_skinny_shapes(tile, M, N, K, hip_device):
    """Few-token (streaming) shapes: column-owning workgroups, 8-way K split inside the workgroup."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    bound = 2e-6 * (a.abs().double() @ w.abs().double().t()).max().item() + 1e-6
    ad, wd, bd, rd = (t.to(hip_device) for t in (a, w, b, r))
    out = diag.gemm(ad, wd, bd, tile=tile)
    assert (out.cpu().double() - ref).abs().max().item() < bound
    assert torch.equal(out, diag.gemm(ad, wd, bd, tile=tile))
    out = diag.gemm(ad, wd, bd, epilogue=1, tile=tile).cpu().double()
    assert (out - torch.nn.functional.gelu(ref)).abs().max().item() < bound
    out = diag.gemm(ad, wd, None, residual=rd, epilogue=2, tile=tile).cpu().double()
    assert (out - (ref - b.double() + r.double())).abs().max().item() < bound


def test_gemm_split_k_rejects_short_k(hip_device):
    from after_amd._lib import AFTERHipError
    a = torch.randn(64, 96, device=hip_device)
    w = torch.randn(64, 96, device=hip_device)
    with pytest.raises(AFTERHipError):
        diag.gemm(a, w, tile=(203, 23))


def test_gemm_strided_operands(hip_device):
    g = torch.Generator().manual_seed(5)
    big = torch.randn(100, 96, generator=g).to(hip_device)
    a = big[:, :64]  # lda = 96
    w = torch.randn(40, 64, generator=g).to(hip_device)
    out = diag.gemm(a, w)
    ref = a.double() @ w.double().t()
    assert (out.double() - ref).abs().max().item() < 1e-4


X6_SHAPES = [(768, 1536, 512, 0), (768, 512, 1536, 2), (6144, 1536, 512, 1), (6144, 1536, 512, 0), (6144, 512, 1536, 2),
             (18432, 1536, 512, 1), (100, 96, 128, 1),
             (49, 33, 256, 0), (1536, 768, 256, 1), (200, 768, 768, 2)]


def _x6_case(M, N, K, epi, seed):
    """Operands in the sampler's regime: LayerNorm-scale activations with outlier rows / columns (x 25), weights
    ~ N(0, 1/K), bias; fp64 reference of the fp32 operands."""
    g = torch.Generator().manual_seed(seed)
    a = 1.3 * torch.randn(M, K, generator=g)
    a[::5, ::11] *= 25.0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if epi == 2 else None
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res.double()
    return a, w, bias, res, ref


# k-parts of each gemm_x6 tile (ACC2 tiles count double) and the fp32 MFMA kernel with the same split
X6_KPARTS = {1: 2, 2: 4, 3: 1, 4: 2, 5: 1, 6: 2, 7: 1, 8: 2, 9: 1, 11: 2, 12: 4, 15: 1}
X6_KDIV = {11: 256, 12: 512}  # the W-in-register tiles walk K in groups of four slabs per k-part
F32_TILE_BY_KPARTS = {1: (304, 23), 2: (103, 21), 4: (203, 21)}


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 15])
@pytest.mark.parametrize("M,N,K,epi", X6_SHAPES)
def test_gemm_x6(tile, M, N, K, epi, hip_device):
    """gemm_x6.hip (the qkv / MLP Linears on the bf16-split path: operands as three bf16 planes, products as six
    bf16 MFMAs, fp32 accumulation) against fp64 at the sampler's own shapes (M = 768 and 6144 token rows, both
    GEMM orientations) and on ragged ones, every tile, bias + GELU / residual epilogues, operands with outliers.
    Bar: its error against fp64 is no larger than that of the fp32 MFMA kernel (an exact fp32 fma chain) with the
    same number of k-parts: rms <= 1.0 x (measured 0.79 - 0.82 x on every shape and tile, profiles/r3_gemm_x6_sweep.jsonl);
    max -- an extreme-value statistic over up to 9.4 M outputs of two different summation orders, measured
    0.55 - 1.08 x -- <= 1.5 x.  And the tile the dispatch picks (tile 0) against whatever the fp32 dispatch picks for
    the same shape, i.e. production against production: rms <= 1.0 x, max <= 1.25 x."""
    from after_amd import _lib, diag
    if K % X6_KDIV.get(tile, 32):
        pytest.skip(f"tile {tile} needs K to be a multiple of {X6_KDIV[tile]}")
    a, w, bias, res, ref = _x6_case(M, N, K, epi, M + N + K + tile)
    dev = hip_device
    w3, a3 = diag.split_x6(w.to(dev)), diag.split_x6(a.to(dev))
    assert torch.equal(a3.join(), a), "the bf16 planes must add up to the fp32 operand exactly"
    kw = dict(bias=bias.to(dev), residual=res.to(dev) if res is not None else None, epilogue=epi)
    got = diag.gemm_x6(a3, w3, tile=tile, **kw).cpu()
    t = tile if tile else _lib.lib().after_gemm_x6_pick_tile(M, N, K)
    same_split = diag.gemm(a.to(dev), w.to(dev), tile=F32_TILE_BY_KPARTS[X6_KPARTS[t]], **kw).cpu()
    default = diag.gemm(a.to(dev), w.to(dev), **kw).cpu()
    e6, es, ed = ((x.double() - ref).abs() for x in (got, same_split, default))
    floor = 1e-7 * ref.abs().max().item()
    rms = lambda e: e.pow(2).mean().sqrt().item()
    assert rms(e6) <= max(rms(es), floor), (rms(e6), rms(es))
    assert e6.max().item() <= max(1.5 * es.max().item(), floor), (e6.max().item(), es.max().item())
    if tile == 0:
        assert rms(e6) <= max(rms(ed), floor), (rms(e6), rms(ed))
        assert e6.max().item() <= max(1.25 * ed.max().item(), floor), (e6.max().item(), ed.max().item())
    if epi != 2 and N % 32 == 0:  # plane output (the next GEMM's A operand): the same numbers, split exactly
        got3 = diag.gemm_x6(a3, w3, tile=tile, planes=True, **kw)
        assert torch.equal(got3.join(), got)


def test_gemm_x6_layout_and_determinism(hip_device):
    from after_amd import _lib, diag
    from after_amd._lib import AFTERHipError
    idx = diag.X6Planes(None, 50, 96).index()
    L = _lib.lib()
    for (r, p, k) in [(0, 0, 0), (1, 0, 8), (5, 2, 9), (17, 1, 33), (49, 2, 95), (15, 0, 31), (16, 0, 0)]:
        assert int(idx[r, p, k]) == L.after_gemm_x6_offset(r, p, k, 96)
    assert idx.unique().numel() == idx.numel() and int(idx.max()) < 64 * 3 * 96
    a, w, bias, _, _ = _x6_case(768, 1536, 512, 0, 3)
    a3, w3 = diag.split_x6(a.to(hip_device)), diag.split_x6(w.to(hip_device))
    outs = [diag.gemm_x6(a3, w3, bias=bias.to(hip_device)).clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    a96, w96 = diag.split_x6(a[:, :96].contiguous().to(hip_device)), diag.split_x6(w[:, :96].contiguous().to(hip_device))
    with pytest.raises(AFTERHipError):  # K not a multiple of the tile's k-parts
        diag.gemm_x6(a96, w96, tile=2)
    with pytest.raises(AFTERHipError):
        diag.gemm_x6(a3, w3, tile=77)


@pytest.mark.parametrize("M,N,K", [(768, 1536, 512), (768, 512, 1536), (192, 192, 512)])
@pytest.mark.parametrize("regime", ["sampler", "tiny_weights", "wide_range"])
def test_h3_two_piece_fp16_products_against_the_fp32_chain(M, N, K, regime, hip_device):
    """The batch sampler's Linears (gemm_h3_pipe.h): every operand as TWO fp16 pieces under an exact power-of-two scale, three
    f16 MFMAs per product block (Wh Al + Wh Ah + Wl Ah), fp32 accumulate -- on the sampler's shapes, operands with outliers
    (x 25), weights of trained-model scale and tiny ones, and activations spanning eight decades.  One wave per 16 x 16 block
    (after_diag_split_gemm), the SAME summation order in all three arithmetics, so the comparison isolates the operand form:
    bar: error vs fp64 -- rms <= 1.0 x and max <= 1.25 x that of the exact fp32 fma chain (v_mfma_f32_16x16x4_f32: the reference's
    own arithmetic); the three-plane bf16 form under the same bar beside it."""
    from after_amd import diag
    a, w, _, _, _ = _x6_case(M, N, K, 0, M + N + K)
    if regime == "tiny_weights":
        w = w * 1e-3
    if regime == "wide_range":  # magnitudes from 1e-6 to 1e2 side by side in every row: the pieces' absolute-error floor
        g = torch.Generator().manual_seed(9)
        a = a * torch.pow(10.0, torch.randint(-6, 2, a.shape, generator=g).float())
    ref = a.double() @ w.double().T
    dev = hip_device
    c32, c6, c3 = (diag.split_gemm(a.to(dev), w.to(dev), m).cpu() for m in (0, 1, 2))
    e32, e6, e3 = ((x.double() - ref).abs() for x in (c32, c6, c3))
    rms = lambda e: e.pow(2).mean().sqrt().item()
    floor = 1e-7 * ref.abs().max().item()
    assert rms(e6) <= max(rms(e32), floor), (rms(e6), rms(e32))
    assert rms(e3) <= max(rms(e32), floor), (rms(e3), rms(e32), rms(e6))
    assert e3.max().item() <= max(1.25 * e32.max().item(), floor), (e3.max().item(), e32.max().item())
    print(f"h3 {M}x{N}x{K} {regime}: rms vs fp64 -- fp32 chain {rms(e32):.3e}, bf16 x 6 {rms(e6):.3e} ({rms(e6) / rms(e32):.2f} x), "
          f"fp16 x 3 {rms(e3):.3e} ({rms(e3) / rms(e32):.2f} x); max {e32.max().item():.3e} / {e6.max().item():.3e} / {e3.max().item():.3e}")
