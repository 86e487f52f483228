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


@pytest.mark.parametrize("tile", [0, 332, 312, 322, 314, 431, 421, 631, 1431, 1332, 831, 1831, 861])
@pytest.mark.parametrize("M,N,K,epi", [(768, 1536, 512, 0), (100, 96, 128, 1), (6144, 512, 1536, 2), (49, 33, 256, 0)])
def test_gemm_x6_experimental(tile, M, N, K, epi, hip_device):
    """The experimental bf16-split GEMM (gemm_x6.hip: fp32 products as six bf16 MFMAs, fp32 accumulation)
    against fp64, every tile, ragged shapes, bias + GELU / residual epilogues: it has to be at least as
    accurate as the fp32 MFMA kernel up to the spread between summation orders (3 x its error)."""
    from after_amd import diag
    g = torch.Generator().manual_seed(M + N + K + tile)
    a = (1.3 * torch.randn(M, K, generator=g))
    a[::5, ::11] *= 25.0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if epi == 2 else None
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res.double()
    w3 = diag.split_x6(w.to(hip_device))
    got = diag.gemm_x6(a.to(hip_device), w3, bias=bias.to(hip_device),
                       residual=res.to(hip_device) if res is not None else None, epilogue=epi, tile=tile).cpu()
    base = diag.gemm(a.to(hip_device), w.to(hip_device), bias=bias.to(hip_device),
                     residual=res.to(hip_device) if res is not None else None, epilogue=epi).cpu()
    e6 = (got.double() - ref).abs().max().item()
    e32 = (base.double() - ref).abs().max().item()
    assert e6 <= max(3.0 * e32, 2e-7 * ref.abs().max().item()), (e6, e32)
