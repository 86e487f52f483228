"""SURVEY 8(f3) at the reference's batch sizes (-m gpu): `prepare_dataset.py:84, 314-340` and
`update_dataset.py:59-61` call `emb_model.encode` on [8 .. 32, 1, 524288] batches of whole clips.

base AutoEncoder.encode on 16 and 32 whole clips through `embed_dataset` (one codec call per batch):
the first and the last clip against the CPU oracle (<= 1e-4 x max|oracle|, the codec's bar everywhere
else), every clip against its own single-clip run of the same handle (other tile shapes and
statistics slots: fp32 round-off only), the regulariser-free latents finite."""
import os

import pytest
import torch

import oracle
from after_amd import pipeline
from after_amd.autoencoder import ExportedAutoEncoder, embed_dataset
from fixtures import max_abs, rel_l2

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def base_codec(hip_device):
    old = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    model, _, acfg = pipeline.build_models("base", "baseAE", hip_device, seed=11)
    yield model.emb_model, acfg
    torch.set_num_threads(old)


@pytest.mark.parametrize("B", [16, 32])
def test_whole_clip_batches_of_the_dataset_scripts(B, base_codec, hip_device):
    ae, acfg = base_codec
    sd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    g = torch.Generator().manual_seed(100 + B)
    w = 0.1 * torch.randn(B, 524288, generator=g)
    w[1] *= 4.0  # clips of different loudness side by side: per-clip GroupNorm statistics must not mix
    w[B - 2, 100000:] = 0.0  # half a clip of digital silence
    z = embed_dataset(ExportedAutoEncoder(ae), w, batch_size=B)  # one encode call over all B clips
    assert z.shape == (B, 64, 256) and torch.isfinite(z).all()
    for i in (0, B - 1):
        want = oracle.ae_encode(sd, w[i:i + 1, None, :], acfg)
        assert max_abs(z[i:i + 1], want) < 1e-4 * want.abs().max().item(), (i, max_abs(z[i:i + 1], want), rel_l2(z[i:i + 1], want))
    for i in range(B):
        zi = ae.encode(w[i:i + 1, None, :].to(hip_device))[0].cpu()
        assert max_abs(z[i:i + 1], zi) < 2e-5 * max(zi.abs().max().item(), 1.0), (i, max_abs(z[i:i + 1], zi))
    # the loop of the dataset scripts: batches of 8 with a ragged last one
    z8 = embed_dataset(ae, w[:B - 3], batch_size=8)
    assert z8.shape == (B - 3, 64, 256) and max_abs(z8, z[:B - 3]) < 2e-5 * z.abs().max().item()
