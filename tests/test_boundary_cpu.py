"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/after_hip.h declares, the host mirrors keep the reference's
state-dict layout, and the product path refuses to run without a GPU (no CPU
fallback).  No compute calls."""
import os
import re

import pytest
import torch

from after_amd import DenoiserV2, RectifiedFlow, _lib, configs
from fixtures import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "after_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(after_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), f"libafter_hip.so does not export {s}"
    # and the ctypes table covers the whole header
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    assert b"gfx950" in L.after_version()


def test_library_is_built_for_gfx950_only():
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    # --offloading extracts the bundles next to its input: work on a copy in a scratch directory
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        so = shutil.copy(_lib.LIB_PATH, tmp)
        out = subprocess.run([objdump, "--offloading", so], capture_output=True, text=True, cwd=tmp)
    txt = out.stdout + out.stderr
    archs = set(re.findall(r"gfx[0-9a-f]+", txt))
    assert archs == {"gfx950"}, archs


@pytest.mark.parametrize("case", ["denoiser_micro", "denoiser_micro_midi", "denoiser_tiny",
                                  "denoiser_base", "denoiser_midi"])
def test_denoiser_state_dict_matches_reference_layout(case):
    fx = Fixture(case)
    net = DenoiserV2(**configs.diffusion_config(fx.meta["config"])["net"])
    sd = net.state_dict()
    ref = fx.meta["shapes"]
    assert set(sd) == set(ref)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k]), k
    # strict load of a reference-layout state dict
    net.load_state_dict(fx.state_dict(), strict=True)


@pytest.mark.parametrize("case", ["unet_micro", "unet_micro_flat", "unet_micro_attn"])
def test_unet1d_state_dict_matches_reference_layout(case):
    """UNET1D (with and without the self-attention layers of n_attn_layers > 0): the keys and shapes the REFERENCE module had when
    the fixture was generated (meta["shapes"]) are exactly the container's."""
    from after_amd.diffusion.networks.unet1d import UNET1D
    fx = Fixture(case)
    net = UNET1D(**configs.unet_config(fx.meta["config"]))
    sd, ref = net.state_dict(), fx.meta["shapes"]
    assert set(sd) == set(ref)
    assert all(tuple(v.shape) == tuple(ref[k]) for k, v in sd.items())
    net.load_state_dict(fx.state_dict(), strict=True)


@pytest.mark.parametrize("case", ["ae_micro", "ae_micro_nopqmf", "ae_micro_snake1", "ae_micro_noise"])
def test_autoencoder_state_dict_matches_reference_layout(case):
    """The codec with and without a filter bank (pqmf_bands = 1: the reference's DummyIdentity has no parameters, neither has ours):
    every key of the reference module exists with its shape; the container adds nothing but the reference's own streaming buffers."""
    from after_amd.autoencoder import AutoEncoder
    fx = Fixture(case)
    cfg = configs.autoencoder_config(fx.meta["config"])
    cfg.pop("bottleneck", None)
    sd, ref = AutoEncoder(**cfg).state_dict(), fx.meta["shapes"]
    assert set(sd) <= set(ref), sorted(set(sd) - set(ref))[:5]
    assert all(k.endswith(".pad") for k in set(ref) - set(sd)), sorted(set(ref) - set(sd))[:5]  # CachedGroupNorm's stream buffers
    assert all(tuple(v.shape) == tuple(ref[k]) for k, v in sd.items())


def test_product_path_refuses_cpu_tensors():
    net = DenoiserV2(**configs.diffusion_config("micro")["net"])
    model = RectifiedFlow(net=net, sr=44100)
    x = torch.zeros(1, 16, 8)
    with pytest.raises(_lib.AFTERHipError):
        net(x, torch.zeros(1), torch.zeros(1, 6), torch.zeros(1, 12, 8))
    with pytest.raises(_lib.AFTERHipError):
        model.sample(x, torch.zeros(1, 6), torch.zeros(1, 12, 8), 2)


def test_rectified_flow_rejects_foreign_networks():
    with pytest.raises(TypeError):
        RectifiedFlow(net=torch.nn.Linear(2, 2), sr=44100)


def test_linspace_formula_matches_torch():
    """The device computes t = linspace(0,1,N+1)[:-1] with at::linspace's fp32 formula
    (embed_rows_kernel); check the formula itself against torch on the CPU."""
    import numpy as np
    for n in (1, 2, 3, 4, 10, 15, 16, 50, 100, 333):
        pts = n + 1
        step = np.float32(1.0) / np.float32(pts - 1)
        got = []
        for s in range(n):
            if s < pts // 2:
                got.append(np.float32(step * np.float32(s)))
            else:
                # fused multiply-add: one rounding (exact product in float64)
                got.append(np.float32(1.0 - float(step) * (pts - s - 1)))
        want = torch.linspace(0, 1, n + 1)[:-1].numpy()
        assert np.array_equal(np.asarray(got, dtype=np.float32), want), n


def test_midi_streamer_refuses_a_post_encoder():
    """export_midi.py:393-394 applies `post_encoder.forward_stream` to the timbre embedding when the model has one; the
    network is not part of this build, so MidiStreamer must refuse such a model (an error, not a silent skip) -- before it
    touches the device."""
    import pytest as _pytest
    from after_amd import MidiStreamer

    class _Blender:  # the attributes MidiStreamer reads before anything else
        encoder_time = None
        post_encoder = torch.nn.Identity()

    with _pytest.raises(NotImplementedError, match="post_encoder"):
        MidiStreamer(_Blender(), emb_model=None)
