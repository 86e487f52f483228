"""Checkpoint + config.gin ingestion: the gin subset reader and the loaders
that mirror after_scripts/export.py:52-101 / export_autoencoder.py:19-45.  CPU only: the
containers are ordinary nn.Modules until a forward builds the HIP handle."""
import os

import pytest
import torch

from after_amd import checkpoint, configs, pipeline
from after_amd.ginfile import GinConfig, GinError, Ref

# written in the style of gin.operative_config_str() (flat bindings, short selectors, wrapped
# lines) -- what model.py:262-265 leaves next to the checkpoints
OPERATIVE = r"""
# Macros:
# ==============================================================================
IN_SIZE = 16
N_SIGNAL = 128
SR = 44100
ZS_CHANNELS = 12
ZT_CHANNELS = 6
LOCAL_ATTENTION_SIZE = 8

# Parameters for Base:
# ==============================================================================
Base.drop_rate = 0.2
Base.drop_value = -4.0
Base.encoder = @encoder/ECAPATDNN()
Base.encoder_time = \
    @encoder_time/Encoder1D()
Base.classifier = @classifier/Encoder1D()
Base.net = @DenoiserV2()
Base.sr = %SR
Base.time_transform = None

# Parameters for DenoiserV2:
DenoiserV2.attention_chunk_size = 4
DenoiserV2.causal = True
DenoiserV2.cond_dim = %ZT_CHANNELS
DenoiserV2.dropout = 0.1
DenoiserV2.embed_dim = 128
DenoiserV2.local_attention_size = %LOCAL_ATTENTION_SIZE
DenoiserV2.mlp_multiplier = 3
DenoiserV2.n_channels = %IN_SIZE
DenoiserV2.n_layers = 2
DenoiserV2.noise_embed_dims = 64
DenoiserV2.pos_emb_type = 'rotary'
DenoiserV2.seq_len = %N_SIGNAL
DenoiserV2.tcond_dim = %ZS_CHANNELS

# Parameters for encoder/ECAPATDNN:
encoder/ECAPATDNN.attention_channels = 16
encoder/ECAPATDNN.channels = [64, 64, 64, 128]
encoder/ECAPATDNN.dilations = [1, 1, 1, 1]
encoder/ECAPATDNN.global_context = True
encoder/ECAPATDNN.groups = [1, 1, 1, 1]
encoder/ECAPATDNN.in_size = %IN_SIZE
encoder/ECAPATDNN.kernel_sizes = [3, 3, 3, 3]
encoder/ECAPATDNN.out_dim = %ZT_CHANNELS
encoder/ECAPATDNN.pooling = True
encoder/ECAPATDNN.regularisation = 'ac'
encoder/ECAPATDNN.res2net_scale = 8
encoder/ECAPATDNN.se_channels = 16
encoder/ECAPATDNN.spherical_normalisation = False
encoder/ECAPATDNN.use_tanh = False

# Parameters for encoder_time/Encoder1D:
encoder_time/Encoder1D.ac_regularisation = True
encoder_time/Encoder1D.average_out = False
encoder_time/Encoder1D.channels = [16, 32, 32, 32, %ZS_CHANNELS]
encoder_time/Encoder1D.in_size = %IN_SIZE
encoder_time/Encoder1D.ratios = [1, 1, 1, 1]
encoder_time/Encoder1D.spherical_normalization = False
encoder_time/Encoder1D.upscale_out = False
encoder_time/Encoder1D.use_tanh = False
encoder_time/Encoder1D.vae_regularisation = False

# Parameters for encoder_time/get_padding:
encoder_time/get_padding.mode = 'causal'

# Parameters for classifier/Encoder1D:
classifier/Encoder1D.channels = [64, 64, 64, 64, %ZT_CHANNELS]
classifier/Encoder1D.ratios = [1, 2, 2, 2, 1]
classifier/Encoder1D.average_out = True

# Parameters for Base.fit:
Base.fit.drop_targets = [0, 1]
Base.fit.lr = 0.0001
"""

BLOCK_AE = """
from __gin__ import dynamic_registration
from after.autoencoder.networks import SimpleNetsStream
import cached_conv

LATENT_SIZE = 16
PQMF_BANDS = 16   # Set to 1 if no pqmf
BASE_CHANNELS = 8

cached_conv.get_padding:
    mode = "causal"

SimpleNetsStream.AutoEncoder:
    in_channels = %PQMF_BANDS
    channels = %BASE_CHANNELS
    pqmf_bands =  %PQMF_BANDS
    z_channels = %LATENT_SIZE
    multipliers = [1, 2, 4, 4, 8, 8] # per block
    factors = [2, 2,
               2, 4, 4]
    dilations = [1, 3, 9]
    kernel_size = 3
    bottleneck  =  @SimpleNetsStream.ReluBottleneck()
    use_norm = False
    decoder_ratio = 1.5
    use_loudness = True
    use_noise = False

some.Trainer:
    weights={
        "feature_matching": 20.0,
        "adversarial": 0.
        }
    waveform_losses = [(1., @core.MultiResolutionSTFTLoss())]
    window = "hann#not-a-comment @nor-a-ref %NOR_A_MACRO"
"""


def test_gin_subset_reader():
    c = GinConfig.parse_string(OPERATIVE)
    assert c.macro("%SR") == 44100
    base = c.kwargs("Base")
    assert isinstance(base["net"], Ref) and base["net"].selector == "DenoiserV2" and base["net"].call
    assert base["encoder_time"].scope == "encoder_time" and base["sr"] == 44100
    assert "lr" not in base  # Base.fit is a different configurable
    net = c.kwargs("diffusion.networks.transformerv2.DenoiserV2")  # dotted-suffix matching
    assert net == configs.diffusion_config("micro")["net"]
    et = c.kwargs("Encoder1D", "encoder_time")
    assert et["channels"] == [16, 32, 32, 32, 12] and "average_out" in et
    assert c.kwargs("Encoder1D", "classifier")["average_out"] is True
    assert c.query("get_padding", "mode", "encoder_time") == "causal"
    assert c.query("get_padding", "mode", "", default="centered") == "centered"
    c.bind("%ZS_CHANNELS", 20)  # macros resolve lazily
    assert c.kwargs("Encoder1D", "encoder_time")["channels"][-1] == 20
    with pytest.raises(GinError):
        c.macro("NOPE")
    a = GinConfig.parse_string(BLOCK_AE)
    kw = a.kwargs("AutoEncoder")
    assert kw["factors"] == [2, 2, 2, 4, 4] and kw["decoder_ratio"] == 1.5 and kw["channels"] == 8
    tr = a.kwargs("Trainer")
    assert tr["weights"] == {"feature_matching": 20.0, "adversarial": 0.0}
    assert tr["window"] == "hann#not-a-comment @nor-a-ref %NOR_A_MACRO"
    assert tr["waveform_losses"][0][0] == 1.0 and tr["waveform_losses"][0][1].selector.endswith("STFTLoss")
    with pytest.raises(GinError):
        GinConfig.parse_string("A.b = __import__('os')")


def test_load_diffusion_and_autoencoder_from_run_folders(tmp_path):
    # a "trained" micro model: random weights saved the way model.py:144-176 saves them
    src, dcfg, acfg = pipeline.build_models("micro", "microAE_causal", "cpu", seed=5)
    run = tmp_path / "run"
    run.mkdir()
    (run / "config.gin").write_text(OPERATIVE)
    sd = {k: v for k, v in src.state_dict().items() if "emb_model" not in k}  # model.py:149-153
    sd["classifier.net.0.weight"] = torch.zeros(3)  # networks outside the sampling path
    torch.save({"model_state": sd, "opt_state": {}}, run / "checkpoint1000_EMA.pt")
    torch.save({"model_state": {k: torch.zeros_like(v) for k, v in sd.items()}, "opt_state": {}},
               run / "checkpoint500_EMA.pt")
    assert checkpoint.find_checkpoint(str(run)).endswith("checkpoint1000_EMA.pt")
    model = checkpoint.load_diffusion(str(run), device="cpu")
    got = model.state_dict()
    want = {k: v for k, v in src.state_dict().items() if "emb_model" not in k}
    assert set(got) == set(want)
    assert all(torch.equal(got[k], want[k]) for k in want)
    assert model.encoder_time.padding_mode == "causal" and model.sr == 44100
    old = checkpoint.load_diffusion(str(run), step=500, device="cpu")
    assert float(old.state_dict()["net.embedding.0.weight"].abs().sum()) == 0.0
    with pytest.raises(FileNotFoundError):
        checkpoint.load_diffusion(str(run), step=7, device="cpu")
    # a checkpoint that does not fit the configured architecture is an error, not a silent skip
    bad = dict(sd)
    bad.pop("net.embedding.0.weight")
    torch.save({"model_state": bad}, run / "checkpoint2000_EMA.pt")
    with pytest.raises(RuntimeError):
        checkpoint.load_diffusion(str(run), device="cpu")

    # training-only hooks (Base.time_transform: applied by prep_data to training batches, model.py:136-137; Base.post_encoder:
    # stored, never called, model.py:38): a run that bound them loads, the sampling path ignores them as the reference's does
    hooks = tmp_path / "hooks"
    hooks.mkdir()
    (hooks / "config.gin").write_text(OPERATIVE.replace("Base.time_transform = None",
                                                        "Base.time_transform = @some_training_transform\nBase.post_encoder = @post/Encoder1D()"))
    sdh = dict(sd)
    sdh["post_encoder.net.0.weight"] = torch.zeros(3)
    torch.save({"model_state": sdh, "opt_state": {}}, hooks / "checkpoint10_EMA.pt")
    mh = checkpoint.load_diffusion(str(hooks), device="cpu")
    assert all(torch.equal(mh.state_dict()[k], want[k]) for k in want)

    ae_run = tmp_path / "codec"
    ae_run.mkdir()
    (ae_run / "config.gin").write_text(BLOCK_AE)
    ae_sd = dict(src.emb_model.state_dict())
    torch.save({"model_state": ae_sd, "opt_state": {}, "dis_state": {}}, ae_run / "checkpoint42.pt")
    ae = checkpoint.load_autoencoder(str(ae_run), device="cpu")
    assert ae.cfg["padding_mode"] == "causal" and not ae.cfg["use_norm"] and ae.ratio == 2048
    got = ae.state_dict()
    assert all(torch.equal(got[k], v) for k, v in ae_sd.items())


def test_bottleneck_binding_decides_and_unsupported_activation_is_refused(tmp_path):
    """The bottleneck has no parameters, so a checkpoint cannot reveal that the codec was trained with
    TanhBottleneck (z = scale * tanh(z)) or VAEBottleneck (2x encoder channels): the config decides."""
    from after_amd.autoencoder import TanhBottleneck, VAEBottleneck
    for bn, cls in (("TanhBottleneck", TanhBottleneck), ("VAEBottleneck", VAEBottleneck)):
        cfg = GinConfig.parse_string(BLOCK_AE.replace("@SimpleNetsStream.ReluBottleneck()", f"@SimpleNetsStream.{bn}()"))
        ae_b = checkpoint.autoencoder_from_config(cfg, device="cpu")
        assert isinstance(ae_b.bottleneck, cls)
        assert ae_b.encoder_out_channels == (2 if cls is VAEBottleneck else 1) * ae_b.z_channels
    cfg = GinConfig.parse_string(BLOCK_AE.replace("@SimpleNetsStream.ReluBottleneck()", "@SimpleNetsStream.FSQ()"))
    with pytest.raises(NotImplementedError):
        checkpoint.autoencoder_from_config(cfg, device="cpu")
    cfg = GinConfig.parse_string(BLOCK_AE + "\nSimpleNetsStream.AutoEncoder.activation = @torch.nn.ReLU\n")
    with pytest.raises(NotImplementedError):
        checkpoint.autoencoder_from_config(cfg, device="cpu")
    ae = checkpoint.autoencoder_from_config(GinConfig.parse_string(BLOCK_AE), device="cpu")
    assert ae.bottleneck.scale == 3.0
    # torch.load stays weights_only unless the caller vouches for the file
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ("unpickled",))

    run = tmp_path / "run"
    run.mkdir()
    (run / "config.gin").write_text(OPERATIVE)
    with open(run / "checkpoint1_EMA.pt", "wb") as f:
        pickle.dump({"model_state": {}, "x": Evil()}, f)
    with pytest.raises(RuntimeError):
        checkpoint.load_diffusion(str(run), device="cpu")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reference_run():
    """tests/golden/ckpt_nano: a run folder + codec folder written by the REFERENCE's own code
    (make_golden.py:checkpoint_case -- `Base.save_model`, model.py:144-176, on reference modules;
    the codec in `Trainer.fit`'s layout, trainer.py:350-361) and the reference's outputs."""
    from fixtures import Fixture
    root = os.path.join(GOLDEN, "ckpt_nano")
    return Fixture("ckpt_nano"), os.path.join(root, "run"), os.path.join(root, "codec")


def test_reference_written_checkpoint_loads_and_pins_the_oracle():
    """The files the reference saved go through config.gin + checkpoint ingestion: every key of the
    sampling path is consumed (weight_g/weight_v pairs, the duplicated BatchNorm names of Encoder1D,
    the three rotary aliases), `classifier.*` is skipped, `emb_model.*` was never saved; and the
    oracle, fed the LOADED state dict, reproduces the reference's vectors."""
    import oracle
    from fixtures import max_abs
    fx, run, codec = _reference_run()
    saved = torch.load(checkpoint.find_checkpoint(run), map_location="cpu", weights_only=True)
    assert set(saved) == {"model_state", "opt_state"}
    assert sorted(saved["model_state"]) == fx.meta["run_keys"]
    assert not any("emb_model" in k for k in saved["model_state"])  # model.py:149-153
    assert any(k.startswith("classifier.") for k in saved["model_state"])
    model = checkpoint.load_diffusion(run, device="cpu")
    ae = checkpoint.load_autoencoder(codec, device="cpu")
    got = model.state_dict()
    for k, v in saved["model_state"].items():
        if not k.startswith("classifier."):
            assert k in got and torch.equal(got[k], v), k
    sd_codec = torch.load(checkpoint.find_checkpoint(codec, ema=False), map_location="cpu",
                          weights_only=True)["model_state"]
    assert sorted(sd_codec) == fx.meta["codec_keys"]
    gota = ae.state_dict()
    assert all(torch.equal(gota[k], v) for k, v in sd_codec.items())
    # oracle on the loaded weights vs the reference's outputs
    pick = lambda pre: {k[len(pre):]: v for k, v in got.items() if k.startswith(pre)}
    cfg = GinConfig.parse_file(os.path.join(run, "config.gin"))
    ncfg = cfg.kwargs("DenoiserV2")
    ecfg = cfg.kwargs("ECAPATDNN", "encoder")
    tcfg = dict(cfg.kwargs("Encoder1D", "encoder_time"), padding_mode="causal")
    nsig, (gt, gs) = fx.meta["n_signal_timbre"], fx.meta["guidance"]
    cond = oracle.ecapa_forward(pick("encoder."), fx.t("zt")[..., :nsig], ecfg)
    tc = oracle.encoder1d_forward(pick("encoder_time."), fx.t("zs"), tcfg)
    assert max_abs(cond, fx.t("cond")) < 2e-5 and max_abs(tc, fx.t("time_cond")) < 2e-5
    z = oracle.sample(pick("net."), ncfg, fx.t("x0"), cond, tc, fx.meta["nb_steps"], gt, gs)
    assert max_abs(z, fx.t("z")) < 2e-5
    y = oracle.ae_decode(gota, z, ae.cfg)
    assert max_abs(y, fx.t("y")) < 2e-5 * fx.t("y").abs().max().item()
    assert max_abs(oracle.ae_encode(gota, fx.t("audio"), ae.cfg), fx.t("z_enc")) < \
        2e-5 * fx.t("z_enc").abs().max().item()


@pytest.mark.gpu
def test_reference_written_checkpoint_reproduces_reference_outputs(hip_device):
    """f2 end to end on the MI355X path: the reference's checkpoint files + config.gin ->
    load_diffusion / load_autoencoder -> encoders, 5-step CFG sampler, decode, encode ==
    the vectors the reference computed from the same weights."""
    from fixtures import max_abs
    fx, run, codec = _reference_run()
    model = checkpoint.load_diffusion(run, device=hip_device)
    model.emb_model = checkpoint.load_autoencoder(codec, device=hip_device)
    d = lambda n: fx.t(n).to(hip_device)
    nsig, (gt, gs) = fx.meta["n_signal_timbre"], fx.meta["guidance"]
    cond = model.encoder(d("zt")[..., :nsig].contiguous())
    tc = model.encoder_time(d("zs"))
    assert max_abs(cond.cpu(), fx.t("cond")) < 1e-4 and max_abs(tc.cpu(), fx.t("time_cond")) < 1e-4
    z = model.sample(d("x0"), cond, tc, fx.meta["nb_steps"], gt, gs)
    assert max_abs(z.cpu(), fx.t("z")) < 2e-4, max_abs(z.cpu(), fx.t("z"))
    y = model.emb_model.decode(d("z")).cpu()
    assert max_abs(y, fx.t("y")) < 1e-4 * fx.t("y").abs().max().item()
    ze, reg = model.emb_model.encode(d("audio"))
    assert max_abs(ze.cpu(), fx.t("z_enc")) < 1e-4 * fx.t("z_enc").abs().max().item()
    # SimpleLatentReg of the ReluBottleneck (SimpleNetsStream.py:742-760, core.py:189-198)
    assert abs(float(reg) - float(fx.t("reg"))) < 1e-4 * max(1.0, abs(float(fx.t("reg"))))
