"""Checkpoint + config ingestion: run a trained AFTER model on the MI355X path.

What the reference does at export time (after_scripts/export.py:52-101,
export_autoencoder.py:19-45): parse `<folder>/config.gin`, construct the gin-configured
`RectifiedFlow()` / `AutoEncoder()`, pick `checkpoint<step>_EMA.pt` (largest step unless
given) and `load_state_dict(ckpt["model_state"], strict=False)`.  Here the gin file is read by
`after_amd.ginfile` (no gin dependency) and the constructors are after_amd's; weight-norm
folding and eval-BatchNorm folding happen inside `*_create` when the HIP handle is built, so the
state dict is consumed in the reference's own layout.

    model = load_diffusion("runs/my_model", device="cuda:0")            # RectifiedFlow
    ae    = load_autoencoder("runs/my_codec", device="cuda:0")          # AutoEncoder
"""
import os
import re
from typing import Optional

import torch

from .autoencoder.model import AutoEncoder, ReluBottleneck, TanhBottleneck, VAEBottleneck
from .diffusion.model import RectifiedFlow
from .diffusion.networks.ecapa_encoder import ECAPATDNN
from .diffusion.networks.encoder import Encoder1D
from .diffusion.networks.transformerv2 import DenoiserV2
from .ginfile import GinConfig, GinError, Ref

# arguments of the reference constructors that do not change the inference graph
_IGNORED = {
    "DenoiserV2": set(),
    "ECAPATDNN": {"activation"},
    "Encoder1D": set(),
    "AutoEncoder": set(),
}


def find_checkpoint(folder: str, step: Optional[int] = None, ema: bool = True) -> str:
    """export.py:52-66 / export_autoencoder.py:32-41: `checkpoint<step>[_EMA].pt`, the largest
    step when none is given."""
    suffix = "_EMA.pt" if ema else ".pt"
    if step is None:
        steps = []
        for f in os.listdir(folder):
            m = re.fullmatch(r"checkpoint(\d+)" + re.escape(suffix), f)
            if m:
                steps.append(int(m.group(1)))
        if not steps:
            raise FileNotFoundError(f"no checkpoint*{suffix} in {folder}")
        step = max(steps)
    path = os.path.join(folder, f"checkpoint{step}{suffix}")
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return path


def _padding_mode(cfg: GinConfig, scope: str) -> str:
    # `encoder_time/convs.get_padding.mode = 'causal'` (base.gin:55) / baseAE.gin:32-33
    return cfg.query("get_padding", "mode", scope, default="centered")


def _build(cfg: GinConfig, ref, overrides=None):
    """Instantiate the after_amd twin of a gin reference `@[scope/]module.Class()`."""
    if ref is None:
        return None
    if not isinstance(ref, Ref):
        raise GinError(f"expected a configurable reference, got {ref!r}")
    name = ref.selector.split(".")[-1]
    kw = cfg.kwargs(ref.selector, ref.scope)
    for k in _IGNORED.get(name, ()):
        kw.pop(k, None)
    kw.update(overrides or {})
    if name == "DenoiserV2":
        return DenoiserV2(**kw)
    if name == "ECAPATDNN":
        return ECAPATDNN(**kw)
    if name == "Encoder1D":
        return Encoder1D(padding_mode=_padding_mode(cfg, ref.scope), **kw)
    raise NotImplementedError(f"after_amd has no MI355X implementation of {ref.selector} "
                              "(built: DenoiserV2, ECAPATDNN, Encoder1D, AutoEncoder)")


def diffusion_from_config(cfg: GinConfig, device="cuda:0", in_size: Optional[int] = None,
                          n_signal: Optional[int] = None) -> RectifiedFlow:
    """`RectifiedFlow()` under the parsed config.  IN_SIZE / N_SIGNAL are bound at train time
    from the codec probe (after_scripts/train.py:71-86); pass them when the file has None."""
    if in_size is not None:
        cfg.bind("%IN_SIZE", in_size)
    if n_signal is not None:
        cfg.bind("%N_SIGNAL", n_signal)
    if cfg.macro("IN_SIZE", None) is None:
        raise GinError("config has IN_SIZE = None: pass in_size= (the codec's latent size)")
    if cfg.macro("N_SIGNAL", None) is None:
        cfg.bind("%N_SIGNAL", 128)  # only sizes an unused index buffer with rotary embeddings
    base = cfg.kwargs("Base")
    if not base:
        base = cfg.kwargs("RectifiedFlow")
    if "net" not in base:
        raise GinError("config has no Base.net binding")
    # Base.time_transform (model.py:136-137) is applied by prep_data, i.e. to TRAINING batches only; Base.post_encoder is
    # stored and never called (model.py:38).  Neither is on the sampling path: a config that binds them loads, the bindings
    # are not built (their state-dict entries are in the allowed-unexpected list of load_diffusion).
    net = _build(cfg, base["net"])
    enc = _build(cfg, base.get("encoder"))
    enc_t = _build(cfg, base.get("encoder_time"))
    model = RectifiedFlow(net=net, sr=base.get("sr", 44100), encoder=enc, encoder_time=enc_t,
                          drop_value=base.get("drop_value", -4.0),
                          drop_rate=base.get("drop_rate", 0.2), device=device)
    return model


def autoencoder_from_config(cfg: GinConfig, device="cuda:0") -> AutoEncoder:
    kw = cfg.kwargs("AutoEncoder")
    if not kw:
        raise GinError("config has no AutoEncoder bindings")
    # the bottleneck has no parameters, so the checkpoint cannot reveal a mismatch: resolve the
    # binding (baseAE.gin:41-43,58) and refuse what the MI355X path does not compute
    b = kw.pop("bottleneck", None)
    if isinstance(b, Ref):
        name = b.selector.split(".")[-1]
        if name == "ReluBottleneck":
            b = ReluBottleneck(**cfg.kwargs(b.selector, b.scope))
        elif name == "TanhBottleneck":
            b = TanhBottleneck(**cfg.kwargs(b.selector, b.scope))
        elif name == "VAEBottleneck":
            b = VAEBottleneck()
        elif name == "Identity":
            raise NotImplementedError("bottleneck = nn.Identity is not callable as a bottleneck in the "
                                      "reference either (encode unpacks `z, regloss`)")
        else:
            b = name  # anything else -> refused by AutoEncoder with the reason
    a = kw.pop("activation", None)
    if isinstance(a, Ref):
        a = a.selector
    return AutoEncoder(padding_mode=_padding_mode(cfg, ""), bottleneck=b, activation=a, **kw).to(device)


def _load_state(path, trust_pickle: bool = False):
    """`weights_only=True` unless the caller vouches for the file: a full unpickle executes
    arbitrary code from the checkpoint (the reference's own `torch.load`, export.py:87, does)."""
    try:
        d = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if not trust_pickle:
            raise RuntimeError(f"{path} holds objects beyond tensors / plain containers ({e}); pass "
                               "trust_pickle=True to unpickle it fully if you trust its origin") from e
        d = torch.load(path, map_location="cpu", weights_only=False)
    return d["model_state"] if "model_state" in d else d


def _check_load(res, allowed_missing=(), allowed_unexpected=()):
    bad_m = [k for k in res.missing_keys if not any(re.search(p, k) for p in allowed_missing)]
    bad_u = [k for k in res.unexpected_keys if not any(re.search(p, k) for p in allowed_unexpected)]
    if bad_m or bad_u:
        raise RuntimeError(f"checkpoint does not fit the configured model: missing {bad_m[:8]} "
                           f"unexpected {bad_u[:8]}")


def load_diffusion(folder: str, step: Optional[int] = None, device="cuda:0", ema: bool = True,
                   in_size: Optional[int] = None, n_signal: Optional[int] = None,
                   trust_pickle: bool = False) -> RectifiedFlow:
    """export.py:52-101.  Networks outside the sampling path that the checkpoint also holds
    (classifier, post_encoder, EMA shadow of the codec) are skipped, like `strict=False`
    skips them in the reference once those attributes are None."""
    cfg = GinConfig.parse_file(os.path.join(folder, "config.gin"))
    model = diffusion_from_config(cfg, device, in_size, n_signal)
    sd = _load_state(find_checkpoint(folder, step, ema), trust_pickle)
    res = model.load_state_dict(sd, strict=False)
    _check_load(res, allowed_unexpected=(r"^classifier\.", r"^post_encoder\.", r"^emb_model\.",
                                         r"^extra_modules\.", r"^time_transform\."))
    return model


def load_autoencoder(folder: str, step: Optional[int] = None, device="cuda:0",
                     trust_pickle: bool = False) -> AutoEncoder:
    """export_autoencoder.py:19-45 (codec checkpoints carry no EMA suffix, trainer.py:352-361)."""
    cfg = GinConfig.parse_file(os.path.join(folder, "config.gin"))
    ae = autoencoder_from_config(cfg, device)
    sd = _load_state(find_checkpoint(folder, step, ema=False), trust_pickle)
    res = ae.load_state_dict(sd, strict=False)
    # bottleneck has no parameters; CachedGroupNorm `pad` buffers are re-created lazily
    _check_load(res, allowed_missing=(r"\.pad$", ), allowed_unexpected=(r"\.pad$", r"^bottleneck\."))
    return ae
