"""The exported-codec surface: what `after_scripts/export_autoencoder.py` packages as
`export.ts` / `export_stream.ts` and what the rest of AFTER calls as `emb_model`
(`emb_model.encode(x) -> z`, `.decode(z) -> x`, `.forward(x)`; prepare_dataset.py:317-323,
update_dataset.py:59-61, export.py:161-166).

Three variants, as in the reference:
  offline            AE_notcausal / AE_causal, `export.ts`          (export_autoencoder.py:235-265)
  causal streaming   AE_causal under cc.use_cached_conv(True), `export_stream.ts` (:293-303):
                     the HIP codec keeps every conv's left context in HBM (after_ae_enable_streaming)
  non-causal stream  `export_stream.ts` of a non-causal codec (:305-312): `model.encoder` is the
                     cached twin (cached centred-padding convs with delay compensation:
                     after_ae_enable_encoder_streaming; its latents lag by `encoder_delay` frames),
                     every GroupNorm is CachedGroupNorm(stream=True) (sliding-window statistics,
                     also in the offline decoder twin: after_ae_set_decoder_gn_window), and
                     AE_notcausal.decode (:128-153) decodes the last `n_fade` latent frames again in
                     front of each chunk and cross-fades the overlap; the output lags by n_fade frames.
                     The windows are the lengths of the export script's first calls (131072 samples,
                     64 latent frames); pass others through `gn_window_samples` / `gn_window_frames`.
"""
import torch

from .. import _lib
from .model import AutoEncoder


class ExportedAutoEncoder:

    def __init__(self, model: AutoEncoder, stream: bool = False, n_fade: int = 4, max_batch: int = 4,
                 chunk_frames: int = 4, gn_window_samples: int = 131072, gn_window_frames: int = 64,
                 prime_with_silence: bool = False):
        """prime_with_silence: the reference runs one encode + decode of 131072 zero samples through the streaming twin
        before scripting it (export_autoencoder.py:296-298, :314-316), so export_stream.ts starts -- and its saved state
        is -- the codec's response to silence (non-zero: biases, Snake), not zeros.  True reproduces that start (here and
        after every reset()): `gn_window_samples` of zeros, fed chunk by chunk; False (default) starts from zero history.
        The reference feeds the silence as ONE 131072-sample call: for the causal codec the two are the same stream, for the
        non-causal twin (windowed GroupNorm, cross-fade buffers) the chunked priming is an approximation of that start, not
        a bit-equal copy -- and costs one encode + decode per chunk on every reset()."""
        if stream:
            # The reference builds the streaming graph as SEPARATE twins of the trained codec
            # (export_autoencoder.py:283-312: new modules under cc.use_cached_conv(True) + load_state_dict), so
            # the caller's `model` -- the offline export.ts, RectifiedFlow.emb_model, embed_dataset -- stays
            # stateless.  Streaming state (cached convs, windowed GroupNorm) therefore lives in a private copy.
            twin = AutoEncoder(**model.cfg_kwargs())
            twin.load_state_dict(model.state_dict(), strict=True)  # (built from the same cfg: a missing or extra key is a bug)
            model = twin.to(next(model.parameters()).device)
        self.model = model
        self.comp_ratio = model.ratio
        self.latent_size = model.z_channels
        self.target_channels = 1
        self.n_fade = int(n_fade)
        self.stream = bool(stream)
        self.causal = model.cfg["padding_mode"] == "causal"
        self.max_batch = int(max_batch)
        dev = next(model.parameters()).device
        if self.stream and self.causal:
            model.enable_streaming(self.max_batch, chunk_frames * self.comp_ratio)
        elif self.stream:
            # :305-312: cached encoder twin + CachedGroupNorm.stream on both twins
            self.encoder_delay = model.enable_encoder_streaming(self.max_batch, chunk_frames * self.comp_ratio,
                                                                gn_window_samples=gn_window_samples)
            if model.cfg["use_norm"]:
                model.set_decoder_gn_window(self.max_batch, (chunk_frames + self.n_fade) * self.comp_ratio,
                                            window_latent_frames=gn_window_frames)
            # export_autoencoder.py:62-65 (4 = nn~'s maximum batch; here max_batch)
            self.out_buffer = torch.zeros(self.max_batch, 1, self.comp_ratio * self.n_fade, device=dev)
            self.z_buffer = torch.zeros(self.max_batch, self.latent_size, self.n_fade, device=dev)
            self.alpha = torch.linspace(0, 1, self.n_fade * self.comp_ratio, device=dev)[None, None, :]
        self.prime_with_silence = bool(prime_with_silence) and self.stream
        self._chunk_samples = chunk_frames * self.comp_ratio
        self._prime_chunks = max(1, gn_window_samples // self._chunk_samples)
        if self.prime_with_silence:
            self._prime()

    @torch.no_grad()
    def _prime(self):
        dev = next(self.model.parameters()).device
        zeros = torch.zeros(self.max_batch, 1, self._chunk_samples, device=dev)
        for _ in range(self._prime_chunks):
            self.decode(self.encode(zeros))

    def reset(self):
        if self.stream and self.causal:
            self.model.reset_state()
        elif self.stream:
            self.model.reset_state()
            self.out_buffer.zero_()
            self.z_buffer.zero_()
        if self.prime_with_silence:
            self._prime()

    @torch.no_grad()
    def encode(self, x):
        return self.model.encode(x)[0]

    @torch.no_grad()
    def decode(self, z):
        if not self.stream or self.causal:
            return self.model.decode(z)
        z = _lib.require_gpu_tensor(z, "z")
        n = z.shape[0]
        if n > self.max_batch:
            raise ValueError(f"batch {n} exceeds max_batch {self.max_batch}")
        nf = self.comp_ratio * self.n_fade
        z = torch.cat((self.z_buffer[:n], z), -1)
        x = self.model.decode(z.contiguous())
        self.z_buffer[:n] = z[:, :, -self.n_fade:]
        x[..., :nf] = (1 - self.alpha) * self.out_buffer[:n] + self.alpha * x[..., :nf]
        self.out_buffer[:n] = x[:, :, -nf:]
        return x[..., :-nf].contiguous()

    @torch.no_grad()
    def forward(self, x):
        """:106-121 / :235-249: decode(encode(x)) without the cross-fade.  On a streaming export this runs the
        streaming twin (stateful: cached convs / windowed GroupNorm advance), as export_stream.ts's forward does."""
        return self.model.decode(self.model.encode(x)[0])

    __call__ = forward


@torch.no_grad()
def embed_dataset(emb_model, waveforms, batch_size: int = 32):
    """prepare_dataset.py:313-323 / update_dataset.py:52-61: z for every `num_signal`-sample
    chunk, `batch_size` chunks per codec call.  waveforms: [N, L] or [N, 1, L] (any device;
    moved to the codec's device per batch).  Returns [N, Z, L / ratio] on the CPU."""
    if isinstance(emb_model, ExportedAutoEncoder):
        if emb_model.stream:
            raise ValueError("embed_dataset needs the offline codec (export.ts): a streaming export is stateful -- its "
                             "latents lag by the encoder delay and depend on the history of earlier calls")
        model = emb_model.model
    else:
        model = emb_model
    dev = next(model.parameters()).device
    w = waveforms if waveforms.dim() == 3 else waveforms[:, None, :]
    model.reserve(min(batch_size, w.shape[0]), w.shape[-1])
    out = []
    for i in range(0, w.shape[0], batch_size):
        z = model.encode(w[i:i + batch_size].to(dev, torch.float32).contiguous())[0]
        out.append(z.cpu())
    return torch.cat(out, 0)
