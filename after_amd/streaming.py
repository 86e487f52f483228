"""Real-time (chunk by chunk) audio-to-audio transfer: the `Streamer` of
after_scripts/export.py:145-506 on MI355X, minus the nn~ / TorchScript packaging.

Per chunk of `chunk_size` latent frames (= chunk_size * ae_ratio audio samples):

    structure(x)  emb_model_structure.encode (streaming codec) -> encoder_time.forward_stream
    timbre(x)     emb_model_timbre.encode -> rolling `previous_timbre` window -> encoder
    diffuse(c)    nb_steps Euler steps of the CFG'd denoiser, each step attending over its own
                  K/V cache slot which is rolled by the chunk length (export.py:398-416)
    decode(z)     emb_model_structure.decode (streaming codec)

All state lives in HBM inside the C-ABI handles (conv left contexts, PQMF tails, K/V caches);
one call of `forward` enqueues ~1.2k kernels on the current stream and never synchronises.

Differences from the reference, all opt-in:
  * `share_first_stream=False` samples every row of the batch independently; the reference
    (`x[:1]` then `.repeat(n)`, export.py:447-452) runs ONE diffusion per nn~ instance and copies
    it -- keep True for drop-in behaviour.
  * the initial noise can be supplied (`noise=`) so that runs are reproducible.
"""
from typing import Optional

import torch

from . import _lib
from .autoencoder.model import AutoEncoder
from .diffusion.model import RectifiedFlow


def clone_codec(ae: AutoEncoder) -> AutoEncoder:
    """A second instance of the codec with its own streaming state (the reference loads the
    TorchScript file twice: emb_model_structure / emb_model_timbre, export.py:161-166)."""
    twin = AutoEncoder(**ae.cfg_kwargs())
    twin.load_state_dict(ae.state_dict(), strict=False)
    return twin.to(next(ae.parameters()).device)


class Streamer:
    cfg_mode = _lib.CFG_EXPORT  # export.py:364-394 (clamp 0.1)
    uses_structure_encoder = True

    def __init__(self, blender: RectifiedFlow, emb_model: AutoEncoder, chunk_size: int = 4,
                 n_signal_timbre: int = 128, latent_range: float = 1.0, max_batch: int = 4,
                 max_nb_steps: int = 16, emb_model_timbre: Optional[AutoEncoder] = None,
                 share_first_stream: bool = True):
        self.blender = blender
        self.net = blender.net
        self.encoder = blender.encoder
        self.encoder_time = blender.encoder_time
        self.emb_model_structure = emb_model
        # The reference loads the codec twice (emb_model_structure / emb_model_timbre, export.py:161-166): two streaming
        # states of the same weights.  Here they are two LANES of one streaming encoder (after_ae_set_stream_lanes):
        # `structure` / `timbre` encode their own lane, and `forward` encodes both in ONE pass of the codec -- its ~120
        # launches per chunk are latency chains whose cost does not depend on the batch.  A caller-supplied second
        # codec (other weights) keeps its own handle.
        self._lanes = self.uses_structure_encoder and emb_model_timbre is None
        if not self.uses_structure_encoder or self._lanes:  # (export_midi.py: ONE codec instance, timbre encode + decode)
            self.emb_model_timbre = emb_model
        else:
            self.emb_model_timbre = emb_model_timbre
        self.chunk_size = int(chunk_size)
        self.n_signal_timbre = int(n_signal_timbre)
        self.latent_range = float(latent_range)
        self.drop_value = blender.drop_value
        self.ae_ratio = emb_model.ratio
        self.ae_latents = emb_model.z_channels
        self.zs_channels = self.net.tcond_dim
        self.zt_channels = self.net.cond_dim
        self.sr = getattr(blender, "sr", 44100)
        self.max_batch = int(max_batch)
        self.share_first_stream = bool(share_first_stream)
        # nn~ attributes (export.py:184-186)
        self.nb_steps = 1
        self.guidance_timbre = 1.0
        self.guidance_structure = 1.0
        dev = next(self.net.parameters()).device
        self.device = dev
        # export.py:189-191 (4 = nn~'s maximum batch; here max_batch)
        self.previous_timbre = torch.zeros(self.max_batch, self.ae_latents, self.n_signal_timbre,
                                           device=dev)
        # ---- streaming state in the handles
        chunk_samples = self.chunk_size * self.ae_ratio
        self.emb_model_structure.enable_streaming((2 if self._lanes else 1) * self.max_batch, chunk_samples)
        self.emb_model_structure.reset_state()  # (the modules may have streamed before: a Streamer starts from silence)
        self.emb_model_structure.set_stream_lanes(self.max_batch if self._lanes else 0)
        if self.emb_model_timbre is not self.emb_model_structure:
            self.emb_model_timbre.enable_streaming(self.max_batch, chunk_samples)
            self.emb_model_timbre.reset_state()
        self._lane_passes = [0, 0]  # chunks each lane has seen (one pass over both needs them at the same parity)
        if self.encoder_time is not None:
            self.encoder_time.enable_streaming(self.max_batch, self.chunk_size)
        rows = 3 * (1 if share_first_stream else self.max_batch)
        # export.py:73-79: MHAttention.max_cache_size = LOCAL_ATTENTION_SIZE
        self.net.enable_streaming_cache(max_diffusion_steps=int(max_nb_steps), max_batch_size=rows,
                                        max_frames=self.chunk_size)
        self.max_nb_steps = int(max_nb_steps)
        self.blender.cfg_mode = self.cfg_mode
        if self.encoder_time is not None:
            self.encoder_time.reset_state()
        self.net.reset_cache()

    # ------------------------------------------------------------ nn~ attribute accessors
    def get_guidance_timbre(self):
        return self.guidance_timbre

    def set_guidance_timbre(self, v: float):
        self.guidance_timbre = float(v)
        return 0

    def get_guidance_structure(self):
        return self.guidance_structure

    def set_guidance_structure(self, v: float):
        self.guidance_structure = float(v)
        return 0

    def get_nb_steps(self):
        return self.nb_steps

    def set_nb_steps(self, n: int):
        if not 0 < int(n) <= self.max_nb_steps:
            raise ValueError(f"nb_steps {n} outside (0, {self.max_nb_steps}] (cache slots)")
        self.nb_steps = int(n)
        return 0

    def reset(self):
        """Start of a new stream: zero the codec / encoder contexts, K/V caches and timbre window."""
        self.emb_model_structure.reset_state()
        if self.emb_model_timbre is not self.emb_model_structure:
            self.emb_model_timbre.reset_state()
        if self.encoder_time is not None:
            self.encoder_time.reset_state()
        self.net.reset_cache()
        self.previous_timbre.zero_()
        self._lane_passes = [0, 0]

    # ------------------------------------------------------------ export.py:398-416
    @torch.no_grad()
    def sample(self, x_last, cond, time_cond):
        return self.net.cfg_sample(x_last, cond, time_cond, self.nb_steps, self.guidance_timbre,
                                   self.guidance_structure, self.drop_value, self.cfg_mode)

    # ------------------------------------------------------------ export.py:418-441
    @torch.no_grad()
    def timbre(self, x):
        if self._lanes:
            z = self.emb_model_timbre.encode(x, row0=self.max_batch)[0]
            self._lane_passes[1] += 1
        else:
            z = self.emb_model_timbre.encode(x)[0]
        return self._timbre_from_latents(z)

    def _timbre_from_latents(self, z):
        n = z.shape[0]
        self.previous_timbre[:n] = torch.cat((self.previous_timbre[:n], z), -1)[..., z.shape[-1]:]
        zsem = self.encoder.forward_stream(self.previous_timbre[:n].contiguous())
        zsem = zsem / self.latent_range
        return zsem.unsqueeze(-1).repeat(1, 1, self.chunk_size)

    @torch.no_grad()
    def structure(self, x):
        if self._lanes:
            z = self.emb_model_structure.encode(x, row0=0)[0]
            self._lane_passes[0] += 1
        else:
            z = self.emb_model_structure.encode(x)[0]
        return self.encoder_time.forward_stream(z)

    # ------------------------------------------------------------ export.py:443-455
    @torch.no_grad()
    def diffuse(self, x, noise=None):
        n = x.shape[0]
        zsem = x[:, -self.zt_channels:].mean(-1) * self.latent_range
        time_cond = x[:, :self.zs_channels].contiguous()
        if noise is None:
            noise = torch.randn(n, self.ae_latents, x.shape[-1], device=x.device)
        if self.share_first_stream:
            out = self.sample(noise[:1].contiguous(), zsem[:1].contiguous(), time_cond[:1].contiguous())
            return out.repeat(n, 1, 1) if n > 1 else out
        return self.sample(noise.contiguous(), zsem.contiguous(), time_cond)

    # ------------------------------------------------------------ export.py:457-470
    @torch.no_grad()
    def diffuse_timbre(self, x, noise=None):
        """x[n, 1 + zt_channels, chunk audio samples]: channel 0 = structure AUDIO, the rest the
        timbre embedding as audio-rate signals (their mean over the chunk is used)."""
        n = x.shape[0]
        zsem = x[:, 1:].mean(-1) * self.latent_range
        time_cond = self.structure(x[:, :1].contiguous())
        if noise is None:
            noise = torch.randn(n, self.ae_latents, time_cond.shape[-1], device=x.device)
        if self.share_first_stream:
            out = self.sample(noise[:1].contiguous(), zsem[:1].contiguous(), time_cond[:1].contiguous())
            return out.repeat(n, 1, 1) if n > 1 else out
        return self.sample(noise.contiguous(), zsem.contiguous(), time_cond.contiguous())

    @torch.no_grad()
    def decode(self, z):
        return self.emb_model_structure.decode(z)

    @torch.no_grad()
    def generate(self, x, noise=None):
        return self.decode(self.diffuse(x, noise))

    @torch.no_grad()
    def generate_timbre(self, x, noise=None):
        return self.decode(self.diffuse_timbre(x, noise))

    # ------------------------------------------------------------ export.py:495-506
    project_model = None  # optional 2-D latent-map autoencoder (export.py --latent_project); None = DummyIdentity

    @torch.no_grad()
    def map2latent(self, x):
        v = x.mean(-1)
        if self.project_model is not None:
            v = self.project_model.decoder(v)
        return v.unsqueeze(-1).repeat(1, 1, x.shape[-1])

    @torch.no_grad()
    def latent2map(self, x):
        v = x.mean(-1)
        if self.project_model is not None:
            v = self.project_model.encoder(v)
        return v.unsqueeze(-1).repeat(1, 1, x.shape[-1])

    # ------------------------------------------------------------ export.py:486-493
    @torch.no_grad()
    def forward(self, x, noise=None):
        """x[n, 2, chunk_size * ae_ratio]: channel 0 = structure audio, 1 = timbre audio."""
        x = _lib.require_gpu_tensor(x, "x")
        if x.dim() != 3 or x.shape[1] != 2 or x.shape[-1] != self.chunk_size * self.ae_ratio:
            raise ValueError(f"forward expects [n, 2, {self.chunk_size * self.ae_ratio}], got {tuple(x.shape)}")
        n = x.shape[0]
        if self._lanes and n == self.max_batch and (self._lane_passes[0] - self._lane_passes[1]) % 2 == 0:
            # both lanes in one pass of the codec: rows [0, n) the structure inputs, [n, 2 n) the timbre inputs
            z2 = self.emb_model_structure.encode(x.transpose(0, 1).reshape(2 * n, 1, x.shape[-1]).contiguous(), row0=0)[0]
            self._lane_passes[0] += 1
            self._lane_passes[1] += 1
            structure = self.encoder_time.forward_stream(z2[:n].contiguous())
            timbre = self._timbre_from_latents(z2[n:].contiguous())
        else:
            structure = self.structure(x[:, :1].contiguous())
            timbre = self.timbre(x[:, 1:].contiguous())
        return self.generate(torch.cat((structure, timbre), 1), noise)

    __call__ = forward



class MidiStreamer(Streamer):
    """`Streamer` of after_scripts/export_midi.py:150-470 (MIDI-conditioned models: no structure
    encoder, `time_cond` is a 128-row piano roll built from `n_poly` (pitch, velocity) signal pairs,
    CFG rows (c, tc) / (c, -4) / (-4, -4) with factor g_s / max(g_t, 0.1), :329-358)."""
    cfg_mode = _lib.CFG_MIDI
    uses_structure_encoder = False

    def __init__(self, blender: RectifiedFlow, emb_model: AutoEncoder, n_poly: int = 4, **kw):
        if blender.encoder_time is not None:
            raise ValueError("MidiStreamer is for MIDI-structure models (encoder_time = None, midi.gin:66)")
        if getattr(blender, "post_encoder", None) is not None:
            # export_midi.py:393-394 applies `post_encoder.forward_stream(zsem)` behind the timbre encoder when the model has
            # one (and :109-110 chains it for the embedding plot).  No shipped configuration binds a post_encoder and its
            # network class is not part of this build: refuse instead of silently conditioning on different timbre vectors.
            raise NotImplementedError(
                "MidiStreamer: the model has a post_encoder, which export_midi.py:393-394 applies to the timbre embedding "
                "(post_encoder.forward_stream); it is not built here -- pass a model without it")
        self.n_poly = int(n_poly)
        super().__init__(blender, emb_model, **kw)

    def structure(self, x):
        raise AttributeError("the MIDI streamer has no structure encoder (export_midi.py)")

    @torch.no_grad()
    def timbre(self, x):
        """export_midi.py:399-414: like the audio streamer, repeated over the chunk's latent frames."""
        z = self.emb_model_timbre.encode(x)[0]
        n = z.shape[0]
        self.previous_timbre[:n] = torch.cat((self.previous_timbre[:n], z), -1)[..., z.shape[-1]:]
        zsem = self.encoder.forward_stream(self.previous_timbre[:n].contiguous())
        return zsem.unsqueeze(-1).repeat(1, 1, z.shape[-1]) / self.latent_range

    @torch.no_grad()
    def piano_roll(self, notes):
        """export_midi.py:424-432.  notes: [n, 2 * n_poly, T] = (pitch, velocity) rows per voice.
        Where voice i sounds at frame j (velocity > 0), the rows of every pitch that voice holds
        within the chunk get velocity(j) / 128; later voices overwrite earlier ones.  Only stream 0
        conditions the diffusion (the reference samples `x[:1]`), so the roll is [1, 128, T]."""
        T = notes.shape[-1]
        n_rows = self.zs_channels  # 128 for MIDI models (midi.gin:13); the reference hard-codes it
        tc = torch.zeros(1, n_rows, T, device=notes.device)
        jj = torch.arange(T, device=notes.device)
        for i in range(self.n_poly):
            pitch = notes[0, 2 * i].long().clamp(0, n_rows - 1)
            vel = notes[0, 2 * i + 1]
            on = vel > 0
            rows = pitch[:, None].expand(T, T)[:, on]          # every held pitch x sounding frames
            cols = jj[None, :].expand(T, T)[:, on]
            tc[0].index_put_((rows.reshape(-1), cols.reshape(-1)), (vel[None, :].expand(T, T)[:, on] / 128).reshape(-1))
        return tc

    @torch.no_grad()
    def diffuse(self, x, noise=None):
        """export_midi.py:416-441.  x: [n, 2 * n_poly + zt_channels, T]."""
        n = x.shape[0]
        zsem = x[:, -self.zt_channels:].mean(-1) * self.latent_range
        time_cond = self.piano_roll(x[:, :2 * self.n_poly])
        if noise is None:
            noise = torch.randn(n, self.ae_latents, x.shape[-1], device=x.device)
        out = self.sample(noise[:1].contiguous(), zsem[:1].contiguous(), time_cond)
        return out.repeat(n, 1, 1) if n > 1 else out

    def forward(self, x, noise=None):
        raise AttributeError("export_midi.py registers timbre / diffuse / generate / decode only")

    __call__ = forward
