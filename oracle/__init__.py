"""CPU oracle for the AFTER latent-diffusion sampling path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64)
restatement of the reference's algorithm for the hot path:
each function cites the reference file:line it follows.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it, and only as the checker / timed baseline -- never from `after_amd/`.

Parity status: PINNED.  The reference publishes no golden vectors (SURVEY.md
§4), so the oracle is pinned against outputs of the reference's own Python,
imported in the build container through `tests/golden/refimport.py`; the
resulting input/output vectors are committed under `tests/golden/*.npz` with
the generating script `tests/golden/make_golden.py`, and
`tests/test_oracle_golden.py` checks the oracle against them on every run.
One caveat is inherited from SURVEY.md §8(c): the third-party `cached_conv`
package is absent from the image, so its offline padding arithmetic
(`get_padding`) is restated from its published behaviour in
`tests/golden/_refshim/cached_conv` and is pinned only by the identities the
reference itself asserts (decode(encode(x)).shape == x.shape, ratio 2048).

The streaming codec (cached_conv state across chunks) is PARITY UNPINNED for
the same reason -- see oracle/streaming.py (causal codec) and oracle/cached.py
(the non-causal codec's cached encoder twin; its CachedGroupNorm part IS pinned,
tests/golden/cached_gn.npz) for what they are anchored on instead.  The
checkpoint / gin ingestion is pinned by a run folder the reference itself wrote
(tests/golden/ckpt_nano).

All functions are *functional*: they take a state dict whose keys are the
reference's own `state_dict()` keys, so weights move
1:1 between the reference, the oracle and the HIP path.
"""
from .denoiser import (denoiser_forward, band_bounds, rope_tables,  # noqa: F401
                       positional_embedding, DenoiserCache, banded_attention,
                       banded_attention_loop)
from .sampler import model_forward, sample  # noqa: F401
from .autoencoder import (ae_encode, ae_decode, pqmf_forward, pqmf_inverse,  # noqa: F401
                          fold_weight_norm, tanh_bottleneck, vae_bottleneck, ae_encode_raw)
from .encoders import encoder1d_forward, ecapa_forward  # noqa: F401
from .streaming import stream_forward  # noqa: F401
from .cached import NonCausalStreamEncoder, StreamGroupNorm, StreamNormDecoder  # noqa: F401
from .unet1d import unet1d_forward  # noqa: F401
