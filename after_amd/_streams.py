"""A side stream per device for the host-level overlap of independent launch chains (the two conditioning encoders in
pipeline.generate_from_latents): one fork / join per clip around chains of ~20 small launches.
AFTER_ENCODERS_CONCURRENT=0 runs them one after the other (A/B switch).  (Measured for Streamer.forward as well: 18.91 -> 18.86 ms
per chunk -- the timbre side of a chunk is too small to matter; not kept.)"""
import os

import torch

CONCURRENT = os.environ.get("AFTER_ENCODERS_CONCURRENT", "1") != "0"
_SIDE = {}


def side_stream(device):
    key = torch.device(device).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]
