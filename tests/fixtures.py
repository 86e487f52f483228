"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

import detweights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:

    def __init__(self, case):
        self.case = case
        z = np.load(os.path.join(GOLDEN, case + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.arrays = {k: z[k] for k in z.files if k != "meta" and not k.startswith("keep:")}
        self.keep = {k[5:]: z[k] for k in z.files if k.startswith("keep:")}

    def t(self, name, dtype=torch.float32):
        return torch.from_numpy(np.asarray(self.arrays[name])).to(dtype)

    def state_dict(self, shapes_key="shapes", seed_offset=0, dtype=torch.float32):
        shapes = self.meta[shapes_key]
        sd = detweights.fill(shapes, self.meta["seed"] + seed_offset, keep=self.keep)
        return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
