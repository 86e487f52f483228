"""Import the reference's Python (read-only /root/reference) in the BUILD
CONTAINER ONLY, through the stand-ins under tests/golden/_refshim for the
third-party modules the image lacks.  Nothing here is used at
test time on the GPU box: it exists so that make_golden.py can (re)generate the
committed fixtures and so the oracle can be pinned against the real reference.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "after"))


def setup():
    if not available():
        raise RuntimeError("reference tree not present")
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_refshim")
    for p in (REFERENCE_ROOT, shim):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, shim)
    # scipy 1.15 vs the reference's scipy==1.12 pin (pqmf.py:10,72)
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    _firwin = scipy.signal.firwin
    if not getattr(_firwin, "_nyq_shim", False):

        def firwin(*a, nyq=None, **k):
            if nyq is not None:
                k["fs"] = 2 * nyq
            return _firwin(*a, **k)

        firwin._nyq_shim = True
        scipy.signal.firwin = firwin
    # torch.utils.tensorboard needs the tensorboard package at import time
    if "torch.utils.tensorboard" not in sys.modules:
        m = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:  # pragma: no cover

            def __init__(self, *a, **k):
                pass

        m.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = m
    # the autoencoder package __init__ pulls the trainer (accelerate etc.):
    # pre-register a bare package so only the network files are executed.
    import importlib.machinery
    for name, rel in (("after", "after"), ("after.autoencoder", "after/autoencoder"),
                      ("after.autoencoder.networks", "after/autoencoder/networks")):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
            pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
            sys.modules[name] = pkg
    if "after.diffusion" not in sys.modules:
        for name, rel in (("after.diffusion", "after/diffusion"),
                          ("after.diffusion.networks", "after/diffusion/networks")):
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, rel)]
            pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
            sys.modules[name] = pkg


def modules():
    """Returns the reference modules on the hot path."""
    setup()
    import importlib
    import cached_conv as cc
    out = types.SimpleNamespace()
    out.cc = cc
    out.model = importlib.import_module("after.diffusion.model")
    out.transformerv2 = importlib.import_module("after.diffusion.networks.transformerv2")
    out.rotary = importlib.import_module("after.diffusion.networks.rotary_embedding")
    out.encoder = importlib.import_module("after.diffusion.networks.encoder")
    out.ecapa = importlib.import_module("after.diffusion.networks.ecapa_encoder")
    out.ae = importlib.import_module("after.autoencoder.networks.SimpleNetsStream")
    out.pqmf = importlib.import_module("after.autoencoder.networks.pqmf")
    out.unet1d = importlib.import_module("after.diffusion.networks.unet1d")
    return out
