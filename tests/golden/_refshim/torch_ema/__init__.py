"""Import-time stand-in for torch_ema (training only).  `average_parameters()` is the null
context (shadow parameters == live parameters): make_golden.py's checkpoint case calls the
reference's own `save_model`, which wraps `state_dict()` in it (model.py:145-146)."""
import contextlib


class ExponentialMovingAverage:

    def __init__(self, *a, **k):
        pass

    @contextlib.contextmanager
    def average_parameters(self):
        yield
