"""Import-time stand-in for einops_exts (only `rearrange_many` is referenced, by
networks that are off the hot path)."""
from einops import rearrange


def rearrange_many(tensors, pattern, **kw):
    return [rearrange(t, pattern, **kw) for t in tensors]
