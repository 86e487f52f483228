"""Import-time stand-in for torchaudio (only referenced by training losses)."""
from . import transforms  # noqa: F401
