class _Unavailable:

    def __init__(self, *a, **k):
        raise RuntimeError("torchaudio shim: transforms are not available")


MelSpectrogram = _Unavailable
Spectrogram = _Unavailable
