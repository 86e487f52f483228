from .model import AutoEncoder, ReluBottleneck, TanhBottleneck, VAEBottleneck  # noqa: F401
from .export import ExportedAutoEncoder, embed_dataset  # noqa: F401
