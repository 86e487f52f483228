from .model import AutoEncoder, ReluBottleneck  # noqa: F401
from .export import ExportedAutoEncoder, embed_dataset  # noqa: F401
