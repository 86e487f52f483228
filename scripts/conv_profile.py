"""Per-layer conv timing of one AE decode: run under rocprofv3 --kernel-trace with AFTER_CONV_LOG=1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import AutoEncoder, configs
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = configs.autoencoder_config("baseAE"); cfg.pop("bottleneck")
ae = AutoEncoder(**cfg).to(dev)
z = torch.randn(1, 64, 256, device=dev)
ae.decode(z); torch.cuda.synchronize()
print("MARK decode start", file=sys.stderr)
ae.decode(z); torch.cuda.synchronize()
