import sys; sys.path.insert(0,'/root/repo')
import torch
from after_amd import pipeline
torch.set_grad_enabled(False)
model, dcfg, acfg = pipeline.build_models("base","baseAE","cuda:0",seed=1)
z = torch.randn(1,64,128,device="cuda:0")
model.encoder(z); torch.cuda.synchronize()
