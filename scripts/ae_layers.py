import os, sys, torch
sys.path.insert(0, '/root/repo')
from after_amd import pipeline
torch.set_grad_enabled(False)
model, dcfg, acfg = pipeline.build_models("base", "baseAE", "cuda:0")
ae = model.emb_model
B = int(sys.argv[1])
z = torch.randn(B, 64, 256, device="cuda:0")
ae.decode(z); torch.cuda.synchronize()
os.environ["X"]="1"
sys.stderr.write("==== decode B=%d\n" % B)
