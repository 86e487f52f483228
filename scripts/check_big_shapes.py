import sys; sys.path.insert(0,'/root/repo')
import torch, time
from after_amd import pipeline
torch.set_grad_enabled(False)
dev='cuda:0'
model, dcfg, acfg = pipeline.build_models("base","baseAE",dev,seed=1)
g=torch.Generator().manual_seed(2)
for B,T in ((16,256),(3,512),(1,1024),(5,128)):
    x=torch.randn(B,64,T,generator=g).to(dev); c=torch.randn(B,6,generator=g).to(dev); tc=torch.randn(B,12,T,generator=g).to(dev)
    t0=time.time(); a=model.sample(x,c,tc,3,2.0,1.0); torch.cuda.synchronize(); dt=time.time()-t0
    b=model.sample(x[:1].contiguous(),c[:1].contiguous(),tc[:1].contiguous(),3,2.0,1.0)
    print(B,T,"max diff clip0 vs alone",(a[:1]-b).abs().max().item(),"finite",bool(torch.isfinite(a).all()), round(dt*1e3,1),"ms")
    y=model.emb_model.decode(a[:2].contiguous()); print("  decode",tuple(y.shape), bool(torch.isfinite(y).all()))
