"""Handle re-creation / capacity growth stress: many shapes in sequence through every handle type;
checks finiteness and agreement of a repeated first call (catches arena overflows / stale state)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import pipeline, Streamer
torch.set_grad_enabled(False)
dev = "cuda:0"
for dname, aname in (("tiny", "baseAE"), ("midi", "baseAE"), ("base", "baseAE_causal")):
    model, dcfg, acfg = pipeline.build_models(dname, aname, dev, seed=3)
    ae = model.emb_model
    g = torch.Generator().manual_seed(0)
    first = None
    for B, T in ((1, 16), (2, 64), (1, 256), (5, 32), (3, 384), (1, 16), (8, 256), (2, 8)):
        x = torch.randn(B, 64, T, generator=g).to(dev)
        c = torch.randn(B, dcfg["net"]["cond_dim"], generator=g).to(dev)
        tc = torch.randn(B, dcfg["net"]["tcond_dim"], T, generator=g).to(dev)
        y = model.sample(x, c, tc, 3, 2.0, 1.0)
        assert torch.isfinite(y).all(), (dname, B, T)
        a = ae.decode(y)
        assert a.shape == (B, 1, T * 2048) and torch.isfinite(a).all()
        z = ae.encode(a)[0]
        assert z.shape == y.shape and torch.isfinite(z).all()
        if model.encoder_time is not None:
            assert torch.isfinite(model.encoder_time(z)).all()
        if T >= 8:
            assert torch.isfinite(model.encoder(z[..., :max(8, T // 2)].contiguous())).all()
        if (B, T) == (1, 16):
            if first is None:
                first = (x, c, tc, y)
            else:
                y2 = model.sample(first[0], first[1], first[2], 3, 2.0, 1.0)
                assert torch.equal(y2, first[3]), "result changed after capacity growth"
    print(dname, aname, "ok")
    if aname.endswith("causal"):
        st = Streamer(model, ae, chunk_size=4, max_batch=3, max_nb_steps=4, share_first_stream=False)
        st.set_nb_steps(4)
        for n in (1, 3, 2, 3):
            out = st(torch.randn(n, 2, 4 * st.ae_ratio, device=dev))
            assert out.shape == (n, 1, 4 * st.ae_ratio) and torch.isfinite(out).all()
        print("  streamer ok")
print("stress ok")
