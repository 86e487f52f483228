"""Cost of the conv epilogue features on the codec's k = 1 / k = 3 layers: bare conv, + statistics,
+ residual, both; heuristic tile and the 64 x 96 / 64 x 64 tiles.

    python scripts/conv_epilogue_modes.py [batch]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from after_amd import diag
from scripts.bench_conv import timeit, PEAK
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, cin, cout, T, k, dil in (("dec2 k1 384@8192", 384, 384, 8192, 1, 1), ("dec0 k1 768@1024", 768, 768, 1024, 1, 1),
                                   ("dec3 k1 192@16384", 192, 192, 16384, 1, 1), ("dec2 k3 384@8192", 384, 384, 8192, 3, 1)):
    w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
    b = torch.randn(cout, device=dev)
    lp = (k - 1) * dil // 2
    c = diag.ConvTm(w, b, B, T, dil, 1, lp, (k - 1) * dil - lp, act=1)
    c.run(None, None, 1)
    line = {"layer": name, "B": B}
    for mode, tag in ((2, "bare"), (2 | 4, "stats"), (2 | 8, "res"), (2 | 4 | 8, "full")):
        for t in (0, 1, 2):
            diag.set_conv_tile(t)
            c.run(None, None, mode)
            torch.cuda.synchronize()
            dt = timeit(lambda: c.run(None, None, mode), 30)
            line[f"{tag}_t{t}"] = round(dt * 1e6, 1)
    diag.set_conv_tile(0)
    print(json.dumps(line), flush=True)
    c.close()
