"""Per-workgroup s_memtime timeline of the DMA GEMM (diagnostics)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from after_amd import diag, _lib
dev = torch.device("cuda:0")
EPI = int(os.environ.get("GEMM_EPI", "0"))
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (768, 1536, 512)
tile = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1, 2)
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
if tile[0] >= 100:  # balanced split-K kernels: (100 + MB, 10 * NS + NB)
    bm, bn = 16 * (tile[0] % 100), 32 * (tile[1] % 10)
else:
    bm, bn = 32 * tile[0], 32 * tile[1]
nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
dbg = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
for _ in range(5): diag.gemm(a, w, tile=tile, out=out)
torch.cuda.synchronize()
_lib.lib().after_gemm_set_debug(dbg.data_ptr())
diag.gemm(a, w, tile=tile, out=out, epilogue=EPI)
torch.cuda.synchronize()
_lib.lib().after_gemm_set_debug(None)
d = dbg.cpu().numpy().reshape(nwg, 8).astype(np.float64)
life = d[:, 3] - d[:, 0]
rs = (d[:, 4] - d[:, 4].min()) * 10.0   # ns
re = (d[:, 5] - d[:, 4].min()) * 10.0
smid = d[:, 6].astype(np.int64)
print(f"shape {M}x{N}x{K} tile {tile}: {nwg} WGs")
print(f"  per-WG cycles: prologue med {np.median(d[:,1]-d[:,0]):.0f} loop med {np.median(d[:,2]-d[:,1]):.0f} (min {np.min(d[:,2]-d[:,1]):.0f} max {np.max(d[:,2]-d[:,1]):.0f}) epilogue med {np.median(d[:,3]-d[:,2]):.0f}; lifetime med {np.median(life):.0f} max {life.max():.0f}")
print(f"  wall (100 MHz clock): first start 0 ns, last start {rs.max():.0f} ns, first end {re.min():.0f} ns, last end {re.max():.0f} ns")
print("  start-time histogram (ns):", [(int(e), int(h)) for h, e in zip(*np.histogram(rs, bins=8))])
pk = dbg.cpu().numpy().reshape(nwg, 8)[:, 7]
pf, pv, pb = pk & 0xFFFFF, (pk >> 20) & 0xFFFFF, (pk >> 40) & 0xFFFFF
print(f"  loop phases (wave 0, cycles summed over slabs): frag-fence med {np.median(pf):.0f}  vmcnt-wait med {np.median(pv):.0f}  barrier med {np.median(pb):.0f}  (loop med {np.median(d[:,2]-d[:,1]):.0f})")
u, c = np.unique(smid, return_counts=True)
print(f"  distinct CU ids {len(u)}; WGs per CU histogram:", dict(zip(*np.unique(c, return_counts=True))))
# per-CU residency slots: start time of the k-th workgroup that landed on each CU
order = np.argsort(rs)
slots = {}
for i in order:
    slots.setdefault(int(smid[i]), []).append(rs[i])
mx = max(len(v) for v in slots.values())
for k in range(min(mx, 6)):
    v = np.array([s[k] for s in slots.values() if len(s) > k])
    print(f"  slot {k}: start ns min {v.min():.0f} med {np.median(v):.0f} max {v.max():.0f}  (n={len(v)})")
xcd = smid  # start order within an XCD-sized group
print("  first 24 start times (ns):", [int(x) for x in np.sort(rs)[:24]])
print("  every 32nd start time (ns):", [int(x) for x in np.sort(rs)[::32]])
