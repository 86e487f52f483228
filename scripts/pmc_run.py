"""HBM traffic per kernel of an arbitrary command: two rocprofv3 counter passes (FETCH_SIZE needs 3 TCC
slots, WRITE_SIZE 2 -- separate runs; kernel trace only, no other trace domain) reduced to
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-B requests
at 64 B, MI355X_MICROARCH.md; Infinity-Cache hits are included).

    python scripts/pmc_run.py out.json -- python scripts/time_codec.py --batches 1 --rounds 3

`--mfma` instead: ONE pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE -> per kernel the fraction of its
SIMD-cycles in which the matrix pipe was busy (bench.py --pmc-mfma does the same for the sampler):

    python scripts/pmc_run.py --mfma out.json -- python scripts/time_codec.py --batches 8 --rounds 3"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("after::", "")
    m = re.match(r"\s*([A-Za-z_][\w]*(<[^()]*>)?)", name)
    return m.group(1) if m else name[:80]


def mfma_main(out_json, cmd, root):
    scratch = os.path.join(root, "gpurun_out", "pmc_" + os.path.splitext(os.path.basename(out_json))[0], "mfma")
    ctrs = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", scratch, "--"] + cmd,
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 --pmc {ctrs} failed:\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(scratch, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[(short(row["Kernel_Name"]), int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    n_simd, n_xcc = 256 * 4, 8  # GRBM_GUI_ACTIVE arrives summed over its 8 XCC instances
    rows = []
    for (name, grid), v in agg.items():
        mb, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]), v.get("GRBM_GUI_ACTIVE", [0.0])
        mbm, guim = sum(mb) / len(mb), sum(gui) / max(1, len(gui))
        rows.append({"kernel": name, "grid_threads": grid, "launches": len(mb), "SQ_VALU_MFMA_BUSY_CYCLES_mean": round(mbm),
                     "GRBM_GUI_ACTIVE_mean": round(guim), "mfma_busy_frac": round(mbm / (guim / n_xcc * n_simd), 4) if guim else None,
                     "_tot": mbm * len(mb)})
    rows.sort(key=lambda r_: -r_["_tot"])
    for r_ in rows:
        del r_["_tot"]
    json.dump({"what": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCC instances x 1024 "
                       "SIMDs) per launch, mean over a kernel's launches: the fraction of SIMD-cycles with the matrix pipe busy "
                       "(bf16 kernels: one 16x16x32 MFMA = 16 busy cycles, six per fp32 product block; fp32 kernels: one 16x16x4 = 32)",
               "command": " ".join(sys.argv[sys.argv.index("--") + 1:]), "kernels": rows[:40]}, open(out_json, "w"), indent=1)
    for r_ in rows[:16]:
        print(r_)


def main():
    if sys.argv[1] == "--mfma":
        root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        cmd = sys.argv[sys.argv.index("--") + 1:]
        cmd = [c if not (c.endswith(".py") and not os.path.isabs(c)) else os.path.join(root, c) for c in cmd]
        return mfma_main(sys.argv[2], cmd, root)
    out_json = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cmd = [c if not (c.endswith(".py") and not os.path.isabs(c)) else os.path.join(root, c) for c in cmd]
    scratch = os.path.join(root, "gpurun_out", "pmc_" + os.path.splitext(os.path.basename(out_json))[0])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(scratch, ctr)
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + cmd,
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"rocprofv3 --pmc {ctr} failed:\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == ctr:
                    agg[(short(row["Kernel_Name"]), int(row["Grid_Size"]))][ctr].append(float(row["Counter_Value"]))
    rows = []
    for (name, grid), v in agg.items():
        f, w = v.get("FETCH_SIZE", [0.0]), v.get("WRITE_SIZE", [0.0])
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        rows.append({"kernel": name, "grid_threads": grid, "launches": len(f), "FETCH_SIZE_KiB_mean": round(fm, 1),
                     "WRITE_SIZE_KiB_mean": round(wm, 1), "fetch_bytes_corrected": round(2 * fm * 1024),
                     "bytes_per_launch_corrected": round((2 * fm + wm) * 1024)})
    rows.sort(key=lambda r: -r["bytes_per_launch_corrected"] * r["launches"])
    json.dump({"note": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch; Infinity-Cache hits included",
               "command": " ".join(sys.argv[sys.argv.index("--") + 1:]), "kernels": rows[:60]},
              open(out_json, "w"), indent=1)
    for r in rows[:25]:
        print(r)


if __name__ == "__main__":
    main()
