"""HBM traffic per kernel of an arbitrary command: two rocprofv3 counter passes (FETCH_SIZE needs 3 TCC
slots, WRITE_SIZE 2 -- separate runs; kernel trace only, no other trace domain) reduced to
bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-B requests
at 64 B, MI355X_MICROARCH.md; Infinity-Cache hits are included).

    python scripts/pmc_run.py out.json -- python scripts/time_codec.py --batches 1 --rounds 3"""
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


def main():
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
