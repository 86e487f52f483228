"""Reduce a rocprofv3 --kernel-trace CSV to one pass of a pipeline: the launches between the last two
occurrences of a terminating kernel (e.g. pqmf_inverse for the decoder), each with its duration and
the idle gap before it; totals per kernel family.

    python scripts/trace_reduce.py <dir with *_kernel_trace.csv> --end pqmf_inverse [--rows]"""
import argparse
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(?:after::)?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--end", default="pqmf_inverse")
    ap.add_argument("--rows", action="store_true")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel trace under " + a.dir)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if a.end in r[2]]
    if len(ends) < 2:
        sys.exit("terminating kernel seen %d times" % len(ends))
    seg = rows[ends[-2] + 1:ends[-1] + 1]
    fam, busy, gaps = {}, 0, 0
    prev_end = rows[ends[-2]][1]
    out_rows = []
    for s, e, n in seg:
        k = short(n)
        d = fam.setdefault(k, {"n": 0, "us": 0.0})
        d["n"] += 1
        d["us"] += (e - s) / 1e3
        busy += e - s
        gap = max(0, s - prev_end)
        gaps += gap
        out_rows.append({"k": k, "us": round((e - s) / 1e3, 2), "gap_us": round(gap / 1e3, 2)})
        prev_end = max(prev_end, e)
    span = (seg[-1][1] - rows[ends[-2]][1]) / 1e3
    res = {"launches": len(seg), "span_us": round(span, 1), "busy_us": round(busy / 1e3, 1),
           "gap_us": round(gaps / 1e3, 1),
           "families": {k: {"n": v["n"], "us": round(v["us"], 1)} for k, v in
                        sorted(fam.items(), key=lambda kv: -kv[1]["us"])}}
    print(json.dumps(res))
    if a.rows:
        for r in out_rows:
            print(json.dumps(r))


if __name__ == "__main__":
    main()
