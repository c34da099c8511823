"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel (template arguments kept, argument list dropped) the number of
dispatches, the mean of every counter, and the mean duration.  usage: pmc_agg.py <out.json> <dir> [<dir> ...]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("showo::", "").replace("g2p::", "").replace("g3w::", ""))
                name = re.sub(r"\((?!.*<).*$", "", name)
                a = acc[name][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                t = acc[name]["_duration_ns"]
                t[0] += 1
                t[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    res = {}
    for k, cs in acc.items():
        n = max(v[0] for v in cs.values())
        res[k] = {"dispatches": n}
        for c, (cnt, s) in cs.items():
            res[k][c] = s / cnt
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -kv[1]["dispatches"] * kv[1].get("_duration_ns", 0))[:6]
    for k, v in top:
        print(k[:70], {c: (round(x, 1) if isinstance(x, float) else x) for c, x in v.items()})


if __name__ == "__main__":
    main()
