"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one counter per pass as the TCC block requires) into
per-kernel HBM traffic per launch.  usage: pmc_summary.py <tag> <fetch_dir> <write_dir> <out.json>

Units / corrections (MI355X_MICROARCH.md, "HBM"): the counters are in KiB per dispatch; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide coalesced streaming reads (16 B per lane, global_load and buffer_load...lds alike -- the
access pattern of every kernel here), so fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is taken at face value
(uncalibrated in the guide).  Infinity-Cache hits are counted, i.e. this is fabric-side traffic, an upper bound of DRAM bytes."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("showo::", "").replace("g2p::", "").replace("g3w::", ""))
            name = re.sub(r"\(.*", "", name)
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    tag, fd, wd, out = sys.argv[1:5]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        kernels[k] = {"launches": max(nf, nw),
                      "fetch_bytes_per_launch": 2.0 * 1024.0 * sf / nf if nf else None,
                      "write_bytes_per_launch": 1024.0 * sw / nw if nw else None}

    def group(prefix):  # prefix: one name prefix or a tuple of them
        n = sum(v["launches"] for k, v in kernels.items() if k.startswith(prefix))
        if not n:
            return None
        f = sum(v["launches"] * (v["fetch_bytes_per_launch"] or 0) for k, v in kernels.items() if k.startswith(prefix)) / n
        w = sum(v["launches"] * (v["write_bytes_per_launch"] or 0) for k, v in kernels.items() if k.startswith(prefix)) / n
        return {"launches": n, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "bytes_per_launch": f + w}
    import hashlib
    h = hashlib.sha1()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("gemm_common.h", "gemm2p.hip", "gemm3w.hip"):  # same definition as bench.py::gemm_src_sha1
        with open(os.path.join(root, "show-o_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    res = {"tag": tag, "kernel_src_sha1": h.hexdigest(), "command": "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events (one --pmc pass per counter)",
           "corrections": "fetch = 2 x FETCH_SIZE KiB (gfx950 wide-read halving), write = WRITE_SIZE KiB (uncalibrated)",
           "gemm2p_kernel": group(("gemm2p_kernel", "gemm3w_kernel")),  # the bf16 GEMM family (gemm2p.hip + its weight-ring form gemm3w.hip)
            "attn_fwd_lds_kernel": group("attn_fwd_lds_kernel"),
           "conv2p_split_kernel": group("conv2p_split_kernel"), "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("gemm2p_kernel", "attn_fwd_lds_kernel", "conv2p_split_kernel")}))


if __name__ == "__main__":
    main()
