"""MFMA busy fraction per kernel from a rocprofv3 --pmc pass aggregated by tools/pmc_agg.py.
usage: pmc_mfma.py <tag> <pmc_agg.json> <out.json>

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock_GHz = GRBM_GUI_ACTIVE / 8 / duration
(profiled passes clock lower than un-profiled ones).  The file is stamped with the sha1 of the production GEMM sources, the same
definition bench.py uses (gemm_src_sha1): a bench line only quotes it when the kernels are the ones it was measured on."""
import hashlib
import json
import os
import sys


def main():
    tag, src, out = sys.argv[1:4]
    agg = json.load(open(src))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for f in ("gemm_common.h", "gemm2p.hip", "gemm3w.hip"):
        with open(os.path.join(root, "show-o_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    kernels = {}
    for k, v in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v or not v.get("_duration_ns"):
            continue
        gui = v["GRBM_GUI_ACTIVE"]
        if gui <= 0:
            continue
        kernels[k] = {"dispatches": v["dispatches"], "avg_us": v["_duration_ns"] / 1e3, "clock_GHz": gui / 8.0 / v["_duration_ns"],
                      "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8.0)}
    res = {"tag": tag, "kernel_src_sha1": h.hexdigest(),
           "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -- python bench.py --steps 1 --warmup 1 (no events, no extra legs)",
           "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock_GHz = GRBM_GUI_ACTIVE / 8 / duration "
                      "(profiled passes clock lower than un-profiled ones)", "kernels": kernels}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["dispatches"] * kv[1]["avg_us"])[:8]
    for k, v in top:
        print(f"{k[:80]:80s} n={v['dispatches']:5d} avg {v['avg_us']:8.1f} us  mfma busy {v['mfma_busy_frac']:.3f}  clock {v['clock_GHz']:.2f} GHz")


if __name__ == "__main__":
    main()
