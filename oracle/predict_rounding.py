"""TEST INFRASTRUCTURE ONLY -- CPU go / no-go for a 16-bit operand type (VERDICT r5 "Next round" #1a).

Runs the rounding-point oracle (showo_oracle.Bf16Points: the fp32 restatement with the HIP path's rounding points) at MODEL SCALE
(1.45 B parameters, oracle/weights.py) with bf16 or fp16 operand rounding and prints the predicted error of the logits against the
committed fp32 REFERENCE fixtures (tests/golden/showo_full_*.npz, written by the real reference through oracle/make_golden.py):

    python oracle/predict_rounding.py                 # fp16 + bf16 on the [2,387] fixture, then one site at a time (fp16)
    python oracle/predict_rounding.py --cfg3          # rows 1 / 5 of the [8,1155] batch as well
    python oracle/predict_rounding.py --exempt w_lm,hf   # fp16 everywhere except the named sites (those stay fp32 = a (hi, lo) pair)

"site" = one of Bf16Points.SITES: w (GEMM weights of the 24 blocks), w_lm, h (LayerNorm output), q, k, v, p (soft-max numerator),
o (attention output), gelu, hf (final hidden state).  `--only SITE` rounds ONLY that site (its share of the error); `--exempt`
rounds everything BUT the listed sites (what a (hi, lo) treatment of those operands would buy).
Results of the run that decided precision 2 are in profiles/r6_fp16_predict.txt.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import showo_oracle as O  # noqa: E402
import weights as Wt  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def relerr(a, b):
    a, b = a.double(), b.double()
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def case_387():
    g = np.load(os.path.join(GOLD, "showo_full_logits_subset.npz"))
    d = Wt.ShowoDims()
    sd = O.to_torch(Wt.make_showo_state(d, seed=int(g["seed"])))
    ids = torch.from_numpy(g["ids"])
    mask = O.mask_t2i(ids, d.pad_id, d.soi_id, d.eoi_id)
    rows, cols = torch.from_numpy(g["rows"]), torch.from_numpy(g["cols"])
    ref = torch.from_numpy(g["logits"])

    def run(pts):
        lg = O.showo_logits(sd, d, ids, attention_mask=mask, pts=pts)
        return relerr(lg[:, rows][:, :, cols], ref)
    return "[2,387]", run


def case_cfg3():
    g = np.load(os.path.join(GOLD, "showo_full_cfg3.npz"))
    d = Wt.ShowoDims(num_vq_tokens=1024)
    sd = O.to_torch(Wt.make_showo_state(d, seed=int(g["seed"])))
    pick = torch.tensor([1, 5])  # one conditional, one unconditional row of the [8,1155] batch (sequences are independent)
    ids = torch.from_numpy(g["ids"].astype(np.int64))[pick]
    mask = O.mask_t2i(ids, d.pad_id, d.soi_id, d.eoi_id)
    rows, cols = torch.from_numpy(g["rows"]), torch.from_numpy(g["cols"])
    ref = torch.from_numpy(g["logits"])[pick]

    def run(pts):
        lg = O.showo_logits(sd, d, ids, attention_mask=mask, pts=pts)
        return relerr(lg[:, rows][:, :, cols], ref)
    return "cfg3 rows 1,5 of [8,1155]", run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg3", action="store_true")
    ap.add_argument("--no-387", action="store_true", help="skip the [2,387] fixture")
    ap.add_argument("--no-sites", action="store_true", help="skip the one-site-at-a-time table")
    ap.add_argument("--only", default="", help="comma list: round ONLY these sites (fp16)")
    ap.add_argument("--exempt", default="", help="comma list: fp16 everywhere except these sites; ';' separates several runs")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    cases = ([] if a.no_387 else [case_387]) + ([case_cfg3] if a.cfg3 else [])
    for mk in cases:
        t0 = time.time()
        name, run = mk()
        print(f"== {name} (weights + inputs built in {time.time() - t0:.0f} s)", flush=True)

        def show(label, pts):
            t1 = time.time()
            rmax, rrms = run(pts)
            print(f"   {label:58s} rel_max={rmax:.3e} rel_rms={rrms:.3e}   ({time.time() - t1:.0f} s)", flush=True)
        show("fp32 restatement (no rounding points)", None)
        show("bf16 at every site (precision 0)", O.Bf16Points(attn_tiles=True))
        show("fp16 at every site (precision 2 candidate)", O.Bf16Points(attn_tiles=True, dtype=torch.float16))
        if a.only:
            show(f"fp16, ONLY {a.only}", O.Bf16Points(attn_tiles=True, dtype=torch.float16, sites=a.only.split(",")))
        for ex in filter(None, a.exempt.split(";")):
            sites = [s for s in O.Bf16Points.SITES if s not in ex.split(",")]
            show(f"fp16, all sites EXCEPT {ex}", O.Bf16Points(attn_tiles=True, dtype=torch.float16, sites=sites))
        if not a.no_sites and mk is case_387:
            for s in O.Bf16Points.SITES:
                show(f"fp16, only site '{s}'", O.Bf16Points(attn_tiles=True, dtype=torch.float16, sites=[s]))


if __name__ == "__main__":
    main()
