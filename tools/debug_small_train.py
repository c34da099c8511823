"""bisect helper: gradient errors of the 324-row SMALL training fixture under the current env switches"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import util
from util import Wt, dev
g = util.golden("showo_small_train.npz")
d = Wt.ShowoDims(**Wt.SMALL)
sd = Wt.make_showo_state(d, seed=13)
m = util.build_showo(d, sd, max_batch=12, max_seq=32).train()
ids, mask, labels = dev(g["ids"]), dev(g["mask"]), dev(g["labels"])
bt, bl, bm = (int(x) for x in g["b"])
logits, l1, l2, l3 = m(ids, attention_mask=mask, labels=labels, batch_size_t2i=bt, batch_size_lm=bl, batch_size_mmu=bm, max_seq_length=d.max_text_len)
(1.0 * l1 + 0.1 * l2 + 1.0 * l3).backward()
named = dict(m.named_parameters())
print("env", {k: v for k, v in os.environ.items() if k.startswith("SHOWO_")})
print("losses", float(l1), float(l2), float(l3), "ref", g["losses"])
for k in g.files:
    if k.startswith("grad::showo"):
        name = k[6:]
        a, b = named[name].grad, torch.from_numpy(g[k])
        rmax, rrms = util.relerr(a, b)
        flag = "  <<<<" if rrms > 3e-2 else ""
        print(f"{name:60s} rel_max={rmax:.3e} rel_rms={rrms:.3e} |got|={float(a.abs().max()):.3e} |ref|={float(b.abs().max()):.3e}{flag}")
