import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch, showo_amd, weights as Wt
d = Wt.ShowoDims()
with torch.device("meta"):
    m = showo_amd.Showo(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=256, max_batch=1, max_seq=768)
m = m.to_empty(device="cuda").eval()
with torch.no_grad():
    for n, p in m.named_parameters():
        if "layernorm" in n and n.endswith("weight"): p.fill_(1.0)
        elif n.endswith("bias"): p.zero_()
        else: p.normal_(0.0, 0.02)
ids = torch.randint(0, 50000, (1, 631), device="cuda")
m.decode_graph = 0
for _ in range(2):
    toks = m.mmu_generate(ids, max_new_tokens=33, top_k=1)
torch.cuda.synchronize()
