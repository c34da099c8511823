"""Full-size runs of the BASELINE.json configs that are not the bench line (cfg3: 512x512 batch 4 + inpainting, cfg4: mmu
w_clip_vit AR decode): functional checks at size + timings.  Synthetic data, random-init weights of the true architecture."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
import showo_amd, showo_oracle as O, weights as Wt
P = showo_amd.prompting_utils

d = Wt.ShowoDims()
torch.manual_seed(0)


def build_model(w_clip_vit, max_batch, max_seq):
    with torch.device("meta"):
        m = showo_amd.Showo(w_clip_vit, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=1024, max_batch=max_batch,
                            max_seq=max_seq)
    m = m.to_empty(device="cuda").eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layernorm" in n and n.endswith("weight"): p.fill_(1.0)
            elif n.endswith("bias"): p.zero_()
            else: p.normal_(0.0, 0.02)
    return m


def timeit(f, n=2):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r


# ---------------------------------------------------------------- cfg3: t2i 512x512, batch 4, inpainting, CFG
B, N = 4, 1024
L = 129 + 1 + N + 1
vq = showo_amd.MAGVITv2(max_batch=4, max_res=512).cuda().eval()
x = torch.rand(1, 3, 512, 512, device="cuda") * 2 - 1
t_enc, codes = timeit(lambda: vq.get_code(x.expand(B, -1, -1, -1).contiguous()))
assert tuple(codes.shape) == (B, N) and int(codes.min()) >= 0 and int(codes.max()) < 8192
grid = torch.zeros(32, 32, dtype=torch.bool); grid[8:24, 8:24] = True  # centred 16x16 block is to be generated
img = torch.where(grid.reshape(-1).cuda()[None], torch.full((B, N), d.mask_token_id, device="cuda"), codes + d.image_offset)
rs = np.random.RandomState(0)
rows_c, rows_u = [], []
for i in range(B):
    k = 6 + 7 * i
    text = [d.t2i_id, 50256] + rs.randint(0, 50256, size=k - 3).tolist() + [50256]
    rows_c.append([d.pad_id] * (129 - k) + text + [d.soi_id] + img[i].tolist() + [d.eoi_id])
    rows_u.append([d.pad_id] * 126 + [d.t2i_id, 50256, 50256] + [d.soi_id] + img[i].tolist() + [d.eoi_id])
ic, iu = torch.tensor(rows_c).cuda(), torch.tensor(rows_u).cuda()
model = build_model(False, 2 * B, L)
cfg = showo_amd.gen_config(num_vq_tokens=N)
mask = P.intervals_predict_next(torch.cat([ic, iu]), pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id, rm_pad_in_image=True)
gen = torch.Generator(device="cuda").manual_seed(1)


def t2i():
    ids = ic.clone()
    return model.t2i_generate(input_ids=ids, uncond_input_ids=iu, attention_mask=mask, timesteps=18, guidance_scale=5.0,
                              generator=gen, config=cfg), ids

t_gen, (toks, ids_after) = timeit(t2i, 1)
keep = ~grid.reshape(-1).cuda()
assert torch.equal(toks[:, keep], codes[:, keep])  # known (unmasked) image tokens are returned untouched
assert int(toks.min()) >= 0 and int(toks.max()) < 8192
t_dec, out = timeit(lambda: vq.decode_code(toks))
assert tuple(out.shape) == (B, 3, 512, 512) and torch.isfinite(out).all()
tot = t_enc + t_gen + t_dec
print(f"cfg3 t2i 512x512 batch 4 inpaint CFG (L={L}): get_code {t_enc*1e3:.1f} ms, 18-step generate {t_gen*1e3:.1f} ms, decode {t_dec*1e3:.1f} ms"
      f" -> {B / tot:.2f} images/s ({122.5 * B / tot:.0f} TF/s algorithmic)")
del model, vq
torch.cuda.empty_cache()

# ---------------------------------------------------------------- cfg4: mmu w_clip_vit, 631 prompt embeds, 100 new tokens, top_k=1
# the whole flow of inference_mmu.py:96-146 after image decoding: CLIP ViT-L/14-336 tower (random-init weights of the true
# architecture) -> mm_projector -> embed the text ids -> splice -> mmu_vit mask -> prefill + KV-cached decode
model = build_model(True, 1, 768)
tower = showo_amd.CLIPVisionTower("synthetic", config=Wt.CLIP_L336, state_dict=O.to_torch(Wt.make_clip_state(Wt.CLIP_L336, seed=22)),
                                  max_batch=1).cuda()
emb_tab = model.showo.model.embed_tokens.weight
Lp = 1 + 28 + 1 + 576 + 1 + 24
times, t_clip, t_first = [], [], []
for img_i in range(4):  # 4 images = 4 independent batch-1 decodes (reference semantics, modeling_showo.py:204,229)
    g = torch.Generator(device="cuda").manual_seed(3 + img_i)
    pixels = torch.randn(1, 3, 336, 336, device="cuda", generator=g)  # CLIPImageProcessor output (host side, not timed)
    ids = torch.randint(0, 50256, (1, Lp - 576), device="cuda", generator=g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        img_emb = model.mm_projector(tower(pixels))
        txt = emb_tab[ids]
        emb = torch.cat([txt[:, :30], img_emb, txt[:, 30:]], dim=1)
    am = P.create_attention_mask_for_mmu_vit(emb, system_prompt_len=28)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    first = model.mmu_generate(input_embeddings=emb, attention_mask=am[0], max_new_tokens=1, top_k=1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    toks = model.mmu_generate(input_embeddings=emb, attention_mask=am[0], max_new_tokens=100, top_k=1)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    t_clip.append(t1 - t0); t_first.append(t2 - t1); times.append(t3 - t2)
    assert len(toks) == 100 and all(0 <= int(t) < d.vocab for t in toks) and int(toks[0]) == int(first[0])
t, tc, tf = float(np.mean(times[1:])), float(np.mean(t_clip[1:])), float(np.mean(t_first[1:]))
print(f"cfg4 mmu w_clip_vit: CLIP ViT-L/14-336 + mm_projector + splice {tc*1e3:.1f} ms; prefill of {Lp} embeds -> first token {tf*1e3:.1f} ms "
      f"(time to first token {1e3*(tc+tf):.1f} ms); {Lp} embeds + 100 new tokens (KV cache): {t*1e3:.1f} ms per image -> {100 / t:.0f} tokens/s")
