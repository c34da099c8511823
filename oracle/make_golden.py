"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz by running the REAL reference (imported from
/root/reference through oracle/ref_loader.py) on the deterministic weights of oracle/weights.py, and
checks the in-repo restatement (oracle/showo_oracle.py) against the reference while doing so.

Run in the build container only:  python oracle/make_golden.py [--full]
The fixtures it writes are committed; the GPU box never needs /root/reference.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader as R  # noqa: E402
import showo_oracle as O  # noqa: E402
import weights as Wt  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def ns(**k):
    return types.SimpleNamespace(**k)


def gen_config(d):
    return ns(model=ns(showo=ns(num_vq_tokens=d.num_vq_tokens, num_new_special_tokens=d.num_new_special_tokens,
                                llm_vocab_size=d.llm_vocab)),
              dataset=ns(preprocessing=ns(max_seq_length=d.max_text_len)))


def ref_showo_from_state(d, sd_np):
    heads = d.heads
    m = R.build_reference_showo(
        phi_overrides=dict(hidden_size=d.hidden, intermediate_size=d.ffn, num_hidden_layers=d.layers,
                           num_attention_heads=heads, num_key_value_heads=heads, vocab_size=d.vocab,
                           max_position_embeddings=d.max_pos),
        vocab_size=d.vocab, llm_vocab_size=d.llm_vocab, codebook_size=d.codebook,
        num_vq_tokens=d.num_vq_tokens, w_clip_vit=d.w_clip_vit)
    missing = m.load_state_dict(O.to_torch(sd_np), strict=True)
    return m.eval()


def t2i_ids(d, text_lens, rs, bos=None, image_tokens=None):
    """[pad]*(T-k) + [t2i, bos, w.., eos] + [soi] + image + [eoi]  (prompting_utils.py:92-123)."""
    T = d.max_text_len + 1
    bos = d.llm_vocab - 10 if bos is None else bos
    rows = []
    for i, k in enumerate(text_lens):
        words = rs.randint(0, d.llm_vocab - 20, size=k - 3).tolist()
        text = [d.t2i_id, bos] + words + [bos]
        img = [d.mask_token_id] * d.num_vq_tokens if image_tokens is None else image_tokens[i]
        rows.append([d.pad_id] * (T - k) + text + [d.soi_id] + list(img) + [d.eoi_id])
    return torch.tensor(rows, dtype=torch.long)


def mmu_ids(d, n, text_len, rs):
    """[mmu][soi] img [eoi][sot] text... (prompting_utils.py:162-212 layout, no padding needed here)."""
    bos = d.llm_vocab - 10
    rows = []
    for _ in range(n):
        img = (rs.randint(0, d.codebook, size=d.num_vq_tokens) + d.image_offset).tolist()
        txt = rs.randint(0, d.llm_vocab - 20, size=text_len).tolist()
        rows.append([d.mmu_id, d.soi_id] + img + [d.eoi_id, bos] + txt)
    return torch.tensor(rows, dtype=torch.long)


def report(name, a, b):
    err = (a.double() - b.double()).abs().max().item()
    scale = b.double().abs().max().item()
    print(f"  oracle-vs-reference {name}: max|d|={err:.3e} (scale {scale:.3e})")
    return err


def make_tiny_showo():
    print("[tiny showo]")
    d = Wt.ShowoDims(**Wt.TINY)
    sd_np = Wt.make_showo_state(d, seed=11)
    ref = ref_showo_from_state(d, sd_np)
    P = R.load_reference().prompting
    sd = O.to_torch(sd_np)
    rs = np.random.RandomState(3)
    out = {}
    # ---- (1) forward, t2i masks with different pad counts (CFG-style batch), reference mask builder
    ids = t2i_ids(d, [5, 8, 3, 9], rs)
    mask = P.create_attention_mask_predict_next(ids, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id,
                                                rm_pad_in_image=True)
    with torch.no_grad():
        lg = ref(ids, attention_mask=mask)
    assert torch.equal(O.mask_t2i(ids, d.pad_id, d.soi_id, d.eoi_id), mask.float())
    e = report("t2i logits", O.showo_logits(sd, d, ids, attention_mask=mask), lg)
    assert e < 2e-4
    out.update(t2i_ids=ids.numpy(), t2i_mask=mask.numpy().astype(np.float32), t2i_logits=lg.numpy())
    # ---- (2) mmu mask
    ids_m = mmu_ids(d, 2, 8, rs)
    mask_m = P.create_attention_mask_for_mmu(ids_m, eoi_id=d.eoi_id)
    with torch.no_grad():
        lg_m = ref(ids_m, attention_mask=mask_m)
    assert torch.equal(O.mask_mmu(ids_m, d.eoi_id), mask_m.float())
    assert report("mmu logits", O.showo_logits(sd, d, ids_m, attention_mask=mask_m), lg_m) < 2e-4
    out.update(mmu_ids=ids_m.numpy(), mmu_mask=mask_m.numpy().astype(np.float32), mmu_logits=lg_m.numpy())
    # ---- (3) mixed training batch: 2 t2i + 1 lm + 2 mmu, three CE losses + grads of a few tensors
    L = ids.shape[1]
    img_gt = rs.randint(0, d.codebook, size=(2, d.num_vq_tokens)) + d.image_offset
    masked = rs.rand(2, d.num_vq_tokens) < 0.6
    img_in = np.where(masked, d.mask_token_id, img_gt)
    ids_t = t2i_ids(d, [6, 9], rs, image_tokens=img_in)
    lab_t = ids_t.clone()
    lab_t[:, -(d.num_vq_tokens + 1):-1] = torch.from_numpy(np.where(masked, img_gt, -100))
    lab_t[lab_t == d.pad_id] = -100
    ids_l = torch.from_numpy(rs.randint(0, d.llm_vocab - 20, size=(1, L))).long()
    lab_l = ids_l.clone()
    ids_u = mmu_ids(d, 2, L - d.num_vq_tokens - 4, rs)
    lab_u = ids_u.clone()
    lab_u[:, : d.num_vq_tokens + 3] = -100
    ids_all = torch.cat([ids_t, ids_l, ids_u])
    labels = torch.cat([lab_t, lab_l, lab_u])
    m_t = P.create_attention_mask_predict_next(ids_t, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id,
                                               rm_pad_in_image=True)
    # lm flow: rm_pad_in_image is left False by the trainer (training/train.py:530-533)
    m_l = P.create_attention_mask_predict_next(ids_l, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id)
    assert torch.equal(O.mask_t2i(ids_l, d.pad_id, d.soi_id, d.eoi_id, rm_pad_in_image=False), m_l.float())
    m_u = P.create_attention_mask_for_mmu(ids_u, eoi_id=d.eoi_id)
    mask_all = torch.cat([m_t, m_l, m_u]).float()
    ref.zero_grad()
    lg_a, l1, l2, l3 = ref(ids_all, attention_mask=mask_all, labels=labels, batch_size_t2i=2, batch_size_lm=1,
                           batch_size_mmu=2, max_seq_length=d.max_text_len)
    loss = 1.0 * l1 + 0.1 * l2 + 1.0 * l3
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    o = O.showo_forward(sd, d, ids_all, attention_mask=mask_all, labels=labels, batch_size_t2i=2,
                        batch_size_lm=1, batch_size_mmu=2, max_seq_length=d.max_text_len)
    for n_, a, b in zip(("loss_t2i", "loss_lm", "loss_mmu"), o[1:], (l1, l2, l3)):
        assert report(n_, a, b.detach()) < 1e-4
    out.update(train_ids=ids_all.numpy(), train_labels=labels.numpy(), train_mask=mask_all.numpy(),
               train_logits=lg_a.detach().numpy(),
               train_losses=np.array([l1.item(), l2.item(), l3.item()], dtype=np.float64))
    for k in ("showo.model.layers.0.self_attn.q_proj.weight", "showo.model.layers.1.mlp.fc2.weight",
              "showo.lm_head.bias", "showo.model.layers.0.input_layernorm.weight",
              "showo.model.layers.1.self_attn.k_layernorm.bias", "showo.model.final_layernorm.weight",
              "showo.model.layers.0.self_attn.dense.bias", "showo.model.layers.0.mlp.fc1.bias"):
        out["grad::" + k] = grads[k].numpy()
    out["grad::embed_rows"] = grads["showo.model.embed_tokens.weight"][ids_all[0, :8]].numpy()
    out["grad::embed_row_ids"] = ids_all[0, :8].numpy()
    # ---- (4) quirk: batch_size_mmu=0 selects the whole batch (modeling_showo.py:95-98)
    with torch.no_grad():
        _, q1, q2, q3 = ref(ids_all, attention_mask=mask_all, labels=labels, batch_size_t2i=2, batch_size_lm=3,
                            batch_size_mmu=0, max_seq_length=d.max_text_len)
    oq = O.showo_forward(sd, d, ids_all, attention_mask=mask_all, labels=labels, batch_size_t2i=2,
                         batch_size_lm=3, batch_size_mmu=0, max_seq_length=d.max_text_len)
    assert report("quirk loss_mmu", oq[3], q3) < 1e-4
    out["quirk_losses"] = np.array([q1.item(), q2.item(), q3.item()], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "showo_tiny_forward.npz"), **out)

    # ---- (5) t2i_generate trajectory with recorded noise
    print("[tiny t2i_generate]")
    steps, w = 6, 1.75
    rs = np.random.RandomState(4)
    ids_c = t2i_ids(d, [7, 4, 9], rs)
    ids_u = t2i_ids(d, [3, 3, 3], rs)
    # one sample is an inpainting case: some tokens already known
    known = rs.rand(d.num_vq_tokens) < 0.4
    ids_c[2, -(d.num_vq_tokens + 1):-1] = torch.from_numpy(
        np.where(known, rs.randint(0, d.codebook, size=d.num_vq_tokens) + d.image_offset, d.mask_token_id))
    ids_c0 = ids_c.clone()
    mask = P.create_attention_mask_predict_next(torch.cat([ids_c, ids_u]), pad_id=d.pad_id, soi_id=d.soi_id,
                                                eoi_id=d.eoi_id, rm_pad_in_image=True)
    gen = torch.Generator().manual_seed(7)
    rec = dict(exp=[], uni=[], multi=[], fwd_in=[], fwd_out=[])
    real_multinomial = torch.multinomial
    real_uniform = torch.Tensor.uniform_
    real_forward = ref.forward

    def rec_multinomial(p, n, generator=None, **kw):
        st = generator.get_state()
        q = torch.empty_like(p).exponential_(1, generator=generator)
        mine = torch.argmax(p / q, dim=-1, keepdim=True)
        generator.set_state(st)
        res = real_multinomial(p, n, generator=generator, **kw)
        assert torch.equal(res, mine), "torch.multinomial != argmax(p/Exp(1))"
        rec["exp"].append(q.clone())
        rec["multi"].append(res.clone())
        return res

    def rec_uniform(self, a=0, b=1, generator=None):
        r = real_uniform(self, a, b, generator=generator)
        rec["uni"].append(r.clone())
        return r

    def rec_forward(x, **kw):
        rec["fwd_in"].append(x.clone())
        y = real_forward(x, **kw)
        rec["fwd_out"].append(y.clone())
        return y

    torch.multinomial = rec_multinomial
    torch.Tensor.uniform_ = rec_uniform
    ref.forward = rec_forward
    try:
        with torch.no_grad():
            ids_run = ids_c.clone()
            res = ref.t2i_generate(input_ids=ids_run, uncond_input_ids=ids_u.clone(), attention_mask=mask,
                                   temperature=1.0, timesteps=steps, guidance_scale=w,
                                   noise_schedule=R.load_reference().sampling.cosine_schedule, generator=gen,
                                   config=gen_config(d))
    finally:
        torch.multinomial = real_multinomial
        torch.Tensor.uniform_ = real_uniform
        ref.forward = real_forward
    noise = O.RecordedNoise(rec["exp"], rec["uni"])
    tr = []
    o_ids = ids_c0.clone()
    o_res = O.t2i_generate(sd, d, o_ids, ids_u.clone(), mask, 1.0, steps, w, noise=noise, trace=tr)
    assert torch.equal(o_res, res), "oracle t2i_generate trajectory differs from the reference"
    assert torch.equal(o_ids, ids_run)
    for s in range(steps):
        assert torch.equal(tr[s]["input_ids"], rec["fwd_in"][s][:3])
    print("  oracle t2i_generate == reference (ids bit-exact over", steps, "steps)")
    ml, tp = O.t2i_step_constants(steps, d.num_vq_tokens)
    np.savez_compressed(
        os.path.join(GOLD, "showo_tiny_t2i.npz"),
        ids_cond=ids_c0.numpy(), ids_uncond=ids_u.numpy(), mask=mask.numpy().astype(np.float32),
        steps=steps, guidance=w, exp_noise=torch.stack(rec["exp"]).numpy(), uniform=torch.stack(rec["uni"]).numpy(),
        multinomial=torch.stack(rec["multi"]).numpy(), fwd_in=torch.stack(rec["fwd_in"]).numpy(),
        fwd_logits=torch.stack(rec["fwd_out"]).numpy(), result=res.numpy(), final_input_ids=ids_run.numpy(),
        mask_len=np.array(ml), temps=np.array(tp))

    # ---- (6) mmu_generate (top_k=1 → deterministic)
    print("[tiny mmu_generate]")
    rs = np.random.RandomState(5)
    ids_m = mmu_ids(d, 1, 5, rs)
    mask_m = P.create_attention_mask_for_mmu(ids_m, eoi_id=d.eoi_id)
    toks = ref.mmu_generate(ids_m.clone(), attention_mask=mask_m.clone(), max_new_tokens=6, top_k=1)
    toks_o = O.mmu_generate(sd, d, ids_m.clone(), attention_mask=mask_m.float().clone(), max_new_tokens=6, top_k=1)
    assert [int(t) for t in toks] == [int(t) for t in toks_o], (toks, toks_o)
    print("  tokens", [int(t) for t in toks])
    # stochastic decode (top_k / temperature / multinomial, modeling_showo.py:220-228) with the reference's own Exp(1) draws
    real_multinomial = torch.multinomial
    extra = {}
    for tag, kw in (("topk5", dict(top_k=5, temperature=0.7)), ("full", dict(top_k=None, temperature=1.3))):
        draws = []

        def rec_multinomial(p, num_samples=1, **k2):
            st = torch.get_rng_state()
            q = torch.empty_like(p).exponential_(1)
            mine = torch.argmax(p / q, dim=-1, keepdim=True)
            torch.set_rng_state(st)
            res = real_multinomial(p, num_samples, **k2)
            assert torch.equal(res, mine), "torch.multinomial != argmax(p/Exp(1))"
            draws.append(q[0].clone())
            return res
        torch.manual_seed(41)
        torch.multinomial = rec_multinomial
        try:
            toks_s = ref.mmu_generate(ids_m.clone(), attention_mask=mask_m.clone(), max_new_tokens=8, **kw)
        finally:
            torch.multinomial = real_multinomial
        noise = O.RecordedNoise(list(draws), [])
        toks_so = O.mmu_generate(sd, d, ids_m.clone(), attention_mask=mask_m.float().clone(), max_new_tokens=8, noise=noise, **kw)
        assert [int(t) for t in toks_s] == [int(t) for t in toks_so], (tag, toks_s, toks_so)
        print(f"  {tag} tokens", [int(t) for t in toks_s])
        extra[f"tokens_{tag}"] = np.array([int(t) for t in toks_s])
        extra[f"exp_noise_{tag}"] = torch.stack(draws).numpy()
    np.savez_compressed(os.path.join(GOLD, "showo_tiny_mmu.npz"), ids=ids_m.numpy(),
                        mask=mask_m.numpy().astype(np.float32), tokens=np.array([int(t) for t in toks]), **extra)


def make_magvit():
    print("[magvit full-arch, small images]")
    sd_np = Wt.make_magvit_state(seed=21)
    ref = R.build_reference_magvit()
    ref.load_state_dict(O.to_torch(sd_np), strict=True)
    sd = O.to_torch(sd_np)
    rs = np.random.RandomState(6)
    x = torch.from_numpy(rs.uniform(-1, 1, size=(2, 3, 64, 64)).astype(np.float32))
    with torch.no_grad():
        z = ref.quantize(ref.encoder(x))  # dict
        z_enc = ref.encoder(x)
        ids = ref.get_code(x)
        zq_ref, ids2 = ref.encode(x)
        img = ref.decode_code(ids)
        ids_ns = torch.from_numpy(rs.randint(0, 8192, size=(1, 8))).long()
        img_ns = ref.decode_code(ids_ns, shape=(2, 4))
    assert torch.equal(ids, ids2)
    o_ids, o_z = O.magvit_get_code(sd, x, return_z=True)
    report("encoder z", o_z, z_enc)
    agree = (o_ids == ids).float().mean().item()
    print("  get_code agreement oracle/ref:", agree)
    assert agree == 1.0
    assert report("decode_code", O.magvit_decode_code(sd, ids), img) < 1e-3
    assert report("decode_code shape=(2,4)", O.magvit_decode_code(sd, ids_ns, shape=(2, 4)), img_ns) < 1e-3
    np.savez_compressed(os.path.join(GOLD, "magvit_small.npz"), seed=21, x=x.numpy(), z=z_enc.numpy(),
                        ids=ids.numpy(), image=img.numpy(), ids_ns=ids_ns.numpy(), image_ns=img_ns.numpy())
    # LFQ known-answer vectors straight from the reference quantizer (edge cases of SURVEY §8c.1)
    zz = rs.standard_normal(size=(3, 13, 4, 5)).astype(np.float32)
    zz[0, :, 0, 0] = 0.0
    zz[0, :, 0, 1] = -0.0
    zz[0, :, 0, 2] = 1e-9
    zz[0, :, 0, 3] = -1e-9
    zz[0, :, 1, 0] = np.float32(1e-45)  # denormal
    zz[0, :, 1, 1] = -np.float32(1e-45)
    zz[1, ::2, 2, 2] = 0.0
    zt = torch.from_numpy(zz)
    with torch.no_grad():
        q = ref.quantize(zt)
        kid = ref.quantize.get_indices(q["z"]).reshape(3, -1)
        back = ref.quantize.get_codebook_entry(kid[:, :16], shape=(4, 4))
    assert np.array_equal(O.lfq_pack_np(zz), kid.numpy())
    assert np.array_equal(O.lfq_unpack_np(kid[:, :16].numpy(), shape=(4, 4)), back.numpy())
    np.savez_compressed(os.path.join(GOLD, "lfq_kat.npz"), z=zz, ids=kid.numpy(), back=back.numpy())
    print("  LFQ KATs ok")


def make_full_showo():
    print("[full-size showo: logits subset]  (needs ~20 GB RAM, a few minutes)")
    d = Wt.ShowoDims()
    sd_np = Wt.make_showo_state(d, seed=0)
    ref = ref_showo_from_state(d, sd_np)
    P = R.load_reference().prompting
    rs = np.random.RandomState(8)
    ids = t2i_ids(d, [7, 3], rs, bos=50256)
    mask = P.create_attention_mask_predict_next(ids, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id,
                                                rm_pad_in_image=True)
    with torch.no_grad():
        lg = ref(ids, attention_mask=mask)
    rows = np.array([0, 100, 122, 128, 129, 130, 257, 386])
    cols = np.concatenate([np.arange(0, d.vocab, 11), np.arange(d.image_offset, d.image_offset + 64)])
    sub = lg[:, rows][:, :, cols].numpy()
    sd = O.to_torch(sd_np)
    o = O.showo_logits(sd, d, ids, attention_mask=mask)
    report("full logits", o, lg)
    np.savez_compressed(os.path.join(GOLD, "showo_full_logits_subset.npz"), seed=0, ids=ids.numpy(), rows=rows,
                        cols=cols, logits=sub, logit_absmax=float(lg.abs().max()), logit_std=float(lg.std()))


def make_prompting():
    """UniversalPrompting layouts + mask_or_random_replace_tokens: the real reference on a stub tokenizer / seeded RNGs."""
    import json
    import random
    from stub_tokenizer import StubTokenizer
    ref = R.load_reference()
    tu = R.load_reference_training_utils()
    out = {}
    texts = ["a photo of a cat", "", "<bos> already has bos", " ".join(f"w{i}" for i in range(40)), "two words",
             "x " * 11 + "exactly"]
    N = 16
    rs = np.random.RandomState(5)
    up = ref.prompting.UniversalPrompting(StubTokenizer(), max_text_len=12, max_seq_len=12 + N + 3, cond_dropout_prob=0.5)
    img = torch.from_numpy(rs.randint(310, 310 + 128, size=(len(texts), N)))
    lab = torch.where(torch.from_numpy(rs.rand(len(texts), N) < 0.5), img, torch.full_like(img, -100))
    out["texts"] = np.array(json.dumps(texts))
    out["image_ids"], out["labels"] = img.numpy(), lab.numpy()
    out["sptids"] = np.array(json.dumps({k: int(v) for k, v in up.sptids_dict.items()}))
    out["pad_id"], out["max_text_len"] = up.pad_id, up.max_text_len
    cfg = ns(training=ns(batch_size=4))

    def put(name, tup):
        for k, t in zip(("seq", "mask", "lab"), tup):
            out[f"{name}_{k}"] = t.numpy()

    for task in ("t2i", "t2v", "lvg"):
        torch.manual_seed(21)
        put(task, up((list(texts), img, lab), task))
        out[f"{task}_rng_after"] = torch.rand(3).numpy()  # position of the host generator after the call
    for task in ("t2i_gen", "t2v_gen", "lvg_gen"):
        put(task, up((list(texts), img), task))
    put("lm", up((list(texts), 12 + N + 3), "lm"))
    put("lm_short", up((list(texts), 9), "lm"))
    put("mmu", up((img, list(texts)), "mmu"))
    torch.manual_seed(22)
    a, b = up((list(texts), img[:4], lab[:4], 31), "t2i_plus_lm", config=cfg)
    put("plus_t2i", a)
    put("plus_lm", b)

    # ---- mask_or_random_replace_tokens -------------------------------------------------------------------------
    class Sec(dict):
        __getattr__ = dict.__getitem__
    sched = ref.sampling.cosine_schedule
    tokens = torch.from_numpy(rs.randint(310, 310 + 128, size=(5, 64)))
    mask_id = 438
    real_rand = torch.rand
    for name, tr, is_train in (("mlm", dict(min_masking_rate=0.0), True),
                               ("mlm_minrate", dict(min_masking_rate=0.6), True),
                               ("mlm_all", dict(min_masking_rate=0.0, predict_all_tokens=True), True),
                               ("mlm_rr", dict(min_masking_rate=0.0, noise_type="random_replace"), True),
                               ("mlm_rect", dict(min_masking_rate=0.1, mask_contiguous_region_prob=1.0), True),
                               ("mlm_eval", dict(min_masking_rate=0.0, eval_mask_ratios=[0.25, 0.5, 0.9]), False)):
        draws = []

        def rec(*a, **k):
            t = real_rand(*a, **k)
            draws.append(t.clone())
            return t
        torch.manual_seed(31)
        random.seed(32)
        torch.rand = rec
        try:
            inp, labels, lw, mp = tu.mask_or_random_replace_tokens(tokens, mask_id, ns(training=Sec(tr), model=ns(codebook_size=128)),
                                                                   sched, is_train=is_train)
        finally:
            torch.rand = real_rand
        out[f"{name}_in"], out[f"{name}_lab"], out[f"{name}_prob"] = inp.numpy(), labels.numpy(), mp.numpy()
        if lw is not None:
            out[f"{name}_lw"] = lw.numpy()
        for i, dr in enumerate(draws):
            out[f"{name}_draw{i}"] = dr.numpy()
        if name in ("mlm", "mlm_minrate", "mlm_all", "mlm_rr"):
            i2, l2, m2 = O.mask_tokens_np(tokens.numpy(), draws[1].numpy(), mp.numpy(), mask_id, predict_all=lw is not None)
            assert np.array_equal(i2, inp.numpy()) and np.array_equal(l2, labels.numpy()), name
            if lw is not None:
                assert np.allclose(O.loss_weight_np(mp.numpy(), m2), lw.numpy(), atol=1e-7)
    out["mlm_tokens"], out["mlm_mask_id"] = tokens.numpy(), mask_id
    np.savez_compressed(os.path.join(GOLD, "prompting.npz"), **out)
    print("prompting.npz:", len(out), "arrays; oracle restatement == reference on the 4 random-permutation cases")


def make_clip():
    """CLIP vision tower: the installed transformers CLIPVisionModel (what models/clip_encoder.py calls) on the deterministic
    weights of oracle/weights.py -> hidden_states[-2][:, 1:]; the restatement is asserted against it."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    out = {}
    for tag, cfg, B, seed in (("tiny", Wt.CLIP_TINY, 3, 21), ("l336", Wt.CLIP_L336, 1, 22)):
        sd_np = Wt.make_clip_state(cfg, seed=seed)
        hf = CLIPVisionModel(CLIPVisionConfig(**cfg)).eval()
        own = hf.state_dict()
        strip = not any(k.startswith("vision_model.") for k in own)  # transformers >= 5 dropped that level
        hf.load_state_dict({(k[len("vision_model."):] if strip else k): torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
        rs = np.random.RandomState(seed + 100)
        x = torch.from_numpy(rs.standard_normal((B, 3, cfg["image_size"], cfg["image_size"])).astype(np.float32))
        with torch.no_grad():
            ref = hf(x, output_hidden_states=True).hidden_states[-2][:, 1:]
            mine = O.clip_vision_features(O.to_torch(sd_np), cfg, x)
        report(f"clip {tag} features", mine, ref)
        assert (mine - ref).abs().max() <= 2e-4 * float(ref.abs().max())
        out[f"{tag}_seed"] = seed  # weights = Wt.make_clip_state(cfg, seed); x = RandomState(seed + 100).standard_normal(...)
        if tag == "tiny":
            out["tiny_x"], out["tiny_features"] = x.numpy(), ref.numpy()
        else:  # full size: a strided subset of the [1,576,1024] features
            out["l336_rows"], out["l336_cols"] = np.arange(0, 576, 37), np.arange(0, 1024, 13)
            out["l336_features"] = ref[0][out["l336_rows"]][:, out["l336_cols"]].numpy()
            out["l336_absmax"], out["l336_std"] = float(ref.abs().max()), float(ref.std())
    psd = Wt.make_projector_state(128, 192, seed=23)
    px = torch.from_numpy(np.random.RandomState(24).standard_normal((37, 128)).astype(np.float32))
    ref = torch.nn.Sequential(torch.nn.Linear(128, 192), torch.nn.GELU(), torch.nn.Linear(192, 192))  # modeling_showo.py:48-53
    ref.load_state_dict(O.to_torch(psd))
    with torch.no_grad():
        want = ref(px)
    assert torch.allclose(O.mm_projector(O.to_torch(psd), px), want, atol=1e-6)
    out["proj_x"], out["proj_out"] = px.numpy(), want.numpy()
    np.savez_compressed(os.path.join(GOLD, "clip_vision.npz"), **out)
    print("clip_vision.npz written")


def make_image_ops():
    """image_transform: installed Pillow (what torchvision's Resize calls) vs the oracle restatement; fixture for the GPU test"""
    from PIL import Image
    rs = np.random.RandomState(9)
    img = rs.randint(0, 256, size=(167, 250, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:167, 0:250]
    img[..., 1] = (127 + 120 * np.sin(xx / 11.0) * np.cos(yy / 19.0)).astype(np.uint8)
    R = 128
    ow, oh = int(R * 250 / 167), R
    ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BICUBIC))
    ct, cl = int(round((oh - R) / 2.0)), int(round((ow - R) / 2.0))
    t, u8 = O.image_transform_np(img, R, True)
    assert np.array_equal(u8, ref[ct:ct + R, cl:cl + R]), "oracle resize != PIL"
    for (H, W, C, w2, h2) in [(300, 451, 3, 256, 170), (64, 64, 3, 256, 256), (123, 77, 1, 40, 64), (97, 400, 3, 256, 62)]:
        im = rs.randint(0, 256, size=(H, W, C)).astype(np.uint8)
        pil = Image.fromarray(im if C == 3 else im[..., 0], "RGB" if C == 3 else "L").resize((w2, h2), Image.BICUBIC)
        assert np.array_equal(np.asarray(pil).reshape(h2, w2, C), O.pil_resize_bicubic_np(im, w2, h2)), (H, W, C)
    import PIL
    np.savez_compressed(os.path.join(GOLD, "image_ops.npz"), img=img, resolution=R, pil_bytes=u8, tensor=t.numpy(),
                        pillow_version=np.array(PIL.__version__))
    print("image_ops.npz written (oracle resize == Pillow", PIL.__version__, "byte for byte)")


def _record_t2i(ref, d, ids_c, ids_u, mask, steps, w, seed):
    """run the REFERENCE t2i_generate with its random draws recorded (multinomial as argmax(p / Exp(1)), Gumbel uniforms) and
    every forward's input ids / logits; returns (result, final input ids, rec)"""
    gen = torch.Generator().manual_seed(seed)
    rec = dict(exp=[], uni=[], multi=[], fwd_in=[], fwd_out=[])
    real_multinomial, real_uniform, real_forward = torch.multinomial, torch.Tensor.uniform_, ref.forward

    def rec_multinomial(p, n, generator=None, **kw):
        st = generator.get_state()
        q = torch.empty_like(p).exponential_(1, generator=generator)
        mine = torch.argmax(p / q, dim=-1, keepdim=True)
        generator.set_state(st)
        res = real_multinomial(p, n, generator=generator, **kw)
        assert torch.equal(res, mine), "torch.multinomial != argmax(p/Exp(1))"
        rec["exp"].append(q.clone())
        rec["multi"].append(res.clone())
        return res

    def rec_uniform(self, a=0, b=1, generator=None):
        r = real_uniform(self, a, b, generator=generator)
        rec["uni"].append(r.clone())
        return r

    def rec_forward(x, **kw):
        rec["fwd_in"].append(x.clone())
        y = real_forward(x, **kw)
        rec["fwd_out"].append(y.clone())
        return y

    torch.multinomial, torch.Tensor.uniform_, ref.forward = rec_multinomial, rec_uniform, rec_forward
    try:
        with torch.no_grad():
            ids_run = ids_c.clone()
            res = ref.t2i_generate(input_ids=ids_run, uncond_input_ids=ids_u.clone(), attention_mask=mask, temperature=1.0,
                                   timesteps=steps, guidance_scale=w, noise_schedule=R.load_reference().sampling.cosine_schedule,
                                   generator=gen, config=gen_config(d))
    finally:
        torch.multinomial, torch.Tensor.uniform_, ref.forward = real_multinomial, real_uniform, real_forward
    return res, ids_run, rec


def make_tiny_inpaint():
    """cfg3-shaped trajectory on the tiny model (inference_t2i.py:100-113): batch 4, N = 64 image tokens (8x8 grid),
    the centred 4x4 block is generated, every other position is pre-filled with the codes of ONE image repeated over the batch,
    CFG 5.0, 18 steps -> tests/golden/showo_tiny_inpaint.npz"""
    print("[tiny inpainting t2i_generate, N = 64, 18 steps]")
    P = R.load_reference().prompting
    d = Wt.ShowoDims(**dict(Wt.TINY, num_vq_tokens=64))
    sd_np = Wt.make_showo_state(d, seed=11)
    ref = ref_showo_from_state(d, sd_np)
    sd = O.to_torch(sd_np)
    rs = np.random.RandomState(12)
    grid = np.zeros((8, 8), dtype=bool)
    grid[2:6, 2:6] = True
    codes = rs.randint(0, d.codebook, size=64)
    img = np.where(grid.reshape(-1), d.mask_token_id, codes + d.image_offset)
    ids_c = t2i_ids(d, [4, 6, 8, 9], rs, image_tokens=[img] * 4)
    ids_u = t2i_ids(d, [3, 3, 3, 3], rs, image_tokens=[img] * 4)
    ids_c0 = ids_c.clone()
    mask = P.create_attention_mask_predict_next(torch.cat([ids_c, ids_u]), pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id,
                                                rm_pad_in_image=True)
    steps, w = 18, 5.0
    res, ids_run, rec = _record_t2i(ref, d, ids_c, ids_u, mask, steps, w, seed=9)
    noise = O.RecordedNoise(rec["exp"], rec["uni"])
    o_ids = ids_c0.clone()
    o_res = O.t2i_generate(sd, d, o_ids, ids_u.clone(), mask, 1.0, steps, w, noise=noise)
    assert torch.equal(o_res, res) and torch.equal(o_ids, ids_run), "oracle inpainting trajectory differs from the reference"
    keep = ~torch.from_numpy(grid.reshape(-1))
    assert torch.equal(res[:, keep], torch.from_numpy(codes)[keep].expand(4, -1))  # known tokens come back untouched
    print("  oracle == reference over", steps, "steps; known tokens untouched")
    np.savez_compressed(os.path.join(GOLD, "showo_tiny_inpaint.npz"), ids_cond=ids_c0.numpy(), ids_uncond=ids_u.numpy(),
                        mask=mask.numpy().astype(np.float32), steps=steps, guidance=w, num_vq_tokens=64, hole=grid.reshape(-1),
                        exp_noise=torch.stack(rec["exp"]).numpy().astype(np.float32), uniform=torch.stack(rec["uni"]).numpy(),
                        fwd_in=torch.stack(rec["fwd_in"]).numpy(), multinomial=torch.stack(rec["multi"]).numpy(),
                        result=res.numpy(), final_input_ids=ids_run.numpy())


def make_magvit_256():
    """config-size VQ fixture (the bench's decode_code runs at 256x256): the REFERENCE MAGVITv2 on one 256x256 image ->
    all 256 ids, the full latent (13x16x16), and every 4th pixel of decode_code(ids) -> tests/golden/magvit_256.npz.  The input is
    regenerated from its seed by the test (RandomState(31).uniform(-1, 1, (1, 3, 256, 256)))."""
    print("[magvit 256x256]")
    sd_np = Wt.make_magvit_state(seed=21)
    ref = R.build_reference_magvit()
    ref.load_state_dict(O.to_torch(sd_np), strict=True)
    x = torch.from_numpy(np.random.RandomState(31).uniform(-1, 1, size=(1, 3, 256, 256)).astype(np.float32))
    with torch.no_grad():
        z = ref.encoder(x)
        ids = ref.get_code(x)
        img = ref.decode_code(ids)
    sd = O.to_torch(sd_np)
    o_ids, o_z = O.magvit_get_code(sd, x, return_z=True)
    report("encoder z @256", o_z, z)
    assert torch.equal(o_ids, ids)
    assert report("decode_code @256", O.magvit_decode_code(sd, ids), img) < 1e-3
    np.savez_compressed(os.path.join(GOLD, "magvit_256.npz"), seed=21, x_seed=31, z=z.numpy(), ids=ids.numpy(),
                        image_s4=img[:, :, ::4, ::4].contiguous().numpy(), image_absmax=float(img.abs().max()),
                        image_rms=float(img.pow(2).mean().sqrt()))


def make_small_train():
    """Production-path training fixture: the REFERENCE's losses and gradients on a mixed 6 t2i + 2 lm + 4 mmu batch
    of 12 x 27 = 324 token rows on the SMALL geometry (hidden 256: 3 * hidden is a multiple of the 256-wide GEMM tile, every GEMM
    dimension >= 256), i.e. a batch the HIP trainer runs through its T >= 256 branch: fused [Wqkv ; W1] save-for-backward launch,
    gemm2p / gemm3w forward + dgrad + wgrad (K = token count), multi-block transposes.  -> tests/golden/showo_small_train.npz"""
    print("[small showo, 324-row training batch]")
    d = Wt.ShowoDims(**Wt.SMALL)
    sd_np = Wt.make_showo_state(d, seed=13)
    ref = ref_showo_from_state(d, sd_np)
    P = R.load_reference().prompting
    sd = O.to_torch(sd_np)
    rs = np.random.RandomState(17)
    N = d.num_vq_tokens
    bt, bl, bm = 6, 2, 4
    img_gt = rs.randint(0, d.codebook, size=(bt, N)) + d.image_offset
    masked = rs.rand(bt, N) < rs.uniform(0.2, 0.95, size=(bt, 1))
    masked[:, 0] = True  # at least one masked position per row (training/utils.py:101-102)
    img_in = np.where(masked, d.mask_token_id, img_gt)
    ids_t = t2i_ids(d, [6, 9, 3, 8, 5, 7], rs, image_tokens=img_in)
    L = ids_t.shape[1]
    lab_t = ids_t.clone()
    lab_t[:, -(N + 1):-1] = torch.from_numpy(np.where(masked, img_gt, -100))
    lab_t[lab_t == d.pad_id] = -100
    ids_l = torch.from_numpy(rs.randint(0, d.llm_vocab - 20, size=(bl, L))).long()
    ids_l[1, :5] = d.pad_id  # a left-padded lm row
    lab_l = ids_l.clone()
    lab_l[lab_l == d.pad_id] = -100
    ids_u = mmu_ids(d, bm, L - N - 4, rs)
    lab_u = ids_u.clone()
    lab_u[:, : N + 3] = -100
    ids_all = torch.cat([ids_t, ids_l, ids_u])
    labels = torch.cat([lab_t, lab_l, lab_u])
    assert ids_all.shape[0] * L >= 256
    m_t = P.create_attention_mask_predict_next(ids_t, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id, rm_pad_in_image=True)
    m_l = P.create_attention_mask_predict_next(ids_l, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id)
    m_u = P.create_attention_mask_for_mmu(ids_u, eoi_id=d.eoi_id)
    mask_all = torch.cat([m_t, m_l, m_u]).float()
    ref.zero_grad()
    lg, l1, l2, l3 = ref(ids_all, attention_mask=mask_all, labels=labels, batch_size_t2i=bt, batch_size_lm=bl, batch_size_mmu=bm,
                         max_seq_length=d.max_text_len)
    (1.0 * l1 + 0.1 * l2 + 1.0 * l3).backward()
    grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    o = O.showo_forward(sd, d, ids_all, attention_mask=mask_all, labels=labels, batch_size_t2i=bt, batch_size_lm=bl,
                        batch_size_mmu=bm, max_seq_length=d.max_text_len)
    assert report("logits", o[0], lg.detach()) < 2e-4
    for n_, a, b in zip(("loss_t2i", "loss_lm", "loss_mmu"), o[1:], (l1, l2, l3)):
        assert report(n_, a, b.detach()) < 1e-4
    out = dict(ids=ids_all.numpy(), labels=labels.numpy(), mask=mask_all.numpy(), b=np.array([bt, bl, bm]),
               logits_s=lg.detach()[:, ::3].contiguous().numpy(),  # every third position: keeps the file small
               losses=np.array([l1.item(), l2.item(), l3.item()], dtype=np.float64))
    big = ("showo.model.layers.0.self_attn.q_proj.weight", "showo.model.layers.1.self_attn.v_proj.weight",
           "showo.model.layers.0.self_attn.dense.weight", "showo.model.layers.1.mlp.fc1.weight", "showo.model.layers.0.mlp.fc2.weight",
           "showo.model.layers.1.self_attn.k_proj.weight")
    for k, v in grads.items():
        if k in big or (v.dim() == 1 and "embed" not in k):
            out["grad::" + k] = v.numpy()
    rows = torch.unique(ids_all.reshape(-1))[:24]
    out["grad::embed_row_ids"] = rows.numpy()
    out["grad::embed_rows"] = grads["showo.model.embed_tokens.weight"][rows].numpy()
    hrows = torch.arange(0, d.vocab, 7)
    out["grad::lm_head_row_ids"] = hrows.numpy()
    out["grad::lm_head_rows"] = grads["showo.lm_head.weight"][hrows].numpy()
    np.savez_compressed(os.path.join(GOLD, "showo_small_train.npz"), **out)
    print("  rows", ids_all.shape[0] * L, "losses", out["losses"], "tensors", len(out))


def make_full_cfg3():
    """BASELINE cfg3 at MODEL SCALE (configs/showo_demo_512x512.yaml: 1 024 image tokens, batch 4, CFG-doubled -> [8,1155], centred
    16x16-token inpainting hole, inference_t2i.py:100-113): the REAL reference's logits on one mask-predict forward
    (models/modeling_showo.py:104-181 calls self(...) on exactly this tensor), a rows x cols subset -> tests/golden/showo_full_cfg3.npz.
    Weights = make_showo_state(seed 0) as in showo_full_logits_subset.npz (num_vq_tokens does not enter the state dict)."""
    print("[full-size showo, cfg3: [8,1155] inpainting batch]  (needs ~30 GB RAM, several minutes)")
    d = Wt.ShowoDims(num_vq_tokens=1024)
    sd_np = Wt.make_showo_state(d, seed=0)
    ref = ref_showo_from_state(d, sd_np)
    P = R.load_reference().prompting
    rs = np.random.RandomState(23)
    B, N = 4, d.num_vq_tokens
    grid = np.zeros((32, 32), dtype=bool)
    grid[8:24, 8:24] = True
    hole = grid.reshape(-1)
    codes = rs.randint(0, d.codebook, size=(B, N)) + d.image_offset
    img = np.where(hole[None], d.mask_token_id, codes)
    ids_c = t2i_ids(d, [3 + 7 * i for i in range(B)], rs, bos=50256, image_tokens=img.tolist())
    ids_u = t2i_ids(d, [3] * B, rs, bos=50256, image_tokens=img.tolist())
    ids = torch.cat([ids_c, ids_u])
    assert tuple(ids.shape) == (8, 1155)
    mask = P.create_attention_mask_predict_next(ids, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id, rm_pad_in_image=True)
    with torch.no_grad():
        lg = ref(ids, attention_mask=mask)
    del ref
    # rows: pads, text, <soi>, kept tokens, hole tokens (the rows t2i_generate samples from), <eoi>
    T0 = 130
    rows = np.array([0, 64, 122, 127, 128, 129, T0, T0 + 31, T0 + 8 * 32 + 7, T0 + 8 * 32 + 8, T0 + 12 * 32 + 16, T0 + 23 * 32 + 23,
                     T0 + 23 * 32 + 24, T0 + 1023, 1154])
    cols = np.concatenate([np.arange(0, d.vocab, 29), np.arange(d.image_offset, d.image_offset + 128)])
    sub = lg[:, rows][:, :, cols].numpy()
    sd = O.to_torch(sd_np)
    o = O.showo_logits(sd, d, ids, attention_mask=mask)
    assert report("full cfg3 logits", o, lg) < 2e-3
    np.savez_compressed(os.path.join(GOLD, "showo_full_cfg3.npz"), seed=0, ids=ids.numpy().astype(np.int32), rows=rows, cols=cols,
                        logits=sub, hole=hole, logit_absmax=float(lg.abs().max()), logit_std=float(lg.std()))


def make_full_cfg4():
    """BASELINE cfg4 at MODEL SCALE (inference_mmu.py:100-150, w_clip_vit): 631 prompt embeddings = [<mmu>, 28 system ids, <soi>] +
    mm_projector(576 CLIP patch features) + [<eoi>, 24 question ids], the reference's create_attention_mask_for_mmu_vit
    (training/prompting_utils.py:606-624) and its own `mmu_generate` (models/modeling_showo.py:183-240: no KV cache, whole sequence
    re-run per token) for 8 greedy tokens.  Recorded: the tokens, every step's last-row logits (column subset + top-2 gap), and a
    rows x cols subset of the PREFILL logits -> tests/golden/showo_full_cfg4.npz.  The CLIP features are regenerated from their seed
    by the test (RandomState(41).standard_normal((1, 576, 1024))); the tower itself is pinned by clip_vision.npz."""
    print("[full-size showo, cfg4: 631-embedding mmu_vit prefill + 8 greedy tokens]  (several minutes)")
    d = Wt.ShowoDims(w_clip_vit=True)
    sd_np = Wt.make_showo_state(d, seed=0)
    ref = ref_showo_from_state(d, sd_np)
    P = R.load_reference().prompting
    rs = np.random.RandomState(41)
    feats = torch.from_numpy(rs.standard_normal((1, 576, 1024)).astype(np.float32))
    sys_ids = rs.randint(0, 50256, size=28).tolist()
    q_ids = rs.randint(0, 50256, size=24).tolist()
    ids_llava = torch.tensor([[d.mmu_id] + sys_ids + [d.soi_id, d.eoi_id] + q_ids], dtype=torch.long)
    NEW = 8
    rec = []
    with torch.no_grad():
        img_emb = ref.mm_projector(feats)
        txt = ref.showo.model.embed_tokens(ids_llava)
        emb = torch.cat([txt[:, :30], img_emb, txt[:, 30:]], dim=1)
        assert emb.shape[1] == 631
        am = P.create_attention_mask_for_mmu_vit(emb, system_prompt_len=28)
        orig_forward = ref.forward

        def rec_forward(*a, **kw):
            out = orig_forward(*a, **kw)
            rec.append(out.detach().clone())
            return out
        ref.forward = rec_forward
        toks = ref.mmu_generate(input_embeddings=emb, attention_mask=am[0].unsqueeze(0), max_new_tokens=NEW, top_k=1)
        ref.forward = orig_forward
    toks = np.array([int(t) for t in toks])
    assert len(rec) == NEW and rec[0].shape[1] == 631
    cols = np.concatenate([np.arange(0, d.vocab, 29), np.arange(d.image_offset, d.image_offset + 64)])
    last = np.stack([r[0, -1].numpy() for r in rec])  # [NEW, V]: the logits each token was drawn from
    srt = np.sort(last, axis=1)
    gap = srt[:, -1] - srt[:, -2]
    assert np.array_equal(last.argmax(1), toks)
    rows = np.array([0, 1, 15, 28, 29, 30, 31, 300, 605, 606, 607, 620, 630])
    pre = rec[0][0][rows][:, cols].numpy()
    print("  tokens", toks.tolist(), "top-2 gaps", np.round(gap, 4).tolist(), "logit std", float(rec[0].std()))
    sd = O.to_torch(sd_np)
    o_emb = torch.cat([txt[:, :30], O.mm_projector({k[len("mm_projector."):]: v for k, v in sd.items() if k.startswith("mm_projector.")},
                                                    feats), txt[:, 30:]], dim=1)
    report("oracle mm_projector splice", o_emb, emb)
    o = O.showo_logits(sd, d, None, input_embeddings=o_emb, attention_mask=am)
    assert report("full cfg4 prefill logits", o, rec[0]) < 2e-3
    np.savez_compressed(os.path.join(GOLD, "showo_full_cfg4.npz"), seed=0, feat_seed=41, ids_llava=ids_llava.numpy().astype(np.int32),
                        tokens=toks, cols=cols, last_logits=last[:, cols], last_top2_gap=gap.astype(np.float32),
                        last_absmax=np.abs(last).max(1).astype(np.float32), rows=rows, prefill_logits=pre,
                        logit_absmax=float(rec[0].abs().max()), logit_std=float(rec[0].std()),
                        emb_absmax=float(emb.abs().max()), img_emb_rows=img_emb[0, ::64].numpy())


def make_magvit_512():
    """512x512 VQ fixture (BASELINE cfg3 / cfg4 encode and decode at this size: 1 024 tokens, 32x32 latent): the REFERENCE MAGVITv2
    (models/modeling_magvitv2.py:416-433) on one image -> all 1 024 ids, the full latent, every 8th pixel of decode_code(ids) and one
    dense 64x64 crop -> tests/golden/magvit_512.npz.  Input regenerated by the test from RandomState(33)."""
    print("[magvit 512x512]")
    sd_np = Wt.make_magvit_state(seed=21)
    ref = R.build_reference_magvit()
    ref.load_state_dict(O.to_torch(sd_np), strict=True)
    x = torch.from_numpy(np.random.RandomState(33).uniform(-1, 1, size=(1, 3, 512, 512)).astype(np.float32))
    with torch.no_grad():
        z = ref.encoder(x)
        ids = ref.get_code(x)
        img = ref.decode_code(ids)
    assert tuple(ids.shape) == (1, 1024) and tuple(img.shape) == (1, 3, 512, 512)
    sd = O.to_torch(sd_np)
    o_ids, o_z = O.magvit_get_code(sd, x, return_z=True)
    report("encoder z @512", o_z, z)
    assert torch.equal(o_ids, ids)
    assert report("decode_code @512", O.magvit_decode_code(sd, ids), img) < 1e-3
    np.savez_compressed(os.path.join(GOLD, "magvit_512.npz"), seed=21, x_seed=33, z=z.numpy(), ids=ids.numpy(),
                        image_s8=img[:, :, ::8, ::8].contiguous().numpy(), image_crop=img[:, :, 224:288, 192:256].contiguous().numpy(),
                        image_absmax=float(img.abs().max()), image_rms=float(img.pow(2).mean().sqrt()),
                        z_absmin=float(z.abs().min()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also make the model-scale fixtures (1.45B logits [2,387], cfg3 [8,1155], cfg4 631-embedding decode, MAGVIT 512x512)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_grad_enabled(True)
    if a.only in ("", "tiny"):
        make_tiny_showo()
    if a.only in ("", "magvit"):
        make_magvit()
    if a.only in ("", "image"):
        make_image_ops()
    if a.only in ("", "clip"):
        make_clip()
    if a.only in ("", "prompting"):
        make_prompting()
    if a.only in ("", "inpaint"):
        make_tiny_inpaint()
    if a.only in ("", "magvit256"):
        make_magvit_256()
    if a.only in ("", "small_train"):
        make_small_train()
    if a.full or a.only == "full":
        make_full_showo()
    if a.full or a.only == "magvit512":
        make_magvit_512()
    if a.full or a.only == "full_cfg3":
        make_full_cfg3()
    if a.full or a.only == "full_cfg4":
        make_full_cfg4()
    print("done")
