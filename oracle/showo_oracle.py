"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the Show-o hot path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the
product package (`show-o_amd/`) never does.  Everything here is plain fp32 PyTorch/numpy on the CPU,
written from the behaviour of the reference (file:line citations are relative to /root/reference) and
pinned against the *real* reference executed in the build container (oracle/make_golden.py →
tests/golden/*.npz; tests/test_oracle_vs_golden.py).  The reference ships no tests / golden vectors of
its own, so "pinned" here means: pinned to outputs of the reference code itself on seeded
synthetic weights (oracle/weights.py).

All functions take a flat state dict `sd` with the reference's state-dict keys (torch fp32 tensors).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

NEG_MASK = float(torch.iinfo(torch.int64).min)  # -9.2234e18: the reference's "masked" value
                                                 # (training/prompting_utils.py:505-509)


# ----------------------------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim (models/phi.py:744, 264-271)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_new(x):
    """ACT2FN['gelu_new'] used by PhiMLP (models/phi.py:204-212)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def rope_tables(rotary_dim, max_pos, theta):
    """PhiRotaryEmbedding cache: emb = cat(freqs, freqs) (models/phi.py:86-102)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, rotary_dim, 2, dtype=torch.int64).float() / rotary_dim))
    t = torch.arange(max_pos, dtype=torch.int64).float()
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def apply_partial_rope(x, cos, sin, rotary_dim):
    """x: [B,H,L,Dh]; rotate dims [0,rotary_dim) pairing i with i+rotary_dim/2 (models/phi.py:163-196, 681-694)."""
    L = x.shape[2]
    xr, xp = x[..., :rotary_dim], x[..., rotary_dim:]
    half = rotary_dim // 2
    rot = torch.cat((-xr[..., half:], xr[..., :half]), dim=-1)
    c = cos[:L][None, None]
    s = sin[:L][None, None]
    return torch.cat((xr * c + rot * s, xp), dim=-1)


# ----------------------------------------------------------------------------------------------
# Phi transformer forward (models/phi.py:655-729, 774-790, 953-1081, 1169-1183)
# ----------------------------------------------------------------------------------------------
def bf16r(t):
    """round-to-nearest-even to bfloat16 and back to fp32: the value a bf16 GEMM / attention operand carries"""
    return t.to(torch.bfloat16).to(torch.float32)


class Bf16Points:
    """Rounding points of the HIP path: WHERE the MI355X kernels round to bf16, applied to this fp32
    restatement so that the remaining difference to the GPU is accumulation order + fast-math only (the fp32 reference
    comparison measures operand-rounding noise, this mode separates kernel error from it):
      * GEMM weights (q/k/v/dense/fc1/fc2/lm_head) are bf16; biases, LayerNorm parameters and the embedding table fp32;
      * h = bf16(LayerNorm(x)); the residual stream x stays fp32;
      * q, k: bias + LayerNorm(64) + RoPE on the fp32 accumulators, q scaled by 1/8 (exact), then bf16; v = bf16(acc + bias).
        `qkv_round=True` (rows < 256: the projection writes a bf16 [T,3H] buffer first) rounds q|k|v once more before that;
      * softmax: P = bf16(exp(s - rowmax)) multiplies bf16 V with fp32 accumulation, the denominator sums the UNROUNDED exp;
        o = bf16(acc / l);
      * ffn = bf16(gelu_new(acc + b1)); x += [o | ffn] [Wd | W2]^T + (bd + b2) in fp32;
      * hf = bf16(final LayerNorm(x)); logits fp32."""

    SITES = ("w", "w_lm", "h", "q", "k", "v", "p", "o", "gelu", "hf")

    def __init__(self, qkv_round=False, attn_tiles=False, dtype=torch.bfloat16, sites=None):
        self.qkv_round = qkv_round
        # attn_tiles: model the rounding SCALE of P in the LDS-tiled attention kernel as well (attention_lds_model below); without it
        # P is rounded at the row's final maximum, which the kernel only does for rows whose maximum sits in their first key tile
        self.attn_tiles = attn_tiles
        # dtype: the 16-bit operand type of the kernels -- torch.bfloat16 (Showo.set_precision(0)) or torch.float16 (set_precision(2):
        # same MFMA rate, 3 more mantissa bits, saturating converts).  sites: which rounding points are active (None = all of SITES);
        # a site left out keeps its fp32 value -- the model of an operand carried as a (hi, lo) pair, and the per-site error budget
        # of oracle/predict_rounding.py
        self.dtype = dtype
        self.sites = set(self.SITES if sites is None else sites)
        self._w = {}

    def r(self, site, t):
        if site not in self.sites:
            return t
        if self.dtype == torch.float16:
            t = t.clamp(-65504.0, 65504.0)  # the kernels' converts saturate (csrc/common.h Op16<true>)
        return t.to(self.dtype).to(torch.float32)

    def w(self, sd, key):
        if key not in self._w:
            self._w[key] = self.r("w_lm" if key.endswith("lm_head.weight") else "w", sd[key])
        return self._w[key]


AT_DEFER = 8.0  # csrc/attention.hip: deferred-rescale threshold of the running maximum (natural-log units)
_LOG2E = 1.4426950408889634


def attention_lds_model(q, k, v, vis, rp=bf16r, ro=bf16r):
    """The arithmetic of csrc/attention.hip::attn_lds_body with its rounding points, restated for the per-block parity gate.
    q (pre-scaled by 1/8), k, v: bf16-rounded fp32 [B,H,L,64]; vis: bool [B,1|H,L,L] (True = key visible).
    One wave owns 32 consecutive query rows and walks the keys in 32-key sub-tiles (multiples of 32 inside the wave's key hull); the
    running maximum m of a row is advanced -- for EVERY row of the wave -- only when some row's sub-tile maximum exceeds its m by more
    than AT_DEFER; P = bf16(exp2((s - m) log2 e)) multiplies bf16 V with fp32 accumulation, the denominator sums the unrounded P, both
    are rescaled in fp32 when m moves; o = bf16(acc * (1 / l)).  The SCALE at which P is rounded (m at that sub-tile, not the row's
    final maximum) is what this adds over rounding exp(s - rowmax): with it the remaining GPU difference is accumulation order only."""
    B, H, L, D = q.shape
    s_all = q @ k.transpose(2, 3)
    vis = vis.expand(B, H, L, L) if vis.shape[1] == 1 else vis
    out = torch.empty_like(q)
    for g0 in range(0, L, 32):
        g1 = min(g0 + 32, L)
        vg = vis[:, :, g0:g1]                                   # [B,H,R,L]
        sg = torch.where(vg, s_all[:, :, g0:g1], torch.full((), float("-inf")))
        anyk = vg.any(dim=2)                                    # [B,H,L]: key visible to some row of the wave
        m = torch.full((B, H, g1 - g0), float("-inf"))
        l = torch.zeros((B, H, g1 - g0))
        acc = torch.zeros((B, H, g1 - g0, D))
        for ks in range(0, L, 32):
            ke = min(ks + 32, L)
            # hull skip is per (b, head) wave: sub-tiles outside [wmin, wmax) are not visited; inside the hull a sub-tile with no
            # visible key contributes p = 0 and never moves m, so visiting it is equivalent -- except through the `any` rule, which
            # such a sub-tile cannot trigger (its maxima are -inf).  Hence no explicit hull logic is needed here.
            sv = sg[:, :, :, ks:ke]
            mx = sv.max(dim=-1).values                          # [B,H,R]
            upd = (mx > m + AT_DEFER).any(dim=-1, keepdim=True)  # wave-wide vote [B,H,1]
            m_new = torch.where(upd, torch.maximum(m, mx), m)
            alpha = torch.where(torch.isinf(m_new) | (m_new == m), torch.ones(()), torch.exp2((m - torch.where(torch.isinf(m_new), torch.zeros(()), m_new)) * _LOG2E))
            alpha = torch.where(torch.isinf(m) & ~torch.isinf(m_new), torch.zeros(()), alpha)
            l = l * alpha
            acc = acc * alpha.unsqueeze(-1)
            m = m_new
            mb = torch.where(torch.isinf(m), torch.zeros(()), m) * _LOG2E
            pr = torch.exp2(sv * _LOG2E - mb.unsqueeze(-1))    # exp2(fma(s, log2 e, -m log2 e)); masked keys: exp2(-inf) = 0
            l = l + pr.sum(dim=-1)
            acc = acc + rp(pr) @ v[:, :, ks:ke]
        out[:, :, g0:g1] = ro(acc * (1.0 / l).unsqueeze(-1))
    return out


def phi_attention(sd, p, d, h, mask, cos, sin, pts=None):
    B, L, Hd = h.shape
    W = (lambda k: pts.w(sd, k)) if pts is not None else (lambda k: sd[k])
    q = h @ W(p + "q_proj.weight").T + sd[p + "q_proj.bias"]
    k = h @ W(p + "k_proj.weight").T + sd[p + "k_proj.bias"]
    v = h @ W(p + "v_proj.weight").T + sd[p + "v_proj.bias"]
    if pts is not None and pts.qkv_round:
        q, k, v = pts.r("q", q), pts.r("k", k), pts.r("v", v)
    q = q.view(B, L, d.heads, d.head_dim).transpose(1, 2)
    k = k.view(B, L, d.heads, d.head_dim).transpose(1, 2)
    v = v.view(B, L, d.heads, d.head_dim).transpose(1, 2)
    # per-head QK LayerNorm, forced on by PhiForCausalLM (models/phi.py:1088, 665-667)
    q = layer_norm(q, sd[p + "q_layernorm.weight"], sd[p + "q_layernorm.bias"], d.ln_eps)
    k = layer_norm(k, sd[p + "k_layernorm.weight"], sd[p + "k_layernorm.bias"], d.ln_eps)
    q = apply_partial_rope(q, cos, sin, d.rotary_dim)
    k = apply_partial_rope(k, cos, sin, d.rotary_dim)
    if pts is not None:
        q, k, v = pts.r("q", q / math.sqrt(d.head_dim)), pts.r("k", k), pts.r("v", v)
        s = q @ k.transpose(2, 3)
    else:
        s = (q @ k.transpose(2, 3)) / math.sqrt(d.head_dim)
    if mask is not None:
        s = s + mask
    else:
        # SDPA is_causal path when no mask is given (models/phi.py:713)
        causal = torch.tril(torch.ones(L, L, dtype=torch.bool))
        s = s.masked_fill(~causal, float("-inf"))
    if pts is not None and pts.attn_tiles:
        vis = (mask == 0) if mask is not None else torch.tril(torch.ones(L, L, dtype=torch.bool)).reshape(1, 1, L, L)
        return attention_lds_model(q, k, v, vis, lambda t: pts.r("p", t), lambda t: pts.r("o", t)).transpose(1, 2).reshape(B, L, Hd)
    if pts is not None:
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        o = pts.r("o", (pts.r("p", e) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, L, Hd)
        return o  # the dense projection joins fc2 in ONE K-concatenated GEMM (phi_hidden)
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, Hd)
    return o @ sd[p + "dense.weight"].T + sd[p + "dense.bias"]


def phi_layer(sd, d, i, x, mask, cos, sin, pts=None):
    """one PhiDecoderLayer (models/phi.py:774-790): x + Attn(LN(x)) + MLP(LN(x)); pts = round where the HIP path rounds"""
    p = f"showo.model.layers.{i}."
    h = layer_norm(x, sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"], d.ln_eps)
    if pts is not None:
        h = pts.r("h", h)
        o = phi_attention(sd, p + "self_attn.", d, h, mask, cos, sin, pts)
        m = pts.r("gelu", gelu_new(h @ pts.w(sd, p + "mlp.fc1.weight").T + sd[p + "mlp.fc1.bias"]))
        return x + (o @ pts.w(sd, p + "self_attn.dense.weight").T + m @ pts.w(sd, p + "mlp.fc2.weight").T
                    + (sd[p + "self_attn.dense.bias"] + sd[p + "mlp.fc2.bias"]))
    a = phi_attention(sd, p + "self_attn.", d, h, mask, cos, sin)
    m = gelu_new(h @ sd[p + "mlp.fc1.weight"].T + sd[p + "mlp.fc1.bias"])
    m = m @ sd[p + "mlp.fc2.weight"].T + sd[p + "mlp.fc2.bias"]
    return a + m + x  # parallel residual (models/phi.py:790)


def phi_head(sd, d, x, pts=None):
    """final LayerNorm + biased lm_head + .float() (models/phi.py:1078, 1182-1183)"""
    hid = layer_norm(x, sd["showo.model.final_layernorm.weight"], sd["showo.model.final_layernorm.bias"], d.ln_eps)
    if pts is not None:
        return (pts.r("hf", hid) @ pts.w(sd, "showo.lm_head.weight").T + sd["showo.lm_head.bias"]).float()
    return (hid @ sd["showo.lm_head.weight"].T + sd["showo.lm_head.bias"]).float()


def phi_hidden(sd, d, input_ids=None, inputs_embeds=None, attention_mask=None, collect=None, pts=None):
    """Returns final-LayerNorm'ed hidden states [B,L,H].  pts: a Bf16Points object = round where the HIP path rounds."""
    if inputs_embeds is None:
        x = sd["showo.model.embed_tokens.weight"][input_ids]
    else:
        x = inputs_embeds
    cos, sin = rope_tables(d.rotary_dim, d.max_pos, d.rope_theta)
    mask = None if attention_mask is None else attention_mask.to(torch.float32)
    for i in range(d.layers):
        x = phi_layer(sd, d, i, x, mask, cos, sin, pts)
        if collect is not None:
            collect.append(x)
    hid = layer_norm(x, sd["showo.model.final_layernorm.weight"], sd["showo.model.final_layernorm.bias"], d.ln_eps)
    return pts.r("hf", hid) if pts is not None else hid


def showo_logits(sd, d, input_ids=None, input_embeddings=None, attention_mask=None, pts=None):
    """pts = Bf16Points(...): the same forward with the HIP path's bf16 rounding points (test_modules_gpu.py gates the GPU logits
    against it at north_star's 1e-3; the plain fp32 call stays the reference-parity number)"""
    hid = phi_hidden(sd, d, input_ids, input_embeddings, attention_mask, pts=pts)
    wlm = pts.w(sd, "showo.lm_head.weight") if pts is not None else sd["showo.lm_head.weight"]
    return (hid @ wlm.T + sd["showo.lm_head.bias"]).float()


def cross_entropy(logits, labels, ignore_index=-100):
    """F.cross_entropy(mean, ignore_index) on [T,V] / [T] (models/modeling_showo.py:83-98)."""
    lse = torch.logsumexp(logits, dim=-1)
    valid = labels != ignore_index
    safe = labels.clamp(min=0)
    nll = lse - logits.gather(1, safe[:, None])[:, 0]
    return (nll * valid).sum() / valid.sum()


def showo_forward(sd, d, input_ids, input_embeddings=None, attention_mask=None, labels=None,
                  batch_size_t2i=0, batch_size_lm=0, batch_size_mmu=0, max_seq_length=128):
    """Showo.forward (models/modeling_showo.py:59-102), including the `[-0:]` quirk for batch_size_mmu=0."""
    logits = showo_logits(sd, d, input_ids, input_embeddings, attention_mask)
    if labels is None:
        return logits
    V = d.vocab
    l_t2i = cross_entropy(logits[:batch_size_t2i, max_seq_length + 1:].reshape(-1, V),
                          labels[:batch_size_t2i, max_seq_length + 1:].reshape(-1))
    l_lm = cross_entropy(logits[batch_size_t2i:batch_size_t2i + batch_size_lm, :-1].reshape(-1, V),
                         labels[batch_size_t2i:batch_size_t2i + batch_size_lm, 1:].reshape(-1))
    l_mmu = cross_entropy(logits[-batch_size_mmu:, :-1].reshape(-1, V), labels[-batch_size_mmu:, 1:].reshape(-1))
    return logits, l_t2i, l_lm, l_mmu


# ----------------------------------------------------------------------------------------------
# masks (training/prompting_utils.py:466-511, 591-624)
# ----------------------------------------------------------------------------------------------
def _invert(bool_mask):
    out = torch.zeros(bool_mask.shape, dtype=torch.float32)
    out[~bool_mask] = NEG_MASK
    return out


def mask_t2i(sequence, pad_id, soi_id, eoi_id, rm_pad_in_image=True):
    """create_attention_mask_predict_next → additive float mask [N,1,L,L] (prompting_utils.py:466-511)."""
    N, L = sequence.shape
    vis = torch.zeros(N, L, L, dtype=torch.bool)
    for b in range(N):
        seq = sequence[b].tolist()
        started = ended = 0
        in_img = []
        for t in seq:
            is_s, is_e = t == soi_id, t == eoi_id
            started += is_s
            ended += is_e
            in_img.append(started > ended or is_s or is_e)
        pads = [i for i, t in enumerate(seq) if t == pad_id]
        last_pad = pads[-1] if pads else -1
        sois = [i for i, t in enumerate(seq) if t == soi_id]
        for r in range(L):
            if in_img[r]:
                row = [True] * L  # image rows see every text/image position (is_text_image is all-ones)
                if rm_pad_in_image and sois and r >= sois[0]:
                    for c in pads:
                        row[c] = False
            else:
                row = [c <= r for c in range(L)]
                if rm_pad_in_image and pads and r > last_pad:
                    for c in range(last_pad + 1):
                        row[c] = False
            vis[b, r] = torch.tensor(row)
    return _invert(vis).unsqueeze(1)


def mask_mmu(sequence, eoi_id):
    """create_attention_mask_for_mmu (prompting_utils.py:591-604): causal + first sample's image prefix fully visible."""
    N, L = sequence.shape
    vis = torch.tril(torch.ones(L, L, dtype=torch.bool))[None].repeat(N, 1, 1)
    e = int((sequence == eoi_id).nonzero()[0, 1])  # eoi column of the first hit (row-major order)
    vis[:, :, : e + 1] = True
    return _invert(vis).unsqueeze(1)


def mask_mmu_vit(N, L, system_prompt_len=0):
    """create_attention_mask_for_mmu_vit (prompting_utils.py:606-624)."""
    vis = torch.tril(torch.ones(L, L, dtype=torch.bool))[None].repeat(N, 1, 1)
    vis[:, :, 1 + system_prompt_len + 1: 1 + system_prompt_len + 1 + 576] = True
    return _invert(vis).unsqueeze(1)


# ----------------------------------------------------------------------------------------------
# sampling (models/sampling.py:10-40; models/modeling_showo.py:104-181)
# ----------------------------------------------------------------------------------------------
def cosine_schedule(t):
    return torch.cos(t * math.pi * 0.5)


def t2i_step_constants(timesteps, num_vq_tokens, temperature=1.0, noise_schedule=cosine_schedule):
    """Host-side per-step scalars exactly as the reference derives them (modeling_showo.py:157-173):
    floor(N * schedule((k+1)/T)) in fp32 and the *compounding* temperature."""
    mask_len, temps = [], []
    temp = temperature
    for step in range(timesteps):
        ratio = 1.0 * (step + 1) / timesteps
        mask_ratio = noise_schedule(torch.tensor(ratio))
        mask_len.append(float((num_vq_tokens * mask_ratio).floor()))
        temp = temp * (1.0 - ratio)
        temps.append(temp)
    return mask_len, temps


class TorchNoise:
    """Noise source that reproduces the reference's calls on a torch.Generator:
    multinomial(probs,1) and uniform_(0,1) for the Gumbel noise."""

    def __init__(self, generator=None):
        self.g = generator

    def multinomial(self, probs2d):
        return torch.multinomial(probs2d, 1, generator=self.g)[:, 0]

    def uniform(self, like):
        return torch.zeros_like(like).uniform_(0, 1, generator=self.g)


class RecordedNoise:
    """Noise-injection source: multinomial is realised as argmax(p / E) with E~Exp(1)
    (the algorithm torch.multinomial uses for one draw: q.exponential_(1); argmax(p/q)), and Gumbel from U."""

    def __init__(self, exp_noise_per_step, uniform_per_step):
        self.e, self.u, self.i, self.j = exp_noise_per_step, uniform_per_step, 0, 0

    def multinomial(self, probs2d):
        e = self.e[self.i]
        self.i += 1
        return torch.argmax(probs2d / e.reshape(probs2d.shape), dim=-1)

    def uniform(self, like):
        u = self.u[self.j]
        self.j += 1
        return u.reshape(like.shape)


# MLM corruption of the image tokens of a training batch (training/utils.py:77-154, random-permutation branch)
def mask_tokens_np(tokens, noise, mask_prob, mask_id, predict_all=False, ignore_id=-100):
    """tokens int64 [B,N]; noise fp32 [B,N] (= the reference's torch.rand(B, N)); mask_prob fp32 [B] (after the schedule and
    the min_masking_rate clip).  num_masked = round_half_even(N * p) clamped to >= 1 (:92); perm = argsort(noise) (:101);
    mask = perm < num_masked (:102); input = mask_id where masked (:133); labels = token where masked else -100 (:150)."""
    tokens = np.asarray(tokens)
    B, N = tokens.shape
    num = np.maximum(np.rint(np.float32(N) * np.asarray(mask_prob, np.float32)), 1).astype(np.int64)
    perm = np.argsort(np.asarray(noise, np.float32), axis=-1, kind="stable")
    mask = perm < num[:, None]
    inp = np.where(mask, mask_id, tokens)
    lab = tokens.copy() if predict_all else np.where(mask, tokens, ignore_id)
    return inp, lab, mask


def loss_weight_np(t, mask, min_val=0.3):
    """get_loss_weight (training/utils.py:73-74)"""
    t = np.asarray(t, np.float32)
    return 1 - (1 - mask.astype(np.int64)) * ((1 - t) * np.float32(1 - min_val))[:, None]


def mask_by_random_topk(mask_len, probs, temperature, uniform):
    """models/sampling.py:31-36 with the uniform draw made explicit."""
    g = -torch.log((-torch.log(uniform.clamp(min=1e-20))).clamp(min=1e-20))
    confidence = torch.log(probs.clamp(min=1e-20)) + temperature * g
    sorted_confidence = torch.sort(confidence, dim=-1).values
    cut_off = torch.gather(sorted_confidence, 1, mask_len.long())
    return confidence < cut_off


def t2i_generate(sd, d, input_ids, uncond_input_ids=None, attention_mask=None, temperature=1.0, timesteps=18,
                 guidance_scale=0.0, noise=None, noise_schedule=cosine_schedule, trace=None):
    """Showo.t2i_generate (models/modeling_showo.py:104-181).  Mutates input_ids in place like the reference."""
    noise = noise or TorchNoise()
    N = d.num_vq_tokens
    off = d.image_offset
    mask_id = d.mask_token_id
    cur = input_ids[:, -(N + 1):-1].clone()
    cur = torch.where(cur == mask_id, mask_id, cur - off)
    if uncond_input_ids is not None:
        uncond_prefix = uncond_input_ids[:, : d.max_text_len + 1]
    sampled_ids = None
    for step in range(timesteps):
        if uncond_input_ids is not None and guidance_scale > 0:
            uncond_input_ids = torch.cat([uncond_prefix, input_ids[:, d.max_text_len + 1:]], dim=1)
            model_input = torch.cat([input_ids, uncond_input_ids])
            lg = showo_logits(sd, d, model_input, attention_mask=attention_mask)
            cond, uncond = lg.chunk(2)
            lg = (1 + guidance_scale) * cond - guidance_scale * uncond
        else:
            lg = showo_logits(sd, d, input_ids, attention_mask=attention_mask)
        lg = lg[:, -(N + 1):-1, off:-1]
        if trace is not None:
            trace.append(dict(input_ids=input_ids.clone(), logits=lg.clone()))
        probs = lg.softmax(dim=-1)
        sampled_ids = noise.multinomial(probs.reshape(-1, lg.size(-1))).view(*lg.shape[:-1])
        unknown = cur == mask_id
        sampled_ids = torch.where(unknown, sampled_ids, cur)
        ratio = 1.0 * (step + 1) / timesteps
        mask_ratio = noise_schedule(torch.tensor(ratio))
        sel = torch.gather(probs, -1, sampled_ids.long()[..., None]).squeeze(-1)
        sel = torch.where(unknown, sel, torch.finfo(sel.dtype).max)
        mask_len = (N * mask_ratio).floor().unsqueeze(0)
        mask_len = torch.max(torch.tensor([1]), torch.min(unknown.sum(dim=-1, keepdim=True) - 1, mask_len))
        temperature = temperature * (1.0 - ratio)
        masking = mask_by_random_topk(mask_len, sel, temperature, noise.uniform(sel))
        input_ids[:, -(N + 1):-1] = torch.where(masking, mask_id, sampled_ids + off)
        cur = torch.where(masking, mask_id, sampled_ids)
        if trace is not None:
            trace[-1].update(sampled_ids=sampled_ids.clone(), masking=masking.clone())
    return sampled_ids


def mmu_generate(sd, d, idx=None, input_embeddings=None, attention_mask=None, max_new_tokens=100,
                 temperature=1.0, top_k=None, eot_token=None, noise=None):
    """Showo.mmu_generate (models/modeling_showo.py:183-240): batch-1, no KV cache, mask grown by a row/col."""
    noise = noise or TorchNoise()
    result = []
    for _ in range(max_new_tokens):
        logits = showo_logits(sd, d, idx, input_embeddings, attention_mask)
        L = attention_mask.shape[-1]
        am = attention_mask.reshape(L, L)
        a = torch.cat([am, torch.full((L, 1), torch.finfo(logits.dtype).min)], dim=1)
        new_row = torch.cat([am[-1, :], torch.zeros(1)])[None]
        attention_mask = torch.cat([a, new_row], dim=0)
        lg = logits[:, -1, :] / temperature
        if top_k is not None:
            v, _ = torch.topk(lg, min(top_k, lg.size(-1)))
            lg = lg.masked_fill(lg < v[:, [-1]], float("-inf"))
        probs = F.softmax(lg, dim=-1)
        nxt = noise.multinomial(probs)[:, None]
        result.append(nxt[0][0])
        if input_embeddings is not None:
            input_embeddings = torch.cat([input_embeddings, sd["showo.model.embed_tokens.weight"][nxt]], dim=1)
        else:
            idx = torch.cat((idx, nxt), dim=1)
        if eot_token is not None and int(nxt) == eot_token:
            break
    return result


# ----------------------------------------------------------------------------------------------
# MAGVIT-v2 (models/modeling_magvitv2.py, models/common_modules.py)
# ----------------------------------------------------------------------------------------------
def lfq_pack_np(z):
    """LFQuantizer sign-pack (modeling_magvitv2.py:201-206, 239-241): z [B,C,h,w] float -> ids [B,h*w] int64.
    bit for channel c (MSB first) is [z_c > 0]; NaN and -0.0/+0.0 give 0 (IEEE `>`)."""
    z = np.asarray(z)
    B, C = z.shape[0], z.shape[1]
    bits = (z > 0).reshape(B, C, -1).astype(np.int64)
    w = (1 << np.arange(C - 1, -1, -1, dtype=np.int64))[None, :, None]
    return (bits * w).sum(axis=1)


def lfq_unpack_np(ids, C=13, shape=None):
    """get_codebook_entry (modeling_magvitv2.py:208-221): ids [B,n] -> ±1 float32 [B,C,h,w]."""
    ids = np.asarray(ids, dtype=np.int64)
    B, n = ids.shape
    h, w = shape if shape is not None else (int(math.sqrt(n)), int(math.sqrt(n)))
    bits = (ids[:, :, None] >> np.arange(C - 1, -1, -1, dtype=np.int64)[None, None, :]) & 1
    zq = bits.astype(np.float32) * 2 - 1
    return np.ascontiguousarray(zq.reshape(B, h, w, C).transpose(0, 3, 1, 2))


def group_norm(x, w, b, groups=32, eps=1e-6):
    """Normalize = GroupNorm(32, eps 1e-6, affine) (common_modules.py:21-24)."""
    B, C, H, W = x.shape
    g = x.reshape(B, groups, -1)
    mu = g.mean(dim=-1, keepdim=True)
    var = ((g - mu) ** 2).mean(dim=-1, keepdim=True)
    g = (g - mu) / torch.sqrt(var + eps)
    return g.reshape(B, C, H, W) * w[None, :, None, None] + b[None, :, None, None]


def swish(x):
    return x * torch.sigmoid(x)


def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, p, x):
    """ResnetBlock.forward with temb=None, dropout 0 (common_modules.py:337-357)."""
    h = swish(group_norm(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]))
    h = _conv(sd, p + ".conv1", h, padding=1)
    h = swish(group_norm(h, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]))
    h = _conv(sd, p + ".conv2", h, padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd, p, x):
    """AttnBlock.forward: single-head attention over h*w with scale C^-0.5 (common_modules.py:187-211)."""
    B, C, H, W = x.shape
    h = group_norm(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"])
    q = _conv(sd, p + ".q", h).reshape(B, C, H * W).permute(0, 2, 1)
    k = _conv(sd, p + ".k", h).reshape(B, C, H * W)
    v = _conv(sd, p + ".v", h).reshape(B, C, H * W)
    w_ = torch.softmax((q @ k) * (int(C) ** -0.5), dim=2)
    o = (v @ w_.permute(0, 2, 1)).reshape(B, C, H, W)
    return x + _conv(sd, p + ".proj_out", o)


def _levels(sd, prefix):
    lv = {}
    for k in sd:
        if k.startswith(prefix):
            parts = k[len(prefix):].split(".")
            if parts[1] == "block":
                lv.setdefault(int(parts[0]), set()).add(int(parts[2]))
    return {l: len(s) for l, s in lv.items()}


def magvit_encoder(sd, x):
    """VQGANEncoder.forward (modeling_magvitv2.py:143-169)."""
    lv = _levels(sd, "encoder.down.")
    h = _conv(sd, "encoder.conv_in", x, padding=1)
    for l in range(len(lv)):
        for j in range(lv[l]):
            h = resnet_block(sd, f"encoder.down.{l}.block.{j}", h)
        if f"encoder.down.{l}.downsample.conv.weight" in sd:
            h = F.pad(h, (0, 1, 0, 1))  # asymmetric pad then stride-2 (common_modules.py:83-88)
            h = _conv(sd, f"encoder.down.{l}.downsample.conv", h, stride=2)
    h = resnet_block(sd, "encoder.mid.block_1", h)
    h = attn_block(sd, "encoder.mid.attn_1", h)
    h = resnet_block(sd, "encoder.mid.block_2", h)
    h = swish(group_norm(h, sd["encoder.norm_out.weight"], sd["encoder.norm_out.bias"]))
    h = _conv(sd, "encoder.conv_out", h, padding=1)
    return _conv(sd, "encoder.quant_conv", h)


def magvit_decoder(sd, z):
    """VQGANDecoder.forward (modeling_magvitv2.py:365-399)."""
    lv = _levels(sd, "decoder.up.")
    h = _conv(sd, "decoder.post_quant_conv", z)
    h = _conv(sd, "decoder.conv_in", h, padding=1)
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    for l in reversed(range(len(lv))):
        for j in range(lv[l]):
            h = resnet_block(sd, f"decoder.up.{l}.block.{j}", h)
        if l != 0:
            h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)  # nearest 2x (common_modules.py:36-40)
            h = _conv(sd, f"decoder.up.{l}.upsample.conv", h, padding=1)
    h = swish(group_norm(h, sd["decoder.norm_out.weight"], sd["decoder.norm_out.bias"]))
    return _conv(sd, "decoder.conv_out", h, padding=1)


def magvit_get_code(sd, pixel_values, return_z=False):
    """MAGVITv2.get_code (modeling_magvitv2.py:423-427)."""
    z = magvit_encoder(sd, pixel_values)
    ids = torch.from_numpy(lfq_pack_np(z.numpy()))
    return (ids, z) if return_z else ids


def magvit_decode_code(sd, ids, shape=None):
    """MAGVITv2.decode_code (modeling_magvitv2.py:429-433)."""
    C = sd["quantize.embedding"].shape[1]
    zq = torch.from_numpy(lfq_unpack_np(ids.numpy(), C=C, shape=shape))
    return magvit_decoder(sd, zq)


def to_torch(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# ----------------------------------------------------------------------------------------------
# CLIP ViT vision tower + mm_projector (models/clip_encoder.py:29-49 -> transformers CLIPVisionModel; modeling_showo.py:48-53)
# The arithmetic is third-party (transformers, pinned 4.41.1 by requirements.txt:203; 5.15 installed here): this restates
# CLIPVisionEmbeddings / CLIPEncoderLayer / CLIPAttention / CLIPMLP and is asserted against the installed class in
# oracle/make_golden.py.
# ----------------------------------------------------------------------------------------------
def clip_vision_features(sd, cfg, images, select_layer=-2):
    """sd: {vision_model.* key: tensor}; images fp32 [B,3,S,S] -> hidden_states[select_layer][:, 1:]  [B, P, H]"""
    H, nH, ps, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["patch_size"], cfg["layer_norm_eps"]
    g = lambda k: sd["vision_model." + k]
    B = images.shape[0]
    patches = F.conv2d(images, g("embeddings.patch_embedding.weight"), stride=ps).flatten(2).transpose(1, 2)  # [B,P,H]
    x = torch.cat([g("embeddings.class_embedding").expand(B, 1, H), patches], dim=1) + g("embeddings.position_embedding.weight")[None]
    x = F.layer_norm(x, (H,), g("pre_layrnorm.weight"), g("pre_layrnorm.bias"), eps)
    n_run = cfg["num_hidden_layers"] + 1 + select_layer
    L = x.shape[1]
    for i in range(n_run):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (H,), g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), eps)
        q = F.linear(h, g(p + "self_attn.q_proj.weight"), g(p + "self_attn.q_proj.bias")) * (H // nH) ** -0.5
        k = F.linear(h, g(p + "self_attn.k_proj.weight"), g(p + "self_attn.k_proj.bias"))
        v = F.linear(h, g(p + "self_attn.v_proj.weight"), g(p + "self_attn.v_proj.bias"))
        sp = lambda t: t.view(B, L, nH, H // nH).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(2, 3), dim=-1) @ sp(v)
        a = a.transpose(1, 2).reshape(B, L, H)
        x = x + F.linear(a, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (H,), g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), eps)
        f = F.linear(h, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))
        f = f * torch.sigmoid(1.702 * f)  # quick_gelu
        x = x + F.linear(f, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    return x[:, 1:]


def mm_projector(sd, x):
    """nn.Sequential(Linear, GELU(), Linear) (modeling_showo.py:48-53); sd keys 0.weight, 0.bias, 2.weight, 2.bias"""
    return F.linear(F.gelu(F.linear(x, sd["0.weight"], sd["0.bias"])), sd["2.weight"], sd["2.bias"])


# ----------------------------------------------------------------------------------------------
# Image pre / post-processing (training/utils.py:178-185 image_transform -> torchvision Resize(BICUBIC) on a PIL image ->
# PIL.Image.resize; third-party: Pillow src/libImaging/Resample.c, restated here and asserted against the installed Pillow)
# ----------------------------------------------------------------------------------------------
def _pil_bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc: bounds int32 [out,2] (first tap, tap count), kk int32 [out,ksize]"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)  # left-to-right double accumulation like the C loop
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_bicubic_np(img, out_w, out_h):
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C]: horizontal pass, uint8, vertical pass (ImagingResampleInner)"""
    img = np.asarray(img)
    H, W, C = img.shape
    cur = img.astype(np.int64)
    if out_w != W:
        b, kk = pil_bicubic_coeffs(W, out_w)
        nxt = np.empty((H, out_w, C), np.int64)
        for xo in range(out_w):
            x0, n = b[xo]
            acc = (1 << 21) + (cur[:, x0:x0 + n, :] * kk[xo, :n].astype(np.int64)[None, :, None]).sum(1)
            nxt[:, xo, :] = np.clip(acc >> 22, 0, 255)
        cur = nxt
    if out_h != H:
        b, kk = pil_bicubic_coeffs(H, out_h)
        nxt = np.empty((out_h, cur.shape[1], C), np.int64)
        for yo in range(out_h):
            y0, n = b[yo]
            acc = (1 << 21) + (cur[y0:y0 + n] * kk[yo, :n].astype(np.int64)[:, None, None]).sum(0)
            nxt[yo] = np.clip(acc >> 22, 0, 255)
        cur = nxt
    return cur.astype(np.uint8)


def image_transform_np(img, resolution=256, normalize=True):
    """image_transform (training/utils.py:178-185) on a uint8 [H,W,C] array: Resize(shorter side -> resolution, bicubic,
    torchvision size rule long = int(resolution * long / short)), CenterCrop, ToTensor, Normalize(0.5, 0.5) -> fp32 [C,R,R]"""
    img = np.asarray(img)
    H, W = img.shape[:2]
    if W <= H:
        ow, oh = resolution, int(resolution * H / W)
    else:
        oh, ow = resolution, int(resolution * W / H)
    r = pil_resize_bicubic_np(img, ow, oh) if (ow, oh) != (W, H) else img
    ct, cl = int(round((oh - resolution) / 2.0)), int(round((ow - resolution) / 2.0))
    r = r[ct:ct + resolution, cl:cl + resolution]
    t = torch.from_numpy(np.ascontiguousarray(r)).permute(2, 0, 1).float().div(255)
    if normalize:
        t = t.sub(0.5).div(0.5)
    return t, r


def images_to_uint8_np(x):
    """inference_t2i.py:157-159: clamp((x + 1) / 2, 0, 1) * 255 -> NHWC uint8 (truncation)"""
    y = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0) * 255.0
    return y.permute(0, 2, 3, 1).numpy().astype(np.uint8)


def inpainting_token_mask(mask01, resolution):
    """inference_t2i.py:100-108: mask fp32 [1,R,R] in [0,1] -> bool [(R/16)^2] (bicubic down-sample, >= 0.5)"""
    m = F.interpolate(mask01[None], size=resolution // 16, mode="bicubic")
    return (m >= 0.5).reshape(-1), m.reshape(-1)
