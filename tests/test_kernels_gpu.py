"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes) and checked against the
CPU oracle (oracle/showo_oracle.py) / committed golden fixtures.  Tolerances are stated per test:
integer / index work is bit-exact; bf16 MFMA paths are compared with the fp32 oracle evaluated on the SAME
bf16-rounded operands, so the bound is the accumulation-order + output-rounding error only."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import util
from util import O, dev, from_bf16_bits, to_bf16_bits, bf16_round

pytestmark = pytest.mark.gpu


def L():
    return util.lib()


def S():
    return util.lib().stream()


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------- LFQ
def test_lfq_pack_unpack_bit_exact():
    g = util.golden("lfq_kat.npz")
    z = dev(g["z"])
    B, Cc, h, w = z.shape
    ids = torch.empty((B, h * w), dtype=torch.int64, device="cuda")
    L().call("showo_lfq_pack_nchw", L().ptr(z), L().ptr(ids), B, Cc, h * w, S())
    assert np.array_equal(ids.cpu().numpy(), g["ids"])  # includes +-0, +-1e-9, denormals
    sub = dev(g["ids"][:, :16])
    back = torch.empty((B, 13, 4, 4), dtype=torch.float32, device="cuda")
    L().call("showo_lfq_unpack_nchw", L().ptr(sub), L().ptr(back), B, 13, 16, S())
    assert np.array_equal(back.cpu().numpy(), g["back"])
    # full-size random: 8 images x 1024 tokens, NCHW and NHWC forms, vs the numpy oracle
    rs = np.random.RandomState(0)
    zz = rs.standard_normal((8, 13, 32, 32)).astype(np.float32)
    zz[rs.rand(*zz.shape) < 0.01] = 0.0
    ids = torch.empty((8, 1024), dtype=torch.int64, device="cuda")
    zd = dev(zz)
    L().call("showo_lfq_pack_nchw", L().ptr(zd), L().ptr(ids), 8, 13, 1024, S())
    want = O.lfq_pack_np(zz)
    assert np.array_equal(ids.cpu().numpy(), want)
    znhwc = zd.permute(0, 2, 3, 1).contiguous()
    ids2 = torch.empty_like(ids)
    L().call("showo_lfq_pack_nhwc", L().ptr(znhwc), L().ptr(ids2), 8, 13, 1024, 13, S())
    assert torch.equal(ids, ids2)
    # padded rows (ldz = 16: the bench's bandwidth shape) and a token count that is not a multiple of the 256-token block, NaN in the pad
    zpad = torch.full((8 * 1024, 16), float("nan"), device="cuda")
    zpad[:, :13] = znhwc.reshape(-1, 13)
    for ntok in (8 * 1024, 8 * 1024 - 77, 3):
        ids3 = torch.full((ntok,), -1, dtype=torch.int64, device="cuda")
        L().call("showo_lfq_pack_nhwc", L().ptr(zpad), L().ptr(ids3), 1, 13, ntok, 16, S())
        assert torch.equal(ids3, ids.reshape(-1)[:ntok])
    # property at full size: unpack(pack(z)) == sign pattern, pack(unpack(id)) == id over the whole codebook
    allids = torch.arange(8192, dtype=torch.int64, device="cuda").reshape(2, 4096)
    zq = torch.empty((2, 13, 4096), dtype=torch.float32, device="cuda")
    L().call("showo_lfq_unpack_nchw", L().ptr(allids), L().ptr(zq), 2, 13, 4096, S())
    rt = torch.empty_like(allids)
    L().call("showo_lfq_pack_nchw", L().ptr(zq), L().ptr(rt), 2, 13, 4096, S())
    assert torch.equal(rt, allids)
    assert set(zq.unique().tolist()) == {-1.0, 1.0}
    zq2 = torch.empty((2, 4096, 13), dtype=torch.float32, device="cuda")
    L().call("showo_lfq_unpack_nhwc", L().ptr(allids), L().ptr(zq2), 2, 13, 4096, S())
    assert torch.equal(zq2.permute(0, 2, 1).contiguous(), zq)
    # empty input is a no-op
    L().call("showo_lfq_pack_nchw", None, None, 0, 13, 16, S())


# ------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,H", [(37, 2048), (5, 128), (3, 100)])
def test_layernorm(rows, H):
    torch.manual_seed(0)
    x = torch.randn(rows, H) * 3 + 0.5
    w, b = torch.randn(H) * 0.1 + 1, torch.randn(H) * 0.05
    y = torch.empty((rows, H), dtype=torch.int16, device="cuda")
    L().call("showo_layernorm_f32_bf16", L().ptr(dev(x)), L().ptr(dev(w)), L().ptr(dev(b)), L().ptr(y), None, rows, H, 1e-5, S())
    want = O.layer_norm(x, w, b, 1e-5)
    got = from_bf16_bits(y).cpu()
    assert (got - want).abs().max() <= 2 ** -8 * want.abs().max() + 1e-6  # one bf16 rounding of the fp32 result
    idx = torch.tensor([rows - 1, 0, rows // 2], dtype=torch.int32)
    y2 = torch.empty((3, H), dtype=torch.int16, device="cuda")
    L().call("showo_layernorm_f32_bf16", L().ptr(dev(x)), L().ptr(dev(w)), L().ptr(dev(b)), L().ptr(y2), L().ptr(dev(idx)), 3, H, 1e-5, S())
    assert torch.equal(y2.cpu(), y.cpu()[idx.long()])


# ------------------------------------------------------------------------------------------------ GEMM
def _gemm(A, W, bias, epi, resid=None, per_row=False, ldo=None):
    M, K = A.shape
    N = W.shape[0]
    Ad, Wd = dev(to_bf16_bits(A)), dev(to_bf16_bits(W))
    bd = None if bias is None else dev(bias)
    ldo = ldo or N
    f32 = epi in (2, 3)
    out = torch.full((M, ldo), float("nan") if f32 else 0, dtype=torch.float32 if f32 else torch.int16, device="cuda")
    rd = None if resid is None else dev(resid)
    L().call("showo_gemm_bf16", L().ptr(Ad), K, L().ptr(Wd), K, L().ptr(bd), int(per_row), L().ptr(out), ldo, L().ptr(rd),
             ldo if rd is not None else 0, M, N, K, epi, S())
    sync()
    return (out if f32 else from_bf16_bits(out)).cpu()[:, :N]


@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (129, 130, 64), (1, 512, 256), (774, 2048, 2048), (300, 439, 128)])
def test_gemm_epilogues(M, N, K):
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K)
    W = torch.randn(N, K) * 0.05
    bias = torch.randn(N)
    ref = bf16_round(A).double() @ bf16_round(W).double().T + bias.double()
    tol = 1e-3 * float(ref.abs().max())
    got = _gemm(A, W, bias, 2)  # fp32 out
    assert (got.double() - ref).abs().max() < tol, (got.double() - ref).abs().max()
    got = _gemm(A, W, bias, 0)  # bf16 out
    assert (got.double() - ref).abs().max() < 2 ** -8 * float(ref.abs().max()) + tol
    got = _gemm(A, W, bias, 1)  # gelu_new bf16
    want = O.gelu_new(ref.float()).double()
    assert (got.double() - want).abs().max() < 2 ** -8 * float(want.abs().max()) + tol
    resid = torch.randn(M, N)
    got = _gemm(A, W, bias, 3, resid=resid)
    assert (got.double() - (ref + resid.double())).abs().max() < tol
    if M >= 100:
        brow = torch.randn(M)
        got = _gemm(A, W, brow, 2, per_row=True)
        ref2 = bf16_round(A).double() @ bf16_round(W).double().T + brow.double()[:, None]
        assert (got.double() - ref2).abs().max() < tol


def test_gemm_inplace_residual_and_errors():
    torch.manual_seed(1)
    M, N, K = 130, 256, 128
    A, W, bias, x = torch.randn(M, K), torch.randn(N, K) * 0.05, torch.randn(N), torch.randn(M, N)
    Ad, Wd = dev(to_bf16_bits(A)), dev(to_bf16_bits(W))
    xd = dev(x)
    L().call("showo_gemm_bf16", L().ptr(Ad), K, L().ptr(Wd), K, L().ptr(dev(bias)), 0, L().ptr(xd), N, L().ptr(xd), N, M, N, K, 3, S())
    ref = bf16_round(A).double() @ bf16_round(W).double().T + bias.double() + x.double()
    assert (xd.cpu().double() - ref).abs().max() < 1e-3 * float(ref.abs().max())
    with pytest.raises(RuntimeError):  # K not a multiple of 64 is rejected, not silently mis-computed
        L().call("showo_gemm_bf16", L().ptr(Ad), K, L().ptr(Wd), K, None, 0, L().ptr(xd), N, None, 0, M, N, 100, 2, S())


# ------------------------------------------------------------------------------- qk-prep + attention
def _rope_tables(rot=32, max_pos=2048):
    return O.rope_tables(rot, max_pos, 10000.0)


def _prep(qkv, qw, qb, kw, kb, B, Lq, nH, pos0=0, Lcap=None, Lp=None, Kbuf=None, Vbuf=None):
    cos, sin = _rope_tables()
    Lcap = Lcap or Lq
    Lp = Lp or ((Lq + 63) // 64) * 64
    Q = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
    K = Kbuf if Kbuf is not None else torch.zeros((B, nH, Lcap, 64), dtype=torch.int16, device="cuda")
    Vt = Vbuf if Vbuf is not None else torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
    L().call("showo_qk_prep", L().ptr(dev(to_bf16_bits(qkv))), L().ptr(dev(qw)), L().ptr(dev(qb)), L().ptr(dev(kw)),
             L().ptr(dev(kb)), L().ptr(dev(cos)), L().ptr(dev(sin)), L().ptr(Q), L().ptr(K), L().ptr(Vt), B, Lq, nH, 32, 1e-5,
             pos0, Lcap, Lp, S())
    sync()
    return Q, K, Vt


def _prep_oracle(qkv, qw, qb, kw, kb, B, Lq, nH, pos0=0):
    x = bf16_round(qkv).view(B, Lq, 3, nH, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))  # [B,nH,L,64]
    cos, sin = _rope_tables()
    q = O.layer_norm(q, qw, qb, 1e-5)
    k = O.layer_norm(k, kw, kb, 1e-5)
    cs, sn = cos[pos0:pos0 + Lq], sin[pos0:pos0 + Lq]
    return O.apply_partial_rope(q, cs, sn, 32), O.apply_partial_rope(k, cs, sn, 32), v


@pytest.mark.parametrize("B,Lq,nH", [(2, 27, 2), (1, 387, 3), (2, 70, 1)])
def test_qk_prep(B, Lq, nH):
    torch.manual_seed(0)
    qkv = torch.randn(B * Lq, 3 * nH * 64)
    qw, qb, kw, kb = torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05
    Q, K, Vt = _prep(qkv, qw, qb, kw, kb, B, Lq, nH)
    q, k, v = _prep_oracle(qkv, qw, qb, kw, kb, B, Lq, nH)
    assert (from_bf16_bits(Q).cpu() - q * 0.125).abs().max() < 2 ** -8 * float((q * 0.125).abs().max()) + 1e-6
    assert (from_bf16_bits(K).cpu() - k).abs().max() < 2 ** -8 * float(k.abs().max()) + 1e-6
    vt = from_bf16_bits(Vt).cpu()
    assert torch.equal(vt[..., :Lq], v.transpose(2, 3).contiguous())  # pure relayout: bit-exact
    assert (vt[..., Lq:] == 0).all()  # pad keys are zero (finite) by contract


@pytest.mark.parametrize("B,Lq,nH,pos0", [(2, 27, 2, 0), (3, 387, 4, 0), (1, 300, 3, 0), (2, 5, 2, 40)])
def test_fused_qkv_projection(B, Lq, nH, pos0):
    """showo_gemm_qkv_bf16 = QKV GEMM whose epilogue does bias + q/k LayerNorm + partial RoPE + relayout.  Checked
    against the oracle evaluated on the fp32 product of the bf16-rounded operands (one output rounding: 2^-8)."""
    torch.manual_seed(B * 1000 + Lq)
    H = nH * 64
    h = torch.randn(B * Lq, H)
    W = torch.randn(3 * H, H) * 0.05
    bias = torch.randn(3 * H) * 0.1
    qw, qb, kw, kb = torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05
    cos, sin = _rope_tables()
    Lcap = pos0 + Lq + 3
    Lp = ((pos0 + Lq + 63) // 64) * 64
    Q = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
    K = torch.zeros((B, nH, Lcap, 64), dtype=torch.int16, device="cuda")
    Vt = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
    L().call("showo_gemm_qkv_bf16", L().ptr(dev(to_bf16_bits(h))), H, L().ptr(dev(to_bf16_bits(W))), H, L().ptr(dev(bias)),
             L().ptr(dev(qw)), L().ptr(dev(qb)), L().ptr(dev(kw)), L().ptr(dev(kb)), L().ptr(dev(cos)), L().ptr(dev(sin)),
             L().ptr(Q), L().ptr(K), L().ptr(Vt), B, Lq, nH, 32, 1e-5, pos0, Lcap, Lp, S())
    sync()
    qkv = (bf16_round(h).double() @ bf16_round(W).double().T + bias.double()).float()
    x = qkv.view(B, Lq, 3, nH, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    cs, sn = cos[pos0:pos0 + Lq], sin[pos0:pos0 + Lq]
    q = O.apply_partial_rope(O.layer_norm(q, qw, qb, 1e-5), cs, sn, 32) * 0.125
    k = O.apply_partial_rope(O.layer_norm(k, kw, kb, 1e-5), cs, sn, 32)
    assert (from_bf16_bits(Q).cpu() - q).abs().max() < 2 ** -8 * float(q.abs().max()) + 1e-5
    Kc = from_bf16_bits(K).cpu()
    assert (Kc[:, :, pos0:pos0 + Lq] - k).abs().max() < 2 ** -8 * float(k.abs().max()) + 1e-5
    assert (Kc[:, :, :pos0] == 0).all() and (Kc[:, :, pos0 + Lq:] == 0).all()  # only the new rows are written
    vt = from_bf16_bits(Vt).cpu()
    assert (vt[..., pos0:pos0 + Lq] - v.transpose(2, 3)).abs().max() < 2 ** -8 * float(v.abs().max()) + 1e-5
    assert (vt[..., :pos0] == 0).all() and (vt[..., pos0 + Lq:] == 0).all()


def _attn(Q, K, Vt, B, nH, Lq, Lk, mask=None, causal=False, Lcap=None):
    Lp = Vt.shape[-1]
    Lcap = Lcap or K.shape[2]
    Od = torch.zeros((B, Lq, nH * 64), dtype=torch.int16, device="cuda")
    if causal:
        L().call("showo_attn_fwd", L().ptr(Q), L().ptr(K), L().ptr(Vt), None, None, None, L().ptr(Od), B, nH, Lq, Lk, Lcap, Lp, nH * 64, S())
        return from_bf16_bits(Od).cpu(), None
    md = dev(mask)
    iv = torch.zeros((B, Lq, 4), dtype=torch.int32, device="cuda")
    flag = torch.zeros(4, dtype=torch.int32, device="cuda")
    L().call("showo_mask_compress", L().ptr(md), L().ptr(iv), L().ptr(flag), B, Lq, Lk, S())
    L().call("showo_attn_fwd", L().ptr(Q), L().ptr(K), L().ptr(Vt), L().ptr(iv), L().ptr(flag), L().ptr(md), L().ptr(Od), B, nH, Lq, Lk, Lcap,
             Lp, nH * 64, S())
    sync()
    return from_bf16_bits(Od).cpu(), (iv.cpu(), int(flag[0]))


def _attn_oracle(Q, K, Vt, mask, Lq, Lk):
    q = from_bf16_bits(Q).cpu()  # already scaled by 1/8
    k = from_bf16_bits(K).cpu()[:, :, :Lk]
    v = from_bf16_bits(Vt).cpu()[..., :Lk].transpose(2, 3)
    s = q @ k.transpose(2, 3) + mask
    a = torch.softmax(s, dim=-1)
    o = a @ v  # [B,nH,Lq,64]
    B, nH = o.shape[:2]
    return o.transpose(1, 2).reshape(B, Lq, nH * 64)


def _mask_cases(d, Lq):
    """the mask families of SURVEY.md §8a A5 at sequence length Lq (image block = Lq - text - 2)"""
    rs = np.random.RandomState(1)
    T = min(129, Lq // 3)
    N = Lq - T - 2
    rows = []
    for k in (3, T, T // 2 + 1):
        words = rs.randint(0, 100, size=k).tolist()
        rows.append([d.pad_id] * (T - k) + words + [d.soi_id] + [d.mask_token_id] * N + [d.eoi_id])
    ids = torch.tensor(rows)
    yield "t2i", O.mask_t2i(ids, d.pad_id, d.soi_id, d.eoi_id)
    ids_m = torch.tensor([[d.mmu_id, d.soi_id] + [7] * N + [d.eoi_id] + [5] * (Lq - N - 3)] * 2)
    yield "mmu", O.mask_mmu(ids_m, d.eoi_id)
    yield "lm_causal", O.mask_t2i(torch.full((2, Lq), 5), d.pad_id, d.soi_id, d.eoi_id, rm_pad_in_image=False)


@pytest.fixture(params=[1, 2], ids=["gather", "lds-tiled"])
def attn_impl(request):
    """run the attention tests on both kernels: 1 = gather form (also the decode kernel), 2 = LDS-tiled form"""
    L().call("showo_attn_set_impl", request.param)
    yield request.param
    L().call("showo_attn_set_impl", 0)


@pytest.mark.parametrize("Lq,nH", [(27, 2), (387, 2), (1155, 1)])
def test_attention_mask_families(Lq, nH, attn_impl):
    d = util.tiny_dims()
    torch.manual_seed(Lq)
    for name, mask in _mask_cases(d, Lq):
        B = mask.shape[0]
        qkv = torch.randn(B * Lq, 3 * nH * 64) * 2
        qw, qb, kw, kb = torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05
        Q, K, Vt = _prep(qkv, qw, qb, kw, kb, B, Lq, nH)
        got, (iv, flag) = _attn(Q, K, Vt, B, nH, Lq, Lq, mask)
        assert flag == 0, f"{name}: reference-built masks must be interval-representable"
        # intervals reproduce the dense mask exactly
        vis = (mask[:, 0] == 0)
        c = torch.arange(Lq)[None, None, :]
        rec = ((c >= iv[..., 0:1]) & (c < iv[..., 1:2])) | ((c >= iv[..., 2:3]) & (c < iv[..., 3:4]))
        assert torch.equal(rec, vis), name
        want = _attn_oracle(Q, K, Vt, mask, Lq, Lq)
        err = (got - want).abs().max()
        # P and O are rounded to bf16 once each: 2^-8 relative to max|V| is the expected scale
        assert err < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3, (name, float(err))


def test_attention_mmu_vit_and_dense_fallback(attn_impl):
    torch.manual_seed(3)
    nH, Lq = 2, 700
    mask = O.mask_mmu_vit(1, Lq, system_prompt_len=28)  # causal + columns [30,606) visible: two intervals
    qkv = torch.randn(Lq, 3 * nH * 64)
    p = [torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05]
    Q, K, Vt = _prep(qkv, *p, 1, Lq, nH)
    got, (iv, flag) = _attn(Q, K, Vt, 1, nH, Lq, Lq, mask)
    assert flag == 0
    want = _attn_oracle(Q, K, Vt, mask, Lq, Lq)
    assert (got - want).abs().max() < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
    # arbitrary (non-interval) visibility + a soft additive bias -> dense path must be taken and be right
    Lq = 150
    vis = torch.rand(2, 1, Lq, Lq) < 0.5
    vis |= torch.eye(Lq, dtype=torch.bool)[None, None]
    mask = torch.where(vis, torch.zeros(()), torch.full((), O.NEG_MASK))
    mask[:, :, :, 3] = -1.5  # soft bias column
    qkv = torch.randn(2 * Lq, 3 * nH * 64)
    Q, K, Vt = _prep(qkv, *p, 2, Lq, nH)
    got, (iv, flag) = _attn(Q, K, Vt, 2, nH, Lq, Lq, mask)
    assert flag == 1
    want = _attn_oracle(Q, K, Vt, mask, Lq, Lq)
    assert (got - want).abs().max() < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
    # no mask -> causal
    got, _ = _attn(Q, K, Vt, 2, nH, Lq, Lq, causal=True)
    cm = torch.where(torch.tril(torch.ones(Lq, Lq, dtype=torch.bool)), torch.zeros(()), torch.full((), O.NEG_MASK))[None, None]
    want = _attn_oracle(Q, K, Vt, cm, Lq, Lq)
    assert (got - want).abs().max() < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3


def test_attention_online_softmax_rescale_branch(attn_impl):
    """force the running-max rescale: one key in a late tile dominates one query row (CDNA playbook rule 26)"""
    torch.manual_seed(5)
    nH, Lq = 1, 200
    qkv = torch.randn(Lq, 3 * 64)
    p = [torch.ones(64), torch.zeros(64), torch.ones(64), torch.zeros(64)]
    Q, K, Vt = _prep(qkv, *p, 1, Lq, nH)
    Qf, Kf = from_bf16_bits(Q), from_bf16_bits(K)
    Kf[0, 0, 170] = Qf[0, 0, 180] * 8 * 6.0  # q180 . k170 is huge -> max jumps at key tile 160..191
    K2 = dev(to_bf16_bits(Kf.cpu()))
    mask = torch.zeros(1, 1, Lq, Lq)
    got, _ = _attn(Q, K2, Vt, 1, nH, Lq, Lq, mask)
    want = _attn_oracle(Q, K2, Vt, mask, Lq, Lq)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3


def test_kv_cache_append_matches_full():
    """qk_prep append path + Lq=1 attention == last row of the full computation (exactness of the KV cache)"""
    torch.manual_seed(7)
    nH, L0, cap = 2, 45, 128
    qkv = torch.randn(L0 + 1, 3 * nH * 64)
    p = [torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05]
    Qf, Kf, Vf = _prep(qkv, *p, 1, L0 + 1, nH)
    mask = O.mask_t2i(torch.full((1, L0 + 1), 5), 0, 1, 2, rm_pad_in_image=False)
    full, _ = _attn(Qf, Kf, Vf, 1, nH, L0 + 1, L0 + 1, mask)
    Kc = torch.zeros((1, nH, cap, 64), dtype=torch.int16, device="cuda")
    Vc = torch.zeros((1, nH, 64, cap), dtype=torch.int16, device="cuda")
    _prep(qkv[:L0], *p, 1, L0, nH, pos0=0, Lcap=cap, Lp=cap, Kbuf=Kc, Vbuf=Vc)
    Q1, _, _ = _prep(qkv[L0:], *p, 1, 1, nH, pos0=L0, Lcap=cap, Lp=cap, Kbuf=Kc, Vbuf=Vc)
    assert torch.equal(Kc[:, :, :L0 + 1].cpu(), Kf.cpu())
    assert torch.equal(Vc[..., :L0 + 1].cpu(), Vf[..., :L0 + 1].cpu())
    one, _ = _attn(Q1, Kc, Vc, 1, nH, 1, L0 + 1, causal=True, Lcap=cap)
    # the cache contents are bit-exact (above); the single-query kernel sums q.k and p.v in a different order than the MFMA
    # tiles, so its output row agrees to the bf16 rounding of O
    assert (one[0, 0] - full[0, L0]).abs().max() <= 2 ** -7 * float(full[0, L0].abs().max())


def test_single_query_decode_attention_two_intervals_and_dense():
    """Lq = 1 (AR decode kernel, keys spread over the block): batch 2, long cache, two visible intervals and the dense fallback"""
    torch.manual_seed(11)
    nH, Lk, cap, B = 3, 701, 768, 2
    qkv = torch.randn(B * Lk, 3 * nH * 64)
    p = [torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05]
    Kc = torch.zeros((B, nH, cap, 64), dtype=torch.int16, device="cuda")
    Vc = torch.zeros((B, nH, 64, cap), dtype=torch.int16, device="cuda")
    Qf, _, _ = _prep(qkv, *p, B, Lk, nH, pos0=0, Lcap=cap, Lp=cap, Kbuf=Kc, Vbuf=Vc)
    Q1 = Qf[:, :, Lk - 1:Lk].contiguous()
    vis = torch.zeros(B, 1, 1, Lk, dtype=torch.bool)
    vis[0, ..., 3:40] = True; vis[0, ..., 300:Lk] = True   # two intervals
    vis[1, ..., 17:650] = True                              # one interval, last keys hidden
    mask = torch.where(vis, torch.zeros(()), torch.full((), O.NEG_MASK))
    got, (iv, flag) = _attn(Q1, Kc, Vc, B, nH, 1, Lk, mask, Lcap=cap)
    assert flag == 0
    want = _attn_oracle(Q1, Kc, Vc, mask, 1, Lk)
    assert (got - want).abs().max() < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
    vis = torch.rand(B, 1, 1, Lk) < 0.5
    vis[..., 0] = True
    mask = torch.where(vis, torch.zeros(()), torch.full((), O.NEG_MASK))
    mask[..., 5] = -0.75
    got, (iv, flag) = _attn(Q1, Kc, Vc, B, nH, 1, Lk, mask, Lcap=cap)
    assert flag == 1
    want = _attn_oracle(Q1, Kc, Vc, mask, 1, Lk)
    assert (got - want).abs().max() < 2.5 * 2 ** -8 * float(want.abs().max()) + 1e-3


# --------------------------------------------------------------------------------------------- sampler
def test_sampler_teacher_forced_against_reference_trajectory():
    """feed the reference's own per-step logits + its recorded noise into the HIP sampler: ids must match."""
    g = util.golden("showo_tiny_t2i.npz")
    d = util.tiny_dims()
    steps, w = int(g["steps"]), float(g["guidance"])
    B, N, V = g["ids_cond"].shape[0], d.num_vq_tokens, d.codebook
    off, mask_id = d.image_offset, d.mask_token_id
    Lseq = g["ids_cond"].shape[1]
    img0 = Lseq - (N + 1)
    ml, tp = util.pkg().sampling.t2i_step_constants(steps, N)
    mismatches = 0
    for s in range(steps):
        ids_in = torch.from_numpy(g["fwd_in"][s])  # [2B, L] as fed to the reference model at this step
        lg = torch.from_numpy(g["fwd_logits"][s])[:, img0:img0 + N, off:off + V].contiguous()  # [2B,N,V]
        cur = ids_in[:B, img0:img0 + N].clone()
        cur = torch.where(cur == mask_id, cur, cur - off)
        lc, lu = dev(lg[:B].reshape(B * N, V)), dev(lg[B:].reshape(B * N, V))
        curd = dev(cur)
        sampled = torch.empty((B, N), dtype=torch.int64, device="cuda")
        sel = torch.empty((B, N), dtype=torch.float32, device="cuda")
        en = dev(g["exp_noise"][s].reshape(B * N, V))
        L().call("showo_cfg_softmax_sample", L().ptr(lc), L().ptr(lu), V, w, L().ptr(curd), mask_id, L().ptr(en), 0, s,
                 L().ptr(sampled), L().ptr(sel), B, N, V, S())
        # oracle for this step from the same inputs
        z = (1 + w) * lg[:B] - w * lg[B:]
        probs = z.softmax(-1)
        want = torch.argmax(probs.reshape(B * N, V) / torch.from_numpy(g["exp_noise"][s]).reshape(B * N, V), -1).view(B, N)
        unknown = cur == mask_id
        want = torch.where(unknown, want, cur)
        assert torch.equal(want[unknown], torch.from_numpy(g["multinomial"][s]).view(B, N)[unknown])  # oracle == reference draw
        got = sampled.cpu()
        mismatches += int((got != want).sum())
        assert torch.equal(got[~unknown], cur[~unknown])
        selw = torch.where(unknown, torch.gather(probs, -1, want[..., None])[..., 0], torch.tensor(torch.finfo(torch.float32).max))
        ok = got == want
        assert torch.allclose(sel.cpu()[ok], selw[ok], rtol=2e-5, atol=1e-9)
        # mask_by_topk with the reference's uniform draws; feed the ORACLE's sampled/sel so the step is isolated
        ids_c = dev(ids_in[:B].clone())
        ids_u = dev(ids_in[B:].clone())
        masking = torch.zeros((B, N), dtype=torch.uint8, device="cuda")
        cur2 = dev(cur)
        L().call("showo_mask_by_topk", L().ptr(dev(selw)), L().ptr(dev(want)), L().ptr(cur2), L().ptr(ids_c), L().ptr(ids_u), Lseq, img0,
                 mask_id, off, ml[s], tp[s], L().ptr(dev(g["uniform"][s].reshape(B, N))), 0, s, L().ptr(masking), B, N, S())
        sync()
        if s + 1 < steps:
            nxt = torch.from_numpy(g["fwd_in"][s + 1])
            assert torch.equal(ids_c.cpu(), nxt[:B]), f"step {s}: re-masked ids differ from the reference"
            assert torch.equal(ids_u.cpu()[:, img0:img0 + N], nxt[B:, img0:img0 + N])
        else:
            assert torch.equal(ids_c.cpu(), torch.from_numpy(g["final_input_ids"]))
    assert mismatches == 0, f"{mismatches} sampled ids differ from the reference draw"


def test_sampler_philox_distribution():
    """chi-square of the on-device Philox sampler against the softmax it samples from (no injected noise)."""
    torch.manual_seed(0)
    V, rows = 16, 20000
    logit = torch.randn(V)
    lc = dev(logit[None].repeat(rows, 1))
    cur = torch.full((rows,), 99, dtype=torch.int64, device="cuda")
    sampled = torch.empty((rows,), dtype=torch.int64, device="cuda")
    sel = torch.empty((rows,), dtype=torch.float32, device="cuda")
    L().call("showo_cfg_softmax_sample", L().ptr(lc), None, V, 0.0, L().ptr(cur), 99, None, 1234, 0, L().ptr(sampled), L().ptr(sel), rows, 1, V, S())
    p = logit.softmax(-1).double()
    cnt = torch.bincount(sampled.cpu(), minlength=V).double()
    chi2 = float(((cnt - rows * p) ** 2 / (rows * p)).sum())
    assert chi2 < 50.0, chi2  # dof = 15; P(chi2 > 50) ~ 1e-5
    assert torch.allclose(sel.cpu().double(), p[sampled.cpu()], rtol=1e-5)
    # different step -> different draws
    s2 = torch.empty_like(sampled)
    L().call("showo_cfg_softmax_sample", L().ptr(lc), None, V, 0.0, L().ptr(cur), 99, None, 1234, 1, L().ptr(s2), L().ptr(sel), rows, 1, V, S())
    assert (s2 != sampled).float().mean() > 0.3


# ------------------------------------------------------------------------------------ VQGAN kernels
def test_groupnorm_and_conv():
    torch.manual_seed(0)
    B, H, W, Cin, Cout = 2, 12, 10, 128, 256
    x = torch.randn(B, Cin, H, W) * 2 + 0.3
    gam, bet = torch.randn(Cin) * .1 + 1, torch.randn(Cin) * .05
    xn = dev(x.permute(0, 2, 3, 1))
    stats = torch.zeros((L().load().showo_gn_stats_doubles(B, H * W),), dtype=torch.float64, device="cuda")
    y = torch.empty((B, H * W, Cin), dtype=torch.int16, device="cuda")
    ylo = torch.empty_like(y)
    L().call("showo_gn_stats", L().ptr(xn), L().ptr(stats), B, H * W, Cin, S())
    st = stats[:B * 64].reshape(B, 32, 2).cpu()
    xg = x.double().reshape(B, 32, -1)
    assert torch.allclose(st[..., 0], xg.sum(-1), rtol=1e-12, atol=1e-9) and torch.allclose(st[..., 1], (xg * xg).sum(-1), rtol=1e-12)
    stats2 = torch.zeros_like(stats)
    L().call("showo_gn_stats", L().ptr(xn), L().ptr(stats2), B, H * W, Cin, S())
    assert torch.equal(stats[:B * 64], stats2[:B * 64])  # deterministic: no atomics
    L().call("showo_gn_apply", L().ptr(xn), L().ptr(stats), L().ptr(dev(gam)), L().ptr(dev(bet)), L().ptr(y), L().ptr(ylo), B, H * W, Cin,
             1e-6, 1, S())
    want = O.swish(O.group_norm(x, gam, bet)).permute(0, 2, 3, 1).reshape(B, H * W, Cin)
    got = from_bf16_bits(y).cpu()
    assert (got - want).abs().max() < 2 ** -8 * float(want.abs().max()) + 1e-5
    got2 = got + from_bf16_bits(ylo).cpu()  # split precision: hi + lo carries ~16 mantissa bits
    assert (got2 - want).abs().max() < 3e-5 * float(want.abs().max())
    # conv modes on the bf16 image, vs F.conv2d on the same rounded operands
    import torch.nn.functional as F
    a = from_bf16_bits(y).cpu().reshape(B, H, W, Cin).permute(0, 3, 1, 2)
    wgt = torch.randn(Cout, Cin, 3, 3) * 0.03
    bias = torch.randn(Cout)
    wp = dev(to_bf16_bits(wgt.permute(0, 2, 3, 1).contiguous()))
    wr = bf16_round(wgt)
    for mode, (Ho, Wo) in ((0, (H, W)), (1, (2 * H, 2 * W)), (2, (H // 2, W // 2))):
        out = torch.full((B, Ho * Wo, Cout), float("nan"), dtype=torch.float32, device="cuda")
        res = torch.randn(B, Ho * Wo, Cout)
        L().call("showo_conv3x3_bf16", L().ptr(y), L().ptr(wp), L().ptr(dev(bias)), L().ptr(dev(res)), L().ptr(out), B, H, W, Cin, Cout, mode, S())
        if mode == 0:
            ref = F.conv2d(a, wr, bias, padding=1)
        elif mode == 1:
            ref = F.conv2d(a.repeat_interleave(2, 2).repeat_interleave(2, 3), wr, bias, padding=1)
        else:
            ref = F.conv2d(F.pad(a, (0, 1, 0, 1)), wr, bias, stride=2)
        ref = ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout) + res
        err = (out.cpu() - ref).abs().max()
        assert err < 1e-3 * float(ref.abs().max()), (mode, float(err))
    # thin output (Cout = 3, like decoder.conv_out)
    w3 = torch.randn(3, Cin, 3, 3) * 0.03
    out = torch.empty((B, H * W, 3), dtype=torch.float32, device="cuda")
    L().call("showo_conv3x3_bf16", L().ptr(y), L().ptr(dev(to_bf16_bits(w3.permute(0, 2, 3, 1).contiguous()))), None, None, L().ptr(out), B, H, W,
             Cin, 3, 0, S())
    ref = F.conv2d(a, bf16_round(w3), None, padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 3)
    assert (out.cpu() - ref).abs().max() < 1e-3 * float(ref.abs().max())


def test_softmax_rows_and_small_conv():
    torch.manual_seed(1)
    x = torch.randn(40, 100) * 3
    y = torch.full((40, 128), 7, dtype=torch.int16, device="cuda")
    ylo = torch.full((40, 128), 7, dtype=torch.int16, device="cuda")
    L().call("showo_softmax_rows_bf16", L().ptr(dev(x)), L().ptr(y), L().ptr(ylo), 40, 100, 128, 0.5, S())
    got = from_bf16_bits(y).cpu()
    assert (got[:, :100] - torch.softmax(x * 0.5, -1)).abs().max() < 2 ** -8
    assert (got[:, 100:] == 0).all()
    got2 = got + from_bf16_bits(ylo).cpu()
    assert (got2[:, :100] - torch.softmax(x * 0.5, -1)).abs().max() < 2e-6 and (from_bf16_bits(ylo).cpu()[:, 100:] == 0).all()
    import torch.nn.functional as F
    z = torch.randn(2, 13, 5, 4)
    wq, bq = torch.randn(13, 13, 1, 1), torch.randn(13)
    out = torch.empty((2, 20, 13), dtype=torch.float32, device="cuda")
    L().call("showo_conv_small_f32", L().ptr(dev(z.permute(0, 2, 3, 1))), L().ptr(dev(wq.permute(0, 2, 3, 1))), L().ptr(dev(bq)), L().ptr(out), 2, 5, 4,
             13, 13, 1, S())
    ref = F.conv2d(z, wq, bq).permute(0, 2, 3, 1).reshape(2, 20, 13)
    assert (out.cpu() - ref).abs().max() < 1e-5


@pytest.mark.parametrize("impl", [1, 2, 5] + [5 + (v << 16) for v in (256, 240, 224, 208, 176, 160, 144, 1192, 1176, 1160, 1144, 1128, 2256, 2240, 2224, 2208, 3192, 3176, 3160, 3144, 4192, 4176, 4160, 4144)])
def test_gemm_both_tile_kernels(impl):
    """the 128^2 register-staged kernel and the 256^2 global_load_lds kernel give the same result on shapes
    with ragged M/N edges (rows/cols beyond the edge are clamped on load and predicated on store)"""
    L().call("showo_gemm_set_impl", impl & 0xffff)
    L().call("showo_gemm_tune", 8, (impl >> 16) << 8, None)  # impl 5: forced tile variant = rows (+ 1000: n-split phase program); 0 = automatic
    try:
        torch.manual_seed(impl & 0xffff)
        for (M, N, K) in [(1100, 520, 192), (6192 // 4, 2048, 256), (300, 256, 64)]:
            A, W, bias = torch.randn(M, K), torch.randn(N, K) * 0.05, torch.randn(N)
            ref = bf16_round(A).double() @ bf16_round(W).double().T + bias.double()
            got = _gemm(A, W, bias, 2)
            assert (got.double() - ref).abs().max() < 1e-3 * float(ref.abs().max()), (impl, M, N, K)
            resid = torch.randn(M, N)
            got = _gemm(A, W, bias, 3, resid=resid)
            assert (got.double() - (ref + resid.double())).abs().max() < 1e-3 * float(ref.abs().max())
            got = _gemm(A, W, bias, 1)
            want = O.gelu_new(ref.float()).double()
            assert (got.double() - want).abs().max() < 2 ** -8 * float(want.abs().max()) + 1e-3 * float(ref.abs().max())
    finally:
        L().call("showo_gemm_set_impl", 0)
        L().call("showo_gemm_tune", 8, 0, None)


def _tiled(W):
    """device copy of a bf16 weight [N, K] in the tiled layout of showo_gemm_tile_weight"""
    N, K = W.shape
    n = L().load().showo_gemm_tiled_elems(N, K)
    out = torch.full((n,), -1, dtype=torch.int16, device="cuda")
    L().call("showo_gemm_tile_weight", L().ptr(W), K, N, K, L().ptr(out), S())
    return out


def test_tile_weight_layout():
    """[ceil(N/256)][K/64][256][64] blocks, chunk p of row r holds logical chunk p ^ (r & 7), rows beyond N are zero"""
    N, K = 300, 192
    W = torch.arange(N * K, dtype=torch.int32).remainder(30011).to(torch.int16).view(N, K)
    got = _tiled(dev(W)).cpu().view(2, K // 64, 256, 8, 8)
    want = torch.zeros(2, K // 64, 256, 8, 8, dtype=torch.int16)
    Wp = torch.zeros(512, K, dtype=torch.int16)
    Wp[:N] = W
    for r in range(256):
        for p in range(8):
            want[:, :, r, p] = Wp.view(2, 256, K // 64, 8, 8)[:, r, :, p ^ (r & 7)]
    assert torch.equal(got, want)


@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("variant", [0, 256, 208, 1176, 1144, 2240, 2208, 3176, 3144, 4176, 4144])
@pytest.mark.parametrize("M,K0,K1,N", [(700, 128, 512, 384), (1548, 256, 1024, 2048), (300, 64, 64, 256)])
def test_gemm_kcat_residual(variant, M, K0, K1, N, tiled):
    """showo_gemm_kcat_bf16: x += [A0 | A1] [W0 | W1]^T + bias in one launch (Phi's dense + fc2 into the same residual row,
    models/phi.py:774-790), operands with DIFFERENT leading dimensions; fp64 reference on the same bf16-rounded operands."""
    torch.manual_seed(M + variant)
    A0, A1 = torch.randn(M, K0), torch.randn(M, K1)
    W = torch.randn(N, K0 + K1) * 0.05
    bias, x = torch.randn(N), torch.randn(M, N)
    ref = bf16_round(torch.cat([A0, A1], 1)).double() @ bf16_round(W).double().T + bias.double() + x.double()
    xd = dev(x.clone())
    Wd = dev(to_bf16_bits(W))
    if tiled:
        Wd = _tiled(Wd)
    L().call("showo_gemm_tune", 8, variant << 8, None)
    try:
        L().call("showo_gemm_kcat_bf16", L().ptr(dev(to_bf16_bits(A0))), K0, K0, L().ptr(dev(to_bf16_bits(A1))), K1, K1,
                 L().ptr(Wd), K0 + K1, L().ptr(dev(bias)), L().ptr(xd), N, L().ptr(xd), N, M, N, 3, tiled, S())
        sync()
    finally:
        L().call("showo_gemm_tune", 8, 0, None)
    assert (xd.cpu().double() - ref).abs().max() < 1e-3 * float(ref.abs().max())
    with pytest.raises(RuntimeError):  # only the residual epilogue is implemented
        L().call("showo_gemm_kcat_bf16", L().ptr(dev(to_bf16_bits(A0))), K0, K0, L().ptr(dev(to_bf16_bits(A1))), K1, K1,
                 L().ptr(dev(to_bf16_bits(W))), K0 + K1, L().ptr(dev(bias)), L().ptr(xd), N, None, 0, M, N, 2, 0, S())


@pytest.mark.parametrize("variant", [256, 224, 1192, 1160, 1128, 2256, 2224, 3192, 3160, 4192, 4160])
@pytest.mark.parametrize("B,Lq,nH,F", [(2, 387, 4, 512), (3, 130, 4, 1024)])
def test_fused_qkv_fc1_equals_separate_launches_bits(variant, B, Lq, nH, F):
    """showo_gemm_qkv_fc1_bf16 ([Wqkv ; W1], column-split epilogue) is bit-identical to showo_gemm_qkv_bf16 + the fc1 GELU GEMM
    at the same tile variant (same k order per output element)."""
    torch.manual_seed(B * 100 + Lq + variant)
    H, M = nH * 64, B * Lq
    h = dev(to_bf16_bits(torch.randn(M, H)))
    W = dev(to_bf16_bits(torch.randn(3 * H + F, H) * 0.05))
    bias = dev(torch.randn(3 * H + F) * 0.1)
    ln = [dev(t) for t in (torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05)]
    cos, sin = (dev(t) for t in _rope_tables())
    Lp = ((Lq + 63) // 64) * 64
    outs = []
    Wt = _tiled(W)
    L().call("showo_gemm_tune", 8, variant << 8, None)
    try:
        for fused in (0, 1, 2):  # separate launches | fused, row-major weights | fused, tiled weights
            Q = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
            K = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
            Vt = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
            f = torch.zeros((M, F), dtype=torch.int16, device="cuda")
            if fused:
                L().call("showo_gemm_qkv_fc1_bf16", L().ptr(h), H, L().ptr(Wt if fused == 2 else W), H, L().ptr(bias), *[L().ptr(t) for t in ln],
                         L().ptr(cos), L().ptr(sin), L().ptr(Q), L().ptr(K), L().ptr(Vt), L().ptr(f), F, F, B, Lq, nH, 32, 1e-5, 0, Lq, Lp,
                         int(fused == 2), S())
            else:
                L().call("showo_gemm_qkv_bf16", L().ptr(h), H, L().ptr(W), H, L().ptr(bias), *[L().ptr(t) for t in ln], L().ptr(cos),
                         L().ptr(sin), L().ptr(Q), L().ptr(K), L().ptr(Vt), B, Lq, nH, 32, 1e-5, 0, Lq, Lp, S())
                L().call("showo_gemm_bf16", L().ptr(h), H, W.data_ptr() + 3 * H * H * 2, H, bias.data_ptr() + 3 * H * 4, 0, L().ptr(f), F,
                         None, 0, M, F, H, 1, S())
            sync()
            outs.append((Q, K, Vt, f))
    finally:
        L().call("showo_gemm_tune", 8, 0, None)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert int(outs[1][3].ne(0).sum()) > 0.9 * M * F  # the fc1 tail was written


@pytest.mark.parametrize("variant", [0, 256, 1160, 3192, 2224])
def test_fused_qkv_fc1_save_matches_training_launches(variant):
    """showo_gemm_qkv_fc1_save_bf16 (training forward): the saved qkv / fc1 pre-activation and gelu(fc1) carry the bits of the separate
    launches (showo_gemm_bf16 x2 + showo_gelu_bf16), V^T is a pure relayout of the saved v, and Q / K (LayerNorm + RoPE of the ROUNDED
    q / k, like showo_qk_prep) agree with showo_qk_prep on the saved qkv to bf16 rounding"""
    torch.manual_seed(variant + 3)
    B, Lq, nH, F = 2, 387, 4, 512
    H, M = nH * 64, B * Lq
    h = dev(to_bf16_bits(torch.randn(M, H)))
    W = dev(to_bf16_bits(torch.randn(3 * H + F, H) * 0.05))
    bias = dev(torch.randn(3 * H + F) * 0.1)
    ln = [dev(t) for t in (torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05)]
    cos, sin = (dev(t) for t in _rope_tables())
    Lp = ((Lq + 63) // 64) * 64

    def bufs():
        return (torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda"), torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda"),
                torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda"))
    L().call("showo_gemm_tune", 8, variant << 8, None)
    try:
        Q1, K1, V1 = bufs()
        raw1 = torch.zeros((M, 3 * H), dtype=torch.int16, device="cuda")
        pre1 = torch.zeros((M, F), dtype=torch.int16, device="cuda")
        ffn1 = torch.zeros((M, F), dtype=torch.int16, device="cuda")
        L().call("showo_gemm_qkv_fc1_save_bf16", L().ptr(h), H, L().ptr(W), H, L().ptr(bias), *[L().ptr(t) for t in ln], L().ptr(cos), L().ptr(sin),
                 L().ptr(Q1), L().ptr(K1), L().ptr(V1), L().ptr(raw1), 3 * H, L().ptr(pre1), L().ptr(ffn1), F, F, B, Lq, nH, 32, 1e-5, 0, Lq, Lp, 0, S())
        # raw-only form (Q = K = Vt = NULL): the three saved tensors alone, same bits
        raw2 = torch.zeros_like(raw1); pre2 = torch.zeros_like(pre1); ffn2 = torch.zeros_like(ffn1)
        L().call("showo_gemm_qkv_fc1_save_bf16", L().ptr(h), H, L().ptr(W), H, L().ptr(bias), *[L().ptr(t) for t in ln], L().ptr(cos), L().ptr(sin),
                 None, None, None, L().ptr(raw2), 3 * H, L().ptr(pre2), L().ptr(ffn2), F, F, B, Lq, nH, 32, 1e-5, 0, Lq, Lp, 0, S())
        with pytest.raises(RuntimeError):  # K without Q
            L().call("showo_gemm_qkv_fc1_save_bf16", L().ptr(h), H, L().ptr(W), H, L().ptr(bias), *[L().ptr(t) for t in ln], L().ptr(cos), L().ptr(sin),
                     None, L().ptr(K1), None, L().ptr(raw2), 3 * H, L().ptr(pre2), L().ptr(ffn2), F, F, B, Lq, nH, 32, 1e-5, 0, Lq, Lp, 0, S())
        Q0, K0, V0 = bufs()
        raw0 = torch.zeros_like(raw1); pre0 = torch.zeros_like(pre1); ffn0 = torch.zeros_like(ffn1)
        L().call("showo_gemm_bf16", L().ptr(h), H, L().ptr(W), H, L().ptr(bias), 0, L().ptr(raw0), 3 * H, None, 0, M, 3 * H, H, 0, S())
        L().call("showo_qk_prep", L().ptr(raw0), *[L().ptr(t) for t in ln], L().ptr(cos), L().ptr(sin), L().ptr(Q0), L().ptr(K0), L().ptr(V0), B, Lq,
                 nH, 32, 1e-5, 0, Lq, Lp, S())
        L().call("showo_gemm_bf16", L().ptr(h), H, W.data_ptr() + 3 * H * H * 2, H, bias.data_ptr() + 3 * H * 4, 0, L().ptr(pre0), F, None, 0, M, F, H, 0, S())
        L().call("showo_gelu_bf16", L().ptr(pre0), L().ptr(ffn0), M * F, S())
        sync()
    finally:
        L().call("showo_gemm_tune", 8, 0, None)
    assert torch.equal(raw1, raw0) and torch.equal(pre1, pre0) and torch.equal(ffn1, ffn0) and torch.equal(V1, V0)
    assert torch.equal(raw2, raw0) and torch.equal(pre2, pre0) and torch.equal(ffn2, ffn0)
    for a, b in ((Q1, Q0), (K1, K0)):
        fa, fb = from_bf16_bits(a).cpu(), from_bf16_bits(b).cpu()
        assert float(fb.abs().max()) > 0.1
        assert ((fa - fb).abs() <= 2 ** -7 * fb.abs() + 1e-6).all(), float((fa - fb).abs().max())


@pytest.mark.parametrize("M,N,T,lda", [(256, 256, 200, 256), (264, 520, 1000, 320), (512, 768, 2100, 512), (128, 384, 135, 128),
                                       (300, 2048, 777, 304)])
def test_gemm_tn_weight_gradient_on_token_major_operands(M, N, T, lda):
    """showo_gemm_tn_bf16: out[M, N] = A^T B over the T token rows (dW = dY^T X) straight from token-major bf16 operands, against fp64
    on the same bf16 values: T not a multiple of 64 (zero page), ragged / non-multiple-of-8 M, a strided A, the split-K shape,
    accumulation, run-to-run identical bits; showo_colsum_bf16 (the bias gradient) next to it."""
    torch.manual_seed(M + N + T)
    A = torch.randn(T, lda)
    B = torch.randn(T, N) * 0.5
    Ab, Bb = dev(to_bf16_bits(A)), dev(to_bf16_bits(B))
    ref = bf16_round(A[:, :M]).double().T @ bf16_round(B).double()
    scale = float(ref.abs().max())
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    L().call("showo_gemm_tn_bf16", L().ptr(Ab), lda, L().ptr(Bb), N, L().ptr(out), N, M, N, T, 0, 0, S())
    sync()
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 2e-5 * scale * max(1.0, (T / 512) ** 0.5), (err, scale)
    out2 = torch.full_like(out, float("nan"))
    L().call("showo_gemm_tn_bf16", L().ptr(Ab), lda, L().ptr(Bb), N, L().ptr(out2), N, M, N, T, 0, 0, S())
    assert torch.equal(out, out2)
    L().call("showo_gemm_tn_bf16", L().ptr(Ab), lda, L().ptr(Bb), N, L().ptr(out2), N, M, N, T, 1, 0, S())  # accumulate
    assert float((out2.cpu().double() - 2 * ref).abs().max()) < 4e-5 * scale * max(1.0, (T / 512) ** 0.5)
    # bias gradient: column sums of the same dY
    part = torch.empty((((T + 31) // 32 + 8) * M,), dtype=torch.float32, device="cuda")
    cs = torch.full((M,), float("nan"), dtype=torch.float32, device="cuda")
    L().call("showo_colsum_bf16", L().ptr(Ab), lda, T, M, L().ptr(part), L().ptr(cs), 0, S())
    want = bf16_round(A[:, :M]).double().sum(0)
    assert float((cs.cpu().double() - want).abs().max()) < 1e-5 * float(want.abs().max()) + 1e-4
    L().call("showo_colsum_bf16", L().ptr(Ab), lda, T, M, L().ptr(part), L().ptr(cs), 1, S())
    assert float((cs.cpu().double() - 2 * want).abs().max()) < 2e-5 * float(want.abs().max()) + 2e-4
    with pytest.raises(RuntimeError):  # a row must hold the 8-column unit of its last column
        L().call("showo_gemm_tn_bf16", L().ptr(Ab), lda, L().ptr(Bb), N, L().ptr(out), N, lda + 4, N, T, 0, 0, S())


@pytest.mark.parametrize("M,N,T", [(2048, 8192, 1024), (2048, 8192, 1100), (2048, 2048, 11223), (2048, 2048, 11264), (6144, 2048, 700)])
def test_gemm_tn_fast_form_ignores_nan_padding_rows(M, N, T):
    """the PRODUCTION forms of showo_gemm_tn_bf16 (rows_padded = 1: unchecked DMAs; the 3-deep operand ring when the launch
    has more than 128 tiles, split-K with the partial last k-tile masked in registers otherwise) on operands allocated to
    roundup(T, 64) rows whose padding rows hold NaN / Inf: equal bits to the fully checked form (rows_padded = 0), no NaN anywhere.
    Shapes: one ring launch (256 tiles) with T % 64 == 0 and != 0, the stage-1 split-K shape (64 tiles, T = 11 223 and a multiple of
    64), and a dqkv-shaped launch."""
    torch.manual_seed(M + N + T)
    Tp = (T + 63) // 64 * 64
    A, B = torch.randn(Tp, M), torch.randn(Tp, N) * 0.5
    A[T:], B[T:] = float("nan"), float("inf")
    if Tp > T:
        B[T, ::2] = float("nan")
    Ab, Bb = dev(to_bf16_bits(A)), dev(to_bf16_bits(B))
    fast = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    chk = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    L().call("showo_gemm_tn_bf16", L().ptr(Ab), M, L().ptr(Bb), N, L().ptr(fast), N, M, N, T, 0, 1, S())
    L().call("showo_gemm_tn_bf16", L().ptr(Ab), M, L().ptr(Bb), N, L().ptr(chk), N, M, N, T, 0, 0, S())
    sync()
    assert torch.isfinite(fast).all() and torch.isfinite(chk).all()
    assert torch.equal(fast, chk)
    rows = torch.tensor([0, 1, 257, M - 1])
    ref = bf16_round(A[:T][:, rows]).double().T @ bf16_round(B[:T]).double()
    err = float((fast.cpu()[rows].double() - ref).abs().max())
    assert err < 2e-5 * float(ref.abs().max()) * max(1.0, (T / 512) ** 0.5), err


SPLITK_OFF, SPLITK_ON = 64, 128  # showo_gemm_tune flag bits


@pytest.mark.parametrize("variant", [0, 256, 208, 176, 144, 1192, 1176, 1160, 1144, 1128])  # 1176 = 6 + 5, 1144 = 5 + 4: wave groups of different height
@pytest.mark.parametrize("M,N,K", [(577, 1024, 4096), (631, 2048, 2048), (300, 700, 2048)])
def test_gemm_splitk_epilogues(variant, M, N, K):
    """Launches with few tiles and a long K (the M = 631 prefill, the CLIP tower) split K over tiles x splits ~ 256 blocks; the last
    block to arrive sums the fp32 partials in split order and runs the ordinary epilogue.  Every epilogue against the fp64
    reference on the same bf16 operands, run-to-run identical bits, and within fp32 re-association noise of the unsplit launch."""
    torch.manual_seed(M + N + variant)
    A, W, bias, resid = torch.randn(M, K), torch.randn(N, K) * 0.05, torch.randn(N), torch.randn(M, N)
    ref = bf16_round(A).double() @ bf16_round(W).double().T + bias.double()
    scale = float(ref.abs().max())
    got = {}
    try:
        for mode in (SPLITK_ON, SPLITK_ON, SPLITK_OFF):
            L().call("showo_gemm_tune", 8, (variant << 8) | mode, None)
            got.setdefault(mode, []).append((_gemm(A, W, bias, 2), _gemm(A, W, bias, 3, resid=resid), _gemm(A, W, bias, 1), _gemm(A, W, bias, 0)))
    finally:
        L().call("showo_gemm_tune", 8, SPLITK_ON, None)
    a, b = got[SPLITK_ON]
    for x, y in zip(a, b):
        assert torch.equal(x, y)  # the sum order does not depend on which block arrives last
    f32, res, gelu, bf = a
    assert (f32.double() - ref).abs().max() < 2e-5 * scale
    assert (res.double() - (ref + resid.double())).abs().max() < 2e-5 * scale
    want = O.gelu_new(ref.float()).double()
    assert (gelu.double() - want).abs().max() < 2 ** -8 * float(want.abs().max()) + 1e-3 * scale
    assert (bf.double() - ref).abs().max() < 2 ** -8 * scale
    off = got[SPLITK_OFF][0]
    assert (f32 - off[0]).abs().max() < 1e-5 * scale and (res - off[1]).abs().max() < 1e-5 * scale


@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("variant", [0, 208, 144, 1176, 1128])
def test_gemm_splitk_kcat_prefill_shape(variant, tiled):
    """the cfg4 prefill residual GEMM: M = 631, [attn | ffn] K = 2048 + 8192, 8 weight panels -> 32-40 tiles x 8 splits; the split
    boundaries fall inside both K segments (K-concatenated operand, tiled and row-major weights)"""
    M, K0, K1, N = 631, 2048, 8192, 2048
    torch.manual_seed(variant + tiled)
    A0, A1 = torch.randn(M, K0), torch.randn(M, K1)
    W = torch.randn(N, K0 + K1) * 0.02
    bias, x = torch.randn(N), torch.randn(M, N)
    ref = (bf16_round(torch.cat([A0, A1], 1)).double() @ bf16_round(W).double().T + bias.double() + x.double()).float()
    Wd = dev(to_bf16_bits(W))
    if tiled:
        Wd = _tiled(Wd)
    A0d, A1d, bd = dev(to_bf16_bits(A0)), dev(to_bf16_bits(A1)), dev(bias)
    outs = []
    try:
        for mode in (SPLITK_ON, SPLITK_ON, SPLITK_OFF):
            L().call("showo_gemm_tune", 8, (variant << 8) | mode, None)
            xd = dev(x.clone())
            L().call("showo_gemm_kcat_bf16", L().ptr(A0d), K0, K0, L().ptr(A1d), K1, K1, L().ptr(Wd), K0 + K1, L().ptr(bd), L().ptr(xd), N,
                     L().ptr(xd), N, M, N, 3, tiled, S())
            sync()
            outs.append(xd.cpu())
    finally:
        L().call("showo_gemm_tune", 8, SPLITK_ON, None)
    scale = float(ref.abs().max())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0] - ref).abs().max() < 2e-5 * scale and (outs[2] - ref).abs().max() < 2e-5 * scale


def test_gemm_splitk_qkv_epilogue():
    """fused QKV projection with few tiles (K = 2048, 24 weight panels, M = 600 -> 96 tiles x 2 splits): split-K under the LayerNorm +
    RoPE + relayout epilogue; bf16 outputs within 2 ulp of the unsplit launch, identical from run to run"""
    B, Lq, nH = 2, 300, 32
    H, M = nH * 64, B * Lq
    torch.manual_seed(5)
    h = dev(to_bf16_bits(torch.randn(M, H)))
    W = dev(to_bf16_bits(torch.randn(3 * H, H) * 0.03))
    bias = dev(torch.randn(3 * H) * 0.1)
    ln = [dev(t) for t in (torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05)]
    cos, sin = (dev(t) for t in _rope_tables())
    Lp = ((Lq + 63) // 64) * 64
    outs = []
    try:
        for mode in (SPLITK_ON, SPLITK_ON, SPLITK_OFF):
            L().call("showo_gemm_tune", 8, (1160 << 8) | mode, None)
            Q = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
            K = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
            Vt = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
            L().call("showo_gemm_qkv_bf16", L().ptr(h), H, L().ptr(W), H, L().ptr(bias), *[L().ptr(t) for t in ln], L().ptr(cos),
                     L().ptr(sin), L().ptr(Q), L().ptr(K), L().ptr(Vt), B, Lq, nH, 32, 1e-5, 0, Lq, Lp, S())
            sync()
            outs.append([from_bf16_bits(t).cpu() for t in (Q, K, Vt)])
    finally:
        L().call("showo_gemm_tune", 8, SPLITK_ON, None)
    for a, b, c in zip(*outs):
        assert torch.equal(a, b)
        assert float(a.abs().max()) > 0.1
        assert ((a - c).abs() <= 2 ** -7 * c.abs() + 1e-6).all()


# ------------------------------------------------------------------------------------ split-precision (bf16 x3) kernels
def _split(t):
    """(hi, lo) bf16 bit tensors on the device with t ~= hi + lo"""
    hi = torch.empty(t.shape, dtype=torch.int16, device="cuda")
    lo = torch.empty_like(hi)
    L().call("showo_split_f32_bf16", L().ptr(dev(t)), L().ptr(hi), L().ptr(lo), t.numel(), S())
    util._KEEP.extend([hi, lo])
    return hi, lo


def test_split_gemm_and_conv_track_fp32():
    """x3 kernels: operands as (hi, lo) bf16 pairs, product = hi*hi + hi*lo + lo*hi accumulated in fp32.  Compared
    with an fp64 product of the ORIGINAL fp32 operands: bound 3e-5 of the output scale (vs 4e-3 for plain bf16)."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    t = torch.randn(1000) * 3
    hi, lo = _split(t)
    rec = from_bf16_bits(hi).cpu() + from_bf16_bits(lo).cpu()
    assert torch.equal(from_bf16_bits(hi).cpu(), bf16_round(t)) and (rec - t).abs().max() <= 2 ** -16 * float(t.abs().max())
    for (M, N, K) in [(300, 200, 128), (129, 512, 512), (16, 16, 64)]:
        A, W, bias, resid = torch.randn(M, K), torch.randn(N, K) * 0.05, torch.randn(N), torch.randn(M, N)
        ah, al = _split(A)
        wh, wl = _split(W)
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        L().call("showo_gemm_bf16x3", L().ptr(ah), L().ptr(al), K, L().ptr(wh), L().ptr(wl), K, L().ptr(dev(bias)), 0, L().ptr(out), N,
                 L().ptr(dev(resid)), N, M, N, K, S())
        ref = A.double() @ W.double().T + bias.double() + resid.double()
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 3e-5, (M, N, K, err)
    # small M -> 128^2 register-staged kernel; M >= 2048 -> phase-split DMA kernel (ragged pixel / channel edges, thin Cout)
    for (B, H, Wd, Cin, Cout) in [(2, 9, 7, 128, 128), (2, 50, 46, 128, 192), (1, 64, 40, 64, 3), (3, 31, 33, 256, 128)]:
        x = torch.randn(B, Cin, H, Wd)
        wgt = torch.randn(Cout, Cin, 3, 3) * 0.03
        bias = torch.randn(Cout)
        xh, xl = _split(x.permute(0, 2, 3, 1).contiguous())
        wh, wl = _split(wgt.permute(0, 2, 3, 1).contiguous())
        for mode, (Ho, Wo) in ((0, (H, Wd)), (1, (2 * H, 2 * Wd)), (2, (H // 2, Wd // 2))):
            out = torch.full((B, Ho * Wo, Cout), float("nan"), dtype=torch.float32, device="cuda")
            res = torch.randn(B, Ho * Wo, Cout) if mode == 0 else None
            L().call("showo_conv3x3_bf16x3", L().ptr(xh), L().ptr(xl), L().ptr(wh), L().ptr(wl), L().ptr(dev(bias)),
                     None if res is None else L().ptr(dev(res)), L().ptr(out), B, H, Wd, Cin, Cout, mode, S())
            xd, wd_, bd = x.double(), wgt.double(), bias.double()
            if mode == 0:
                ref = F.conv2d(xd, wd_, bd, padding=1)
            elif mode == 1:
                ref = F.conv2d(xd.repeat_interleave(2, 2).repeat_interleave(2, 3), wd_, bd, padding=1)
            else:
                ref = F.conv2d(F.pad(xd, (0, 1, 0, 1)), wd_, bd, stride=2)
            ref = ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)
            if res is not None:
                ref = ref + res.double()
            err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
            assert err < 3e-5, (B, H, Wd, Cin, Cout, mode, err)


def test_split_conv_with_groupnorm_statistics_equals_conv_then_gn_stats():
    """showo_conv3x3_bf16x3_gn: same output bits as the plain call, and the (sum, sumsq) of its output per (image, group) equal to
    showo_gn_stats on that output to the precision of the double sums -- through the epilogue-fused path (256-pixel tiles inside one
    image, channels per group 4 / 8 / 16, with and without residual, all three gather modes) and through the fallback."""
    torch.manual_seed(11)
    cases = [(2, 32, 32, 128, 128, 0, True), (8, 16, 16, 256, 256, 0, True), (8, 16, 16, 128, 512, 0, False), (2, 16, 32, 128, 128, 1, False),
             (8, 32, 32, 128, 256, 2, False), (3, 48, 16, 64, 128, 0, True), (1, 16, 16, 128, 128, 0, False), (2, 30, 34, 128, 128, 0, True)]
    for (B, H, Wd, Cin, Cout, mode, with_res) in cases:
        Ho, Wo = (H, Wd) if mode == 0 else (2 * H, 2 * Wd) if mode == 1 else (H // 2, Wd // 2)
        x = torch.randn(B, H, Wd, Cin) + 0.2
        wgt = torch.randn(Cout, 3, 3, Cin) * 0.03
        bias = torch.randn(Cout)
        res = dev(torch.randn(B, Ho * Wo, Cout)) if with_res else None
        xh, xl = _split(x)
        wh, wl = _split(wgt)
        out0 = torch.full((B, Ho * Wo, Cout), float("nan"), dtype=torch.float32, device="cuda")
        out1 = torch.full_like(out0, float("nan"))
        nd = L().load().showo_gn_stats_doubles(B, Ho * Wo)
        st1 = torch.full((nd,), float("nan"), dtype=torch.float64, device="cuda")
        st0 = torch.zeros_like(st1)
        rp = None if res is None else L().ptr(res)
        L().call("showo_conv3x3_bf16x3", L().ptr(xh), L().ptr(xl), L().ptr(wh), L().ptr(wl), L().ptr(dev(bias)), rp, L().ptr(out0),
                 B, H, Wd, Cin, Cout, mode, S())
        L().call("showo_conv3x3_bf16x3_gn", L().ptr(xh), L().ptr(xl), L().ptr(wh), L().ptr(wl), L().ptr(dev(bias)), rp, L().ptr(out1),
                 L().ptr(st1), B, H, Wd, Cin, Cout, mode, S())
        assert torch.equal(out0, out1), (B, H, Wd, Cin, Cout, mode)
        L().call("showo_gn_stats", L().ptr(out0), L().ptr(st0), B, Ho * Wo, Cout, S())
        a, b = st1[:B * 64].cpu(), st0[:B * 64].cpu()
        assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=1e-12, atol=1e-9), (B, H, Wd, Cin, Cout, mode, float((a - b).abs().max()))
        og = out0.cpu().double().reshape(B, Ho * Wo, 32, Cout // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
        assert torch.allclose(a.reshape(B, 32, 2)[..., 0], og.sum(-1), rtol=1e-11, atol=1e-8)
        assert torch.allclose(a.reshape(B, 32, 2)[..., 1], (og * og).sum(-1), rtol=1e-11)
        st2 = torch.full_like(st1, float("nan"))  # deterministic
        L().call("showo_conv3x3_bf16x3_gn", L().ptr(xh), L().ptr(xl), L().ptr(wh), L().ptr(wl), L().ptr(dev(bias)), rp, L().ptr(out1),
                 L().ptr(st2), B, H, Wd, Cin, Cout, mode, S())
        assert torch.equal(st1[:B * 64], st2[:B * 64])


# --------------------------------------------------------------------------------------------- AR decode sampler
def _topk_reference(logits, top_k, temperature, e):
    """modeling_showo.py:220-228 with multinomial(p, 1) realised as argmax(p / E) (fp32, CPU)"""
    lg = logits[None].clone() / temperature
    if top_k is not None:
        v, _ = torch.topk(lg, min(top_k, lg.size(-1)))
        lg[lg < v[:, [-1]]] = -float("inf")
    p = torch.softmax(lg, dim=-1)
    return int(torch.argmax(p / e[None], dim=-1)), p[0]


@pytest.mark.parametrize("V", [439, 58498])
def test_sample_topk_with_injected_noise_equals_reference_expression(V):
    torch.manual_seed(V)
    tok = torch.zeros(1, dtype=torch.int64, device="cuda")
    for top_k, T in ((None, 1.0), (1, 1.0), (5, 0.7), (50, 1.3), (V, 0.9), (V + 7, 1.0), (3, 2.0)):
        for rep in range(3):
            logits = torch.randn(V) * 3
            if rep == 2:  # ties at the k-th value: the reference keeps every entry equal to it
                logits = (logits * 2).round() / 2
            e = torch.empty(V).exponential_(1)
            want, p = _topk_reference(logits, top_k, T, e)
            L().call("showo_sample_topk", L().ptr(dev(logits)), V, 0 if top_k is None else top_k, T, L().ptr(dev(e)), 0, 0, L().ptr(tok), S())
            got = int(tok.item())
            if got != want:  # only acceptable as an fp32 near-tie of p/E between the two candidates
                a, b = float(p[got] / e[got]), float(p[want] / e[want])
                assert p[got] > 0 and abs(a - b) <= 1e-5 * abs(b), (top_k, T, rep, got, want)
    # row `step` of a noise matrix is used
    logits = torch.randn(V)
    E = torch.empty(3, V).exponential_(1)
    for step in range(3):
        L().call("showo_sample_topk", L().ptr(dev(logits)), V, 7, 1.0, L().ptr(dev(E)), 0, step, L().ptr(tok), S())
        assert int(tok.item()) == _topk_reference(logits, 7, 1.0, E[step])[0]


def test_sample_topk_philox_distribution_and_support():
    """chi-square of the on-device draws against softmax over the top-k support; nothing outside the support is ever drawn"""
    torch.manual_seed(1)
    V, k, n = 64, 6, 6000
    logits = torch.randn(V) * 2
    ld = dev(logits)
    toks = torch.zeros(n, dtype=torch.int64, device="cuda")
    for i in range(n):
        L().call("showo_sample_topk", L().ptr(ld), V, k, 0.8, None, 99, i, toks.data_ptr() + 8 * i, S())
    lg = logits / 0.8
    keep = lg >= torch.topk(lg, k).values[-1]
    p = torch.where(keep, lg, torch.tensor(-float("inf"))).softmax(-1).double()
    cnt = torch.bincount(toks.cpu(), minlength=V).double()
    assert cnt[~keep].sum() == 0
    chi2 = float((((cnt - n * p) ** 2)[keep] / (n * p[keep])).sum())
    assert chi2 < 40.0, chi2  # dof = 5; P(chi2 > 40) ~ 1e-7


@pytest.mark.parametrize("n", [7, 4097, 16384, 58498, 100003])
def test_argmax_first_maximal_index(n):
    """showo_argmax_f32 (single-block and two-stage forms) == torch.argmax incl. ties (first maximal index), NaN-free input"""
    torch.manual_seed(n)
    x = torch.randn(n)
    x[torch.randint(0, n, (5,))] = float(x.max()) + 1.0  # five-way tie for the maximum
    out = torch.zeros(1, dtype=torch.int64, device="cuda")
    L().call("showo_argmax_f32", L().ptr(dev(x)), n, L().ptr(out), S())
    sync()
    assert int(out) == int(torch.nonzero(x == x.max())[0])


# ------------------------------------------------------------------------------------- CU masks, grid budgets, cooperative ownership
def test_cu_masked_stream_takes_the_same_number_of_cus_from_every_xcd():
    """showo_stream_create_cu_mask(r): a census kernel on the stream must find (32 - r / 8) distinct CUs on EVERY one of the 8 XCDs
    (the round-4 mask took all of them from XCC 0), showo_cu_usable reports 256 - r for that stream only, and destroying the
    stream withdraws the reservation"""
    lib = L()
    import ctypes as C2
    cus = C2.c_int()
    lib.call("showo_device_info", C2.byref(cus), None, None, 0)
    total = cus.value
    assert total % 8 == 0

    def census(stream_ptr, torch_stream):
        ids = torch.zeros(16 * total, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(torch_stream):
            lib.call("showo_cu_census", lib.ptr(ids), ids.numel(), 200000, stream_ptr)
        torch.cuda.synchronize()
        per = {}
        for v in ids.cpu().tolist():
            per.setdefault(v >> 16, set()).add(v & 0xffff)
        return {x: len(s) for x, s in per.items()}

    cur = torch.cuda.current_stream()
    full = census(cur.cuda_stream, cur)
    assert len(full) == 8 and all(n == total // 8 for n in full.values()), full
    assert lib.load().showo_cu_usable(C2.c_void_p(cur.cuda_stream)) == total
    with pytest.raises(RuntimeError):  # not a multiple of 8: the XCDs would lose different numbers of CUs
        h = C2.c_void_p()
        lib.check(lib.load().showo_stream_create_cu_mask(12, C2.byref(h)), "showo_stream_create_cu_mask")
    for r in (8, 16):
        h = C2.c_void_p()
        lib.check(lib.load().showo_stream_create_cu_mask(r, C2.byref(h)), "showo_stream_create_cu_mask")
        ext = torch.cuda.ExternalStream(h.value)
        got = census(h.value, ext)
        assert len(got) == 8 and all(n == total // 8 - r // 8 for n in got.values()), (r, got)
        assert lib.load().showo_cu_usable(h) == total - r and lib.load().showo_cu_usable(C2.c_void_p(cur.cuda_stream)) == total
        assert lib.load().showo_cu_reserved_max() == r
        lib.call("showo_stream_destroy", h)
        assert lib.load().showo_cu_reserved_max() == 0


def test_split_k_launches_on_two_streams_at_once_share_the_gpu_without_waiting_for_each_other():
    """two cooperative split-K launches on different streams could each be partly resident and spin until the trap.  One
    stream at a time owns the cooperative reduction, the other gets the last-arriver form -- same bits either way.  Many interleaved
    launches of the M = 631 prefill shapes on two streams: every result equals the single-stream result bit for bit, nothing hangs."""
    torch.manual_seed(3)
    M, N, K = 631, 2048, 4096
    A, W, b = torch.randn(M, K), torch.randn(N, K) * 0.02, torch.randn(N)
    Ad, Wd, bd = dev(to_bf16_bits(A)), dev(to_bf16_bits(W)), dev(b)
    ref = torch.empty((M, N), dtype=torch.float32, device="cuda")
    L().call("showo_gemm_bf16", L().ptr(Ad), K, L().ptr(Wd), K, L().ptr(bd), 0, L().ptr(ref), N, None, 0, M, N, K, 2, S())
    sync()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.empty_like(ref) for _ in range(16)]
    for i, o in enumerate(outs):
        st = s1 if i % 2 == 0 else s2
        with torch.cuda.stream(st):
            L().call("showo_gemm_bf16", L().ptr(Ad), K, L().ptr(Wd), K, L().ptr(bd), 0, L().ptr(o), N, None, 0, M, N, K, 2, st.cuda_stream)
    sync()
    for o in outs:
        assert torch.equal(o, ref)


def test_grad_clip_norm_kernel_matches_torch_and_is_reproducible():
    """showo_grad_clip_norm = torch.nn.utils.clip_grad_norm_ on a flat buffer (reference training/train.py:614-615): total norm to fp32
    accuracy of a double-precision sum, g scaled by min(max / (norm + 1e-6), 1), untouched when the coefficient is 1, same bits twice"""
    lib = L()
    torch.manual_seed(1)
    n = (1 << 22) + 7
    ws = torch.empty(lib.load().showo_grad_clip_ws_doubles(), dtype=torch.float64, device="cuda")
    for max_norm, clipped in ((5.0, True), (1e9, False)):
        g0 = torch.randn(n + 4, device="cuda")[:n] * 0.3
        runs = []
        for _ in range(2):
            g = g0.clone()
            out2 = torch.empty(2, device="cuda")
            lib.call("showo_grad_clip_norm", g.data_ptr(), n, max_norm, lib.ptr(ws), lib.ptr(out2), S())
            sync()
            runs.append((g, out2.clone()))
        total = float(g0.double().pow(2).sum().sqrt())
        assert abs(float(runs[0][1][0]) - total) <= 2e-7 * total
        coef = min(max_norm / (total + 1e-6), 1.0)
        assert abs(float(runs[0][1][1]) - coef) <= 1e-6 * coef
        if clipped:
            assert torch.allclose(runs[0][0], g0 * runs[0][1][1], rtol=0, atol=0)  # one fp32 multiply per element by the stored coefficient
        else:
            assert torch.equal(runs[0][0], g0)
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_valu_only_wave_reductions_give_the_bits_of_the_butterfly_reductions():
    """common.h: wave_sum_swap / wave_max_swap (v_permlane32/16_swap + DPP row rotations / quad permutes, no ds_bpermute) pair every
    lane with the same partner VALUES in the same order as the __shfl_xor butterfly -> identical bits, in every lane."""
    n = 4096
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 64, generator=g) * torch.logspace(-6, 6, n).reshape(n, 1)  # wide dynamic range: rounding differs by order
    x[7, 13] = float("inf"); x[8, :] = -0.0; x[9, 3] = float("-inf")
    xd = x.cuda().contiguous()
    out = torch.full((n, 4), float("nan"), device="cuda")
    L().call("showo_wave_reduce_probe", L().ptr(xd), L().ptr(out), n, S())
    sync()
    o = out.cpu()
    assert torch.equal(o[:, 0].view(torch.int32), o[:, 1].view(torch.int32))
    assert torch.equal(o[:, 2].view(torch.int32), o[:, 3].view(torch.int32))
    assert torch.equal(o[:, 2], x.max(dim=1).values)
    ref = x.double().sum(dim=1)
    mag = x.double().abs().sum(dim=1)
    fin = torch.isfinite(ref) & (mag > 0)
    assert float(((o[:, 0].double() - ref)[fin].abs() / mag[fin]).max()) < 1e-6


def test_split_conv_three_tap_reuse_kernel_tracks_fp64_and_gn_statistics():
    """gemm.hip conv3t_split_kernel (the activation operand staged once per (ky, channel chunk) for all three kx taps; launches that do
    not split K): every tile geometry -- R image rows x TW columns per 256-pixel tile for Wout = 16 ... 512 -- in modes 0 and 1, with and
    without residual, against an fp64 convolution of the original fp32 operands (3e-5 of the output scale, the split kernels' bound),
    plus the epilogue-fused GroupNorm statistics against showo_gn_stats of the stored output.  The launch counter proves coverage."""
    import torch.nn.functional as F
    torch.manual_seed(21)
    lib = L().load()
    cases = [(3, 128, 128, 64, 128, 0, True), (5, 64, 64, 128, 256, 0, False), (1, 128, 512, 64, 128, 0, True), (1, 256, 256, 64, 128, 0, False),
             (160, 16, 16, 64, 128, 0, True), (40, 32, 32, 64, 128, 0, False), (3, 64, 64, 64, 128, 1, False), (40, 16, 16, 64, 128, 1, False),
             (1, 64, 256, 128, 128, 1, False)]
    for (B, H, Wd, Cin, Cout, mode, with_res) in cases:
        Ho, Wo = (H, Wd) if mode == 0 else (2 * H, 2 * Wd)
        x = torch.randn(B, H, Wd, Cin) + 0.2
        wgt = torch.randn(Cout, 3, 3, Cin) * 0.03
        bias = torch.randn(Cout)
        res = torch.randn(B, Ho * Wo, Cout) if with_res else None
        xh, xl = _split(x)
        wh, wl = _split(wgt)
        out = torch.full((B, Ho * Wo, Cout), float("nan"), dtype=torch.float32, device="cuda")
        out1 = torch.full_like(out, float("nan"))
        nd = lib.showo_gn_stats_doubles(B, Ho * Wo)
        st1 = torch.full((nd,), float("nan"), dtype=torch.float64, device="cuda")
        st0 = torch.zeros_like(st1)
        rp = None if res is None else L().ptr(dev(res))
        n0 = lib.showo_conv3t_launches()
        L().call("showo_conv3x3_bf16x3", L().ptr(xh), L().ptr(xl), L().ptr(wh), L().ptr(wl), L().ptr(dev(bias)), rp, L().ptr(out),
                 B, H, Wd, Cin, Cout, mode, S())
        L().call("showo_conv3x3_bf16x3_gn", L().ptr(xh), L().ptr(xl), L().ptr(wh), L().ptr(wl), L().ptr(dev(bias)), rp, L().ptr(out1),
                 L().ptr(st1), B, H, Wd, Cin, Cout, mode, S())
        sync()
        assert lib.showo_conv3t_launches() == n0 + 2, (B, H, Wd, Cin, Cout, mode, "the case must reach the 3-tap-reuse kernel")
        xd = x.permute(0, 3, 1, 2).double()
        wd_ = wgt.permute(0, 3, 1, 2).double()
        if mode == 1:
            xd = xd.repeat_interleave(2, 2).repeat_interleave(2, 3)
        ref = F.conv2d(xd, wd_, bias.double(), padding=1).permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)  # CPU, fp64
        if res is not None:
            ref = ref + res.double()
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 3e-5, (B, H, Wd, Cin, Cout, mode, err)
        assert torch.equal(out, out1)
        L().call("showo_gn_stats", L().ptr(out), L().ptr(st0), B, Ho * Wo, Cout, S())
        a, b = st1[:B * 64].cpu(), st0[:B * 64].cpu()
        assert torch.isfinite(a).all() and torch.allclose(a, b, rtol=1e-12, atol=1e-9), (B, H, Wd, Cin, Cout, mode, float((a - b).abs().max()))


@pytest.mark.parametrize("polls", [0, 3])
def test_cooperative_split_k_falls_back_to_the_last_arriver_without_aborting(polls):
    """a block of a cooperative split-K launch that does not see its tile's siblings in time no longer ends in
    __builtin_trap(): the TILE switches to the last-arriver reduction (gemm_common.h splitk_coop_finish, mode word per tile) and the
    blocks leave.  With the poll budget forced to 0 / 3 almost every tile takes the fallback; the residual epilogue (in-place x += ..,
    the one where a double reduction would show) and the fp32 epilogue give the bits of the default (cooperative) launch, launch after
    launch on the same workspace (the three words of a tile are left zeroed whichever mode it took)."""
    torch.manual_seed(11)
    M, N, K = 516, 2048, 10240  # cfg1's dense|fc2 launch: 24 tiles x 10 splits, all resident
    A, W, bias, resid = torch.randn(M, K), torch.randn(N, K) * 0.03, torch.randn(N), torch.randn(M, N)
    base = (_gemm(A, W, bias, 3, resid=resid), _gemm(A, W, bias, 2))
    cnt = (C.c_int64 * 3)()
    L().call("showo_gemm_counters", cnt, 1)
    try:
        L().call("showo_gemm_set_coop_polls", polls)
        for _ in range(3):
            got = (_gemm(A, W, bias, 3, resid=resid), _gemm(A, W, bias, 2))
            assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1])
    finally:
        L().call("showo_gemm_set_coop_polls", -1)
    L().call("showo_gemm_counters", cnt, 0)
    assert cnt[2] >= 6  # the launches above did split K
    again = (_gemm(A, W, bias, 3, resid=resid), _gemm(A, W, bias, 2))  # cooperative again, same workspace: same bits
    assert torch.equal(again[0], base[0]) and torch.equal(again[1], base[1])
    ref = bf16_round(A).double() @ bf16_round(W).double().T + bias.double()
    assert (base[1].double() - ref).abs().max() < 2e-5 * float(ref.abs().max())
