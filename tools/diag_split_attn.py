"""diagnostic (GPU): which ingredient of the dense-mask case moves showo_attn_fwd_split away from the fp64 SDPA"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import util
from util import O
import test_precise_gpu as T

torch.manual_seed(258)
Lq, nH, B = 258, 2, 2
q = torch.randn(B, nH, Lq, 64) * 0.25
k = torch.randn(B, nH, Lq, 64)
v = torch.randn(B, nH, Lq, 64)
vis = torch.rand(B, 1, Lq, Lq) < 0.5
vis |= torch.eye(Lq, dtype=torch.bool)[None, None]
causal = torch.tril(torch.ones(Lq, Lq, dtype=torch.bool))[None, None].expand(B, 1, Lq, Lq)
cases = {}
for name, vv, soft, neg in (("random NEG", vis, False, O.NEG_MASK), ("random NEG + soft col", vis, True, O.NEG_MASK), ("random -1e4", vis, False, -1e4),
                            ("causal NEG + soft col (dense path)", causal, True, O.NEG_MASK), ("causal NEG (interval path)", causal, False, O.NEG_MASK)):
    m = torch.where(vv, torch.zeros(()), torch.full((), neg)).float().clone()
    if soft:
        m[:, :, :, 3] = -1.5
    got, flag = T._split_attn(q, k, v, m)
    s = q.double() @ k.double().transpose(2, 3) + m.double()
    want = (torch.softmax(s, dim=-1) @ v.double()).transpose(1, 2).reshape(B, Lq, nH * 64)
    d = (got - want).abs()
    print(f"{name:40s} flag={flag} rel_max={float(d.max() / want.abs().max()):.3e} abs_max={float(d.max()):.3e} worst row {int(d.max(dim=2).values.argmax() % Lq)}")

print("---- per 32-row tile: causal, dense path vs interval path (kernel vs fp64), and dense vs interval outputs of the kernel")
m_i = torch.where(causal, torch.zeros(()), torch.full((), O.NEG_MASK)).float().clone()
m_d = m_i.clone(); m_d[:, :, 0, 5] = -1.5e-30  # an "odd" value that changes nothing numerically but forces the dense path (row 0 cannot see key 5 anyway: -1.5e-30 makes it visible!)
m_d = m_i.clone(); m_d[:, :, 200, 3] = -1e-30   # visible key with a negligible soft bias -> dense path, same mathematics
g_i, f_i = T._split_attn(q, k, v, m_i)
g_d, f_d = T._split_attn(q, k, v, m_d)
s = q.double() @ k.double().transpose(2, 3) + m_i.double()
want = (torch.softmax(s, dim=-1) @ v.double()).transpose(1, 2).reshape(B, Lq, nH * 64)
print("flags", f_i, f_d)
for t in range((Lq + 31) // 32):
    sl = slice(32 * t, min(Lq, 32 * t + 32))
    print(f"rows {32*t:3d}..: interval err {float((g_i[:, sl] - want[:, sl]).abs().max()):.2e}  dense err {float((g_d[:, sl] - want[:, sl]).abs().max()):.2e}  dense-interval {float((g_d[:, sl] - g_i[:, sl]).abs().max()):.2e}")
# which head-dim columns
dcol = (g_d - want).abs().amax(dim=(0, 1)).view(nH, 64)
print("dense err by head dim (head 0):", [f"{float(x):.1e}" for x in dcol[0][::4]])

print("---- which operand loses its low half in the dense path?  dense-kernel output vs references with ONE operand rounded to bf16")
from util import bf16_round
def ref(qq, kk, vv, pround=False):
    s_ = qq.double() @ kk.double().transpose(2, 3) + m_i.double()
    p_ = torch.softmax(s_, dim=-1)
    if pround:
        mxs = s_.max(dim=-1, keepdim=True).values
        pu = torch.exp(s_ - mxs)
        p_ = bf16_round(pu.float()).double() / pu.sum(dim=-1, keepdim=True)
    return (p_ @ vv.double()).transpose(1, 2).reshape(B, Lq, nH * 64)
for name, r in (("exact", ref(q, k, v)), ("V -> bf16", ref(q, k, bf16_round(v))), ("K -> bf16", ref(q, bf16_round(k), v)), ("Q -> bf16", ref(bf16_round(q), k, v)),
                ("P -> bf16", ref(q, k, v, True)), ("Q,K,V -> bf16", ref(bf16_round(q), bf16_round(k), bf16_round(v))),
                ("all -> bf16", ref(bf16_round(q), bf16_round(k), bf16_round(v), True))):
    print(f"  dense kernel vs [{name:14s}]: {float((g_d - r).abs().max()):.2e}    interval kernel: {float((g_i - r).abs().max()):.2e}")
print("rows 0..3 dense err:", [f"{float((g_d[:, i] - want[:, i]).abs().max()):.1e}" for i in range(4)], " |v - bf16(v)| max row 0:", float((v[:, :, 0] - bf16_round(v[:, :, 0])).abs().max()))

print("---- full visibility: interval path (zero mask) vs dense path (zero mask + one -1e-30 entry), split kernel; and the bf16 kernel on the causal pair")
z_i = torch.zeros(B, 1, Lq, Lq)
z_d = z_i.clone(); z_d[:, :, 200, 3] = -1e-30
a_i, fi = T._split_attn(q, k, v, z_i)
a_d, fd = T._split_attn(q, k, v, z_d)
s0 = q.double() @ k.double().transpose(2, 3)
w0 = (torch.softmax(s0, dim=-1) @ v.double()).transpose(1, 2).reshape(B, Lq, nH * 64)
print(f"  flags {fi} {fd}: interval err {float((a_i - w0).abs().max()):.2e}  dense err {float((a_d - w0).abs().max()):.2e}")
import test_kernels_gpu as TK
from util import to_bf16_bits, dev
Qb = dev(to_bf16_bits(q)); Kb = dev(to_bf16_bits(k))
Vtb = torch.zeros((B, nH, 64, ((Lq + 63) // 64) * 64), dtype=torch.int16, device="cuda"); Vtb[..., :Lq] = to_bf16_bits(v.transpose(2, 3).contiguous()).cuda()
TK.L().call("showo_attn_set_impl", 2)
o_i, _ = TK._attn(Qb, Kb, Vtb, B, nH, Lq, Lq, m_i)
o_d, (_, fl) = TK._attn(Qb, Kb, Vtb, B, nH, Lq, Lq, m_d)
print(f"  bf16 kernel, causal: dense flag {fl}; max |dense - interval| = {float((o_d - o_i).abs().max()):.2e} (one bf16 ulp of the output is ~8e-3 at |o| ~ 2)")
