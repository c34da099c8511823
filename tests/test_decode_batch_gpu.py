"""GPU parity tests of the BATCHED KV-cached decode (csrc/decode_batch.hip; BASELINE cfg4 "batch=4 images").

The reference's Showo.mmu_generate is batch-1 (models/modeling_showo.py:204,229) and inference_mmu.py:87-177 walks the images one by
one, so the specification of `mmu_generate_batch` is: n calls of `mmu_generate`.  Every sequence of a batch runs the arithmetic of its
batch-1 run (same lane split, accumulation order and epilogue expressions per sequence), so the bar is BIT-EXACT: equal tokens, and equal
logits bits at the C ABI."""
import numpy as np
import pytest
import torch

import util
from util import O, dev

pytestmark = pytest.mark.gpu


def _prompts(d, g, n):
    """n mmu prompts of different lengths derived from the golden one: [<mmu>, <soi>, image tokens, <eoi>, text ...]"""
    rs = np.random.RandomState(7)
    base = g["ids"][0].tolist()
    out = []
    for b in range(n):
        extra = rs.randint(5, 200, size=3 * b + (b % 2)).tolist()
        row = base[:len(base) - (b % 4 if b % 2 else 0)] + extra  # the text tail behind <eoi> is 6 tokens long
        out.append(torch.tensor([row], dtype=torch.int64))
    return out


@pytest.mark.parametrize("precision", [0, 2])  # 2 (round 6): fp16 instances of the batched kernels + the fused split-bf16 head
@pytest.mark.parametrize("n", [2, 3, 4, 8])
def test_tiny_mmu_generate_batch_equals_n_single_calls(n, precision):
    g = util.golden("showo_tiny_mmu.npz")
    d, sd = util.tiny_state()
    m = util.build_showo(d, sd)
    m.set_precision(precision)
    ids = _prompts(d, g, n)
    assert len({t.shape[1] for t in ids}) > 1  # ragged: every sequence has its own length, position and mask row
    masks = [O.mask_mmu(t, d.eoi_id).cuda() for t in ids]
    ids = [t.cuda() for t in ids]
    single = [[int(t) for t in m.mmu_generate(ids[b], attention_mask=masks[b], max_new_tokens=40, top_k=1)] for b in range(n)]
    assert single[0][:len(g["tokens"])] == g["tokens"].tolist()  # sequence 0 is the reference's own prompt
    for graph in (0, 1):  # eager steps == hipGraph replay, across the 16-step chunk boundaries
        m.decode_graph = graph
        got = m.mmu_generate_batch(idx=ids, attention_mask=masks, max_new_tokens=40, top_k=1)
        assert [[int(t) for t in r] for r in got] == single, graph
    # on-device visibility intervals instead of dense masks
    P = util.pkg().prompting_utils
    ivs = [P.intervals_for_mmu(t, eoi_id=d.eoi_id) for t in ids]
    got = m.mmu_generate_batch(idx=ids, attention_mask=ivs, max_new_tokens=40, top_k=1)
    assert [[int(t) for t in r] for r in got] == single
    # <eot>: every sequence stops right after ITS first <eot> (like n separate calls), the others keep going
    eot = single[1][5]
    want = [m.mmu_generate(ids[b], attention_mask=masks[b], max_new_tokens=40, top_k=1, eot_token=eot) for b in range(n)]
    got = m.mmu_generate_batch(idx=ids, attention_mask=masks, max_new_tokens=40, top_k=1, eot_token=eot)
    assert [[int(t) for t in r] for r in got] == [[int(t) for t in r] for r in want]
    assert len(got[1]) == single[1].index(eot) + 1
    # sampling / accuracy mode fall back to n sequential calls (same results as calling mmu_generate)
    m.set_precision(1)
    gp = m.mmu_generate_batch(idx=ids[:2], attention_mask=masks[:2], max_new_tokens=6, top_k=1)
    assert [[int(t) for t in r] for r in gp] == [[int(t) for t in m.mmu_generate(ids[b], attention_mask=masks[b], max_new_tokens=6, top_k=1)] for b in range(2)]


@pytest.mark.parametrize("precision", [0, 2])
def test_batch_decode_logits_are_the_bits_of_the_batch1_run(precision):
    """C ABI: showo_engine_batch_prefill / _batch_decode_greedy against showo_engine_prefill / _decode_greedy, logits workspace compared
    bit for bit after the same number of steps; capacity / order errors are refused"""
    g = util.golden("showo_tiny_mmu.npz")
    d, sd = util.tiny_state()
    m = util.build_showo(d, sd)
    m.set_precision(precision)
    L = util.lib()
    eng = m.engine()
    n, steps = 3, 5
    ids = [t.cuda() for t in _prompts(d, g, n)]
    masks = [O.mask_mmu(t.cpu(), d.eoi_id).cuda().float().contiguous() for t in ids]
    V = d.vocab
    side = torch.cuda.Stream()
    ref_logits, ref_tokens = [], []
    for b in range(n):
        lg = torch.empty((V,), dtype=torch.float32, device="cuda")
        L.call("showo_engine_prefill", eng, L.ptr(ids[b]), None, L.ptr(masks[b]), ids[b].shape[1], L.ptr(lg), L.stream())
        tok = lg.argmax().reshape(1).to(torch.int64)
        out = torch.empty((steps,), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            L.call("showo_engine_decode_greedy", eng, L.ptr(tok), steps, L.ptr(out), L.ptr(lg), 1, L.stream())
        torch.cuda.synchronize()
        ref_logits.append(lg.clone()), ref_tokens.append(out.tolist())
    with pytest.raises(RuntimeError):  # no batch yet
        L.call("showo_engine_batch_prefill", eng, 0, L.ptr(ids[0]), None, L.ptr(masks[0]), ids[0].shape[1], None, L.stream())
    cap = max(t.shape[1] for t in ids) + steps + 1
    L.call("showo_engine_batch_begin", eng, n, cap)
    lgs = torch.empty((n, V), dtype=torch.float32, device="cuda")
    tok = torch.empty((n,), dtype=torch.int64, device="cuda")
    with pytest.raises(RuntimeError):  # decode before every sequence has a prefill
        L.call("showo_engine_batch_decode_greedy", eng, L.ptr(tok), 1, L.ptr(tok), L.ptr(lgs), 0, L.stream())
    for b in range(n):
        L.call("showo_engine_batch_prefill", eng, b, L.ptr(ids[b]), None, L.ptr(masks[b]), ids[b].shape[1], L.ptr(lgs[b]), L.stream())
    tok.copy_(lgs.argmax(dim=1))
    out = torch.empty((n, steps), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        L.call("showo_engine_batch_decode_greedy", eng, L.ptr(tok), steps, L.ptr(out), L.ptr(lgs), 1, L.stream())
    torch.cuda.synchronize()
    assert out.tolist() == ref_tokens
    for b in range(n):
        assert torch.equal(lgs[b], ref_logits[b]), (b, float((lgs[b] - ref_logits[b]).abs().max()))
    big = torch.empty((n, 4096), dtype=torch.int64, device="cuda")
    with pytest.raises(RuntimeError):  # more steps than the caches hold
        L.call("showo_engine_batch_decode_greedy", eng, L.ptr(tok), 4096, L.ptr(big), L.ptr(lgs), 0, L.stream())
    with pytest.raises(RuntimeError):
        L.call("showo_engine_batch_begin", eng, 9, 64)
