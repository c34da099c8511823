"""Acceptance check of the DROP-IN claim against the reference's own, unmodified scripts (north_star: "inference_t2i.py /
inference_mmu.py / training/train.py call the new path unchanged").

The reference reaches the hot path through `from models import ...`, `from training.prompting_utils import ...` and two helpers of
`training.utils`.  This test `ast`-walks the reference's scripts where they lie (/root/reference in the build container; skipped
elsewhere -- the GPU box has no copy), collects EVERY call and attribute access on `model`, `vq_model`, `uni_prompting`,
`vision_tower`, `mask_schedule` and on the names imported from those modules, and binds each one against `showo_amd`:
`inspect.signature(...).bind` for calls (positional count + keyword names as written in the script), `getattr` chains on real
(tiny, CPU-resident) instances for attributes, literal subscripts against the instance's dict.  A renamed keyword, a dropped
attribute or a changed positional order fails here, on CPU, before any GPU run.  Nothing is executed: what the kernels COMPUTE is
pinned by the GPU parity tests; this pins the surface the scripts touch."""
import ast
import inspect
import os
import sys

import pytest
import torch

import util
from util import O

REF = "/root/reference"
SCRIPTS = ["inference_t2i.py", "inference_mmu.py", "training/train.py", "training/train_w_clip_vit.py"]
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")

# reference module -> where the name lives in this package; names of a mapped module that are control plane
# (config loading, logging meters) stay the reference's own and are listed as such
MODULE_MAP = {"models": lambda P: P, "training.prompting_utils": lambda P: P.prompting_utils,
              "training.utils": lambda P: {"mask_or_random_replace_tokens": P.training_utils, "image_transform": P.image_utils}}
CONTROL_PLANE = {"get_config", "flatten_omega_conf", "AverageMeter"}
ROOTS = ("model", "vq_model", "uni_prompting", "vision_tower")


def _instances():
    P = util.pkg()
    sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
    from stub_tokenizer import StubTokenizer
    d, sd = util.tiny_state()

    def showo(w_clip_vit):
        return P.Showo(w_clip_vit, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
                       intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)

    up = P.UniversalPrompting(StubTokenizer(), max_text_len=12, special_tokens=("<|soi|>", "<|eoi|>", "<|sov|>", "<|eov|>", "<|t2i|>", "<|mmu|>",
                                                                                 "<|t2v|>", "<|v2v|>", "<|lvg|>"), ignore_id=-100,
                              cond_dropout_prob=0.1)
    return {"model": showo(True), "vq_model": P.MAGVITv2(ch=32, max_batch=1, max_res=64), "uni_prompting": up,
            "vision_tower": P.CLIPVisionTower}  # the tower needs a checkpoint to instantiate: its surface is checked on the class


def _chain(node):
    """Attribute / Call / Subscript chain -> (root name, [steps]); steps are ('attr', name) | ('call', ast.Call) | ('item', key)"""
    steps = []
    while True:
        if isinstance(node, ast.Attribute):
            steps.append(("attr", node.attr)); node = node.value
        elif isinstance(node, ast.Call):
            steps.append(("call", node)); node = node.func
        elif isinstance(node, ast.Subscript):
            key = node.slice.value if isinstance(node.slice, ast.Constant) else None
            steps.append(("item", key)); node = node.value
        elif isinstance(node, ast.Name):
            return node.id, steps[::-1]
        else:
            return None, []


def _bind(fn, call, where, extra_kwargs=None):
    """bind the call as written: positional count, keyword names; `*x` / `**x` arguments make the bind partial"""
    if isinstance(fn, torch.nn.Module):
        fn = fn.forward
    sig = inspect.signature(fn)
    star = any(isinstance(a, ast.Starred) for a in call.args)
    kw = {k.arg: None for k in call.keywords if k.arg is not None}
    dstar = any(k.arg is None for k in call.keywords)
    kw.update(extra_kwargs or {})
    try:
        if star or (dstar and extra_kwargs is None):
            sig.bind_partial(*[None] * sum(not isinstance(a, ast.Starred) for a in call.args), **kw)
        else:
            sig.bind(*[None] * len(call.args), **kw)
    except TypeError as ex:
        return f"{where}: {getattr(fn, '__qualname__', fn)}{sig} does not accept the call as written ({ex})"
    return None


def _walk_script(path, objs, P):
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    imported = {}  # local name -> object in this package
    problems, seen = [], []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module in MODULE_MAP:
            target = MODULE_MAP[node.module](P)
            for a in node.names:
                if a.name in CONTROL_PLANE:
                    continue
                home = target.get(a.name) if isinstance(target, dict) else target
                if home is None or not hasattr(home, a.name):
                    problems.append(f"{path}:{node.lineno}: `from {node.module} import {a.name}` has no counterpart in showo_amd")
                else:
                    imported[a.asname or a.name] = getattr(home, a.name)
    # every maximal chain rooted at a role variable or an imported name
    inner = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Attribute, ast.Call, ast.Subscript)):
            child = node.value if isinstance(node, (ast.Attribute, ast.Subscript)) else node.func
            inner.add(id(child))
    for node in ast.walk(tree):
        if not isinstance(node, (ast.Attribute, ast.Call, ast.Subscript)) or id(node) in inner:
            continue
        root, steps = _chain(node)
        if root is None or not steps:
            continue
        where = f"{path}:{node.lineno}"
        if root in objs:
            obj = objs[root]
        elif root in imported:
            obj = imported[root]
        else:
            continue
        label = root
        class_stand_in = inspect.isclass(obj) and root in ROOTS  # vision_tower: checked on the class, methods take an explicit self
        if root == "vq_model" and steps[0][0] == "call":  # `vq_model = vq_model()` (training/train.py:183): the role variable is the CLASS there
            obj = type(obj)
        for kind, what in steps:
            if kind == "attr":
                if label.endswith(".text_tokenizer"):  # the HF tokenizer is third-party: its own surface is not ours to provide
                    obj = None
                    break
                if what == "module":  # DDP / accelerate wrapper (training/train.py:454): the wrapped module is the model itself
                    continue
                if root == "vq_model" and what in ("from_pretrained",):  # `vq_model` is first the CLASS (get_vq_model_class), then the instance
                    obj = type(obj) if not inspect.isclass(obj) else obj
                if not hasattr(obj, what):
                    problems.append(f"{where}: {label}.{what} does not exist on {type(obj).__name__ if not inspect.isclass(obj) else obj.__name__}")
                    obj = None
                    break
                obj = getattr(obj, what)
                label += "." + what
            elif kind == "call":
                extra = None
                if obj is P.Showo and any(k.arg is None for k in what.keywords):  # Showo(**config.model.showo): the keys of the reference's yaml
                    import yaml
                    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "showo_pretraining_stage1.yaml")))
                    extra = {k: None for k in cfg["model"]["showo"]}
                target = obj.__init__ if inspect.isclass(obj) else obj
                if inspect.isclass(obj):
                    # bind against __init__ without self
                    sig_fn = lambda *a, __f=obj, **k: None  # noqa: E731
                    try:
                        sig = inspect.signature(obj)
                        npos = sum(not isinstance(a, ast.Starred) for a in what.args)
                        kw = {k.arg: None for k in what.keywords if k.arg is not None}
                        kw.update(extra or {})
                        partial = any(isinstance(a, ast.Starred) for a in what.args) or (any(k.arg is None for k in what.keywords) and extra is None)
                        (sig.bind_partial if partial else sig.bind)(*[None] * npos, **kw)
                    except TypeError as ex:
                        problems.append(f"{where}: {obj.__name__}{sig} does not accept the call as written ({ex})")
                    seen.append((path, node.lineno, label + "()"))
                    obj = None
                    break
                if class_stand_in and inspect.isfunction(target):  # unbound method of the stand-in class: supply self
                    import functools
                    target = functools.partial(target, None)
                err = _bind(target, what, where, extra)
                if err:
                    problems.append(err)
                seen.append((path, node.lineno, label + "()"))
                obj = None  # the value a call returns is not followed (tensors / tuples)
                break
            elif kind == "item":
                if isinstance(obj, dict) and what is not None and what not in obj:
                    problems.append(f"{where}: {label}[{what!r}] is not a key (have {sorted(obj)[:6]}...)")
                obj = None
                break
        else:
            seen.append((path, node.lineno, label))
    return problems, seen


def test_every_reference_call_site_binds_against_showo_amd():
    P = util.pkg()
    objs = _instances()
    problems, seen = [], []
    for s in SCRIPTS:
        p, n = _walk_script(s, objs, P)
        problems += p
        seen += n
    labels = [x[2] for x in seen]
    print(f"[callsites] {len(seen)} call sites / attribute chains of the reference's scripts bound against showo_amd; distinct: {sorted(set(labels))}")
    assert not problems, "\n".join(problems)
    # the walker found the surface it is meant to guard (line lists of VERDICT r4 "missing" #4)
    per_script = {s: [x[2] for x in seen if x[0] == s] for s in SCRIPTS}
    assert per_script["inference_t2i.py"].count("model.t2i_generate()") == 3
    assert per_script["inference_t2i.py"].count("vq_model.decode_code()") >= 2 and "vq_model.get_code()" in per_script["inference_t2i.py"]
    assert per_script["inference_mmu.py"].count("model.mmu_generate()") == 2
    assert "model.mm_projector()" in per_script["inference_mmu.py"] and "model.showo.model.embed_tokens()" in per_script["inference_mmu.py"]
    assert "vision_tower()" in per_script["inference_mmu.py"]
    for s in ("training/train.py", "training/train_w_clip_vit.py"):
        assert "model()" in per_script[s] and "mask_or_random_replace_tokens()" in per_script[s] and "uni_prompting()" in per_script[s], s
        assert "model.save_pretrained()" in per_script[s] and "model.showo.model.embed_tokens.weight.dtype" in per_script[s], s
    assert len(seen) >= 120


def test_noise_schedule_and_prompting_objects_behave_as_the_scripts_use_them():
    """the few places where the scripts rely on VALUES of host-side objects rather than on signatures: `mask_schedule` is called with a
    tensor of ratios (models/modeling_showo.py:169), `len(uni_prompting.text_tokenizer)` offsets the image tokens
    (inference_t2i.py:112), `uni_prompting.sptids_dict[...]` feeds int() (inference_t2i.py:120-122)"""
    P = util.pkg()
    objs = _instances()
    sched = P.get_mask_chedule("cosine")
    r = sched(torch.tensor([0.0, 0.5, 1.0]))
    assert r.shape == (3,) and float(r[0]) == pytest.approx(1.0) and float(r[2]) == pytest.approx(0.0, abs=1e-6)
    up = objs["uni_prompting"]
    assert isinstance(len(up.text_tokenizer), int)
    for k in ("<|pad|>", "<|soi|>", "<|eoi|>", "<|mmu|>", "<|sot|>", "<|eot|>", "<|t2i|>"):
        assert int(up.sptids_dict[k]) >= 0
    m = objs["model"]
    assert m.config.mask_token_id == m.mask_token_id == m.vocab_size - 1 and m.output_size == m.vocab_size
    assert m.showo.model.embed_tokens.weight.dtype == torch.float32


def test_the_walker_reports_a_broken_surface(monkeypatch):
    """non-vacuity: with a keyword of `mmu_generate` renamed, an attribute of the model dropped and a special token missing, the walk
    names the reference lines that would break"""
    P = util.pkg()
    objs = _instances()
    monkeypatch.setattr(P.Showo, "mmu_generate", lambda self, idx=None, input_embeddings=None, mask=None, max_new_tokens=100: None)
    objs["uni_prompting"].sptids_dict.pop("<|soi|>")
    problems, _ = _walk_script("inference_mmu.py", objs, P)
    text = "\n".join(problems)
    assert "inference_mmu.py:146" in text and "inference_mmu.py:169" in text and "attention_mask" in text  # both mmu_generate call sites
    assert "<|soi|>" in text
    vq = objs["vq_model"]
    monkeypatch.delattr(type(vq), "decode_code")
    problems, _ = _walk_script("inference_t2i.py", objs, P)
    assert sum("vq_model.decode_code does not exist" in p for p in problems) >= 2
