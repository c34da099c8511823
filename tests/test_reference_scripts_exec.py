"""EXECUTES the bodies of the reference's own, unmodified `inference_t2i.py` (modes t2i, inpainting, extrapolation) and
`inference_mmu.py` (both branches: VQ tokens in the prompt / CLIP embeddings spliced in) against this package -- the second half of the
acceptance check that `tests/test_reference_callsites.py` starts (which binds every call site statically).

`runpy` runs the script where it lies (/root/reference; never copied) with the bindings INTEGRATION.md section 1 describes:
`models` -> showo_amd (Showo, MAGVITv2, get_mask_chedule), `training.prompting_utils` -> showo_amd.prompting_utils,
`training.utils.image_transform` -> showo_amd.image_utils; and with the control plane stubbed: `get_config` returns a dict namespace
with the keys of configs/showo_demo.yaml the script reads (OmegaConf is not installed), `wandb` records what is logged,
`AutoTokenizer.from_pretrained` returns the deterministic stub tokenizer (no tokenizer files offline), `from_pretrained` of the two
models builds tiny random-weight instances (no checkpoints offline).

Two forms of the same run:
  * CPU (runs in the build container, where the reference is): the three GPU entry points the script reaches -- the mask builder,
    `Showo.t2i_generate`, `MAGVITv2.decode_code` -- are replaced by shape-checking stand-ins (the product has no CPU path by design);
    everything else is the real package: class surface, attributes, `UniversalPrompting`, schedules, argument plumbing.  The stand-in of
    `t2i_generate` binds the script's call against the REAL signature first and checks every tensor's shape / dtype.
  * GPU (`-m gpu`): the same run on the real kernels; it needs a GPU AND the reference tree, which this setup never has on one machine
    (the GPU box has no /root/reference), so it is skipped there and kept for a host that has both."""
import inspect
import math
import os
import runpy
import sys
import types

import numpy as np
import pytest
import torch

import util
from util import O

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


class NS(dict):
    """dict with attribute access and .get(): what the scripts use of an OmegaConf node"""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _ns(d):
    return NS({k: _ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def _run_inference_t2i(tmp_path, monkeypatch, on_gpu, mode="t2i"):
    P = util.pkg()
    sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
    from stub_tokenizer import StubTokenizer
    d, sd = util.tiny_state()
    prompts = ["a red cube on a table", "two dogs", "the sea at night", "a", "green hills far away"]
    pf = tmp_path / "prompts.txt"
    pf.write_text("\n".join(prompts))
    batch = 2
    from PIL import Image
    rs_img = np.random.RandomState(1)
    img_path, mask_path = tmp_path / "image.png", tmp_path / "mask.png"
    Image.fromarray(rs_img.randint(0, 255, (80, 64, 3), dtype=np.uint8)).save(img_path)
    mk = np.zeros((64, 64), dtype=np.uint8)
    mk[:, :32] = 255  # repaint the left half
    Image.fromarray(mk).save(mask_path)
    config = _ns({
        "wandb": {"resume": False}, "experiment": {"name": "acceptance"}, "mode": mode,
        "prompt": "a red cube" if mode == "inpainting" else "a red cube *** a blue sphere", "image_path": str(img_path),
        "inpainting_mask_path": str(mask_path), "extra_direction": "left *** right", "offset": 0,
        "model": {"showo": {"llm_model_path": "stub", "pretrained_model_path": "stub", "num_vq_tokens": d.num_vq_tokens,
                            "codebook_size": d.codebook, "llm_vocab_size": d.llm_vocab, "num_new_special_tokens": d.num_new_special_tokens,
                            "w_clip_vit": False},
                  "vq_model": {"type": "magvitv2", "vq_model_name": "stub"}},
        "dataset": {"preprocessing": {"max_seq_length": d.max_text_len}, "params": {"validation_prompts_file": str(pf), "resolution": 64}},
        "training": {"cond_dropout_prob": 0.1, "generation_temperature": 1.0},
        "validation_prompts_file": str(pf), "batch_size": batch, "guidance_scale": 1.75, "generation_timesteps": 3,
    })
    record = {"wandb_init": [], "wandb_log": [], "t2i": [], "decode": [], "mask": [], "masked": [], "get_code": 0, "transform": []}

    # ---- control plane stubs
    import importlib.machinery

    def module(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)  # importlib.util.find_spec(name) is asked by third-party integrations
        return m

    wandb = module("wandb")
    wandb.util = types.SimpleNamespace(generate_id=lambda: "run0")
    wandb.init = lambda **kw: record["wandb_init"].append(kw)
    wandb.log = lambda data, step=None: record["wandb_log"].append((step, data))
    wandb.Image = lambda image, caption=None: (np.asarray(image), caption)
    monkeypatch.setitem(sys.modules, "wandb", wandb)
    import transformers
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: StubTokenizer()))

    def flatten(cfg, resolve=True, prefix=""):
        for k, v in cfg.items():
            if isinstance(v, dict):
                yield from flatten(v, resolve, prefix + k + ".")
            else:
                yield prefix + k, v

    # ---- the bindings of INTEGRATION.md section 1
    models = module("models")
    models.Showo, models.MAGVITv2, models.get_mask_chedule = P.Showo, P.MAGVITv2, P.get_mask_chedule
    training = module("training")
    training.__path__ = []
    tpu = module("training.prompting_utils")
    tpu.UniversalPrompting = P.UniversalPrompting
    tpu.create_attention_mask_predict_next = P.prompting_utils.create_attention_mask_predict_next
    tu = module("training.utils")
    tu.get_config = lambda: config
    tu.flatten_omega_conf = flatten
    tu.image_transform = P.image_utils.image_transform
    for name, mod in (("models", models), ("training", training), ("training.prompting_utils", tpu), ("training.utils", tu)):
        monkeypatch.setitem(sys.modules, name, mod)

    # ---- no checkpoints offline: from_pretrained builds tiny random-weight instances
    def showo_from_pretrained(cls, path, *a, **k):
        m = P.Showo(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
                    intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads, max_batch=2 * batch, max_seq=64)
        m.load_state_dict(O.to_torch(sd), strict=True)
        return m

    monkeypatch.setattr(P.Showo, "from_pretrained", classmethod(showo_from_pretrained))
    monkeypatch.setattr(P.MAGVITv2, "from_pretrained", classmethod(lambda cls, name, *a, **k: P.MAGVITv2(ch=32, max_batch=batch, max_res=64)))

    if not on_gpu:
        monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
        real_sig = inspect.signature(P.Showo.t2i_generate)
        L = d.max_text_len + 1 + 1 + d.num_vq_tokens + 1

        def fake_mask(sequence, pad_id=128256, soi_id=128257, eoi_id=128258, rm_pad_in_image=False, return_inverse_mask=True):
            record["mask"].append(tuple(sequence.shape))
            assert sequence.dtype == torch.int64 and sequence.shape[1] == L
            return O.mask_t2i(sequence, pad_id, soi_id, eoi_id, rm_pad_in_image=rm_pad_in_image)

        def fake_t2i(self, *args, **kw):
            b = real_sig.bind(self, *args, **kw)  # the script's call must bind against the REAL signature
            b.apply_defaults()
            a = b.arguments
            ids, unc, am = a["input_ids"], a["uncond_input_ids"], a["attention_mask"]
            B = ids.shape[0]
            assert ids.dtype == torch.int64 and tuple(ids.shape) == (B, L) and tuple(unc.shape) == (B, L)
            assert tuple(am.shape) == (2 * B, 1, L, L) and am.dtype == torch.float32
            record["masked"].append(int((ids == self.mask_token_id).sum()))  # t2i: every image slot; inpainting / extrapolation: the masked region
            assert int((unc == self.mask_token_id).sum()) == record["masked"][-1]
            assert a["timesteps"] == 3 and a["guidance_scale"] == 1.75 and callable(a["noise_schedule"]) and a["config"] is config
            assert float(a["noise_schedule"](torch.tensor(0.0))) == pytest.approx(1.0)
            record["t2i"].append(B)
            return torch.randint(0, d.codebook, (B, d.num_vq_tokens))

        def fake_decode(self, code, shape=None):
            inspect.signature(real_decode).bind(self, code, shape=shape)
            side = int(math.isqrt(d.num_vq_tokens))
            h, w = shape if shape is not None else (side, side)
            assert code.dtype == torch.int64 and code.shape[1] == h * w and int(code.max()) < d.codebook and int(code.min()) >= 0
            record["decode"].append(code.shape[0] if shape is None else (code.shape[0], h, w))
            return torch.rand(code.shape[0], 3, 16 * h, 16 * w) * 2 - 1

        def fake_get_code(self, x):
            side = int(math.isqrt(d.num_vq_tokens)) * 16
            assert x.dim() == 4 and tuple(x.shape[1:]) == (3, side, side) and x.dtype == torch.float32
            record["get_code"] += 1
            return torch.randint(0, d.codebook, (x.shape[0], d.num_vq_tokens))

        def fake_transform(image, resolution=256, normalize=True, **kw):
            inspect.signature(P.image_utils.image_transform).bind(image, resolution=resolution, normalize=normalize, **kw)
            ch = len(image.getbands())  # RGB image / "L" mask (inference_t2i.py:82-86)
            record["transform"].append((ch, resolution, normalize))
            arr = torch.from_numpy(np.asarray(image.resize((resolution, resolution)), dtype=np.float32) / 255.0)
            arr = arr.reshape(resolution, resolution, ch).permute(2, 0, 1)
            return (arr - 0.5) / 0.5 if normalize else arr

        real_decode = P.MAGVITv2.decode_code
        tu.image_transform = fake_transform
        monkeypatch.setattr(P.MAGVITv2, "get_code", fake_get_code)

        tpu.create_attention_mask_predict_next = fake_mask
        monkeypatch.setattr(P.Showo, "t2i_generate", fake_t2i)
        monkeypatch.setattr(P.MAGVITv2, "decode_code", fake_decode)

    monkeypatch.setattr(sys, "argv", ["inference_t2i.py"])
    runpy.run_path(os.path.join(REF, "inference_t2i.py"), run_name="__main__")
    return record, prompts, batch, d


def _check(record, prompts, batch, d):
    nb = math.ceil(len(prompts) / batch)
    assert len(record["wandb_init"]) == 1 and record["wandb_init"][0]["name"] == "acceptance_t2i_t2i"
    assert [s for s, _ in record["wandb_log"]] == list(range(0, len(prompts), batch))
    side = int(math.isqrt(d.num_vq_tokens)) * 16
    n_img = 0
    for step, data in record["wandb_log"]:
        for i, (img, caption) in enumerate(data["generated_images"]):
            assert img.shape == (side, side, 3) and img.dtype == np.uint8 and caption == prompts[step + i]
            n_img += 1
    assert n_img == len(prompts)
    return nb


def test_inference_t2i_script_runs_unchanged_against_this_package_cpu(tmp_path, monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test runs the real kernels")
    record, prompts, batch, d = _run_inference_t2i(tmp_path, monkeypatch, on_gpu=False)
    nb = _check(record, prompts, batch, d)
    assert record["t2i"] == [2, 2, 1] and record["decode"] == [2, 2, 1] and len(record["mask"]) == nb  # ragged last batch included
    assert all(s[0] == 2 * b for s, b in zip(record["mask"], record["t2i"]))  # CFG: cond + uncond rows in one mask


def test_inference_t2i_script_inpainting_mode_cpu(tmp_path, monkeypatch):
    """the other two modes of the same unmodified script: inpainting (image + mask files -> image_transform, get_code, masked tokens
    re-generated) and, below, extrapolation (two directions: the token grid grows to 4 x 8, decode_code(ids, shape=(h, w)))"""
    if torch.cuda.is_available():
        pytest.skip("CPU form")
    record, _, batch, d = _run_inference_t2i(tmp_path, monkeypatch, on_gpu=False, mode="inpainting")
    side = int(math.isqrt(d.num_vq_tokens))
    assert record["t2i"] == [batch] and record["decode"] == [batch] and record["get_code"] == 1
    assert record["transform"] == [(3, 16 * side, True), (1, 16 * side, False)]
    assert record["masked"] == [batch * d.num_vq_tokens // 2]  # the left half of every token grid
    (step, data), = record["wandb_log"]
    caps = [c for _, c in data["generated_images"]]
    assert step == 0 and caps == ["original image", "inpainting mask"] + ["a red cube"] * batch


def test_inference_t2i_script_extrapolation_mode_cpu(tmp_path, monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("CPU form")
    record, _, batch, d = _run_inference_t2i(tmp_path, monkeypatch, on_gpu=False, mode="extrapolation")
    side = int(math.isqrt(d.num_vq_tokens))
    assert record["t2i"] == [batch, batch] and record["get_code"] == 1
    assert record["masked"] == [batch * side * (side // 2)] * 2  # half a grid per step
    assert record["decode"] == [(batch, side, 2 * side)]  # left, then right: 4 x 8 tokens
    (step, data), = record["wandb_log"]
    assert all(img.shape == (16 * side, 32 * side, 3) and c == "a red cube a blue sphere" for img, c in data["generated_images"])


@pytest.mark.gpu
def test_inference_t2i_script_runs_unchanged_against_this_package_gpu(tmp_path, monkeypatch):
    """the same run on the real kernels (needs a GPU and the reference tree on one machine)"""
    record, prompts, batch, d = _run_inference_t2i(tmp_path, monkeypatch, on_gpu=True)
    _check(record, prompts, batch, d)


# ------------------------------------------------------------------------------------------------ inference_mmu.py
class _Enc(dict):
    """what a HF tokenizer call returns as far as the script uses it: `.input_ids` and `['input_ids']`"""
    @property
    def input_ids(self):
        return self["input_ids"]


def _mmu_tokenizer():
    sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
    from stub_tokenizer import StubTokenizer

    class Tok(StubTokenizer):
        """the stub tokenizer with the call forms inference_mmu.py uses (return_tensors="pt", padding=..., list or str input,
        batch_decode); the system prompt maps to 28 ids like under the real Phi-1.5 tokenizer (the script asserts it)"""
        def __call__(self, texts, truncation=False, return_tensors=None, padding=None, **kw):
            single = isinstance(texts, str)
            rows = super().__call__(texts, truncation=truncation)["input_ids"]
            if single and texts.startswith("A chat between a curious user"):
                rows = [(rows[0] + [7] * 28)[:28]]
            if return_tensors == "pt":
                width = max(len(r) for r in rows)
                rows = torch.tensor([r + [self.pad_token_id or 0] * (width - len(r)) for r in rows], dtype=torch.int64)
            return _Enc(input_ids=rows)

        def batch_decode(self, rows, skip_special_tokens=True):
            return [" ".join(f"w{int(t)}" for t in r) for r in rows]

    return Tok()


def _run_inference_mmu(tmp_path, monkeypatch, w_clip_vit):
    from PIL import Image
    P = util.pkg()
    d, sd = util.tiny_state()
    root = tmp_path / "imgs"
    root.mkdir()
    rs = np.random.RandomState(0)
    for name, (h, w) in (("a.png", (70, 90)), ("b.png", (64, 64))):
        Image.fromarray(rs.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(root / name)
    NEW, RES = 5, 64
    config = _ns({
        "wandb": {"resume": False}, "experiment": {"name": "acceptance"},
        "model": {"showo": {"llm_model_path": "stub", "pretrained_model_path": "stub", "w_clip_vit": w_clip_vit},
                  "vq_model": {"type": "magvitv2", "vq_model_name": "stub"}},
        "dataset": {"preprocessing": {"max_seq_length": d.max_text_len}, "params": {"resolution": RES}},
        "training": {"cond_dropout_prob": 0.1},
        "mmu_image_root": str(root), "question": "what is in the picture ? *** how many objects ?", "max_new_tokens": NEW,
    })
    record = {"wandb_log": [], "get_code": 0, "mmu": [], "clip": 0, "proj": 0, "transform": 0, "mask": 0}

    import importlib.machinery

    def module(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        return m

    wandb = module("wandb")
    wandb.util = types.SimpleNamespace(generate_id=lambda: "run0")
    wandb.init = lambda **kw: None
    wandb.log = lambda data, step=None: record["wandb_log"].append((step, data))
    wandb.Image = lambda image, caption=None: (np.asarray(image), caption)
    monkeypatch.setitem(sys.modules, "wandb", wandb)
    import transformers
    tok = _mmu_tokenizer()
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: tok))
    proc = types.SimpleNamespace(preprocess=lambda image, return_tensors=None: {"pixel_values": torch.zeros(1, 3, 336, 336)})
    monkeypatch.setattr(transformers.CLIPImageProcessor, "from_pretrained", staticmethod(lambda *a, **k: proc))
    monkeypatch.syspath_prepend(REF)  # `from llava.llava import conversation` comes from the reference tree itself

    def flatten(cfg, resolve=True, prefix=""):
        for k, v in cfg.items():
            if isinstance(v, dict):
                yield from flatten(v, resolve, prefix + k + ".")
            else:
                yield prefix + k, v

    n_vq = d.num_vq_tokens
    HID = d.hidden

    class Tower(torch.nn.Module):  # constructor / .to() / call form of showo_amd.CLIPVisionTower (its weights are files: none offline)
        def __init__(self, name, *a, **k):
            super().__init__()
            inspect.signature(P.CLIPVisionTower.__init__).bind(self, name, *a, **k)
            assert name == "openai/clip-vit-large-patch14-336"

        def forward(self, pixels):
            assert tuple(pixels.shape) == (1, 3, 336, 336)
            record["clip"] += 1
            return torch.zeros(1, 576, 1024)

    def fake_transform(image, resolution=256, **kw):
        inspect.signature(P.image_utils.image_transform).bind(image, resolution=resolution, **kw)
        assert hasattr(image, "convert") and resolution == RES  # a PIL image, the configured resolution
        record["transform"] += 1
        return torch.rand(3, resolution, resolution) * 2 - 1

    def fake_mask_mmu(sequence, eoi_id=128258, return_inverse_mask=True):
        inspect.signature(P.prompting_utils.create_attention_mask_for_mmu).bind(sequence, eoi_id=eoi_id)
        record["mask"] += 1
        return O.mask_mmu(sequence.cpu(), eoi_id)

    def fake_mask_vit(sequence, return_inverse_mask=True, system_prompt_len=0):
        inspect.signature(P.prompting_utils.create_attention_mask_for_mmu_vit).bind(sequence, system_prompt_len=system_prompt_len)
        assert sequence.dim() == 3 and system_prompt_len == 28
        record["mask"] += 1
        n = sequence.shape[1]
        return torch.zeros(sequence.shape[0], 1, n, n)

    models = module("models")
    models.Showo, models.MAGVITv2, models.CLIPVisionTower = P.Showo, P.MAGVITv2, Tower
    training = module("training")
    training.__path__ = []
    tpu = module("training.prompting_utils")
    tpu.UniversalPrompting = P.UniversalPrompting
    tpu.create_attention_mask_for_mmu, tpu.create_attention_mask_for_mmu_vit = fake_mask_mmu, fake_mask_vit
    tu = module("training.utils")
    tu.get_config, tu.flatten_omega_conf, tu.image_transform = (lambda: config), flatten, fake_transform
    for name, mod in (("models", models), ("training", training), ("training.prompting_utils", tpu), ("training.utils", tu)):
        monkeypatch.setitem(sys.modules, name, mod)

    def showo_from_pretrained(cls, path, *a, **k):
        m = P.Showo(w_clip_vit, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=n_vq, hidden_size=HID,
                    intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads, max_batch=2, max_seq=64)
        return m

    monkeypatch.setattr(P.Showo, "from_pretrained", classmethod(showo_from_pretrained))
    monkeypatch.setattr(P.MAGVITv2, "from_pretrained", classmethod(lambda cls, name, *a, **k: P.MAGVITv2(ch=32, max_batch=1, max_res=RES)))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    sig_gen = inspect.signature(P.Showo.mmu_generate)

    def fake_get_code(self, x):
        assert tuple(x.shape) == (1, 3, RES, RES) and x.dtype == torch.float32
        record["get_code"] += 1
        return torch.randint(0, d.codebook, (1, n_vq))

    def fake_mmu_generate(self, *args, **kw):
        a = sig_gen.bind(self, *args, **kw)
        a.apply_defaults()
        a = a.arguments
        assert a["max_new_tokens"] == NEW and a["top_k"] == 1 and a["eot_token"] is not None
        if w_clip_vit:
            emb, am = a["input_embeddings"], a["attention_mask"]
            assert a["idx"] is None and emb.dim() == 3 and emb.shape[0] == 1 and emb.shape[2] == HID
            assert emb.shape[1] > 1 + 28 + 1 + 576 + 1 and tuple(am.shape) == (1, 1, emb.shape[1], emb.shape[1])
        else:
            ids, am = a["idx"], a["attention_mask"]
            assert ids.dtype == torch.int64 and ids.shape[0] == 1 and tuple(am.shape[-2:]) == (ids.shape[1], ids.shape[1])
            assert int(ids[0, 0]) == int(self_prompting["<|mmu|>"]) and ids.shape[1] > n_vq + 4
        record["mmu"].append(1)
        return [torch.tensor([11 + j]) for j in range(NEW)]

    self_prompting = {}
    real_up_init = P.UniversalPrompting.__init__

    def up_init(self, *a, **k):
        real_up_init(self, *a, **k)
        self_prompting.update({k_: int(v) for k_, v in self.sptids_dict.items()})

    monkeypatch.setattr(P.UniversalPrompting, "__init__", up_init)
    monkeypatch.setattr(P.MAGVITv2, "get_code", fake_get_code)
    monkeypatch.setattr(P.Showo, "mmu_generate", fake_mmu_generate)
    if w_clip_vit:
        class Proj(torch.nn.Module):  # the projector's call form (its arithmetic is a HIP kernel: GPU tests)
            def forward(self, feats):
                assert tuple(feats.shape) == (1, 576, 1024)
                record["proj"] += 1
                return torch.zeros(1, 576, HID)
        real_init = P.Showo.__init__

        def init_with_cpu_projector(self, *a, **k):
            real_init(self, *a, **k)
            assert hasattr(self, "mm_projector")  # the attribute the script reaches for exists on the real class
            self.mm_projector = Proj()

        monkeypatch.setattr(P.Showo, "__init__", init_with_cpu_projector)
    monkeypatch.setattr(sys, "argv", ["inference_mmu.py"])
    runpy.run_path(os.path.join(REF, "inference_mmu.py"), run_name="__main__")
    return record


@pytest.mark.parametrize("w_clip_vit", [False, True])
def test_inference_mmu_script_runs_unchanged_against_this_package_cpu(tmp_path, monkeypatch, w_clip_vit):
    """the body of the reference's unmodified inference_mmu.py, both branches (VQ tokens in the prompt / CLIP embeddings spliced in):
    two images x two questions.  GPU entry points are shape-checking stand-ins bound against the real signatures (see the module docstring)."""
    if torch.cuda.is_available():
        pytest.skip("CPU form")
    record = _run_inference_mmu(tmp_path, monkeypatch, w_clip_vit)
    assert len(record["mmu"]) == 4 and record["get_code"] == 2 and record["transform"] == 2 and record["mask"] == 4
    assert (record["clip"], record["proj"]) == ((4, 4) if w_clip_vit else (0, 0))
    (step, data), = record["wandb_log"]
    assert step == 0 and len(data["multimodal understanding"]) == 2
    for img, caption in data["multimodal understanding"]:
        assert img.shape == (64, 64, 3) and img.dtype == np.uint8
        assert caption.count("User: ") == 2 and "w11 w12 w13 w14 w15" in caption  # both questions answered with the decoded tokens
