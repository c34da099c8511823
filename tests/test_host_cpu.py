"""CPU: host-side logic and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import util
from util import O, Wt


def test_library_exports_every_declared_symbol():
    L = util.lib()
    lib = L.load()
    hdr = open(os.path.join(util.ROOT, "include", "showo_hip.h")).read()
    declared = set(re.findall(r"\b(showo_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"showo_hip"}
    assert len(declared) > 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/showo_hip.h but not exported"
    assert set(L.EXPORTED_SYMBOLS) == declared, (set(L.EXPORTED_SYMBOLS) ^ declared)
    assert lib.showo_abi_version() == 1


def test_no_cpu_fallback():
    """the product path must fail loudly without a GPU"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d, sd_np = util.tiny_state()
    S = util.pkg().Showo
    m = S(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
          intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 27, dtype=torch.long))
    with pytest.raises(RuntimeError):
        util.pkg().MAGVITv2().decode_code(torch.zeros(1, 16, dtype=torch.long))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(util.ROOT, "show-o_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "showo_oracle" not in src and "ref_loader" not in src and "/root/reference" not in src, f
    # tools/ and the secondary bench workloads never touch the oracle either; bench.py / bench_train.py only inside cpu_baseline()
    for f in [os.path.join("tools", n) for n in os.listdir(os.path.join(util.ROOT, "tools")) if n.endswith((".py", ".cpp"))] + ["bench_configs.py"]:
        src = open(os.path.join(util.ROOT, f)).read()
        assert "showo_oracle" not in src and "import weights" not in src and "/root/reference" not in src, f
    for f in ("bench.py", "bench_train.py"):
        src = open(os.path.join(util.ROOT, f)).read()
        assert src.count("import showo_oracle") == 1 and src.index("import showo_oracle") > src.index("def cpu_baseline"), f


def test_state_dict_keys_match_reference_layout():
    d, sd_np = util.tiny_state()
    S = util.pkg().Showo
    m = S(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
          intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)
    assert list(m.state_dict().keys()) == list(sd_np.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == sd_np[k].shape, k
    m.load_state_dict(O.to_torch(sd_np), strict=True)
    assert m.mask_token_id == d.vocab - 1 == m.config.mask_token_id
    # full-size key list = reference's: 24*18 + 5 tensors
    full = Wt.ShowoDims()
    assert full.vocab == 58498 and full.image_offset == 50305 and full.mask_token_id == 58497
    v = util.pkg().MAGVITv2()
    ref_keys = Wt.make_magvit_state  # generator mirrors the reference's names (validated in make_golden.py strict load)
    sd_v = ref_keys(seed=1)
    assert set(v.state_dict().keys()) == set(sd_v.keys())
    for k, t in v.state_dict().items():
        assert tuple(t.shape) == sd_v[k].shape, k


def test_schedule_constants_match_reference_golden():
    g = util.golden("showo_tiny_t2i.npz")
    S = util.pkg().sampling
    ml, tp = S.t2i_step_constants(int(g["steps"]), 16, 1.0, S.get_mask_chedule("cosine"))
    assert np.array_equal(np.array(ml), g["mask_len"])
    assert np.array_equal(np.array(tp, dtype=np.float32), g["temps"].astype(np.float32))
    # last step: cos(pi/2) < 0 in fp32 -> floor -> -1 (reference quirk)
    ml18, tp18 = S.t2i_step_constants(18, 256)
    assert ml18[-1] == -1.0 and tp18[-1] == 0.0
    with pytest.raises(ValueError):
        S.get_mask_chedule("nope")
    for name in ("linear", "pow2", "sigmoid"):
        f = S.get_mask_chedule(name)
        assert 0.0 <= float(f(torch.tensor(0.3))) <= 1.0


# ---------------------------------------------------------------------------------------------------------------
# UniversalPrompting: sequence layouts against the real reference (fixture made by oracle/make_golden.py with the stub
# tokenizer of oracle/stub_tokenizer.py)
# ---------------------------------------------------------------------------------------------------------------
def _prompting_fixture():
    import json
    from stub_tokenizer import StubTokenizer
    g = util.golden("prompting.npz")
    N = g["image_ids"].shape[1]
    up = util.pkg().UniversalPrompting(StubTokenizer(), max_text_len=12, max_seq_len=12 + N + 3, cond_dropout_prob=0.5)
    texts = json.loads(str(g["texts"]))
    return g, up, texts, torch.from_numpy(g["image_ids"]), torch.from_numpy(g["labels"])


def _same(g, name, tup):
    for k, t in zip(("seq", "mask", "lab"), tup):
        ref = g[f"{name}_{k}"]
        assert tuple(t.shape) == ref.shape, (name, k, tuple(t.shape), ref.shape)
        assert t.dtype == torch.int64 and np.array_equal(t.numpy(), ref), (name, k)


def test_universal_prompting_special_ids_and_attributes():
    import json
    g, up, *_ = _prompting_fixture()
    assert {k: int(v) for k, v in up.sptids_dict.items()} == json.loads(str(g["sptids"]))
    assert up.pad_id == int(g["pad_id"]) and up.max_text_len == int(g["max_text_len"]) == 13
    assert up.ignore_id == -100 and up.cond_dropout_prob == 0.5


@pytest.mark.parametrize("task", ["t2i", "t2v", "lvg"])
def test_universal_prompting_training_layouts_with_condition_dropout(task):
    g, up, texts, img, lab = _prompting_fixture()
    torch.manual_seed(21)
    _same(g, task, up((list(texts), img, lab), task))
    # the host generator is left exactly where the reference leaves it (one rand(B) per call, two for lvg)
    assert np.array_equal(torch.rand(3).numpy(), g[f"{task}_rng_after"])


def test_universal_prompting_generation_lm_mmu_layouts():
    g, up, texts, img, lab = _prompting_fixture()
    N = img.shape[1]
    for task in ("t2i_gen", "t2v_gen", "lvg_gen"):
        _same(g, task, up((list(texts), img), task))
    _same(g, "lm", up((list(texts), 12 + N + 3), "lm"))
    _same(g, "lm_short", up((list(texts), 9), "lm"))  # longer texts are cut WITHOUT re-adding <eos>
    _same(g, "mmu", up((img, list(texts)), "mmu"))
    torch.manual_seed(22)
    cfg = type("C", (), {"training": type("T", (), {"batch_size": 4})})
    a, b = up((list(texts), img[:4], lab[:4], 31), "t2i_plus_lm", config=cfg)
    _same(g, "plus_t2i", a)
    _same(g, "plus_lm", b)
    with pytest.raises(NotImplementedError):
        up((texts, img), "no_such_task")


def test_universal_prompting_edge_cases():
    g, up, texts, img, lab = _prompting_fixture()
    bos, eos = up.text_tokenizer.bos_token_id, up.text_tokenizer.eos_token_id
    seq, _ = up.t2i_gen_prompt([[]], img[:1])           # empty prompt -> [pad.. t2i bos eos]
    assert seq[0, :13].tolist() == [up.pad_id] * 10 + [int(up.sptids_dict['<|t2i|>']), bos, eos]
    ids = [[5] * 40]
    seq, _ = up.t2i_gen_prompt(ids, img[:1])            # too long -> first window-1 ids, then <eos>; bos added in place
    assert ids[0][0] == bos and seq[0, 12] == eos and seq[0, 0] == int(up.sptids_dict['<|t2i|>'])
    assert seq.shape[1] == 13 + 1 + img.shape[1] + 1
    assert seq[0, 13] == int(up.sptids_dict['<|soi|>']) and seq[0, -1] == int(up.sptids_dict['<|eoi|>'])


# ---------------------------------------------------------------------------------------------------------------
# checkpoint format I/O (reference models/modeling_utils.py:47-49, 270-414, 416-865)
# ---------------------------------------------------------------------------------------------------------------
def _tiny_showo_cpu():
    d, sd_np = util.tiny_state()
    S = util.pkg().Showo
    m = S(d.w_clip_vit, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
          intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)
    m.load_state_dict(O.to_torch(sd_np), strict=True)
    return d, m


@pytest.mark.parametrize("safe,shard", [(True, "10GB"), (False, "10GB"), (True, "200KB")])
def test_save_pretrained_from_pretrained_roundtrip(tmp_path, safe, shard):
    import json
    d, m = _tiny_showo_cpu()
    out = str(tmp_path / "checkpoint-10" / "unwrapped_model")  # the layout training/train.py:851-889 writes
    m.save_pretrained(out, safe_serialization=safe, max_shard_size=shard)
    cfg = json.load(open(os.path.join(out, "config.json")))
    assert cfg["_class_name"] == "Showo" and cfg["vocab_size"] == d.vocab and cfg["llm_vocab_size"] == d.llm_vocab
    assert cfg["codebook_size"] == d.codebook and cfg["num_vq_tokens"] == d.num_vq_tokens and cfg["w_clip_vit"] is False
    files = sorted(os.listdir(out))
    if shard == "10GB":
        assert files == ["config.json", "pytorch_model.safetensors" if safe else "pytorch_model.bin"]
    else:
        assert "pytorch_model.safetensors.index.json" in files and sum(f.endswith(".safetensors") for f in files) > 2
    m2 = util.pkg().Showo.from_pretrained(out, device="cpu", max_batch=2, max_seq=64)
    assert m2.max_batch == 2 and not m2.training and m2.mask_token_id == d.vocab - 1
    a, b = m.state_dict(), m2.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # a second save in the other format replaces nothing it should not, and loading prefers safetensors
    m2.save_pretrained(out, safe_serialization=not safe)
    assert "config.json" in os.listdir(out)


def test_from_pretrained_reads_reference_written_files_and_reports_bad_ones(tmp_path):
    """a directory written the way the reference's ModelMixin does (torch.save of the state dict + its config.json keys)"""
    import json
    d, m = _tiny_showo_cpu()
    out = str(tmp_path / "ref_style")
    os.makedirs(out)
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(out, "pytorch_model.bin"))
    json.dump({"_class_name": "Showo", "_diffusers_version": "0.30.1", "codebook_size": d.codebook, "llm_model_path": "",
               "llm_vocab_size": d.llm_vocab, "load_from_showo": True, "num_vq_tokens": d.num_vq_tokens, "vocab_size": d.vocab,
               "w_clip_vit": False}, open(os.path.join(out, "config.json"), "w"))
    geo = dict(hidden_size=d.hidden, intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)
    m2 = util.pkg().Showo.from_pretrained(out, device="cpu", **geo)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    sd = m.state_dict()
    first = next(iter(sd))
    torch.save({k: v for k, v in sd.items() if k != first}, os.path.join(out, "pytorch_model.bin"))
    with pytest.raises(KeyError):
        util.pkg().Showo.from_pretrained(out, device="cpu", **geo)
    torch.save({k: (v[:1] if k == first else v) for k, v in sd.items()}, os.path.join(out, "pytorch_model.bin"))
    with pytest.raises(ValueError):
        util.pkg().Showo.from_pretrained(out, device="cpu", **geo)
    with pytest.raises(EnvironmentError):
        util.pkg().Showo.from_pretrained(str(tmp_path / "nope"))


def test_magvitv2_config_roundtrip(tmp_path):
    V = util.pkg().MAGVITv2
    with torch.device("meta"):
        v = V()
    v.save_config(str(tmp_path))
    import json
    assert json.load(open(tmp_path / "config.json"))["_class_name"] == "MAGVITv2"
    assert V.load_config(str(tmp_path))["_class_name"] == "MAGVITv2"


def test_clip_tower_loads_a_local_checkpoint_directory(tmp_path):
    """config.json of a full CLIPConfig (vision part nested) + model.safetensors with text-tower tensors next to the vision ones"""
    import json
    from safetensors.torch import save_file
    sd = {k: torch.from_numpy(v) for k, v in Wt.make_clip_state(Wt.CLIP_TINY, seed=3).items()}
    sd["text_model.final_layer_norm.weight"] = torch.ones(8)
    save_file(sd, str(tmp_path / "model.safetensors"))
    json.dump({"model_type": "clip", "vision_config": dict(Wt.CLIP_TINY, model_type="clip_vision_model")}, open(tmp_path / "config.json", "w"))
    t = util.pkg().CLIPVisionTower(str(tmp_path))
    assert t.is_loaded and t.vision_tower_name == str(tmp_path) and t.hidden_size == 128 and t.num_patches == 16
    keys = list(t.state_dict())
    assert keys[0] == "vision_tower.vision_model.embeddings.class_embedding" and len(keys) == 5 + 3 * 16 + 2
    assert torch.equal(t.state_dict()["vision_tower.vision_model.encoder.layers.1.mlp.fc2.weight"], sd["vision_model.encoder.layers.1.mlp.fc2.weight"])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            t(torch.zeros(1, 3, 56, 56))
    with pytest.raises(EnvironmentError):
        util.pkg().CLIPVisionTower(str(tmp_path / "missing"))


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """the drop-in boundary is a C ABI: include/showo_hip.h compiles as C99 (no C++-isms), and a C program linked against
    libshowo_hip.so resolves the entry points and gets an error code + message (not a crash) for a bad call without a GPU"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(util.ROOT, "include", "showo_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include "showo_hip.h"\n'
                   'int main(void) {\n'
                   '  if (showo_abi_version() != 1) return 2;\n'
                   '  int rc = showo_sample_topk(0, 0, 0, 1.0f, 0, 0, 0, 0, 0);   /* null arguments: refused, no launch */\n'
                   '  printf("%d %s\\n", rc, showo_last_error());\n'
                   '  return rc != 0 ? 0 : 3;\n}\n')
    lib = util.lib().LIB_PATH
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.dirname(hdr), str(src), "-o", str(exe), lib,
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "sample_topk" in out.stdout


def test_checkpoint_rotation_and_resume_layout(tmp_path):
    """training/train.py:851-889 / 429-443: checkpoint-N/unwrapped_model/pytorch_model.bin + metadata.json, oldest removed before
    saving so that at most `checkpoints_total_limit` remain, resume picks the highest N"""
    import json
    ck = util.pkg().checkpointing
    d, m = _tiny_showo_cpu()
    out = str(tmp_path / "run")
    assert ck.resume_from_checkpoint(m, out) == 0 and ck.latest_checkpoint(out) == (None, 0)
    for step in (10, 20, 30, 40):
        with torch.no_grad():
            m.showo.model.final_layernorm.bias.fill_(float(step))
        p = ck.save_checkpoint(m, out, step, checkpoints_total_limit=2)
        assert sorted(os.listdir(p)) == ["metadata.json", "unwrapped_model"]
        assert sorted(os.listdir(os.path.join(p, "unwrapped_model"))) == ["config.json", "pytorch_model.bin"]
        assert json.load(open(os.path.join(p, "metadata.json"))) == {"global_step": step}
    assert sorted(os.listdir(out), key=lambda x: int(x.split("-")[1])) == ["checkpoint-30", "checkpoint-40"]
    _, m2 = _tiny_showo_cpu()
    assert ck.resume_from_checkpoint(m2, out) == 40
    assert float(m2.showo.model.final_layernorm.bias[0].detach()) == 40.0
    for (n, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n
    assert ck.save_checkpoint(m, out, 50, is_main_process=False) is None and not os.path.exists(os.path.join(out, "checkpoint-50"))


def test_ctypes_prototypes_match_header_argument_counts_and_kinds():
    """every binding in show-o_amd/_lib.py has as many arguments as the C declaration, pointers where the header has pointers,
    and the right scalar kind (int / int64 / float / uint64) elsewhere: catches ABI drift between include/*.h and the host side"""
    import ctypes as C
    L = util.lib()
    hdr = open(os.path.join(util.ROOT, "include", "showo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decls = re.findall(r"\b(?:int|void|const char\*)\s+(showo_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(decls) > 100
    protos = dict(L._PROTOS)
    protos.update(L._VOID)
    checked = 0
    for name, args in decls:
        if name not in protos:
            continue
        params = [a.strip() for a in args.replace("\n", " ").split(",")] if args.strip() not in ("", "void") else []
        got = protos[name]
        assert len(got) == len(params), (name, len(got), params)
        for p, ct in zip(params, got):
            is_ptr = "*" in p
            if is_ptr:
                assert ct in (C.c_void_p, C.c_char_p) or issubclass(ct, C._Pointer), (name, p, ct)
            elif re.match(r"(const\s+)?(int64_t|long long)\b", p):
                assert ct is C.c_int64, (name, p, ct)
            elif re.match(r"(const\s+)?(uint64_t|unsigned long long)\b", p):
                assert ct is C.c_uint64, (name, p, ct)
            elif re.match(r"(const\s+)?uint32_t\b", p):
                assert ct is C.c_uint32, (name, p, ct)
            elif re.match(r"(const\s+)?float\b", p):
                assert ct is C.c_float, (name, p, ct)
            elif re.match(r"(const\s+)?int\b", p):
                assert ct is C.c_int, (name, p, ct)
            else:
                raise AssertionError(f"{name}: unrecognised parameter type {p!r}")
        checked += 1
    assert checked > 95


def test_decode_knob_setters_validate_their_arguments_without_a_gpu():
    """showo_decode_set_tuning / showo_decode_set_prefetch are host-side state (they select grids of later launches): valid knobs are
    accepted, unknown names and out-of-range values are refused with an error code and a message -- no GPU needed, nothing launched"""
    L = util.lib()
    lib = L.load()
    for name, value in (("co_blocks", 128), ("batch_co_blocks", 128), ("batch_ln_blocks", 1024), ("ln_blocks", 1024), ("out_blocks", 256)):
        assert lib.showo_decode_set_tuning(name.encode(), value) == 0  # (these are the defaults: state unchanged)
    assert lib.showo_decode_set_tuning(b"no_such_knob", 8) != 0 and b"unknown knob" in lib.showo_last_error()
    assert lib.showo_decode_set_tuning(b"co_blocks", 0) != 0
    assert lib.showo_decode_set_tuning(None, 8) != 0
    assert lib.showo_decode_set_prefetch(-1, 0, 0) != 0 and lib.showo_decode_set_prefetch(0, 0, 5000) != 0
    assert lib.showo_decode_set_prefetch(0, 0, 0) == 0
    assert lib.showo_conv3t_launches() == 0  # no convolution was launched in this process


def test_alias_package_shares_module_objects():
    """`showo_amd.x` and `show-o_amd.x` are the same module objects (one instance of every class, so isinstance checks hold no
    matter which spelling user code imports from)"""
    import importlib
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import importlib, showo_amd\n"
            "from showo_amd.prompting_utils import IntervalMask as A\n"
            "import showo_amd.sampling as ls\n"
            "real = importlib.import_module('show-o_amd.prompting_utils')\n"
            "assert A is real.IntervalMask and ls is importlib.import_module('show-o_amd.sampling')\n"
            "assert showo_amd.Showo.__module__ == 'show-o_amd.modeling_showo'\n"
            "print('ok')\n") % util.ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout, out.stderr[-2000:])


def test_load_from_showo_false_initialises_from_a_local_phi_checkpoint(tmp_path):
    """reference models/modeling_showo.py:45-46: PhiForCausalLM.from_pretrained(llm_model_path) + resize_token_embeddings.
    A tiny synthetic HF-layout Phi checkpoint (keys model.* / lm_head.*, vocabulary 96, no q/k LayerNorms like phi-1_5):
    every tensor arrives under the showo. prefix, the first 96 embedding / lm_head rows are the checkpoint's, the q/k
    LayerNorms are freshly initialised (1, 0), and a missing directory raises instead of training from noise."""
    import json
    import showo_amd
    from safetensors.torch import save_file
    H, F, nL, V0, V1 = 128, 256, 2, 96, 140
    g = torch.Generator().manual_seed(3)
    sd = {"model.embed_tokens.weight": torch.randn(V0, H, generator=g), "lm_head.weight": torch.randn(V0, H, generator=g),
          "lm_head.bias": torch.randn(V0, generator=g), "model.final_layernorm.weight": torch.randn(H, generator=g),
          "model.final_layernorm.bias": torch.randn(H, generator=g)}
    for i in range(nL):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "dense"):
            sd[p + f"self_attn.{n}.weight"] = torch.randn(H, H, generator=g)
            sd[p + f"self_attn.{n}.bias"] = torch.randn(H, generator=g)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = torch.randn(F, H, generator=g), torch.randn(F, generator=g)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = torch.randn(H, F, generator=g), torch.randn(H, generator=g)
        sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"] = torch.randn(H, generator=g), torch.randn(H, generator=g)
    d = tmp_path / "phi-tiny"
    d.mkdir()
    save_file(sd, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps({"model_type": "phi", "hidden_size": H, "intermediate_size": F, "num_hidden_layers": nL,
                                               "num_attention_heads": 2, "vocab_size": V0}))
    kw = dict(hidden_size=H, intermediate_size=F, num_hidden_layers=nL, num_attention_heads=2)
    m = showo_amd.Showo(False, V1, V0, llm_model_path=str(d), load_from_showo=False, **kw)
    got = m.state_dict()
    for k, v in sd.items():
        if k in ("model.embed_tokens.weight", "lm_head.weight", "lm_head.bias"):
            assert torch.equal(got["showo." + k][:V0], v) and got["showo." + k].shape[0] == V1
        else:
            assert torch.equal(got["showo." + k], v), k
    assert float(got["showo.lm_head.bias"][V0:].abs().max()) == 0.0          # new bias entries: zero
    assert 0.005 < float(got["showo.model.embed_tokens.weight"][V0:].std()) < 0.05  # new rows: N(0, 0.02)
    for i in range(nL):
        for n in ("q_layernorm", "k_layernorm"):
            assert bool((got[f"showo.model.layers.{i}.self_attn.{n}.weight"] == 1).all())
            assert bool((got[f"showo.model.layers.{i}.self_attn.{n}.bias"] == 0).all())
    with pytest.raises(EnvironmentError):
        showo_amd.Showo(False, V1, V0, llm_model_path="microsoft/phi-1_5", load_from_showo=False, **kw)
    with pytest.raises(ValueError):  # geometry mismatch between config.json and the constructor
        showo_amd.Showo(False, V1, V0, llm_model_path=str(d), load_from_showo=False, hidden_size=H, intermediate_size=F,
                        num_hidden_layers=nL + 1, num_attention_heads=2)
    # a Show-o checkpoint saved with load_from_showo=False in its config.json reloads without the Phi directory
    out = tmp_path / "showo-ckpt"
    m.save_pretrained(str(out))
    (d / "model.safetensors").unlink()
    m2 = showo_amd.Showo.from_pretrained(str(out), device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_bench_gpus_flag_fails_loudly_without_enough_gpus():
    """`python bench.py --gpus N` self-launches N ranks; with fewer visible GPUs it must refuse instead of running one replica
    and reporting n_gpus = 1 (reference launch: accelerate, one process per GPU, training/train.py:91-110)."""
    import subprocess
    import sys
    ROOT = util.ROOT
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    for script in ("bench.py", "bench_train.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "--gpus 64 requested" in (r.stderr + r.stdout), (script, r.stderr[-500:])
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_training_leg_child_launch(monkeypatch):
    """bench.py's `train_step` leg runs bench_train.py as a child of every rank: own rendezvous port (MASTER_PORT + 101, the
    launcher's agent store not inherited), bounded by a timeout, only rank 0 reports, a failing child yields an error object
    instead of taking the headline line down"""
    import json
    import subprocess
    import sys
    sys.path.insert(0, util.ROOT)
    import bench
    seen = {}

    class FakeProc:
        returncode = 0

        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"] = cmd, env

        def communicate(self, timeout=None):
            seen["timeout"] = timeout
            line = {"metric": "train step-time", "value": 200.0, "n_gpus": 2, "steps": 4, "warmup": 2, "scaling": "weak",
                    "config": {"global_batch": 58, "tokens_per_s": 1.0, "gradient_wire": "bf16"}, "roofline": {"achieved": 1000.0, "frac": 0.4}}
            return "noise\n" + json.dumps(line) + "\n", ""

        def kill(self):
            seen["killed"] = True

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setenv("MASTER_PORT", "29511")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    got = bench.train_leg(2, 0)
    assert seen["env"]["MASTER_PORT"] == "29612" and "TORCHELASTIC_USE_AGENT_STORE" not in seen["env"]
    assert seen["cmd"][1].endswith("bench_train.py") and seen["cmd"][2:4] == ["--gpus", "2"] and "--no-cpu-baseline" in seen["cmd"]
    assert seen["timeout"] and got["ms_per_step"] == 200.0 and got["n_gpus"] == 2 and got["gradient_wire"] == "bf16"
    assert bench.train_leg(2, 1) is None  # only rank 0 reports
    FakeProc.returncode = 3
    assert "error" in bench.train_leg(2, 0)

    class Hung(FakeProc):
        def communicate(self, timeout=None):
            if timeout:
                raise subprocess.TimeoutExpired("x", timeout)
            return "", ""

    monkeypatch.setattr(subprocess, "Popen", Hung)
    assert "timed out" in bench.train_leg(2, 0)["error"] and seen.get("killed")


def test_measurement_tools_parse_their_inputs(tmp_path):
    """tools/power_trace.py (sysfs / rocm-smi parsing, summary statistics) and tools/pmc_summary.py (per-kernel HBM traffic with the
    gfx950 FETCH_SIZE correction, GEMM family grouped over gemm2p + gemm3w) on synthetic inputs"""
    import csv
    import importlib.util
    import json
    import subprocess
    import sys

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(util.ROOT, "tools", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    pt = load("power_trace")
    assert pt.dpm_current("0: 500Mhz\n1: 2381Mhz *\n2: 2400Mhz") == 2381 and pt.dpm_current("S: 117Mhz *\n0: 500Mhz") == 117
    assert pt.dpm_current(None) is None and pt.dpm_current("0: 500Mhz") is None
    st = pt.stats([float(i) for i in range(1, 101)])
    assert st["n"] == 100 and st["min"] == 1.0 and st["max"] == 100.0 and st["median"] == 51.0 and abs(st["mean"] - 50.5) < 1e-9
    out = tmp_path / "trace.json"
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "power_trace.py"), str(out), "--", sys.executable, "-c", "print(1)"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.load(open(out))["summary"]["rc"] == 0

    fd, wd = tmp_path / "f", tmp_path / "w"
    for d, counter, vals in ((fd, "FETCH_SIZE", (1000.0, 3000.0, 500.0)), (wd, "WRITE_SIZE", (100.0, 300.0, 50.0))):
        os.makedirs(d)
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writerow(["void showo::(anonymous namespace)::gemm2p_kernel<3, 5, 4, true>(showo::GemmArgs)", counter, vals[0]])
            w.writerow(["void showo::(anonymous namespace)::gemm3w_kernel<4, 6, 6, true, false>(showo::GemmArgs)", counter, vals[1]])
            w.writerow(["void (anonymous namespace)::attn_fwd_lds_kernel<4, 4>((anonymous namespace)::AttnArgs)", counter, vals[2]])
    res = tmp_path / "traffic.json"
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tools", "pmc_summary.py"), "tag", str(fd), str(wd), str(res)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    t = json.load(open(res))
    g = t["gemm2p_kernel"]  # both kernels of the family, launch-weighted; fetch = 2 x KiB x 1024, write = KiB x 1024
    assert g["launches"] == 2 and abs(g["fetch_bytes_per_launch"] - 2 * 1024 * 2000.0) < 1e-6 and abs(g["write_bytes_per_launch"] - 1024 * 200.0) < 1e-6
    assert abs(t["attn_fwd_lds_kernel"]["bytes_per_launch"] - (2 * 1024 * 500.0 + 1024 * 50.0)) < 1e-6


def test_build_script_compiles_every_kernel_file():
    """a new .hip file that is not in build.sh's list would link an old library silently (the GPU box runs what was built here)"""
    import glob
    import re
    csrc = os.path.join(util.ROOT, "show-o_amd", "csrc")
    sh = open(os.path.join(csrc, "build.sh")).read()
    m = re.search(r"for f in ([^;]+); do", sh)
    assert m, "build.sh: file list not found"
    listed = set(m.group(1).split())
    present = {os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(csrc, "*.hip"))}
    assert present == listed, (sorted(present - listed), sorted(listed - present))
