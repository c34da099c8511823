"""`Showo` — drop-in replacement of the reference class `models.modeling_showo.Showo`
(reference models/modeling_showo.py:23-240) backed by the gfx950 engine in libshowo_hip.so.

Same constructor arguments, attributes (`config.mask_token_id`, `mask_token_id`, `vocab_size`,
`output_size`, `showo.model.embed_tokens`, `showo.resize_token_embeddings`, `mm_projector`) and state-dict
keys as the reference, so `inference_t2i.py` / `inference_mmu.py` call it unchanged.
The torch modules below are *parameter containers only* (they give `state_dict()` / `load_state_dict()`
the reference's names and shapes); all arithmetic runs in HIP kernels through the C ABI.  There is no
PyTorch fallback: without a GPU or without the built library every compute entry point raises.
"""
import types

import torch
import torch.nn as nn

from . import _lib
from .persistence import PretrainedMixin
from .sampling import cosine_schedule, t2i_step_constants

# microsoft/phi-1_5 architecture numbers (what AutoConfig.from_pretrained(llm_model_path) yields in the
# reference, models/modeling_showo.py:42; PhiConfig defaults)
PHI_1_5 = dict(hidden_size=2048, intermediate_size=8192, num_hidden_layers=24, num_attention_heads=32,
               max_position_embeddings=2048, layer_norm_eps=1e-5, rope_theta=10000.0, partial_rotary_factor=0.5)


class _Cfg(dict):
    """attribute-style config like the reference's diffusers FrozenDict (`model.config.mask_token_id`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def get(self, k, default=None):
        return dict.get(self, k, default)


class _PhiAttentionParams(nn.Module):
    def __init__(self, hidden, head_dim, eps):
        super().__init__()
        self.q_proj = nn.Linear(hidden, hidden, bias=True)
        self.k_proj = nn.Linear(hidden, hidden, bias=True)
        self.v_proj = nn.Linear(hidden, hidden, bias=True)
        self.dense = nn.Linear(hidden, hidden, bias=True)
        self.q_layernorm = nn.LayerNorm(head_dim, eps=eps)  # forced on by PhiForCausalLM (reference models/phi.py:1088)
        self.k_layernorm = nn.LayerNorm(head_dim, eps=eps)


class _PhiMLPParams(nn.Module):
    def __init__(self, hidden, ffn):
        super().__init__()
        self.fc1 = nn.Linear(hidden, ffn)
        self.fc2 = nn.Linear(ffn, hidden)


class _PhiLayerParams(nn.Module):
    def __init__(self, hidden, ffn, head_dim, eps):
        super().__init__()
        self.self_attn = _PhiAttentionParams(hidden, head_dim, eps)
        self.mlp = _PhiMLPParams(hidden, ffn)
        self.input_layernorm = nn.LayerNorm(hidden, eps=eps)


class _PhiModelParams(nn.Module):
    def __init__(self, vocab, hidden, ffn, layers, head_dim, eps):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, hidden)
        self.layers = nn.ModuleList([_PhiLayerParams(hidden, ffn, head_dim, eps) for _ in range(layers)])
        self.final_layernorm = nn.LayerNorm(hidden, eps=eps)


class _PhiForCausalLMParams(nn.Module):
    """Names mirror reference models/phi.py:1084-1095 (`model`, untied biased `lm_head`)."""

    def __init__(self, vocab, hidden, ffn, layers, head_dim, eps):
        super().__init__()
        self.model = _PhiModelParams(vocab, hidden, ffn, layers, head_dim, eps)
        self.lm_head = nn.Linear(hidden, vocab, bias=True)
        self.vocab_size = vocab
        for m in self.modules():  # Phi init: N(0, 0.02), zero biases (reference models/phi.py:833-842)
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=0.02)

    def resize_token_embeddings(self, new_vocab):
        """reference: PreTrainedModel.resize_token_embeddings called at models/modeling_showo.py:46."""
        old = self.model.embed_tokens.weight.shape[0]
        if new_vocab == old:
            return self.model.embed_tokens
        hidden = self.model.embed_tokens.weight.shape[1]
        dev = self.model.embed_tokens.weight.device
        emb = nn.Embedding(new_vocab, hidden, device=dev)
        head = nn.Linear(hidden, new_vocab, bias=True, device=dev)
        nn.init.normal_(emb.weight, std=0.02)
        nn.init.normal_(head.weight, std=0.02)
        nn.init.zeros_(head.bias)
        n = min(old, new_vocab)
        with torch.no_grad():
            emb.weight[:n] = self.model.embed_tokens.weight[:n]
            head.weight[:n] = self.lm_head.weight[:n]
            head.bias[:n] = self.lm_head.bias[:n]
        self.model.embed_tokens, self.lm_head, self.vocab_size = emb, head, new_vocab
        return emb

    def get_input_embeddings(self):
        return self.model.embed_tokens


class _ShowoTrainFn(torch.autograd.Function):
    """Showo.forward with labels (reference models/modeling_showo.py:80-102) as one autograd node: forward runs the HIP
    training forward (activations stay in the trainer), backward runs the HIP backward for the incoming loss weights
    and hands the parameter gradients (reference names and shapes) back to autograd."""

    @staticmethod
    def forward(ctx, model, input_ids, input_embeddings, attention_mask, labels, b_t2i, b_lm, b_mmu, max_seq_length, *params):
        tr = model.trainer()
        lab = labels.to(torch.int64).contiguous()
        B, L = lab.shape
        if attention_mask is not None and tuple(attention_mask.shape) != (B, 1, L, L):
            raise ValueError(f"Attention mask should be of size {(B, 1, L, L)}, but is {tuple(attention_mask.shape)}")
        from .training import train_mask
        mask = train_mask(tr, attention_mask)  # dense fp32 mask, or an IntervalMask registered with the trainer
        logits = torch.empty((B, L, model.vocab_size), dtype=torch.float32, device=lab.device)
        losses = torch.empty(3, dtype=torch.float32, device=lab.device)
        try:
            if input_embeddings is None:
                ids = input_ids.to(torch.int64).contiguous()
                _lib.call("showo_train_forward", tr, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(lab), B, L, b_t2i, b_lm, b_mmu,
                          max_seq_length, _lib.ptr(logits), _lib.ptr(losses), _lib.stream())
            else:  # `inputs_embeds` flow of the w_clip_vit trainer (reference modeling_showo.py:77-78)
                emb = input_embeddings.detach().float().contiguous()
                if tuple(emb.shape) != (B, L, model.arch["hidden_size"]):
                    raise ValueError(f"input_embeddings should be {(B, L, model.arch['hidden_size'])}, got {tuple(emb.shape)}")
                _lib.call("showo_train_forward_embeds", tr, _lib.ptr(emb), _lib.ptr(mask), _lib.ptr(lab), B, L, b_t2i, b_lm, b_mmu,
                          max_seq_length, _lib.ptr(logits), _lib.ptr(losses), _lib.stream())
        finally:
            _lib.call("showo_trainer_use_intervals", tr, None, None)
        ctx.model, ctx.lab, ctx.meta = model, lab, (b_t2i, b_lm, b_mmu, max_seq_length)
        ctx.emb_shape = None if input_embeddings is None else (tuple(input_embeddings.shape), input_embeddings.dtype)
        ctx.names = ["showo." + n for n, _ in model.showo.named_parameters()]
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.mark_non_differentiable(logits)
        return logits, losses[0].clone(), losses[1].clone(), losses[2].clone()

    @staticmethod
    def backward(ctx, g_logits, g_t2i, g_lm, g_mmu):
        model = ctx.model
        g = [0.0 if x is None else float(x) for x in (g_t2i, g_lm, g_mmu)]
        b_t2i, b_lm, b_mmu, msl = ctx.meta
        _lib.call("showo_train_backward", model._trainer, _lib.ptr(ctx.lab), b_t2i, b_lm, b_mmu, msl, g[0], g[1], g[2], _lib.stream())
        grads = []
        for name, shape in zip(ctx.names, ctx.shapes):
            t = torch.empty(shape, dtype=torch.float32, device=ctx.lab.device)
            _lib.call("showo_train_grad_copy", model._trainer, name.encode(), _lib.ptr(t), t.numel(), _lib.stream())
            grads.append(t)
        g_emb = None
        if ctx.emb_shape is not None and ctx.needs_input_grad[2]:
            g_emb = torch.empty(ctx.emb_shape[0], dtype=torch.float32, device=ctx.lab.device)
            _lib.call("showo_train_input_grad", model._trainer, _lib.ptr(g_emb), g_emb.numel(), _lib.stream())
            g_emb = g_emb.to(ctx.emb_shape[1])
        return (None, None, g_emb) + (None,) * 6 + tuple(grads)


class _ProjectorFn(torch.autograd.Function):
    """mm_projector as one autograd node: HIP forward, HIP backward (csrc/clip_engine.hip: showo_projector_backward).  The
    backward re-runs the (two small GEMMs of the) forward on the saved input, so interleaved calls cannot mix up saved state."""

    @staticmethod
    def forward(ctx, mod, x, w0, b0, w1, b1):
        ctx.mod = mod
        ctx.save_for_backward(x)
        return mod._hip_forward(x)

    @staticmethod
    def backward(ctx, g):
        mod, (x,) = ctx.mod, ctx.saved_tensors
        din, dout_dim = mod._dims
        mod._hip_forward(x)  # restores the engine's saved activations for exactly these rows
        gout = g.detach().float().contiguous()
        T = gout.numel() // dout_dim
        dev = gout.device
        gw0 = torch.empty((dout_dim, din), dtype=torch.float32, device=dev)
        gb0 = torch.empty((dout_dim,), dtype=torch.float32, device=dev)
        gw1 = torch.empty((dout_dim, dout_dim), dtype=torch.float32, device=dev)
        gb1 = torch.empty((dout_dim,), dtype=torch.float32, device=dev)
        dx = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        _lib.call("showo_projector_backward", mod._proj, _lib.ptr(gout), T, _lib.ptr(dx), _lib.ptr(gw0), _lib.ptr(gb0), _lib.ptr(gw1),
                  _lib.ptr(gb1), _lib.stream())
        return None, (None if dx is None else dx.to(x.dtype)), gw0, gb0, gw1, gb1


class _MMProjector(nn.Sequential):
    """`model.mm_projector` (reference modeling_showo.py:48-53): same parameters and state-dict keys (mm_projector.0.*,
    mm_projector.2.*).  Forward and backward run on the HIP projector (two MFMA GEMMs + exact GELU and their autograd,
    csrc/clip_engine.hip): inference_mmu.py:134 as well as the w_clip_vit trainer that fine-tunes these 6 M parameters."""

    def __init__(self, din, dout):
        super().__init__(nn.Linear(din, dout), nn.GELU(), nn.Linear(dout, dout))
        self._dims = (din, dout)
        self._proj, self._rows, self._versions = None, 0, None

    def set_precision(self, precision):
        """0 (default): bf16 GEMM operands.  1: accuracy mode of the projector -- split-bf16 (hi + lo) MFMA GEMMs and the exact GELU in
        fp32 (csrc/clip_engine.hip showo_projector_set_precision): nn.Sequential(Linear, GELU, Linear) in fp32 to ~1e-5.  Inference
        only (the backward differentiates the bf16-operand forward).  `Showo.set_precision` forwards here."""
        if int(precision) not in (0, 1):
            raise ValueError("precision must be 0 (bf16 operands) or 1 (split-bf16, fp32-class)")
        self._precision = int(precision)
        if getattr(self, "_proj", None) is not None:
            _lib.call("showo_projector_set_precision", self._proj, self._precision)
            if self._precision == 1 and not _lib.load().showo_projector_precise_ready(self._proj):
                self._versions = {}  # the low halves are made by the loader (allocated on first use): upload the weights again
        return self

    def mark_weights_dirty(self):
        """re-upload the projector's parameters at the next call (updates through `.data` do not bump tensor versions)"""
        if getattr(self, "_proj", None) is not None:
            self._versions = {}

    def _drop(self):
        if getattr(self, "_proj", None) is not None:
            _lib.load().showo_projector_destroy(self._proj)
        self._proj, self._versions = None, None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def _hip_forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("show-o_amd runs the projector on the GPU (no CPU path exists)")
        import ctypes as C
        din, dout = self._dims
        xin = x.detach().float().contiguous()
        T = xin.numel() // din
        if self._proj is None or T > self._rows:
            self._drop()
            h = C.c_void_p()
            self._rows = max(T, 576)
            _lib.check(_lib.load().showo_projector_create(din, dout, self._rows, C.byref(h)), "showo_projector_create")
            self._proj, self._versions = h, {}
            _lib.call("showo_projector_set_precision", self._proj, int(getattr(self, "_precision", 0)))
        for k, v in self.state_dict().items():
            ver = (v.data_ptr(), v._version)
            if self._versions.get(k) != ver:
                src = v.detach().float().contiguous()
                _lib.call("showo_projector_load", self._proj, k.encode(), _lib.ptr(src), src.numel(), _lib.stream())
                self._versions[k] = ver
                torch.cuda.current_stream().synchronize()
        out = torch.empty(tuple(x.shape[:-1]) + (dout,), dtype=torch.float32, device=x.device)
        _lib.call("showo_projector_forward", self._proj, _lib.ptr(xin), T, _lib.ptr(out), _lib.stream())
        return out.to(x.dtype)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _ProjectorFn.apply(self, x, self[0].weight, self[0].bias, self[2].weight, self[2].bias)
        return self._hip_forward(x)


class Showo(PretrainedMixin, nn.Module):
    _supports_gradient_checkpointing = True
    # the reference's @register_to_config arguments (models/modeling_showo.py:26-37) = the keys of its config.json
    _config_keys = ("w_clip_vit", "vocab_size", "llm_vocab_size", "llm_model_path", "codebook_size", "num_vq_tokens", "load_from_showo")

    def _extra_config(self):
        return {k: v for k, v in self.arch.items() if PHI_1_5.get(k) != v}

    def __init__(self, w_clip_vit, vocab_size, llm_vocab_size, llm_model_path='', codebook_size=8192,
                 num_vq_tokens=256, load_from_showo=True, **kwargs):
        super().__init__()
        arch = dict(PHI_1_5)
        arch.update({k: kwargs[k] for k in PHI_1_5 if k in kwargs})
        self.arch = arch
        self.config = _Cfg(w_clip_vit=w_clip_vit, vocab_size=vocab_size, llm_vocab_size=llm_vocab_size,
                           llm_model_path=llm_model_path, codebook_size=codebook_size, num_vq_tokens=num_vq_tokens,
                           load_from_showo=load_from_showo, mask_token_id=vocab_size - 1)
        self.vocab_size = vocab_size
        hidden, heads = arch["hidden_size"], arch["num_attention_heads"]
        if hidden // heads != 64:
            raise ValueError("the gfx950 attention kernel is built for head_dim 64 (Phi-1.5)")
        # the reference builds Phi with its native vocabulary and then resizes (modeling_showo.py:43-46); the
        # end state (embed + untied lm_head of `vocab_size` rows) is what we allocate directly.
        self.showo = _PhiForCausalLMParams(vocab_size, hidden, arch["intermediate_size"], arch["num_hidden_layers"],
                                           64, arch["layer_norm_eps"])
        if not load_from_showo:
            # reference models/modeling_showo.py:45-46: PhiForCausalLM.from_pretrained(llm_model_path) then
            # resize_token_embeddings(vocab_size) -- stage-1 training starts from the HF Phi-1.5 weights (training/train.py:203)
            self._init_from_phi_checkpoint(llm_model_path)
        self.output_size = self.vocab_size
        if w_clip_vit:
            self.mm_projector = _MMProjector(1024, hidden)  # reference: nn.Sequential(Linear(1024, 2048), GELU(), Linear(2048, 2048))
        self._engine = None
        self._engine_key = None
        self._engine_versions = None
        self._trainer = None
        self._weights_changed = True
        self.max_batch = int(kwargs.get("max_batch", 32))
        self.max_seq = int(kwargs.get("max_seq", 1280))
        self._precision = 0

    def _init_from_phi_checkpoint(self, llm_model_path):
        """`load_from_showo=False`: start from a Hugging Face Phi checkpoint in the local directory `llm_model_path`
        (config.json + model.safetensors | pytorch_model.bin | a sharded index; keys `model.*`, `lm_head.*`).  What the reference does
        (models/modeling_showo.py:45-46; transformers `from_pretrained` + `resize_token_embeddings`):
          * every checkpoint tensor is loaded under the `showo.` prefix; tensors the checkpoint lacks keep their fresh
            initialisation -- for microsoft/phi-1_5 those are the q/k LayerNorms that PhiForCausalLM forces on (models/phi.py:1088):
            weight 1, bias 0;
          * the embedding and the untied, biased lm_head grow from the checkpoint's vocabulary to `vocab_size`: the first
            min(old, new) rows (and bias entries) are the checkpoint's, the new rows are N(0, 0.02) / zero bias (`_init_weights`).
        There is no hub access here: anything but a local directory raises."""
        import json
        import os
        from .persistence import _load_file
        if self.showo.lm_head.weight.is_meta:
            return  # from_pretrained() builds on the meta device and overwrites every tensor from the Show-o checkpoint itself
        d = llm_model_path
        if not d or not os.path.isdir(d):
            raise EnvironmentError(f"load_from_showo=False needs a local Hugging Face Phi checkpoint directory as llm_model_path "
                                   f"(got {llm_model_path!r}); hub downloads are not supported")
        cfg_path = os.path.join(d, "config.json")
        if os.path.isfile(cfg_path):
            with open(cfg_path, encoding="utf-8") as f:
                hf = json.load(f)
            for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads"):
                if k in hf and hf[k] != self.arch[k]:
                    raise ValueError(f"{cfg_path}: {k} = {hf[k]} but the model was built with {self.arch[k]} "
                                     f"(pass {k}=... to Showo for a non-Phi-1.5 geometry)")
        files = None
        for name in ("model.safetensors", "pytorch_model.safetensors", "pytorch_model.bin"):
            if os.path.isfile(os.path.join(d, name)):
                files = [os.path.join(d, name)]
                break
            idx = os.path.join(d, name + ".index.json")
            if os.path.isfile(idx):
                with open(idx, encoding="utf-8") as f:
                    files = [os.path.join(d, fn) for fn in sorted(set(json.load(f)["weight_map"].values()))]
                break
        if files is None:
            raise EnvironmentError(f"no model.safetensors / pytorch_model.bin (or sharded index) in {d}")
        want = self.showo.state_dict()
        for k, v in want.items():  # fresh initialisation of what a Phi-1.5 checkpoint does not hold
            if k.endswith("_layernorm.weight") and ("q_layernorm" in k or "k_layernorm" in k):
                v.data.fill_(1.0)
            elif k.endswith("_layernorm.bias") and ("q_layernorm" in k or "k_layernorm" in k):
                v.data.zero_()
        seen = set()
        resized = ("model.embed_tokens.weight", "lm_head.weight", "lm_head.bias")
        with torch.no_grad():
            for path in files:
                part = _load_file(path, "cpu")
                for k, v in part.items():
                    if k not in want:
                        continue
                    dst = want[k]
                    if k in resized:
                        n = min(v.shape[0], dst.shape[0])
                        if tuple(v.shape[1:]) != tuple(dst.shape[1:]):
                            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} does not match {tuple(dst.shape)} beyond the vocabulary")
                        dst[:n].copy_(v[:n])
                    else:
                        if tuple(v.shape) != tuple(dst.shape):
                            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(dst.shape)}")
                        dst.copy_(v)
                    seen.add(k)
                del part
        allowed_missing = [k for k in want if "q_layernorm" in k or "k_layernorm" in k]
        missing = [k for k in want if k not in seen and k not in allowed_missing]
        if missing:
            raise KeyError(f"Phi checkpoint {d} lacks {len(missing)} tensors, e.g. {missing[:5]}")

    def set_precision(self, precision):
        """0 (default): bf16 GEMM / attention operands with fp32 accumulation -- the timed path.  1: accuracy mode, the reference's
        fp32 inference (inference_t2i.py:67, models/phi.py:1182-1183) to ~1e-4 end to end: split-bf16 (hi + lo) MFMA GEMMs, fp32
        LayerNorm / RoPE / attention / gelu_new.  At Phi-1.5's shape this mode runs on the PRODUCTION kernels (K-concatenated split
        images, csrc/engine.hip run_layers_precise_fast): same launches as precision 0 with three MFMAs per product, prefix reuse,
        hipGraph replay and the KV-cached decode included; tiny test shapes use the fp32 reference kernels of csrc/precise.hip (and
        mmu_generate then runs the reference's own no-cache algorithm).  Applies to forward() without labels, t2i_generate() and
        mmu_generate(); training keeps bf16 operands.  Costs 4x the bf16 weight memory.  With w_clip_vit the mm_projector follows.
        2: fp16 (IEEE half) operands on the SAME kernels, launches, prefix reuse and hipGraph replay as precision 0 -- the MFMA rate is
        the same, operand rounding is 2^-12 instead of 2^-9 -- with the final LayerNorm + lm_head as the split-bf16 product of
        precision 1: logits within 1e-3 of the reference's fp32 inference (rel_rms ~8e-4 at model scale where bf16 operands give
        7e-3) at the speed of the default path.  Converts saturate at +-65504 (`range_check()` counts saturated activations);
        the KV-cached decode steps run the fp16 instances of the fused three-launch layer and of the batched layer (csrc/decode.hip, decode_batch.hip; the lm_head of a decode step is one fused LayerNorm + (hi, lo) GEMV launch), the
        mm_projector runs in its fp32-class mode (it is one small MLP).  Switching to / from 2 re-uploads the weight images."""
        if int(precision) not in (0, 1, 2):
            raise ValueError("precision must be 0 (bf16 operands), 1 (split-bf16, fp32-class) or 2 (fp16 operands)")
        self._precision = int(precision)
        if self.__dict__.get("_modules", {}).get("mm_projector") is not None:
            self.mm_projector.set_precision(1 if int(precision) else 0)
        return self

    def range_check(self, fn):
        """Precision-2 diagnostic: run `fn()` (any forward / generate call of this model) with the engine's range check on and
        return the number of fp16 activation elements that left a convert saturated (|x| = 65504) or non-finite (csrc/engine.hip
        range_check; extra launches, not for timed runs).  0 on random-init weights; a deployment on a real checkpoint runs this once."""
        eng = self.engine()
        cnt = torch.zeros(1, dtype=torch.int64, device=self.showo.lm_head.weight.device)
        _lib.call("showo_engine_set_range_check", eng, _lib.ptr(cnt))
        try:
            fn()
        finally:
            _lib.call("showo_engine_set_range_check", self._engine, None)
        return int(cnt.item())

    def mark_weights_dirty(self):
        """Re-upload every parameter to the HIP engine at the next call.  The engine notices parameter changes by
        (data_ptr, tensor._version); updates made through `.data` (p.data.copy_(), DeepSpeed ZeRO flat-buffer updates, EMA swaps)
        do not bump the version counter, so an optimizer of that kind must call this after `optimizer.step()`."""
        self._engine_versions = {} if self._engine is not None else None
        self._weights_changed = True
        if getattr(self, "mm_projector", None) is not None:
            self.mm_projector.mark_weights_dirty()

    # ---- reference attribute passthrough (`model.mask_token_id`, reference models/modeling_utils.py:139-155)
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            cfg = self.__dict__.get("config")
            if cfg is not None and name in cfg:
                return cfg[name]
            raise

    def _set_gradient_checkpointing(self, module, value=False):
        self.gradient_checkpointing = True

    @classmethod
    def from_state_dict(cls, state_dict, device="cuda", **ctor):
        m = cls(**ctor)
        m.load_state_dict(state_dict, strict=True)
        return m.to(device)

    # ---- engine management -------------------------------------------------------------------------
    def configure_workspace(self, max_batch, max_seq):
        """Size the engine's HBM workspaces (tokens = max_batch * max_seq).  Re-creates the engine."""
        self.max_batch, self.max_seq = int(max_batch), int(max_seq)
        self._drop_engine()

    def _drop_engine(self):
        if getattr(self, "_trainer", None) is not None:
            _lib.load().showo_train_destroy(self._trainer)
            self._trainer = None
        if self._engine is not None:
            _lib.load().showo_engine_destroy(self._engine)
        self._engine, self._engine_versions = None, None

    def trainer(self):
        """HIP training state (saved activations, transposed weight images, fp32 gradient buffers); created lazily."""
        eng = self.engine(for_training=True)
        if self._trainer is None:
            import ctypes as C
            h = C.c_void_p()
            _lib.check(_lib.load().showo_train_create(eng, self.max_batch, self.max_seq, C.byref(h)), "showo_train_create")
            self._trainer = h
            self._weights_changed = True
        if self._weights_changed:
            _lib.call("showo_train_invalidate_weights", self._trainer)
            self._weights_changed = False
        return self._trainer

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _engine_params(self):
        return [(k, v) for k, v in self.showo.state_dict(prefix="showo.").items()]

    def _rope_tables(self, device):
        # exactly PhiRotaryEmbedding._set_cos_sin_cache (reference models/phi.py:86-102), fp32, on the host
        a = self.arch
        dim = int(a["partial_rotary_factor"] * 64)
        inv_freq = 1.0 / (a["rope_theta"] ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
        t = torch.arange(a["max_position_embeddings"], dtype=torch.int64).type_as(inv_freq)
        freqs = torch.outer(t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.cos().contiguous().to(device), emb.sin().contiguous().to(device)

    def engine(self, for_training=False):
        """Create the HIP engine if needed and (re)upload weights whose version changed.  for_training: the trainer works on bf16
        weight images, so a model in precision 2 (fp16 images) trains on a precision-0 engine (and re-uploads on the way back)."""
        _lib.require_gpu()
        lib = _lib.load()
        dev = self.showo.lm_head.weight.device
        if dev.type != "cuda":
            raise RuntimeError("Showo parameters must live on the GPU (model.to('cuda')); no CPU path exists")
        a = self.arch
        if self._engine is None:
            import ctypes as C
            cfg = _lib.EngineConfig(hidden=a["hidden_size"], layers=a["num_hidden_layers"], heads=a["num_attention_heads"],
                                    ffn=a["intermediate_size"], vocab=self.vocab_size,
                                    rotary_dim=int(a["partial_rotary_factor"] * 64), max_pos=a["max_position_embeddings"],
                                    ln_eps=a["layer_norm_eps"], rope_theta=a["rope_theta"],
                                    max_batch=self.max_batch, max_seq=self.max_seq)
            h = C.c_void_p()
            _lib.check(lib.showo_engine_create(C.byref(cfg), C.byref(h)), "showo_engine_create")
            self._engine = h
            self._engine_versions = {}
            cos, sin = self._rope_tables(dev)
            _lib.call("showo_engine_load", self._engine, b"rope.cos", _lib.ptr(cos), cos.numel(), _lib.stream())
            _lib.call("showo_engine_load", self._engine, b"rope.sin", _lib.ptr(sin), sin.numel(), _lib.stream())
            torch.cuda.current_stream().synchronize()
        want = int(getattr(self, "_precision", 0))
        if for_training and want == 2:
            want = 0
        if lib.showo_engine_get_precision(self._engine) != want:
            _lib.call("showo_engine_set_precision", self._engine, want)
        if want == 1 and not lib.showo_engine_precise_ready(self._engine):
            self._engine_versions = {}  # the low halves of the weights are made by the loader: upload everything again
        if lib.showo_engine_missing(self._engine):
            self._engine_versions = {}  # first use, or the weight images changed their element type (precision 2 <-> 0 / 1)
        for k, v in self._engine_params():
            ver = (v.data_ptr(), v._version)
            if self._engine_versions.get(k) != ver:
                self._weights_changed = True
                src = v.detach()
                if src.dtype != torch.float32 or not src.is_contiguous():
                    src = src.float().contiguous()
                _lib.call("showo_engine_load", self._engine, k.encode(), _lib.ptr(src), src.numel(), _lib.stream())
                self._engine_versions[k] = ver
                if src is not v:
                    torch.cuda.current_stream().synchronize()  # keep the temporary alive until consumed
        missing = lib.showo_engine_missing(self._engine)
        if missing:
            raise RuntimeError(f"engine is missing {missing} tensors")
        return self._engine

    def _use_mask(self, eng, attention_mask):
        """dense additive mask -> contiguous fp32 tensor; IntervalMask (prompting_utils.intervals_*) -> registered with the
        engine, returns None (the call then passes no dense mask)"""
        from .prompting_utils import IntervalMask
        if isinstance(attention_mask, IntervalMask):
            attention_mask.check()
            _lib.call("showo_engine_use_intervals", eng, _lib.ptr(attention_mask.iv), _lib.ptr(attention_mask.flag))
            return None
        return attention_mask.detach().float().contiguous()

    # ---- Showo.forward (reference models/modeling_showo.py:59-102) -------------------------------------
    def forward(self, input_ids, input_embeddings=None, attention_mask=None, labels=None, label_smoothing=0.0,
                batch_size_t2i=0, batch_size_lm=0, batch_size_mmu=0, max_seq_length=128, labels_mask_text=None,
                labels_mask_image=None, **kwargs):
        if labels is not None:
            params = [p for _, p in self.showo.named_parameters()]
            return _ShowoTrainFn.apply(self, input_ids, input_embeddings, attention_mask, labels, int(batch_size_t2i),
                                       int(batch_size_lm), int(batch_size_mmu), int(max_seq_length), *params)
        eng = self.engine()
        if input_embeddings is None:
            B, L = input_ids.shape
            ids = input_ids.to(torch.int64).contiguous()
            emb = None
            dev = ids.device
        else:
            B, L = input_embeddings.shape[:2]
            ids = None
            emb = input_embeddings.detach().float().contiguous()
            dev = emb.device
        mask = None
        if attention_mask is not None:
            if tuple(attention_mask.shape) != (B, 1, L, L):  # same check as the eager path, reference models/phi.py:368-372
                raise ValueError(f"Attention mask should be of size {(B, 1, L, L)}, but is {tuple(attention_mask.shape)}")
            mask = self._use_mask(eng, attention_mask)
        logits = torch.empty((B, L, self.vocab_size), dtype=torch.float32, device=dev)
        try:
            _lib.call("showo_engine_forward", eng, _lib.ptr(ids), _lib.ptr(emb), _lib.ptr(mask), B, L, _lib.ptr(logits),
                      _lib.stream())
        finally:
            _lib.call("showo_engine_use_intervals", eng, None, None)
        if labels is None:
            return logits
        raise AssertionError("unreachable")

    # ---- Showo.t2i_generate (reference models/modeling_showo.py:104-181) -----------------------------------
    def t2i_generate(self, input_ids=None, uncond_input_ids=None, attention_mask=None, temperature=1.0, timesteps=18,
                     guidance_scale=0, noise_schedule=cosine_schedule, generator=None, config=None,
                     _exp_noise=None, _uniform=None, **kwargs):
        eng = self.engine()
        N = config.model.showo.num_vq_tokens
        offset = config.model.showo.llm_vocab_size + config.model.showo.num_new_special_tokens
        text_len = config.dataset.preprocessing.max_seq_length
        B, L = input_ids.shape
        if input_ids.dtype != torch.int64 or not input_ids.is_contiguous():
            raise ValueError("input_ids must be a contiguous int64 tensor (it is updated in place like the reference)")
        codebook = self.vocab_size - 1 - offset  # logits[..., offset:-1] (reference :144)
        unc = None
        if uncond_input_ids is not None and guidance_scale > 0:
            unc = uncond_input_ids.to(torch.int64).contiguous()
        mask = None if attention_mask is None else self._use_mask(eng, attention_mask)
        import ctypes as C
        ml, tp = t2i_step_constants(timesteps, N, temperature, noise_schedule)
        ml_a = (C.c_float * timesteps)(*ml)
        tp_a = (C.c_float * timesteps)(*tp)
        if generator is not None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=generator.device).item())
        else:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        out = torch.empty((B, N), dtype=torch.int64, device=input_ids.device)
        # the denoise steps run as replays of ONE captured hipGraph (cached on the engine across calls) unless use_graph=0 is
        # passed or per-launch event timing is on (the engine then runs the steps eagerly)
        use_graph = int(kwargs.get("use_graph", 1))
        # bit 0: hipGraph replay of the denoise step; bit 1: recompute the step-invariant text rows every step (A/B switch)
        flags = (1 if use_graph else 0) | (0 if kwargs.get("reuse_prefix", True) else 2)

        def run():
            _lib.call("showo_engine_t2i_generate", eng, _lib.ptr(input_ids), _lib.ptr(unc), _lib.ptr(mask), B, L, N, text_len,
                      self.config.mask_token_id, offset, codebook, float(guidance_scale), timesteps,
                      C.cast(ml_a, C.c_void_p), C.cast(tp_a, C.c_void_p), seed, _lib.ptr(_exp_noise), _lib.ptr(_uniform),
                      flags, _lib.ptr(out), _lib.stream())

        if use_graph:
            # hipGraph replay of the denoise step: stream capture needs a non-default stream
            if getattr(self, "_graph_stream", None) is None:
                self._graph_stream = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self._graph_stream.wait_stream(cur)
            with torch.cuda.stream(self._graph_stream):
                run()
            cur.wait_stream(self._graph_stream)
        else:
            run()
        _lib.call("showo_engine_use_intervals", eng, None, None)
        return out

    # ---- Showo.mmu_generate (reference models/modeling_showo.py:183-240) -----------------------------------
    @torch.no_grad()
    def mmu_generate(self, idx=None, input_embeddings=None, attention_mask=None, max_new_tokens=100, temperature=1.0,
                     top_k=None, eot_token=None, generator=None, _exp_noise=None):
        """reference signature + `generator` (seeds the on-device Philox stream of the multinomial draws; default: torch's
        global generator) and `_exp_noise` fp32 [max_new_tokens, vocab] (parity tests inject the reference's Exp(1) draws)"""
        eng = self.engine()
        greedy = top_k == 1  # the reference caller's setting (inference_mmu.py:81): multinomial of a one-hot = arg-max
        if not greedy:
            if not temperature > 0:
                raise ValueError("temperature must be > 0")
            k = 0 if top_k is None else int(top_k)
            if generator is not None:
                seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=generator.device).item())
            else:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            noise = None if _exp_noise is None else _exp_noise.detach().float().contiguous()
            if noise is not None and tuple(noise.shape) != (max_new_tokens, self.vocab_size):
                raise ValueError("_exp_noise must be [max_new_tokens, vocab_size]")
        dev = idx.device if idx is not None else input_embeddings.device
        # accuracy mode: KV-cached like precision 0 when the engine runs it on the production kernels (Phi-1.5's shape); otherwise -- or
        # with `self.precise_recompute = True` -- the reference's own no-cache algorithm on the fp32-class path
        if int(getattr(self, "_precision", 0)) == 1 and (getattr(self, "precise_recompute", False) or
                                                         not _lib.load().showo_engine_precise_fast(eng)):
            return self._mmu_generate_recompute(idx, input_embeddings, attention_mask, max_new_tokens, temperature, top_k, eot_token,
                                                greedy, None if greedy else (k, seed, noise))
        if input_embeddings is not None:
            if input_embeddings.shape[0] != 1:
                raise ValueError("mmu_generate has batch-1 semantics (reference modeling_showo.py:204,229)")
            L = input_embeddings.shape[1]
            emb = input_embeddings.detach().float().contiguous()
            ids = None
        else:
            if idx.shape[0] != 1:
                raise ValueError("mmu_generate has batch-1 semantics (reference modeling_showo.py:204,229)")
            L = idx.shape[1]
            ids = idx.to(torch.int64).contiguous()
            emb = None
        from .prompting_utils import IntervalMask
        if isinstance(attention_mask, IntervalMask):  # prompting_utils.intervals_for_mmu / _mmu_vit: no [1,1,L,L] tensor at all
            mask = self._use_mask(eng, attention_mask)
        else:
            mask = None if attention_mask is None else attention_mask.detach().float().reshape(1, 1, L, L).contiguous()
        logits = torch.empty((self.vocab_size,), dtype=torch.float32, device=dev)
        tok = torch.empty((1,), dtype=torch.int64, device=dev)
        try:
            _lib.call("showo_engine_prefill", eng, _lib.ptr(ids), _lib.ptr(emb), _lib.ptr(mask), L, _lib.ptr(logits), _lib.stream())
        finally:
            _lib.call("showo_engine_use_intervals", eng, None, None)
        # logits / temperature does not change the arg-max for temperature > 0; top_k=1 makes the reference's multinomial a
        # deterministic arg-max.  The first token comes from the prefill logits; the continuation runs in
        # chunks of `chunk` steps entirely on the device (embed -> 24 layers on the KV cache -> lm_head -> arg-max), one
        # hipGraph replay per step, and the host looks at the tokens (for <eot>) once per chunk.
        if greedy:
            _lib.call("showo_argmax_f32", _lib.ptr(logits), self.vocab_size, _lib.ptr(tok), _lib.stream())
        else:  # logits / temperature -> top-k filter -> softmax -> multinomial (reference :220-228), one kernel
            _lib.call("showo_sample_topk", _lib.ptr(logits), self.vocab_size, k, float(temperature), _lib.ptr(noise), seed, 0,
                      _lib.ptr(tok), _lib.stream())
        first = int(tok.item())
        result = [torch.tensor(first, device=dev)]
        if (eot_token is not None and first == eot_token) or max_new_tokens <= 1:
            return result[:max_new_tokens]
        use_graph, chunk = int(getattr(self, "decode_graph", 1)), 16
        if getattr(self, "_graph_stream", None) is None:
            self._graph_stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        remaining = max_new_tokens - 1
        while remaining > 0:
            n = min(chunk, remaining)
            outc = torch.empty((n,), dtype=torch.int64, device=dev)
            self._graph_stream.wait_stream(cur)
            with torch.cuda.stream(self._graph_stream):
                if greedy:
                    _lib.call("showo_engine_decode_greedy", eng, _lib.ptr(tok), n, _lib.ptr(outc), _lib.ptr(logits), use_graph,
                              _lib.stream())
                else:
                    _lib.call("showo_engine_decode_sample", eng, _lib.ptr(tok), n, _lib.ptr(outc), _lib.ptr(logits), k,
                              float(temperature), _lib.ptr(noise), seed, max_new_tokens - remaining, use_graph, _lib.stream())
            cur.wait_stream(self._graph_stream)
            toks = outc.tolist()
            for t in toks:
                result.append(torch.tensor(t, device=dev))
                if eot_token is not None and t == eot_token:
                    return result
            remaining -= n
        return result


@torch.no_grad()
def mmu_generate_batch(self, idx=None, input_embeddings=None, attention_mask=None, max_new_tokens=100, temperature=1.0, top_k=None,
                       eot_token=None):
    """n independent `mmu_generate` calls served together (BASELINE cfg4: "batch=4 images"; the reference's mmu_generate is batch-1 and
    inference_mmu.py:87-177 walks the images one by one).  `idx` / `input_embeddings` / `attention_mask` are LISTS (one entry per
    sequence, each what a single `mmu_generate` call takes: [1, L_b] ids or [1, L_b, H] embeddings, its own mask or IntervalMask);
    returns a list of n token lists -- exactly what n separate calls return.  Greedy (`top_k=1`, the reference caller's setting,
    inference_mmu.py:81) at precision 0 or 2 runs on the batched engine path (csrc/decode_batch.hip: n KV caches, ONE weight stream per
    token step, every sequence bit-identical to its batch-1 run); anything else falls back to n sequential calls."""
    seqs = idx if idx is not None else input_embeddings
    n = len(seqs)
    masks = attention_mask if isinstance(attention_mask, (list, tuple)) else [attention_mask] * n
    if top_k != 1 or int(getattr(self, "_precision", 0)) == 1 or n > 8 or n < 2:
        return [self.mmu_generate(idx=None if idx is None else idx[b], input_embeddings=None if input_embeddings is None else input_embeddings[b],
                                  attention_mask=masks[b], max_new_tokens=max_new_tokens, temperature=temperature, top_k=top_k,
                                  eot_token=eot_token) for b in range(n)]
    from .prompting_utils import IntervalMask
    eng = self.engine()
    dev = seqs[0].device
    lens = [int(t.shape[1]) for t in seqs]
    try:
        _lib.call("showo_engine_batch_begin", eng, n, max(lens) + max_new_tokens + 1)
    except RuntimeError:
        # the batch's common capacity (longest prompt + max_new_tokens) does not fit the engine (max_position_embeddings, the single-block
        # decode attention): n sequential calls fail only if a sequence actually reaches the limit before <eot>
        return [self.mmu_generate(idx=None if idx is None else idx[b], input_embeddings=None if input_embeddings is None else input_embeddings[b],
                                  attention_mask=masks[b], max_new_tokens=max_new_tokens, temperature=temperature, top_k=top_k,
                                  eot_token=eot_token) for b in range(n)]
    logits = torch.empty((n, self.vocab_size), dtype=torch.float32, device=dev)
    tok = torch.empty((n,), dtype=torch.int64, device=dev)
    for b in range(n):
        if seqs[b].shape[0] != 1:
            raise ValueError("every sequence of mmu_generate_batch is one batch-1 mmu_generate call (reference modeling_showo.py:204,229)")
        ids = emb = None
        if idx is not None:
            ids = idx[b].to(torch.int64).contiguous()
        else:
            emb = input_embeddings[b].detach().float().contiguous()
        am = masks[b]
        if isinstance(am, IntervalMask):
            mask = self._use_mask(eng, am)
        else:
            mask = None if am is None else am.detach().float().reshape(1, 1, lens[b], lens[b]).contiguous()
        try:
            _lib.call("showo_engine_batch_prefill", eng, b, _lib.ptr(ids), _lib.ptr(emb), _lib.ptr(mask), lens[b], _lib.ptr(logits[b]), _lib.stream())
        finally:
            _lib.call("showo_engine_use_intervals", eng, None, None)
        _lib.call("showo_argmax_f32", _lib.ptr(logits[b]), self.vocab_size, _lib.ptr(tok[b:b + 1]), _lib.stream())
    first = tok.tolist()
    results = [[torch.tensor(t, device=dev)] for t in first]
    done = [(eot_token is not None and t == eot_token) or max_new_tokens <= 1 for t in first]
    use_graph, chunk = int(getattr(self, "decode_graph", 1)), 16
    if getattr(self, "_graph_stream", None) is None:
        self._graph_stream = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    remaining = max_new_tokens - 1
    while remaining > 0 and not all(done):
        m = min(chunk, remaining)
        outc = torch.empty((n, m), dtype=torch.int64, device=dev)
        self._graph_stream.wait_stream(cur)
        with torch.cuda.stream(self._graph_stream):
            _lib.call("showo_engine_batch_decode_greedy", eng, _lib.ptr(tok), m, _lib.ptr(outc), _lib.ptr(logits), use_graph, _lib.stream())
        cur.wait_stream(self._graph_stream)
        rows = outc.tolist()
        for b in range(n):
            for t in rows[b]:
                if done[b]:
                    break
                results[b].append(torch.tensor(t, device=dev))
                if (eot_token is not None and t == eot_token) or len(results[b]) >= max_new_tokens:
                    done[b] = True
        remaining -= m
    return [r[:max_new_tokens] for r in results]


Showo.mmu_generate_batch = mmu_generate_batch


def _dense_mask_of(attention_mask, L, device):
    """[1,1,L,L] additive fp32 mask from whatever mmu_generate accepts (dense tensor, IntervalMask, None = causal)"""
    from .prompting_utils import IntervalMask
    neg = float(torch.iinfo(torch.int64).min)
    if attention_mask is None:
        m = torch.zeros((L, L), dtype=torch.float32, device=device)
        m.masked_fill_(torch.ones((L, L), dtype=torch.bool, device=device).triu(1), neg)
        return m.reshape(1, 1, L, L)
    if isinstance(attention_mask, IntervalMask):
        iv = attention_mask.check().iv.reshape(-1, L, 4)[:1].to(torch.int64)
        col = torch.arange(L, device=iv.device)[None, None, :]
        vis = ((col >= iv[..., 0:1]) & (col < iv[..., 1:2])) | ((col >= iv[..., 2:3]) & (col < iv[..., 3:4]))
        return torch.where(vis, 0.0, neg).to(torch.float32).reshape(1, 1, L, L).to(device)
    return attention_mask.detach().float().reshape(1, 1, L, L).to(device)


def _mmu_generate_recompute(self, idx, input_embeddings, attention_mask, max_new_tokens, temperature, top_k, eot_token, greedy, sampling):
    """Accuracy mode (`set_precision(1)`): the reference's OWN algorithm (models/modeling_showo.py:190-240) -- no KV cache, the whole
    sequence is run again for every token on the grown mask (new column hidden from the old rows, new row = last row + itself,
    :203-217) -- on the fp32-class engine path.  O(tokens x forward): a parity mode, not a serving mode."""
    eng = self.engine()
    dev = idx.device if idx is not None else input_embeddings.device
    if (input_embeddings if input_embeddings is not None else idx).shape[0] != 1:
        raise ValueError("mmu_generate has batch-1 semantics (reference modeling_showo.py:204,229)")
    if input_embeddings is not None:
        emb = input_embeddings.detach().float().contiguous()
    else:
        emb = self.showo.model.embed_tokens.weight.detach().float()[idx.to(torch.int64)].contiguous()
    L = emb.shape[1]
    mask = _dense_mask_of(attention_mask, L, dev)
    neg = float(torch.finfo(torch.float32).min)  # the reference pads the new column with finfo(dtype).min (:206-209)
    row = torch.empty((1,), dtype=torch.int32, device=dev)
    logits = torch.empty((self.vocab_size,), dtype=torch.float32, device=dev)
    tok = torch.empty((1,), dtype=torch.int64, device=dev)
    result = []
    for step in range(max_new_tokens):
        row.fill_(L - 1)
        _lib.call("showo_engine_forward_rows", eng, None, _lib.ptr(emb), _lib.ptr(mask), 1, L, _lib.ptr(row), 1, 0, self.vocab_size,
                  _lib.ptr(logits), _lib.stream())
        if greedy:
            _lib.call("showo_argmax_f32", _lib.ptr(logits), self.vocab_size, _lib.ptr(tok), _lib.stream())
        else:
            k, seed, noise = sampling
            _lib.call("showo_sample_topk", _lib.ptr(logits), self.vocab_size, k, float(temperature), _lib.ptr(noise), seed, step, _lib.ptr(tok),
                      _lib.stream())
        t = int(tok.item())
        result.append(torch.tensor(t, device=dev))
        if eot_token is not None and t == eot_token:
            break
        grown = torch.full((1, 1, L + 1, L + 1), neg, dtype=torch.float32, device=dev)
        grown[0, 0, :L, :L] = mask[0, 0]
        grown[0, 0, L, :L] = mask[0, 0, L - 1]
        grown[0, 0, L, L] = 0.0
        mask = grown
        emb = torch.cat([emb, self.showo.model.embed_tokens.weight.detach().float()[tok].reshape(1, 1, -1)], dim=1).contiguous()
        L += 1
    return result


Showo._mmu_generate_recompute = _mmu_generate_recompute


def gen_config(llm_vocab_size=50295, num_new_special_tokens=10, num_vq_tokens=256, max_seq_length=128):
    """The subset of the OmegaConf config that t2i_generate reads (reference modeling_showo.py:123-133)."""
    ns = types.SimpleNamespace
    return ns(model=ns(showo=ns(num_vq_tokens=num_vq_tokens, num_new_special_tokens=num_new_special_tokens,
                                llm_vocab_size=llm_vocab_size)),
              dataset=ns(preprocessing=ns(max_seq_length=max_seq_length)))
