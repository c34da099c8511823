"""TEST INFRASTRUCTURE ONLY (oracle side): deterministic synthetic weights.

No pretrained Show-o / MAGVIT-v2 checkpoints are reachable offline, so parity is defined on
seeded random weights.  The generator below uses `numpy.random.RandomState` (a frozen, version-stable
stream) so that the build container (where the real reference is imported to make golden fixtures) and
the GPU box (where only this repo exists) construct bit-identical weights without shipping them.

Key names/shapes are the reference's state-dict keys (reference
`models/phi.py:919-1095`, `models/modeling_showo.py:26-54`, `models/modeling_magvitv2.py:48-433`).
Every parameter (including biases and norm affines, which the reference initialises to 0/1) is given a
non-trivial value so that a kernel that drops a bias or an affine term cannot pass parity.
"""
from collections import OrderedDict
import numpy as np


class ShowoDims:
    """Architecture numbers of the Phi-1.5 based Show-o (reference models/phi.py, configs/showo_demo.yaml:19-24)."""

    def __init__(self, hidden=2048, layers=24, heads=32, ffn=8192, vocab=58498, llm_vocab=50295,
                 num_new_special_tokens=10, codebook=8192, num_vq_tokens=256, max_text_len=128,
                 rotary_dim=32, rope_theta=10000.0, ln_eps=1e-5, max_pos=2048, w_clip_vit=False):
        self.hidden, self.layers, self.heads, self.ffn = hidden, layers, heads, ffn
        self.vocab, self.llm_vocab, self.num_new_special_tokens = vocab, llm_vocab, num_new_special_tokens
        self.codebook, self.num_vq_tokens, self.max_text_len = codebook, num_vq_tokens, max_text_len
        self.head_dim = hidden // heads
        self.rotary_dim, self.rope_theta, self.ln_eps, self.max_pos = rotary_dim, rope_theta, ln_eps, max_pos
        self.w_clip_vit = w_clip_vit
        assert self.head_dim * heads == hidden

    @property
    def image_offset(self):  # first image-token id (reference modeling_showo.py:144)
        return self.llm_vocab + self.num_new_special_tokens

    @property
    def mask_token_id(self):  # reference modeling_showo.py:40
        return self.vocab - 1

    # special ids as derived in SURVEY.md §8 (len(tokenizer) after UniversalPrompting.__init__)
    @property
    def pad_id(self):
        return self.llm_vocab

    @property
    def soi_id(self):
        return self.llm_vocab + 1

    @property
    def eoi_id(self):
        return self.llm_vocab + 2

    @property
    def t2i_id(self):
        return self.llm_vocab + 5

    @property
    def mmu_id(self):
        return self.llm_vocab + 6

    def as_dict(self):
        return dict(self.__dict__)


FULL = dict()  # defaults of ShowoDims = full-size Show-o
TINY = dict(hidden=128, layers=2, heads=2, ffn=256, vocab=439, llm_vocab=300, codebook=128,
            num_vq_tokens=16, max_text_len=8, max_pos=2048)
# smallest geometry whose >= 256-row batches take the PRODUCTION kernels of the training step: 3 * hidden is a multiple of the 256-wide
# GEMM tile (fused [Wqkv ; W1] save-for-backward launch) and every GEMM dimension is >= 256 (gemm2p / gemm3w forward, dgrad, wgrad)
SMALL = dict(hidden=256, layers=2, heads=4, ffn=512, vocab=439, llm_vocab=300, codebook=128,
             num_vq_tokens=16, max_text_len=8, max_pos=2048)


def showo_state_spec(d: ShowoDims):
    """state-dict key -> (shape, std, mean) in the reference's parameter order (the RandomState draw order of
    make_showo_state; the golden fixtures depend on it)."""
    sp = OrderedDict()
    H, F = d.hidden, d.ffn
    sp["showo.model.embed_tokens.weight"] = ((d.vocab, H), 0.02, 0.0)
    for i in range(d.layers):
        p = f"showo.model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "dense"):
            sp[p + f"self_attn.{n}.weight"] = ((H, H), 0.02, 0.0)
            sp[p + f"self_attn.{n}.bias"] = ((H,), 0.02, 0.0)
        for n in ("q_layernorm", "k_layernorm"):
            sp[p + f"self_attn.{n}.weight"] = ((d.head_dim,), 0.1, 1.0)
            sp[p + f"self_attn.{n}.bias"] = ((d.head_dim,), 0.05, 0.0)
        sp[p + "mlp.fc1.weight"] = ((F, H), 0.02, 0.0)
        sp[p + "mlp.fc1.bias"] = ((F,), 0.02, 0.0)
        sp[p + "mlp.fc2.weight"] = ((H, F), 0.02, 0.0)
        sp[p + "mlp.fc2.bias"] = ((H,), 0.02, 0.0)
        sp[p + "input_layernorm.weight"] = ((H,), 0.1, 1.0)
        sp[p + "input_layernorm.bias"] = ((H,), 0.05, 0.0)
    sp["showo.model.final_layernorm.weight"] = ((H,), 0.1, 1.0)
    sp["showo.model.final_layernorm.bias"] = ((H,), 0.05, 0.0)
    sp["showo.lm_head.weight"] = ((d.vocab, H), 0.02, 0.0)
    sp["showo.lm_head.bias"] = ((d.vocab,), 0.02, 0.0)
    if d.w_clip_vit:
        sp["mm_projector.0.weight"] = ((2048, 1024), 0.02, 0.0)
        sp["mm_projector.0.bias"] = ((2048,), 0.02, 0.0)
        sp["mm_projector.2.weight"] = ((H, 2048), 0.02, 0.0)
        sp["mm_projector.2.bias"] = ((H,), 0.02, 0.0)
    return sp


def make_showo_state(d: ShowoDims, seed: int = 0, dtype=np.float32):
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for k, (shape, std, mean) in showo_state_spec(d).items():
        sd[k] = (rs.standard_normal(size=shape) * std + mean).astype(dtype)
    return sd


class MagvitDims:
    """VQGAN/LFQ architecture numbers (reference models/modeling_magvitv2.py:62-72, 278-288, 173-197)."""

    def __init__(self, ch=128, enc_ch_mult=(1, 2, 2, 4, 4), enc_blocks=(4, 3, 4, 3, 4),
                 dec_ch_mult=(1, 1, 2, 2, 4), dec_blocks=(4, 4, 3, 4, 3), z_channels=13, in_ch=3, out_ch=3):
        self.ch, self.enc_ch_mult, self.enc_blocks = ch, tuple(enc_ch_mult), tuple(enc_blocks)
        self.dec_ch_mult, self.dec_blocks = tuple(dec_ch_mult), tuple(dec_blocks)
        self.z_channels, self.in_ch, self.out_ch = z_channels, in_ch, out_ch


def make_magvit_state(d: MagvitDims = None, seed: int = 0, dtype=np.float32):
    d = d or MagvitDims()
    rs = np.random.RandomState(seed)
    sd = OrderedDict()

    def conv(name, cout, cin, k):
        bound = 1.0 / np.sqrt(cin * k * k)
        sd[name + ".weight"] = rs.uniform(-bound, bound, size=(cout, cin, k, k)).astype(dtype)
        sd[name + ".bias"] = rs.uniform(-bound, bound, size=(cout,)).astype(dtype)

    def norm(name, c):
        sd[name + ".weight"] = (1.0 + 0.1 * rs.standard_normal(size=(c,))).astype(dtype)
        sd[name + ".bias"] = (0.05 * rs.standard_normal(size=(c,))).astype(dtype)

    def resblock(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1)

    def attn(p, c):
        norm(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(p + "." + n, c, c, 1)

    ch = d.ch
    # ---- encoder (reference modeling_magvitv2.py:62-139)
    conv("encoder.conv_in", ch, d.in_ch, 3)
    in_mult = (1,) + d.enc_ch_mult
    block_in = ch
    for lvl in range(len(d.enc_ch_mult)):
        block_in = ch * in_mult[lvl]
        block_out = ch * d.enc_ch_mult[lvl]
        for j in range(d.enc_blocks[lvl]):
            resblock(f"encoder.down.{lvl}.block.{j}", block_in, block_out)
            block_in = block_out
        if lvl != len(d.enc_ch_mult) - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", block_in, block_in, 3)
    resblock("encoder.mid.block_1", block_in, block_in)
    attn("encoder.mid.attn_1", block_in)
    resblock("encoder.mid.block_2", block_in, block_in)
    norm("encoder.norm_out", block_in)
    conv("encoder.conv_out", d.z_channels, block_in, 3)
    conv("encoder.quant_conv", d.z_channels, d.z_channels, 1)
    # ---- decoder (reference modeling_magvitv2.py:278-362)
    nres = len(d.dec_ch_mult)
    block_in = ch * d.dec_ch_mult[nres - 1]
    conv("decoder.conv_in", block_in, d.z_channels, 3)
    resblock("decoder.mid.block_1", block_in, block_in)
    attn("decoder.mid.attn_1", block_in)
    resblock("decoder.mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * d.dec_ch_mult[lvl]
        for j in range(d.dec_blocks[lvl]):
            resblock(f"decoder.up.{lvl}.block.{j}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", block_in, block_in, 3)
    norm("decoder.norm_out", block_in)
    conv("decoder.conv_out", d.out_ch, block_in, 3)
    conv("decoder.post_quant_conv", d.z_channels, d.z_channels, 1)
    # ---- quantizer buffers (reference modeling_magvitv2.py:186-197)
    nb = d.z_channels
    idx = np.arange(2 ** nb, dtype=np.int64)
    binary = (idx[:, None] >> np.arange(nb - 1, -1, -1, dtype=np.int64)[None, :]) & 1
    sd["quantize.embedding"] = (binary.astype(dtype) * 2 - 1)
    sd["quantize.power_vals"] = (2 ** np.arange(nb - 1, -1, -1, dtype=np.int64))
    return sd


# ----------------------------------------------------------------------------------------------------------------
# CLIP ViT vision tower (transformers CLIPVisionModel; openai/clip-vit-large-patch14-336 in the reference's configs)
# ----------------------------------------------------------------------------------------------------------------
CLIP_L336 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                 patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu")
CLIP_TINY = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56,
                 patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu")


def make_clip_state(cfg, seed=0, dtype=np.float32):
    """deterministic weights under transformers-4.41 / checkpoint key names (vision_model.*); scales chosen so that
    activations stay O(1) through the stack (LayerNorm weights ~ N(1, 0.1), projections ~ N(0, 1/sqrt(fan_in)))"""
    rs = np.random.RandomState(seed)
    H, F, S, ps = cfg["hidden_size"], cfg["intermediate_size"], cfg["image_size"], cfg["patch_size"]
    sd = OrderedDict()

    def put(k, shape, std, mean=0.0):
        sd[k] = (rs.standard_normal(size=shape) * std + mean).astype(dtype)
    put("vision_model.embeddings.class_embedding", (H,), 0.5)
    put("vision_model.embeddings.patch_embedding.weight", (H, 3, ps, ps), 1.0 / np.sqrt(3 * ps * ps))
    put("vision_model.embeddings.position_embedding.weight", ((S // ps) ** 2 + 1, H), 0.3)
    put("vision_model.pre_layrnorm.weight", (H,), 0.1, 1.0)
    put("vision_model.pre_layrnorm.bias", (H,), 0.05)
    for i in range(cfg["num_hidden_layers"]):
        p = f"vision_model.encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            put(p + f"self_attn.{nm}.weight", (H, H), (2.0 if nm in ("q_proj", "k_proj") else 1.0) / np.sqrt(H))
            put(p + f"self_attn.{nm}.bias", (H,), 0.05)
        put(p + "layer_norm1.weight", (H,), 0.1, 1.0)
        put(p + "layer_norm1.bias", (H,), 0.05)
        put(p + "mlp.fc1.weight", (F, H), 1.0 / np.sqrt(H))
        put(p + "mlp.fc1.bias", (F,), 0.05)
        put(p + "mlp.fc2.weight", (H, F), 1.0 / np.sqrt(F))
        put(p + "mlp.fc2.bias", (H,), 0.05)
        put(p + "layer_norm2.weight", (H,), 0.1, 1.0)
        put(p + "layer_norm2.bias", (H,), 0.05)
    put("vision_model.post_layernorm.weight", (H,), 0.1, 1.0)
    put("vision_model.post_layernorm.bias", (H,), 0.05)
    return sd


def make_projector_state(din=1024, dout=2048, seed=0, dtype=np.float32):
    rs = np.random.RandomState(seed)
    return OrderedDict([("0.weight", (rs.standard_normal((dout, din)) / np.sqrt(din)).astype(dtype)),
                        ("0.bias", (rs.standard_normal((dout,)) * 0.1).astype(dtype)),
                        ("2.weight", (rs.standard_normal((dout, dout)) / np.sqrt(dout)).astype(dtype)),
                        ("2.bias", (rs.standard_normal((dout,)) * 0.1).astype(dtype))])
