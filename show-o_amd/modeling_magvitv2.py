"""`MAGVITv2` — drop-in replacement of the reference class `models.modeling_magvitv2.MAGVITv2`
(reference models/modeling_magvitv2.py:402-433) backed by the gfx950 VQ engine in libshowo_hip.so.

`encode / get_code / decode_code` keep the reference signatures and the state-dict keys are the
reference's (`encoder.down.0.block.0.conv1.weight`, ..., buffers `quantize.embedding`,
`quantize.power_vals`).  The torch modules are parameter containers only; there is no PyTorch fallback.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .persistence import PretrainedMixin


def _norm(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class _Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = _norm(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = _norm(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1, 1, 0)


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _norm(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)


class _Resample(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride, 1 if stride == 1 else 0)


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.block_1 = _Res(c, c)
        self.attn_1 = _Attn(c)
        self.block_2 = _Res(c, c)


class _Encoder(nn.Module):
    """parameter layout of VQGANEncoder (reference models/modeling_magvitv2.py:62-139)"""

    def __init__(self, ch, ch_mult, num_res_blocks, z_channels):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch, 3, 1, 1)
        self.down = nn.ModuleList()
        in_mult = (1,) + tuple(ch_mult)
        block_in = ch
        for l in range(len(ch_mult)):
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            block_in, block_out = ch * in_mult[l], ch * ch_mult[l]
            for _ in range(num_res_blocks[l]):
                lvl.block.append(_Res(block_in, block_out))
                block_in = block_out
            if l != len(ch_mult) - 1:
                lvl.downsample = _Resample(block_in, 2)
            self.down.append(lvl)
        self.mid = _Mid(block_in)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, z_channels, 3, 1, 1)
        self.quant_conv = nn.Conv2d(z_channels, z_channels, 1)


class _Decoder(nn.Module):
    """parameter layout of VQGANDecoder (reference models/modeling_magvitv2.py:278-362)"""

    def __init__(self, ch, ch_mult, num_res_blocks, z_channels):
        super().__init__()
        n = len(ch_mult)
        block_in = ch * ch_mult[n - 1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = _Mid(block_in)
        ups = []
        for l in reversed(range(n)):
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            block_out = ch * ch_mult[l]
            for _ in range(num_res_blocks[l]):
                lvl.block.append(_Res(block_in, block_out))
                block_in = block_out
            if l != 0:
                lvl.upsample = _Resample(block_in, 1)
            ups.insert(0, lvl)
        self.up = nn.ModuleList(ups)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, 3, 3, 1, 1)
        self.post_quant_conv = nn.Conv2d(z_channels, z_channels, 1)


class _LFQBuffers(nn.Module):
    """buffers of LFQuantizer (reference models/modeling_magvitv2.py:186-197); the kernels use the bit form."""

    def __init__(self, codebook_dim=13):
        super().__init__()
        idx = torch.arange(2 ** codebook_dim)
        binary = (idx.unsqueeze(1) >> torch.arange(codebook_dim - 1, -1, -1, dtype=torch.long)) & 1
        self.register_buffer("embedding", binary.float() * 2 - 1)
        self.register_buffer("power_vals", 2 ** torch.arange(codebook_dim - 1, -1, -1))
        self.e_dim = codebook_dim
        self.codebook_size = 2 ** codebook_dim

    def get_indices(self, z_q):
        """reference :201-206: z_q [B,C,h,w] -> ids int64 [B,1,h,w], bit of channel c (MSB first) = (z_q[c] > 0)"""
        if not z_q.is_cuda:
            raise RuntimeError("show-o_amd quantizes on the GPU (no CPU path exists)")
        B, C, h, w = z_q.shape
        z = z_q.detach().float().contiguous()
        ids = torch.empty((B, h * w), dtype=torch.int64, device=z.device)
        _lib.call("showo_lfq_pack_nchw", _lib.ptr(z), _lib.ptr(ids), B, C, h * w, _lib.stream())
        return ids.view(B, 1, h, w)

    def get_codebook_entry(self, indices, shape=None):
        """reference :208-221: ids [B,n] -> z_q fp32 [B,C,h,w] of +-1"""
        if not indices.is_cuda:
            raise RuntimeError("show-o_amd de-quantizes on the GPU (no CPU path exists)")
        b, n = indices.shape
        h, w = (int(math.sqrt(n)), int(math.sqrt(n))) if shape is None else shape
        ids = indices.to(torch.int64).contiguous()
        zq = torch.empty((b, self.e_dim, h, w), dtype=torch.float32, device=ids.device)
        _lib.call("showo_lfq_unpack_nchw", _lib.ptr(ids), _lib.ptr(zq), b, self.e_dim, h * w, _lib.stream())
        return zq


class MAGVITv2(PretrainedMixin, nn.Module):
    ENC = dict(ch_mult=(1, 2, 2, 4, 4), num_res_blocks=(4, 3, 4, 3, 4))
    DEC = dict(ch_mult=(1, 1, 2, 2, 4), num_res_blocks=(4, 4, 3, 4, 3))

    def __init__(self, ch=128, z_channels=13, max_batch=8, max_res=256, precision=1):
        """precision: 1 (default) = split-bf16 MFMA operands (hi+lo pairs, fp32-class accuracy: token ids track the
        fp32 reference); 0 = plain bf16 operands (3x fewer MFMAs, ~1e-2 relative error)."""
        super().__init__()
        self.precision = int(precision)
        self.ch, self.z_channels = ch, z_channels
        self.encoder = _Encoder(ch, self.ENC["ch_mult"], self.ENC["num_res_blocks"], z_channels)
        self.decoder = _Decoder(ch, self.DEC["ch_mult"], self.DEC["num_res_blocks"], z_channels)
        self.quantize = _LFQBuffers(z_channels)
        self.max_batch, self.max_res = max_batch, max_res
        self._vq = None
        self._versions = None

    def forward(self, pixel_values, return_loss=False):
        pass  # the reference's forward is empty too (modeling_magvitv2.py:413-414)

    def configure_workspace(self, max_batch, max_res):
        self.max_batch, self.max_res = int(max_batch), int(max_res)
        self._drop()

    def _drop(self):
        if self._vq is not None:
            _lib.load().showo_vq_destroy(self._vq)
        self._vq, self._versions = None, None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def mark_weights_dirty(self):
        """re-upload every parameter at the next call (needed after updates through `.data`, which do not bump tensor versions)"""
        if self._vq is not None:
            self._versions = {}

    def engine(self):
        _lib.require_gpu()
        lib = _lib.load()
        dev = self.decoder.conv_out.weight.device
        if dev.type != "cuda":
            raise RuntimeError("MAGVITv2 parameters must live on the GPU (model.to('cuda')); no CPU path exists")
        if self._vq is None:
            import ctypes as C
            cfg = _lib.VQConfig()
            cfg.ch, cfg.z_channels = self.ch, self.z_channels
            for i, (m, b) in enumerate(zip(self.ENC["ch_mult"], self.ENC["num_res_blocks"])):
                cfg.enc_ch_mult[i], cfg.enc_blocks[i] = m, b
            for i, (m, b) in enumerate(zip(self.DEC["ch_mult"], self.DEC["num_res_blocks"])):
                cfg.dec_ch_mult[i], cfg.dec_blocks[i] = m, b
            cfg.enc_levels, cfg.dec_levels = len(self.ENC["ch_mult"]), len(self.DEC["ch_mult"])
            cfg.max_batch, cfg.max_res = self.max_batch, self.max_res
            cfg.precision = self.precision
            h = C.c_void_p()
            _lib.check(lib.showo_vq_create(C.byref(cfg), C.byref(h)), "showo_vq_create")
            self._vq, self._versions = h, {}
        for k, v in self.state_dict().items():
            if k.startswith("quantize."):
                continue
            ver = (v.data_ptr(), v._version)
            if self._versions.get(k) != ver:
                src = v.detach()
                if src.dtype != torch.float32 or not src.is_contiguous():
                    src = src.float().contiguous()
                _lib.call("showo_vq_load", self._vq, k.encode(), _lib.ptr(src), src.numel(), _lib.stream())
                self._versions[k] = ver
                if src is not v:
                    torch.cuda.current_stream().synchronize()
        missing = lib.showo_vq_missing(self._vq)
        if missing:
            raise RuntimeError(f"VQ engine is missing {missing} tensors")
        return self._vq

    def _run_encoder(self, pixel_values, want_z):
        vq = self.engine()
        B, C, H, W = pixel_values.shape
        if C != 3:
            raise ValueError("pixel_values must be [B,3,H,W]")
        if B > self.max_batch or H * W > self.max_res * self.max_res:
            self.configure_workspace(max(B, self.max_batch), max(int(math.ceil(math.sqrt(H * W))), self.max_res))
            vq = self.engine()
        x = pixel_values.detach().float().contiguous()
        ids = torch.empty((B, (H // 16) * (W // 16)), dtype=torch.int64, device=x.device)
        z = torch.empty((B, self.z_channels, H // 16, W // 16), dtype=torch.float32, device=x.device) if want_z else None
        _lib.call("showo_vq_get_code", vq, _lib.ptr(x), B, H, W, _lib.ptr(ids), _lib.ptr(z), _lib.stream())
        return ids, z

    # ---- reference API (modeling_magvitv2.py:416-433) --------------------------------------------------
    def get_code(self, pixel_values):
        return self._run_encoder(pixel_values, False)[0]

    def get_code_and_latents(self, pixel_values):
        """extra (not in the reference): also returns the pre-quantisation latents z [B,13,h,w]."""
        return self._run_encoder(pixel_values, True)

    def encode(self, pixel_values, return_loss=False):
        ids, z = self._run_encoder(pixel_values, True)
        zq = torch.empty_like(z)  # quantized_states = sign pattern of the ids (:239-241, 208-221)
        B, Cz, h, w = z.shape
        _lib.call("showo_lfq_unpack_nchw", _lib.ptr(ids), _lib.ptr(zq), B, Cz, h * w, _lib.stream())
        return zq, ids

    def decode_code(self, codebook_indices, shape=None):
        vq = self.engine()
        B, n = codebook_indices.shape
        if shape is None:
            h = w = int(math.sqrt(n))
        else:
            h, w = shape
        if h * w != n:
            raise ValueError("codebook_indices does not match the latent shape")
        if B > self.max_batch or (16 * h) * (16 * w) > self.max_res * self.max_res:
            self.configure_workspace(max(B, self.max_batch), max(int(math.ceil(16 * math.sqrt(h * w))), self.max_res))
            vq = self.engine()
        ids = codebook_indices.to(torch.int64).contiguous()
        img = torch.empty((B, 3, 16 * h, 16 * w), dtype=torch.float32, device=ids.device)
        _lib.call("showo_vq_decode_code", vq, _lib.ptr(ids), B, h, w, _lib.ptr(img), _lib.stream())
        return img
