"""Data-parallel training step on the HIP path (reference training/train.py:510-628 + accelerate/DeepSpeed gradient
exchange, accelerate_configs/*.yaml).

`Trainer.step()` = Showo.forward with labels -> backward -> gradient exchange -> AdamW, all on HIP kernels; the only
collective of the step is the gradient average (RCCL through torch.distributed, one all-reduce per gradient bucket:
head, then transformer blocks last to first, then the embedding), issued as soon as the bucket's backward kernels are
queued so that it overlaps the backward of the next block.  One process per GPU; every rank draws its own batch and the
three losses are per-rank means, averaged across ranks by the gradient average exactly as DDP does.
"""
import ctypes as C

import torch

from . import _lib

NO_DECAY = ("bias", "layer_norm.weight", "mlm_ln.weight", "embeddings.weight")  # reference training/train.py:211


class _DevView:
    """zero-copy torch view of a raw device buffer owned by the HIP library (__cuda_array_interface__ v2)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_view(ptr, n, device):
    return torch.as_tensor(_DevView(ptr, n), device=device)


class HipWireOps:
    """staging of the gradient wire on the device: HIP kernels on the current stream (include/showo_hip.h)"""

    @staticmethod
    def pack(grad, wire, scale):
        _lib.call("showo_grad_wire_pack", grad.data_ptr(), wire.data_ptr(), grad.numel(), float(scale), _lib.stream())

    @staticmethod
    def unpack(wire, grad):
        _lib.call("showo_grad_wire_unpack", wire.data_ptr(), grad.data_ptr(), grad.numel(), _lib.stream())

    @staticmethod
    def scale(grad, s):
        _lib.call("showo_scale_f32", grad.data_ptr(), grad.numel(), float(s), _lib.stream())


class GradientExchange:
    """The ONE exchange step of a data-parallel iteration (reference: accelerate / DeepSpeed around
    `accelerator.backward`, training/train.py:449,612): every gradient bucket is averaged over the process group with one
    all-reduce, launched as soon as the bucket's backward kernels are queued (`launch(b)`), and all of them are completed before the
    optimizer runs (`finish()`).

    wire = "bf16" (default, the reference's `mixed_precision: bf16` wire, 2 bytes per gradient): the bucket is scaled by 1/world
    and rounded to a bf16 staging buffer, the staging buffer is summed by the collective and widened back into the fp32 bucket.
    wire = "fp32": the fp32 bucket itself is summed and then scaled (twice the volume, no rounding).
    `ops` supplies pack / unpack / scale for the device the buckets live on (HipWireOps for the GPU path)."""

    def __init__(self, buckets, dist, group=None, wire="bf16", ops=HipWireOps):
        if wire not in ("bf16", "fp32"):
            raise ValueError("wire must be 'bf16' or 'fp32'")
        self.buckets, self.dist, self.group, self.wire, self.ops = buckets, dist, group, wire, ops
        self.world = dist.get_world_size(group)
        self.stage = [torch.empty(b.numel(), dtype=torch.bfloat16, device=b.device) for b in buckets] if wire == "bf16" else None
        self.works = []
        self._measure, self._spans, self._bspans = False, [], []

    def measure(self, on=True):
        """record the GPU time of every finish() (device events on the compute stream) from now on; exposed_ms() reads the mean,
        exposed_ms_per_bucket() the same split by bucket (wait for that bucket's collective + widening its wire)"""
        self._measure, self._spans, self._bspans = bool(on), [], []

    def exposed_ms_per_bucket(self):
        """{bucket index: mean GPU ms per step the compute stream spent waiting for / unpacking that bucket inside finish()}: buckets
        whose all-reduce hid behind the backward show only their unpack time; None when nothing was measured"""
        if not self._bspans:
            return None
        torch.cuda.synchronize()
        acc, cnt = {}, {}
        for b, e0, e1 in self._bspans:
            acc[b] = acc.get(b, 0.0) + e0.elapsed_time(e1)
            cnt[b] = cnt.get(b, 0) + 1
        return {b: acc[b] / cnt[b] for b in sorted(acc)}

    def exposed_ms(self):
        """mean GPU time per step that the compute stream spent in finish(): waiting for collectives that did not hide behind the
        backward, plus widening / scaling the wire.  None when nothing was measured (CPU buckets, measure() off)."""
        if not self._spans:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._spans) / len(self._spans)

    def wire_bytes(self):
        """bytes every rank hands to the collective per step"""
        return sum(b.numel() for b in self.buckets) * (2 if self.wire == "bf16" else 4)

    def launch(self, b):
        if self.wire == "bf16":
            self.ops.pack(self.buckets[b], self.stage[b], 1.0 / self.world)
            t = self.stage[b]
        else:
            t = self.buckets[b]
        self.works.append((self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), b))

    def finish(self):
        span = None
        if self._measure and self.buckets and self.buckets[0].is_cuda:
            span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            span[0].record()
        self._finish()
        if span is not None:
            span[1].record()
            self._spans.append(span)

    def _finish(self):
        timed = self._measure and self.buckets and self.buckets[0].is_cuda
        for w, b in self.works:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if w is not None:
                w.wait()
            if self.wire == "bf16":
                self.ops.unpack(self.stage[b], self.buckets[b])
            else:
                self.ops.scale(self.buckets[b], 1.0 / self.world)
            if timed:
                e1.record()
                self._bspans.append((b, e0, e1))
        self.works = []


def train_mask(tr, attention_mask):
    """dense additive [B,1,L,L] mask -> contiguous fp32 tensor; IntervalMask (prompting_utils.intervals_* /
    training_utils.build_training_batch) -> registered with the trainer, returns None (no dense mask is passed)"""
    from .prompting_utils import IntervalMask
    if attention_mask is None:
        return None
    if isinstance(attention_mask, IntervalMask):
        # no host-side check(): a mask the intervals cannot represent turns the losses into NaN on the device
        _lib.call("showo_trainer_use_intervals", tr, _lib.ptr(attention_mask.iv.contiguous()), _lib.ptr(attention_mask.flag))
        return None
    return attention_mask.detach().float().contiguous()


class Trainer:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, coeffs=(1.0, 0.1, 1.0), group=None,
                 wire="bf16", max_grad_norm=None, force_exchange=False, reserve_cus=None):
        """wire: "bf16" | "fp32" gradient wire of the exchange (GradientExchange).  max_grad_norm: global-norm clipping before the
        optimizer (training/train.py:614-615; null in the shipped stage-1 YAMLs).  force_exchange: run the exchange even in a
        process group of one rank (tests: drives the RCCL path on a single GPU).  reserve_cus: while an exchange is active the step's
        kernels run on a stream that keeps this many CUs free for RCCL's channel kernels (showo_stream_create_cu_mask; None = the
        SHOWO_RESERVE_CUS environment variable, default 0 = no mask: profiles/r4_exchange_contention.txt)."""
        import os
        self.reserve_cus = int(os.environ.get("SHOWO_RESERVE_CUS", "0")) if reserve_cus is None else int(reserve_cus)
        self._masked_stream = None
        self.model, self.lr, self.betas, self.eps, self.wd, self.coeffs, self.group = model, lr, betas, eps, weight_decay, coeffs, group
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.params = [("showo." + n, p) for n, p in model.showo.named_parameters()]
        self.m = {n: torch.zeros_like(p) for n, p in self.params}
        self.v = {n: torch.zeros_like(p) for n, p in self.params}
        self._wire, self._force_exchange = wire, force_exchange
        self._bound = None
        self._bind()

    def _binding_key(self, params=None):
        params = self.params if params is None else params
        return (self.tr.value if hasattr(self.tr, "value") else int(self.tr),) + tuple(p.data_ptr() for _, p in params)

    def _bind(self):
        """(Re)create everything that holds raw pointers into the native trainer or the parameters: the gradient-bucket views, the
        exchange's staging buffers and the optimizer's parameter table.  Called by __init__ and by step() whenever the trainer
        handle changed (configure_workspace / _drop_engine re-create it) or a parameter's storage moved (load_state_dict(assign=True),
        .to(), ...): a stale handle would have no parameters bound and the native AdamW would write through dangling pointers."""
        self.tr = self.model.trainer()
        dev = self.params[0][1].device
        for n, p in self.params:
            if self.m[n].device != p.device or self.m[n].shape != p.shape:
                self.m[n] = torch.zeros_like(p)
                self.v[n] = torch.zeros_like(p)
        lib = _lib.load()
        self.buckets = []
        for b in range(lib.showo_train_num_buckets(self.tr)):
            ptr, n = C.c_void_p(), C.c_int64()
            _lib.check(lib.showo_train_bucket(self.tr, b, C.byref(ptr), C.byref(n)), "showo_train_bucket")
            self.buckets.append(device_view(ptr.value, n.value, dev))
        for n, p in self.params:  # master weights + moments are registered once per binding; the step is one C call
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise ValueError(f"{n}: the native optimizer updates contiguous fp32 master parameters in place")
            _lib.call("showo_train_bind_param", self.tr, n.encode(), p.data_ptr(), self.m[n].data_ptr(), self.v[n].data_ptr(), p.numel())
        import torch.distributed as dist
        group = self.group
        active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self._force_exchange)
        self.exchange = GradientExchange(self.buckets, dist, group, wire=self._wire) if active else None
        self._bound = self._binding_key()

    # ---- checkpoint / resume of the optimizer (reference: accelerator.save_state / load_state, training/train.py:851-889, 429-443)
    def state_dict(self):
        """torch.optim.AdamW layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} with the parameters
        numbered in `model.showo.named_parameters()` order, so the file is interchangeable with an AdamW built over that list"""
        state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[n].detach().clone(),
                     "exp_avg_sq": self.v[n].detach().clone()} for i, (n, _) in enumerate(self.params)}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group], "param_names": [n for n, _ in self.params]}

    def load_state_dict(self, sd):
        """moments are copied INTO the tensors bound to the native optimizer (showo_train_bind_param); lr / betas / eps / weight
        decay / step count are taken from the file"""
        names = [n for n, _ in self.params]
        if "param_names" in sd and list(sd["param_names"]) != names:
            raise ValueError("optimizer state was saved for a different parameter list")
        if len(sd["state"]) not in (0, len(names)):
            raise ValueError(f"optimizer state has {len(sd['state'])} entries, the model has {len(names)} parameters")
        steps = set()
        with torch.no_grad():
            for i, n in enumerate(names):
                if i not in sd["state"]:
                    continue
                st = sd["state"][i]
                if tuple(st["exp_avg"].shape) != tuple(self.m[n].shape):
                    raise ValueError(f"{n}: moment shape {tuple(st['exp_avg'].shape)} != {tuple(self.m[n].shape)}")
                self.m[n].copy_(st["exp_avg"])
                self.v[n].copy_(st["exp_avg_sq"])
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: not an optimizer state this trainer can resume")
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.wd = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        self.step_count = steps.pop() if steps else 0

    def set_lr(self, lr):
        """learning-rate schedulers (training/train.py:303-310 `get_scheduler`) call this between steps"""
        self.lr = float(lr)

    def compute_stream(self):
        """the CU-masked stream the step runs on while a gradient exchange is active (None: the caller's current stream)"""
        if self.exchange is None or self.reserve_cus <= 0:
            return None
        if self._masked_stream is None:
            h = C.c_void_p()
            _lib.check(_lib.load().showo_stream_create_cu_mask(self.reserve_cus, C.byref(h)), "showo_stream_create_cu_mask")
            self._masked_handle = h
            self._masked_stream = torch.cuda.ExternalStream(h.value)
        return self._masked_stream

    def logging_means(self, losses, mask_prob=None):
        """The reference gathers four tensors per step for logging (training/train.py:603-610: accelerator.gather of the three losses
        and the masking rate, each repeated batch-size times, then .mean() = the mean over ranks).  Here that is ONE all-reduce of a
        4-vector, issued next to the gradient buckets; returns a device tensor [loss_t2i, loss_lm, loss_mmu, masking_rate] averaged
        over the ranks (no host sync).  Without a process group the inputs come back unchanged."""
        v = torch.cat([losses.detach().float().reshape(3), (mask_prob.detach().float().mean().reshape(1) if mask_prob is not None
                                                             else torch.zeros(1, device=losses.device))])
        ex = self.exchange
        if ex is None or ex.world == 1:
            return v
        ex.dist.all_reduce(v, op=ex.dist.ReduceOp.SUM, group=ex.group)
        return v / ex.world

    def close(self):
        """release the CU-masked compute stream (one HIP stream per Trainer that ran with reserve_cus > 0)"""
        h = getattr(self, "_masked_handle", None)
        if h is not None:
            self._masked_stream = None
            self._masked_handle = None
            try:
                _lib.call("showo_stream_destroy", h)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, input_ids, attention_mask, labels, batch_size_t2i, batch_size_lm, batch_size_mmu, max_seq_length):
        """one optimisation step; returns the three losses (fp32 device tensor [3])"""
        ms = None
        if getattr(self, "exchange", None) is not None and self.reserve_cus > 0:
            ms = self.compute_stream()
        if ms is None:
            return self._step(input_ids, attention_mask, labels, batch_size_t2i, batch_size_lm, batch_size_mmu, max_seq_length)
        cur = torch.cuda.current_stream()
        ms.wait_stream(cur)
        with torch.cuda.stream(ms):
            out = self._step(input_ids, attention_mask, labels, batch_size_t2i, batch_size_lm, batch_size_mmu, max_seq_length)
        cur.wait_stream(ms)
        out.record_stream(cur)  # allocated on the masked stream's pool, consumed on the caller's stream
        return out

    def _step(self, input_ids, attention_mask, labels, batch_size_t2i, batch_size_lm, batch_size_mmu, max_seq_length):
        # model.trainer() re-uploads parameters whose (data_ptr, version) changed since the last call (load_state_dict, resume,
        # manual edits); the native AdamW below updates them through raw pointers and refreshes the engine images itself
        self.tr = self.model.trainer()
        cur = [("showo." + n, p) for n, p in self.model.showo.named_parameters()]  # assign=True loads replace the Parameter objects
        if self._binding_key(cur) != self._bound:  # new native trainer, or a parameter's storage moved: rebuild every raw-pointer binding
            self.params = cur
            self._bind()
        m, tr, s = self.model, self.tr, _lib.stream
        B, L = input_ids.shape
        ids = input_ids.to(torch.int64).contiguous()
        lab = labels.to(torch.int64).contiguous()
        mask = train_mask(tr, attention_mask)
        losses = torch.empty(3, dtype=torch.float32, device=ids.device)
        # the weights of the three losses are known up front here (training/train.py:600): the forward's cross-entropy pass also
        # writes d(loss)/d(logits), so the backward does not read the [B*L, V] logits again
        ex = self.exchange
        nL = m.arch["num_hidden_layers"]
        _lib.call("showo_train_set_loss_weights", tr, self.coeffs[0], self.coeffs[1], self.coeffs[2], 1)
        try:  # ONE scope for the announcement: a forward that raises must withdraw it too
            try:
                _lib.call("showo_train_forward", tr, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(lab), B, L, batch_size_t2i, batch_size_lm,
                          batch_size_mmu, max_seq_length, None, _lib.ptr(losses), s())
            finally:
                _lib.call("showo_trainer_use_intervals", tr, None, None)
            # `lab` is this step's own contiguous copy / view and is not written between the forward above and this call: the d(logits)
            # of the forward's cross-entropy pass is reused only for the SAME forward (every forward invalidates it), pointer, split and weights.
            _lib.call("showo_train_backward_head", tr, _lib.ptr(lab), batch_size_t2i, batch_size_lm, batch_size_mmu, max_seq_length,
                      self.coeffs[0], self.coeffs[1], self.coeffs[2], s())
        finally:
            # the announcement is per step: a later forward with labels outside Trainer.step (the module's autograd path, an eval loop)
            # must not write the T x Vp d(logits) tensor (1.3 GB at the stage-1 batch) for a backward that never comes
            _lib.call("showo_train_set_loss_weights", tr, 0.0, 0.0, 0.0, 0)
        if ex is not None:
            ex.launch(nL + 1)
        for i in range(nL - 1, -1, -1):
            _lib.call("showo_train_backward_layer", tr, i, s())
            if ex is not None:
                ex.launch(i + 1)  # the collective of block i overlaps the backward of block i - 1
        _lib.call("showo_train_backward_embed", tr, s())
        if ex is not None:
            ex.launch(0)
            ex.finish()
        if self.max_grad_norm is not None:
            self.clip_grad_norm_(self.max_grad_norm)
        self.step_count += 1
        _lib.call("showo_train_adamw_step", tr, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, s())
        return losses

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ over the flat gradient buffer (reference training/train.py:614-615
        `accelerator.clip_grad_norm_(model.parameters(), max_grad_norm)`): g *= max_norm / (||g||_2 + 1e-6) when that is < 1.
        ONE fixed-order HIP reduction over the library's flat gradient buffer + a scale pass that reads the coefficient from device
        memory (csrc/train_kernels.hip showo_grad_clip_norm): no ATen arithmetic, no host synchronisation, bit-reproducible.
        Returns the total norm (device scalar)."""
        b0, bl = self.buckets[0], self.buckets[-1]
        n = (bl.data_ptr() + bl.numel() * 4 - b0.data_ptr()) // 4
        if sum(b.numel() for b in self.buckets) != n:
            raise RuntimeError("gradient buckets are not one contiguous range")
        if getattr(self, "_clip_ws", None) is None:
            self._clip_ws = torch.empty(_lib.load().showo_grad_clip_ws_doubles(), dtype=torch.float64, device=b0.device)
        out2 = torch.empty(2, dtype=torch.float32, device=b0.device)
        _lib.call("showo_grad_clip_norm", b0.data_ptr(), n, float(max_norm), _lib.ptr(self._clip_ws), _lib.ptr(out2), _lib.stream())
        return out2[0]
