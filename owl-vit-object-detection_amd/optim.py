"""Fused AdamW over the model's flat trainable bucket.

Drop-in for the reference's optimizer (ref main.py:56-60: `torch.optim.AdamW(model.parameters(), lr,
weight_decay)`, single param group -- weight decay also hits LayerNorm affine, biases and the query bank):
same constructor keywords, `zero_grad()` / `step()`.  One HIP kernel touches (param, grad, m, v) once and
refreshes the bf16 compute copy.  torch.optim.AdamW on `model.parameters()` also works (the parameters are
ordinary f32 leaves); this class is the fast path.
"""
import torch

from . import _lib, ops
from .autograd import _attach_grads, _grads_attached


class FusedAdamW:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, overlap: bool = False):
        """overlap=True (keyword-only addition): backward + this step run on the model's tail stream under the NEXT forward's frozen prefix
        (models.OwlViT.overlap_tail) -- bitwise the in-line schedule.  The model's own forward / backward / zero_grad / state_dict order
        themselves behind the deferred tail; anything else that reads `p.grad` or the parameters calls `model.finish()` first (and finds the
        gradient bucket already zeroed after a step: the zeroing of the next `zero_grad()` is part of the tail)."""
        self.model = model
        if overlap:
            model.overlap_tail = True
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(model.flat_param)
        self.exp_avg_sq = torch.zeros_like(model.flat_param)
        self.step_count = 0
        self.grad_scale = 1.0          # set to 1/world_size by the data-parallel wrapper

    def _attach(self):
        """Re-attach detached `.grad` views (nn.Module.zero_grad(set_to_none=True) drops them).  Attaching zero-fills the bucket on the CURRENT
        stream: behind a deferred tail that may still be reading / zeroing it."""
        m = self.model
        if not _grads_attached(m):          # (detached, or foreign .grad tensors: re-attaching zero-fills / copies into the bucket)
            self._sync_tail()
            _attach_grads(m)

    def _sync_tail(self):
        """Order the current stream behind everything the tail stream holds (deferred backward / all-reduce / AdamW / zeroing)."""
        m = self.model
        m._wait_params()
        if m.flat_param.is_cuda and getattr(m, "_tail_stream_", None) is not None and torch.cuda.current_stream() != m._tail_stream_:
            torch.cuda.current_stream().wait_stream(m._tail_stream_)

    def zero_grad(self, set_to_none: bool = False):
        self._attach()
        if getattr(self.model, "_grad_clean", False):
            return                       # a deferred step (ddp.DataParallel(overlap=True)) already zeroed the bucket on its stream
        self.model._wait_params()
        self.model.flat_grad.zero_()

    @torch.no_grad()
    def step(self):
        m = self.model
        self._attach()
        m._grad_clean = False
        self.step_count += 1
        deferred = getattr(m, "overlap_tail", False) and m.flat_param.is_cuda and torch.cuda.current_stream() != m._tail_stream
        if deferred:
            # the backward of this step is (or may be) still running on the tail stream: the update and the zeroing of the bucket follow it there
            ts = m._tail_stream
            ts.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ts):
                self._launch()
                m.flat_grad.zero_()               # the next zero_grad(), done where nothing races with the reads above
                ev = torch.cuda.Event()
                ev.record(ts)
            m._param_event = ev
            m._grad_clean = True
        else:
            self._launch()
        m._mark_bf16_current()          # one-shot token for the next forward (models.OwlViT.__init__)

    def _launch(self):
        m = self.model
        _lib.call("owl_adamw_step", ops.stream(), m.flat_param, m.flat_grad, self.exp_avg, self.exp_avg_sq, m.flat_bf16,
                  m.flat_numel, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                  float(self.grad_scale))

    def state_dict(self):
        self._sync_tail()               # a deferred step may still be writing the moments on the tail stream
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr, betas=self.betas,
                    eps=self.eps, weight_decay=self.weight_decay)

    def load_state_dict(self, sd):
        self._sync_tail()
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
