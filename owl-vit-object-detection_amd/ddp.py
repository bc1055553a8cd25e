"""Data-parallel training: one process per GPU, ONE all-reduce per step.

The reference has no distributed code; images are independent units, so the build
shards the batch across ranks (pure data parallel, full weight replica per GPU) and exchanges exactly one
contiguous buffer per step: the flat gradient bucket (8 684 292 f32 = 34.7 MB for B/16, C = 10).  On ROCm
the "nccl" backend IS RCCL; over xGMI a ring all-reduce of 34.7 MB on 8 GPUs is ~0.4 ms, far below the
compute time of a step.
Loss semantics: mean over images; with equal per-rank batches the mean of per-rank
means equals the global mean, so the summed bucket is scaled by 1/world (inside the fused optimizer, or by
one multiply for any other optimizer).
Two schedules:
  * in-line (default): all-reduce -> AdamW on the compute stream right after backward;
  * `overlap=True`: all-reduce -> AdamW -> zero the bucket run on a SIDE stream and the next step's forward
    starts at once -- embeddings and encoder layers 0..10 are frozen (ref src/models.py:173-184), so nothing
    before the trainable layer depends on the update; the compute stream waits for the side stream exactly
    where the first trainable tensor is read (models.OwlViT._wait_params).  Same arithmetic, same order:
    parameters are bitwise equal to the in-line schedule (tests/test_autograd_contract_gpu.py::test_overlapped_optimizer_schedule_is_bitwise_the_inline_schedule,
    tests/test_ddp_rccl_gpu.py).  Until the side stream is done the bucket is being reduced, read and then ZEROED there: the model's
    forward / backward / zero_grad and `state_dict()` order themselves behind it; anything else that reads parameters or gradients
    (logging a gradient norm from `p.grad`, cloning `flat_param`) must call `finish()` first -- and reads zeros from the gradients
    afterwards (inspect them with the in-line schedule).
The same code runs on CPU tensors with the gloo backend (tests/test_ddp_cpu.py; in-line schedule).
"""
import os
import socket

import torch
import torch.distributed as dist


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_from_env(backend: str = None):
    """torchrun-style bootstrap (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("OWL_FORCE_DIST", "0") == "1"      # exercise the RCCL path with a single rank (testing)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError("WORLD_SIZE > 1 needs MASTER_PORT (launch with torch.distributed.run, or `bench.py --gpus N`)")
            os.environ["MASTER_PORT"] = str(_free_port())    # single forced rank: any free port
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            if torch.cuda.device_count() <= local:
                raise RuntimeError(f"LOCAL_RANK={local} but only {torch.cuda.device_count()} GPU(s) are visible")
            torch.cuda.set_device(local)
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("OWL_FORCE_DIST", "0") == "1")


def allreduce_flat(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of the flat gradient bucket, in place (single collective per step)."""
    if _active(group):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def broadcast_flat(flat_param: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every rank start from rank `src`'s trainable parameters."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_param, src=src, group=group)
    return flat_param


class DataParallel:
    """Couples a model (anything exposing .flat_grad) with an optimizer: `sync_and_step()` = all-reduce(sum) -> scale
    1/world -> optimizer step.  A fused optimizer (anything exposing .grad_scale) folds the 1/world into its own pass; for
    any other optimizer (e.g. torch.optim.AdamW over model.parameters()) the summed bucket is scaled explicitly."""

    def __init__(self, model, optimizer, group=None, overlap: bool = False):
        self.model, self.optimizer, self.group = model, optimizer, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.fused = hasattr(optimizer, "grad_scale")
        if self.fused:
            self.optimizer.grad_scale = 1.0 / self.world
        if hasattr(model, "flat_param"):
            broadcast_flat(model.flat_param, 0, group)
            if hasattr(model, "refresh_compute_weights"):
                model.refresh_compute_weights(force=True)
        self.overlap = bool(overlap) and model.flat_grad.is_cuda and self.fused
        if self.overlap:
            model.overlap_tail = True                   # backward, all-reduce and AdamW all run on the model's tail stream
        self.side = model._tail_stream if self.overlap else None
        self._checked_batch = False

    def check_equal_batches(self, batch_size: int):
        """mean-of-means == global mean only for equal per-rank batches: verify once (one tiny collective)."""
        if self.world > 1 and not self._checked_batch:
            t = torch.tensor([batch_size, -batch_size], dtype=torch.int64, device=self.model.flat_grad.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            if int(t[0]) != -int(t[1]):
                raise RuntimeError(f"per-rank batch sizes differ (min {-int(t[1])}, max {int(t[0])}): the data-parallel mean would be biased")
        self._checked_batch = True

    def _step(self):
        # A model whose backward runs on its tail stream (FusedAdamW(overlap=True) switched overlap_tail on, or a previous deferred step is still
        # in flight) is still accumulating into / zeroing the bucket THERE: the in-line collective on this stream must be ordered behind it.
        self._order_behind_tail()
        allreduce_flat(self.model.flat_grad, self.group)
        if not self.fused and self.world > 1:
            self.model.flat_grad.mul_(1.0 / self.world)
        self.optimizer.step()

    def sync_and_step(self):
        # (model.overlap_tail switched off for a step -- bench.py does it on its kernel-timing steps -- means THIS step is in-line, backward included)
        if not (self.overlap and self.model.overlap_tail):
            self._step()
            return
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)                     # (the backward itself already runs on the tail stream: models.OwlViT.overlap_tail)
        with torch.cuda.stream(self.side):
            allreduce_flat(self.model.flat_grad, self.group)     # RCCL on the tail stream, behind the backward
        self.optimizer.step()                           # FusedAdamW: update + bucket zeroing on the tail stream, event for the next forward

    def _order_behind_tail(self):
        m = self.model
        if getattr(m, "flat_grad", None) is None or not m.flat_grad.is_cuda:
            return
        if hasattr(m, "_wait_params"):
            m._wait_params()                            # a deferred step's event (all-reduce / AdamW / zeroing of the previous step)
        if getattr(m, "overlap_tail", False) or getattr(m, "_tail_stream_", None) is not None:
            torch.cuda.current_stream().wait_stream(m._tail_stream)    # a backward that was sent to the tail stream

    def finish(self):
        """Make the current stream wait for a deferred step (before reading parameters outside the model's forward)."""
        ev = getattr(self.model, "_param_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.model._param_event = None
