"""N > 1 path on CPU: two processes, gloo backend, the SAME DataParallel / allreduce code that runs over
RCCL on the GPUs.  Per-rank gradients come from the CPU oracle on per-rank shards; the check is the
data-parallel identity:  mean over ranks of per-rank mean-loss gradients == gradient of the global batch,
and that every rank ends the step with identical parameters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _CpuModel:
    """Stand-in exposing the flat-bucket surface of OwlViT (flat_param / flat_grad) on CPU."""

    def __init__(self, cfg, W):
        from owl_vit_object_detection_amd import weights
        self.names = [n for n in W if weights.is_trainable(n)]
        self.shapes = {n: W[n].shape for n in self.names}
        self.flat_param = torch.cat([torch.from_numpy(W[n]).reshape(-1) for n in self.names]).clone()
        self.flat_grad = torch.zeros_like(self.flat_param)

    def set_grads(self, grads):
        self.flat_grad.copy_(torch.cat([grads[n].reshape(-1) for n in self.names]))


class _CpuAdamW:
    def __init__(self, model, lr, wd):
        self.model, self.lr, self.wd, self.grad_scale, self.t = model, lr, wd, 1.0, 0
        self.m = torch.zeros_like(model.flat_param); self.v = torch.zeros_like(model.flat_param)

    def zero_grad(self):
        self.model.flat_grad.zero_()

    def step(self):
        from oracle import owl_oracle as O
        self.t += 1
        p, self.m, self.v = O.adamw_step(self.model.flat_param, self.model.flat_grad * self.grad_scale, self.m, self.v,
                                         self.t, lr=self.lr, wd=self.wd)
        self.model.flat_param.copy_(p)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import owl_oracle as O
    from owl_vit_object_detection_amd import ddp, synth, weights
    from owl_vit_object_detection_amd.config import get_config
    r, w, _ = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = get_config("tiny")
    W = weights.make_weights(cfg)
    model = _CpuModel(cfg, W)
    if rank == 1:
        model.flat_param.add_(1.0)                       # broadcast from rank 0 must undo this
    opt = _CpuAdamW(model, lr=1e-3, wd=0.1)
    dp = ddp.DataParallel(model, opt)
    assert opt.grad_scale == 1.0 / world
    per = 2                                               # images per rank
    imgs = synth.make_images(cfg, per, first=rank * per)
    labels, boxes = synth.make_targets(cfg, per, first=rank * per, max_boxes=5)
    scales = torch.tensor([3.0, 3.5, 4.0, 3.2])
    wt = {k: torch.from_numpy(v) for k, v in W.items()}
    _, losses, grads = O.train_step(cfg, wt, torch.from_numpy(imgs), [torch.from_numpy(l) for l in labels],
                                    [torch.from_numpy(b) for b in boxes], scales)
    opt.zero_grad()
    model.set_grads(grads)
    local_grad = model.flat_grad.clone()
    dp.sync_and_step()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), local=local_grad.numpy(), summed=model.flat_grad.numpy(),
             param=model.flat_param.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_data_parallel_step(tmp_path):
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    # one SUM all-reduce: both ranks hold the same summed bucket = local0 + local1
    np.testing.assert_allclose(r0["summed"], r0["local"] + r1["local"], rtol=1e-6, atol=1e-8)
    np.testing.assert_array_equal(r0["summed"], r1["summed"])
    np.testing.assert_array_equal(r0["param"], r1["param"])          # identical replicas after the step
    # data-parallel identity vs the single-process global batch (mean-over-images loss semantics)
    sys.path.insert(0, ROOT)
    from oracle import owl_oracle as O
    from owl_vit_object_detection_amd import synth, weights
    from owl_vit_object_detection_amd.config import get_config
    cfg = get_config("tiny")
    W = weights.make_weights(cfg)
    imgs = synth.make_images(cfg, 4)
    labels, boxes = synth.make_targets(cfg, 4, max_boxes=5)
    wt = {k: torch.from_numpy(v) for k, v in W.items()}
    _, _, g = O.train_step(cfg, wt, torch.from_numpy(imgs), [torch.from_numpy(l) for l in labels],
                           [torch.from_numpy(b) for b in boxes], torch.tensor([3.0, 3.5, 4.0, 3.2]))
    names = [n for n in W if weights.is_trainable(n)]
    glob = torch.cat([g[n].reshape(-1) for n in names]).numpy()
    scale = float(np.abs(glob).max())
    np.testing.assert_allclose(r0["summed"] / world, glob, rtol=1e-3, atol=1e-5 * scale)


def test_single_process_is_a_noop():
    from owl_vit_object_detection_amd import ddp
    t = torch.arange(8, dtype=torch.float32)
    assert torch.equal(ddp.allreduce_flat(t.clone()), t) and torch.equal(ddp.broadcast_flat(t.clone()), t)
