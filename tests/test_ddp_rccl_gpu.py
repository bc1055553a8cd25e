"""-m gpu: the RCCL leg of the data-parallel path and the bench launcher contract.

* `bench.py --gpus N` launches N ranks itself; on a box with fewer GPUs it must exit non-zero with a clear message
  instead of quietly benchmarking one GPU;
* one forced rank over the real `nccl` (= RCCL) backend: the bench line carries rccl_ranks / backend / allreduce_ms;
* two ranks over RCCL (skipped with fewer than 2 GPUs): identical replicas after a step, and the data-parallel identity
  against the single-process global batch.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OWL_FORCE_DIST"):
        env.pop(k, None)
    env.update(kw)
    return env


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


@pytest.mark.timeout(600)
def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0",
                        "--arch", "tiny", "--batch", "2", "--no-cpu-baseline"], capture_output=True, text=True, env=_env(), timeout=500)
    assert r.returncode != 0
    assert f"only {n} GPU" in r.stderr and "n_gpus" not in r.stdout


@pytest.mark.timeout(900)
@pytest.mark.parametrize("schedule", ["in-line", "overlap"])
def test_bench_single_forced_rccl_rank_reports_the_collective(schedule):
    """One forced rank over the real RCCL backend, both optimizer schedules (with more than one rank the bench defaults to the overlapped
    one: backward + RCCL + AdamW on the model's tail stream while the next step's sub-batch streams already run the frozen prefix); batch 8 so that the
    encoder runs its two sub-batch streams beside them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "1", "--arch", "small",
                        "--batch", "8", "--no-cpu-baseline"] + (["--overlap"] if schedule == "overlap" else []),
                       capture_output=True, text=True, env=_env(OWL_FORCE_DIST="1"), timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["backend"].startswith("nccl")
    assert out["allreduce_ms"] > 0 and out["allreduce_bytes"] > 0 and out["value"] > 0
    assert out["config"]["optimizer_schedule"].startswith("backward + all-reduce + AdamW on the tail stream" if schedule == "overlap" else "in-line")
    assert out["config"]["encoder_streams"] == 2
    assert out["replicas_equal"] is True
    # the fields a multi-GPU line must carry (VERDICT r04 #6a): the collective's time / bytes and the phase they were measured under, per-rank throughput,
    # the schedule reported, and under which phase every HIP-event figure was taken
    assert out["allreduce_measured_under"] == "in-line schedule"
    assert 0 < out["images_per_sec_per_rank_min"] <= out["images_per_sec_per_rank_max"]
    assert out["images_per_sec_per_rank_min"] == pytest.approx(out["value"], rel=1e-3)           # one rank: the slowest rank IS the job
    assert out["schedule"] == out["config"]["optimizer_schedule"]
    for r_ in [out["roofline"]] + out["roofline_other"]:
        assert r_["measured_under"] == "in-line schedule", r_
    assert "images_per_sec_incl_h2d" in out["config"]
    if schedule == "overlap":      # the unloseable flow: in-line measured first, pre-flight bitwise check, then the deferred tail
        assert out["config"]["schedule_check"].startswith("pre-flight: 2 steps from one state, deferred tail == in-line bitwise")
        assert out["config"]["images_per_sec_inline_schedule"] > 0 and out["config"]["images_per_sec_deferred_tail_schedule"] > 0
        assert out["deferred_phase"] == "ok"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fault,expect", [("mismatch", "pre-flight MISMATCH"), ("raise", "deferred-tail phase raised RuntimeError"), ("hang", "did not finish within")])
def test_bench_multi_rank_flow_keeps_the_inline_line_when_the_deferred_phase_fails(fault, expect):
    """VERDICT r03 #3, the other half: whatever the deferred-tail phase does on the first real multi-GPU node -- disagree with the in-line schedule, raise,
    or never come back -- the in-line measurement taken before it is printed, says why, and the process leaves with 0."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--arch", "small", "--batch", "4",
                        "--no-cpu-baseline", "--no-compare", "--overlap", "--inject-fault", fault, "--watchdog-seconds", "20"],
                       capture_output=True, text=True, env=_env(OWL_FORCE_DIST="1"), timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["config"]["optimizer_schedule"] == "in-line" and expect in out["config"]["schedule_check"], out["config"]
    assert out["value"] > 0 and out["replicas_equal"] is True and "images_per_sec_deferred_tail_schedule" not in out["config"]
    assert out["deferred_phase"] == {"mismatch": "mismatch", "raise": "raised", "hang": "watchdog fired"}[fault]      # a machine-readable verdict beside the prose


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,arch", [(2, "small"), (8, "tiny")])
def test_bench_two_gloo_ranks_on_one_gpu_run_the_multi_rank_flow(world, arch):
    """bench.py's flow for more than one rank with REAL ranks on the one visible GPU (gloo moves the bucket; `--backend gloo` is
    test-only): in-line measurement first, pre-flight deferred == in-line bitwise, deferred measurement, replicas equal after both.  Two ranks on the
    `small` config, and the driver's rank count -- eight -- on the `tiny` one."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "4", "--warmup", "1",
                        "--arch", arch, "--batch", "4", "--no-cpu-baseline", "--no-compare"], capture_output=True, text=True, env=_env(), timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["backend"].startswith("gloo")
    assert out["replicas_equal"] is True
    assert out["config"]["schedule_check"].startswith("pre-flight: 2 steps from one state, deferred tail == in-line bitwise"), out["config"]
    assert out["config"]["optimizer_schedule"].startswith("backward + all-reduce + AdamW on the tail stream")
    assert out["config"]["global_batch"] == 4 * world and out["value"] > 0
    assert out["deferred_phase"] == "ok" and 0 < out["images_per_sec_per_rank_min"] <= out["images_per_sec_per_rank_max"]
    assert out["images_per_sec_per_rank_min"] * world == pytest.approx(out["value"], rel=1e-3)    # the slowest rank sets the job's throughput



_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from owl_vit_object_detection_amd import ddp, synth, weights
from owl_vit_object_detection_amd.config import get_config
from owl_vit_object_detection_amd.losses import PushPullLoss
from owl_vit_object_detection_amd.models import OwlViT
from owl_vit_object_detection_amd.optim import FusedAdamW
rank, world, local = ddp.init_from_env({backend!r})
dev = torch.device("cuda", local if {backend!r} == "nccl" else 0)       # (gloo variant: both ranks on the one visible GPU)
cfg = get_config("small")
model = OwlViT(cfg, weights.make_weights(cfg), dev)
if rank == 1:
    model.flat_param.add_(0.5)                    # the broadcast from rank 0 must undo this
opt = FusedAdamW(model, lr=1e-3, weight_decay=0.1, overlap={opt_overlap})
dp = ddp.DataParallel(model, opt, overlap={overlap})
per = 2
dp.check_equal_batches(per)
img = torch.from_numpy(synth.make_images(cfg, per, first=rank * per)).to(dev)
labels, boxes = synth.make_targets(cfg, per, first=rank * per, max_boxes=5)
crit = PushPullLoss(cfg.n_classes, None)
opt.zero_grad()
pb, _, ps, _ = model(img)
l = crit(ps, [torch.from_numpy(x).to(dev) for x in labels], pb, [torch.from_numpy(x).to(dev) for x in boxes])
(l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward()
dp.finish()                                       # deferred tail: the backward runs on the tail stream -- order this stream behind it before reading .grad
local_grad = model.flat_grad.clone()
dp.sync_and_step(); dp.finish(); torch.cuda.synchronize()
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), local=local_grad.cpu().numpy(), param=model.flat_param.cpu().numpy())
hist = []
for k in range(3):                                # a few more steps through the steady-state schedule (step k+1's frozen prefix under step k's tail)
    opt.zero_grad()
    pb, _, ps, _ = model(img)
    l = crit(ps, [torch.from_numpy(x).to(dev) for x in labels], pb, [torch.from_numpy(x).to(dev) for x in boxes])
    tot = l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]
    tot.backward(); dp.sync_and_step(); hist.append(tot.detach())
dp.finish(); torch.cuda.synchronize()
np.savez(os.path.join({out!r}, f"steady{{rank}}.npz"), param=model.flat_param.cpu().numpy(), loss=torch.stack(hist).cpu().numpy())
dist.barrier(); dist.destroy_process_group()
'''


def _run_two_ranks(tmp_path, overlap, backend, opt_overlap=False):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path), overlap=overlap, backend=backend, opt_overlap=opt_overlap))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, env=_env(), timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    return [np.load(tmp_path / f"{name}{k}.npz") for name in ("rank", "steady") for k in (0, 1)]


@pytest.mark.timeout(900)
def test_two_gloo_ranks_on_one_gpu_deferred_tail_is_bitwise_the_inline_schedule(tmp_path):
    """Two real ranks on the ONE visible GPU (gloo all-reduce of the CUDA bucket): the schedule with more than one rank -- backward + all-reduce + AdamW on
    the tail stream under the next forward's frozen prefix -- leaves both replicas identical and bitwise where the in-line schedule leaves them, after one
    step and after three more."""
    res = {}
    for overlap in (False, True):
        d = tmp_path / ("overlap" if overlap else "inline"); d.mkdir()
        res[overlap] = _run_two_ranks(d, overlap, "gloo")
    for overlap in (False, True):
        r0, r1, s0, s1 = res[overlap]
        np.testing.assert_array_equal(r0["param"], r1["param"]); np.testing.assert_array_equal(s0["param"], s1["param"])
    np.testing.assert_array_equal(res[False][0]["param"], res[True][0]["param"])
    np.testing.assert_array_equal(res[False][0]["local"], res[True][0]["local"])
    np.testing.assert_array_equal(res[False][2]["param"], res[True][2]["param"])
    np.testing.assert_array_equal(res[False][2]["loss"], res[True][2]["loss"])
    np.testing.assert_array_equal(res[False][3]["loss"], res[True][3]["loss"])


@pytest.mark.timeout(900)
def test_two_gloo_ranks_tail_stream_backward_with_the_inline_collective(tmp_path):
    """FusedAdamW(overlap=True) sends the backward to the tail stream while DataParallel(overlap=False) all-reduces in-line on the compute
    stream -- the collective must order itself behind the tail (ddp.DataParallel._order_behind_tail), or it reads a bucket the backward is still writing.
    Replicas identical, bitwise the plain in-line schedule."""
    d0 = tmp_path / "inline"; d0.mkdir()
    d1 = tmp_path / "mixed"; d1.mkdir()
    ref = _run_two_ranks(d0, False, "gloo")
    mix = _run_two_ranks(d1, False, "gloo", opt_overlap=True)
    np.testing.assert_array_equal(mix[0]["param"], mix[1]["param"]); np.testing.assert_array_equal(mix[2]["param"], mix[3]["param"])
    np.testing.assert_array_equal(ref[0]["local"], mix[0]["local"])
    np.testing.assert_array_equal(ref[0]["param"], mix[0]["param"])
    np.testing.assert_array_equal(ref[2]["param"], mix[2]["param"])
    np.testing.assert_array_equal(ref[2]["loss"], mix[2]["loss"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_rccl_step(tmp_path, overlap):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's 8-GPU node); the same code runs on 2 gloo ranks in tests/test_ddp_cpu.py and, on one GPU, in the test above")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path), overlap=overlap, backend="nccl", opt_overlap=False))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, env=_env(), timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["param"], r1["param"])            # identical replicas after the step
    # data-parallel identity: mean of the per-rank gradients == gradient of the 4-image global batch on one GPU
    from owl_vit_object_detection_amd import synth, weights
    from owl_vit_object_detection_amd.config import get_config
    from owl_vit_object_detection_amd.losses import PushPullLoss
    from owl_vit_object_detection_amd.models import OwlViT
    cfg = get_config("small")
    model = OwlViT(cfg, weights.make_weights(cfg), "cuda")
    img = torch.from_numpy(synth.make_images(cfg, 4)).cuda()
    labels, boxes = synth.make_targets(cfg, 4, max_boxes=5)
    crit = PushPullLoss(cfg.n_classes, None)
    pb, _, ps, _ = model(img)
    l = crit(ps, [torch.from_numpy(x).cuda() for x in labels], pb, [torch.from_numpy(x).cuda() for x in boxes])
    (l["loss_ce"] + l["loss_bg"] + l["loss_bbox"] + l["loss_giou"]).backward()
    glob = model.flat_grad.cpu().numpy()
    mean = 0.5 * (r0["local"] + r1["local"])
    np.testing.assert_allclose(mean, glob, rtol=1e-3, atol=1e-5 * float(np.abs(glob).max()))


def test_c_abi_allreduce_entry_single_rank():
    """the C ABI's own collective entry (for hosts without PyTorch): `owl_allreduce_sum_f32` over a communicator the CALLER built with RCCL's C API
    (here through ctypes: ncclGetUniqueId + ncclCommInitRank with one rank -- all a 1-GPU box allows).  One rank's sum is the bucket itself, bit for bit; the
    call is enqueued on the given stream; a null communicator is refused with a message."""
    import ctypes
    from owl_vit_object_detection_amd import _lib, ops
    try:
        rccl = ctypes.CDLL("librccl.so")          # the instance torch already mapped (its bundled copy): the library binds the same one (RTLD_NOLOAD first)
    except OSError:
        pytest.skip("librccl.so not loadable")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        g = torch.randn(8_684_292, device="cuda")
        ref = g.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _lib.call("owl_allreduce_sum_f32", ops.stream(), comm.value, g, g.numel())
        side.synchronize()
        assert torch.equal(g, ref)
        with pytest.raises(_lib.OwlLibError, match="null communicator"):
            _lib.call("owl_allreduce_sum_f32", ops.stream(), None, g, g.numel())
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
