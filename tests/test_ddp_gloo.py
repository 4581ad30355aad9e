"""Data-parallel path with world_size 2 on gloo/CPU (kernels emulated): the flat-gradient all-reduce buckets of
engine.MVAEStep must reproduce 'average of the per-shard gradients, then one Adam step' (local BatchNorm)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, sync_bn=False, reduce_bf16=None, precision="fp32"):
    for p in (ROOT, os.path.join(ROOT, "multimodal-dynamics_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmdyn_hip import ops
    from mmdyn_hip.engine import MVAEStep
    from mmdyn_hip.models import InjectedNoise
    from mmdyn_hip.utils.seeded_init import seeded_batch, seeded_noise
    from emu_backend import EmuBackend
    import test_model_emu as T
    ops.set_backend(EmuBackend())
    B = 2
    inputs, targets = seeded_batch(B * world, 1234)
    sl = slice(rank * B, (rank + 1) * B)
    eps, masks = seeded_noise(B, 256, 7, 8, 100 + rank)
    m = T.build("cnn-mvae", True, True, "cpu")
    step = MVAEStep(m, noise=InjectedNoise(eps, masks), process_group=dist.group.WORLD, world_size=world,
                    sync_bn=sync_bn, grad_reduce_bf16=reduce_bf16, precision=precision)
    loss = step.train_step([x[sl] for x in inputs], [x[sl] for x in targets], 0.02)
    torch.save({"flat": step.params.flat.clone(), "order": step.params.order, "offsets": step.params.offsets,
                "loss": float(loss), "grad": step.params.grad.clone(), "buckets_bf16": step._grad16 is not None}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_matches_averaged_gradients(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    torch.testing.assert_close(r0["flat"], r1["flat"], rtol=0, atol=0)       # replicas stay identical

    from oracle import mvae_oracle as O
    from mmdyn_hip.models.shapes import state_dict_shapes
    from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    inputs, targets = seeded_batch(4, 1234)
    grads, losses = [], []
    for rank in range(world):
        prm, buf = O.split_state(sd)
        sl = slice(rank * 2, (rank + 1) * 2)
        eps, masks = seeded_noise(2, 256, 7, 8, 100 + rank)
        _, loss, _ = O.evaluate_mvae(prm, [x[sl] for x in inputs], [x[sl] for x in targets], eps, masks, 0.02, 1000.0, True, buf)
        loss.backward()
        grads.append({k: v.grad for k, v in prm.items()})
        losses.append(float(loss.detach()))
    assert r0["loss"] == pytest.approx(losses[0], rel=1e-4) and r1["loss"] == pytest.approx(losses[1], rel=1e-4)
    prm, _ = O.split_state(sd)
    names = list(prm)
    for k in names:
        prm[k].grad = 0.5 * (grads[0][k] + grads[1][k])
    O.Adam([prm[k] for k in names], lr=1e-3).step()
    for k in names:
        o = r0["offsets"][k]
        got = r0["flat"][o:o + prm[k].numel()].view_as(prm[k])
        err = (got - prm[k].detach()).abs()
        # first Adam step moves every element by ~lr*sign(g): allow sign flips only where the gradient is at noise level
        assert float((err > 1e-5).float().mean()) < 0.02, (k, float(err.max()))
        assert float(err.max()) <= 2.1e-3, k


@pytest.mark.timeout(600)
def test_two_rank_step_in_the_split_arithmetic(tmp_path):
    """precision="fp32x3" (bench.py's default arithmetic) through the data-parallel schedule: fp32 gradient buckets (the split changes
    the matrix pipe, not the storage), replicas identical after the step, the parameters those of the fp32 two-rank step (the CPU
    emulation of the kernels computes the split arithmetic's fp32 contract)."""
    world = 2
    a, b = tmp_path / "x3", tmp_path / "f32"
    a.mkdir()
    b.mkdir()
    mp.start_processes(_worker, args=(world, _free_port(), str(a), False, None, "fp32x3"), nprocs=world, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(world, _free_port(), str(b)), nprocs=world, join=True, start_method="spawn")
    r0, r1 = torch.load(a / "rank0.pt", weights_only=False), torch.load(a / "rank1.pt", weights_only=False)
    f0 = torch.load(b / "rank0.pt", weights_only=False)
    torch.testing.assert_close(r0["flat"], r1["flat"], rtol=0, atol=0)
    assert not r0["buckets_bf16"]
    assert r0["loss"] == pytest.approx(f0["loss"], rel=1e-6)
    torch.testing.assert_close(r0["flat"], f0["flat"], rtol=0, atol=1e-6)


@pytest.mark.timeout(600)
def test_two_rank_step_with_bf16_gradient_buckets(tmp_path):
    """Gradient buckets on the wire in bf16 (the rule of the 16-bit storage modes, forced here on the fp32 arithmetic so
    that the bucket rounding is the only difference): the reduced gradient every rank feeds to Adam is the oracle's
    rank-averaged gradient to bf16 precision (stated bound: 1e-2 relative L2 per tensor = the bf16s gradient bound of
    tests/test_model_gpu.py; measured ~2e-3), every element is a bf16 value, replicas stay identical, and the Adam step
    lands where the oracle's does."""
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), False, True), nprocs=world, join=True,
                       start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    torch.testing.assert_close(r0["flat"], r1["flat"], rtol=0, atol=0)
    torch.testing.assert_close(r0["grad"], r1["grad"], rtol=0, atol=0)
    assert int((r0["grad"].view(torch.int32) & 0xFFFF).abs().max()) == 0          # widened bf16 values: low halves are zero

    from oracle import mvae_oracle as O
    from mmdyn_hip.models.shapes import state_dict_shapes
    from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    inputs, targets = seeded_batch(4, 1234)
    grads = []
    for rank in range(world):
        prm, buf = O.split_state(sd)
        sl = slice(rank * 2, (rank + 1) * 2)
        eps, masks = seeded_noise(2, 256, 7, 8, 100 + rank)
        _, loss, _ = O.evaluate_mvae(prm, [x[sl] for x in inputs], [x[sl] for x in targets], eps, masks, 0.02, 1000.0, True, buf)
        loss.backward()
        grads.append({k: v.grad for k, v in prm.items()})
    prm, _ = O.split_state(sd)
    names = list(prm)
    worst = 0.0
    for k in names:
        want = 0.5 * (grads[0][k] + grads[1][k])
        o = r0["offsets"][k]
        got = 0.5 * r0["grad"][o:o + want.numel()].view_as(want)                   # the buffer holds the SUM over ranks
        rel = float((got - want).norm() / (want.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 1e-2, (k, rel)
        prm[k].grad = want
    assert worst > 1e-5                                                            # ... and it really was rounded
    O.Adam([prm[k] for k in names], lr=1e-3).step()
    for k in names:
        o = r0["offsets"][k]
        got = r0["flat"][o:o + prm[k].numel()].view_as(prm[k])
        err = (got - prm[k].detach()).abs()
        assert float((err > 1e-5).float().mean()) < 0.03, (k, float(err.max()))
        assert float(err.max()) <= 2.1e-3, k


@pytest.mark.timeout(600)
def test_two_rank_sync_bn_equals_global_batch(tmp_path):
    """sync_bn=True: BatchNorm statistics over the global batch.  Two ranks with 2 samples each must then reproduce
    the single-process reference step on the concatenated batch of 4: loss = mean of the rank losses, parameters after
    one Adam step equal (the gradient average over ranks is the global-batch gradient)."""
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True,
                       start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    torch.testing.assert_close(r0["flat"], r1["flat"], rtol=0, atol=0)

    from oracle import mvae_oracle as O
    from mmdyn_hip.models.shapes import state_dict_shapes
    from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    inputs, targets = seeded_batch(4, 1234)
    per_rank = [seeded_noise(2, 256, 7, 8, 100 + r) for r in range(world)]
    eps = [torch.cat([per_rank[r][0][i] for r in range(world)]) for i in range(7)]
    masks = [torch.cat([per_rank[r][1][i] for r in range(world)]) for i in range(8)]
    prm, buf = O.split_state(sd)
    _, loss, _ = O.evaluate_mvae(prm, inputs, targets, eps, masks, 0.02, 1000.0, True, buf)
    loss.backward()
    assert 0.5 * (r0["loss"] + r1["loss"]) == pytest.approx(float(loss.detach()), rel=1e-4)
    names = list(prm)
    O.Adam([prm[k] for k in names], lr=1e-3).step()
    for k in names:
        o = r0["offsets"][k]
        got = r0["flat"][o:o + prm[k].numel()].view_as(prm[k])
        err = (got - prm[k].detach()).abs()
        assert float((err > 1e-5).float().mean()) < 0.02, (k, float(err.max()))
        assert float(err.max()) <= 2.1e-3, k


def _worker_uneven(rank, world, port, out_dir, sizes):
    for p in (ROOT, os.path.join(ROOT, "multimodal-dynamics_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmdyn_hip import ops
    from mmdyn_hip.engine import MVAEStep
    from mmdyn_hip.models import InjectedNoise, setup_model
    from mmdyn_hip.utils.seeded_init import seeded_batch, seeded_noise, seeded_state_dict
    from emu_backend import EmuBackend
    import test_model_emu as T
    ops.set_backend(EmuBackend())
    lo, B = sum(sizes[:rank]), sizes[rank]
    inputs, targets = seeded_batch(sum(sizes), 1234)
    sl = slice(lo, lo + B)
    eps, masks = seeded_noise(B, 256, 7, 8, 100 + rank)
    # every rank seeds its replica DIFFERENTLY: the engine must start all of them from rank 0's weights and buffers
    m = setup_model("cnn-mvae", cross_modal=True, use_pose=True, **T.MODEL_KW)
    m.load_state_dict(seeded_state_dict(m.state_dict(), rank))
    m.train()
    step = MVAEStep(m, noise=InjectedNoise(eps, masks), process_group=dist.group.WORLD, world_size=world)
    start = step.params.flat.clone()
    loss = step.train_step([x[sl] for x in inputs], [x[sl] for x in targets], 0.02)
    torch.save({"flat": step.params.flat.clone(), "start": start, "offsets": step.params.offsets, "loss": float(loss)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_four_ranks_uneven_shards_and_unequal_seeds(tmp_path):
    """world_size 4 with a global batch of 7 that does not divide evenly (shards of 2, 2, 2, 1 samples) and replicas
    seeded differently per rank: after construction every rank holds rank 0's parameters; after one step all ranks hold
    the same parameters, equal to 'mean over ranks of the per-shard gradients, one Adam step' (DistributedDataParallel
    semantics: each rank's loss is the mean over ITS samples, local BatchNorm)."""
    world, sizes = 4, (2, 2, 2, 1)
    mp.start_processes(_worker_uneven, args=(world, _free_port(), str(tmp_path), sizes), nprocs=world, join=True,
                       start_method="spawn")
    rs = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(world)]
    for r in rs[1:]:
        torch.testing.assert_close(r["start"], rs[0]["start"], rtol=0, atol=0)
        torch.testing.assert_close(r["flat"], rs[0]["flat"], rtol=0, atol=0)

    from oracle import mvae_oracle as O
    from mmdyn_hip.models.shapes import state_dict_shapes
    from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)          # rank 0's seed
    inputs, targets = seeded_batch(sum(sizes), 1234)
    grads = []
    for rank in range(world):
        prm, buf = O.split_state(sd)
        lo, B = sum(sizes[:rank]), sizes[rank]
        sl = slice(lo, lo + B)
        eps, masks = seeded_noise(B, 256, 7, 8, 100 + rank)
        _, loss, _ = O.evaluate_mvae(prm, [x[sl] for x in inputs], [x[sl] for x in targets], eps, masks, 0.02, 1000.0, True, buf)
        loss.backward()
        assert rs[rank]["loss"] == pytest.approx(float(loss.detach()), rel=1e-4)
        grads.append({k: v.grad for k, v in prm.items()})
    prm, _ = O.split_state(sd)
    names = list(prm)
    for k in names:
        prm[k].grad = sum(g[k] for g in grads) / world
    O.Adam([prm[k] for k in names], lr=1e-3).step()
    for k in names:
        o = rs[0]["offsets"][k]
        got = rs[0]["flat"][o:o + prm[k].numel()].view_as(prm[k])
        err = (got - prm[k].detach()).abs()
        assert float((err > 1e-5).float().mean()) < 0.02, (k, float(err.max()))
        assert float(err.max()) <= 2.1e-3, k


@pytest.mark.timeout(900)
def test_bench_multi_rank_dry_run(tmp_path):
    """bench.py under torch.distributed.run with 2 ranks, rehearsed on CPU (gloo, emulated kernels): the rendezvous,
    barriers, max-over-ranks timing and the ONE JSON line on rank 0 -- n_gpus, global_batch, parallelism, scaling --
    are exercised before a multi-GPU node ever runs it."""
    import json
    import subprocess
    env = dict(os.environ, MMDYN_BENCH_DRYRUN="emu", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "samples/s"
    assert d["value"] == pytest.approx(4 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-6)
    assert "dry_run" in d and "cpu_baseline" not in d


def test_bench_gpus_n_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher and no WORLD_SIZE must start its two ranks itself (a child
    torch.distributed.run) and relay ONE JSON line saying n_gpus == 2 -- never silently measure one rank."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMDYN_BENCH_DRYRUN="emu")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert "launching 2 ranks" in out.stderr
