#!/usr/bin/env python3
"""bench.py -- cnn-mvae visuotactile+pose training throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step of the cnn-mvae seq_modeling problem on one synthetic batch that is already
resident in HBM: the 7 modality-subset ELBO passes, backward and Adam (problems.py:148-156, 473-546), fp32,
64x64 visual + tactile + 7-DoF pose, 256 samples per GPU (weak scaling: pure data parallel, gradients
all-reduced over RCCL).  Prints ONE JSON line on rank 0.

Other BASELINE configs (same JSON contract, the workload named in config.workload):
    configs[2]  python bench.py --dtype bf16s --batch 128
    configs[3]  python bench.py --image-size 128 --problem dyn_modeling --batch 128
    configs[4]  python bench.py --image-size 256 --dtype fp16 --batch 256
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16), no sparsity
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0      # same rate as bf16 (MI355X_MICROARCH.md, matrix cores table)
KL_WEIGHT = 1.0 / 50               # epoch 0 of the reference's annealing schedule (problems.py:212-216)


def algo_gflop_per_sample(size=64, use_pose=True):
    """Algorithmic FLOPs of one train step per sample, counted as SURVEY.md section 8(d) does: every mathematically
    distinct contraction once (encoder trunks once per modality, image heads per pass, the 4 live passes of each image
    decoder, the pose encoder once, 4 live pose-decoder passes), forward + input gradient + weight gradient, no input
    gradient for a network's first layer.  size = 64 gives the survey's 1 012 188 160 MAC = 2.024 GFLOP; 128 / 256 apply
    the same count to the extended stacks of models/shapes.py."""
    from mmdyn_hip.models.shapes import extra_stages, encoder_channels, decoder_channels, FEAT, HID
    e = extra_stages(size)
    L = 256
    enc, H = [], size // 2
    first = H * H * 32 * 3 * 16
    for j, (cin, cout) in enumerate(encoder_channels(e)):
        last = j == 2 + e
        Ho = H - 3 if last else H // 2
        enc.append(Ho * Ho * cout * cin * 16)
        H = Ho
    fc = FEAT * HID
    trunk = first + sum(enc) + fc
    heads = HID * 2 * L
    enc_mod = trunk * 2 + (trunk - first) + 4 * 3 * heads            # fwd + wgrad + dgrad (not for conv1); heads x 4 passes
    dec, H = [], 5
    for j, (cin, cout) in enumerate(decoder_channels(e)):
        Ho = H + 3 if j == 0 else 2 * H
        dec.append(H * H * cin * cout * 16)          # transposed: every INPUT pixel meets all 16 taps
        H = Ho
    dec_pass = L * FEAT + sum(dec) + (2 * H) * (2 * H) * 3 * 32 * 4   # last layer: each output pixel sees 4 of the 16 taps
    macs = 2 * enc_mod + 2 * 4 * 3 * dec_pass
    if use_pose:
        macs += (7 * HID + HID * HID) * 2 + HID * HID + 3 * heads + 4 * 3 * (L * HID + HID * HID + HID * 7)
    return 2.0 * macs / 1e9


ALGO_GFLOP_PER_SAMPLE = 2.024      # SURVEY.md section 8(d), 64 x 64 (== algo_gflop_per_sample(64), asserted in main)


def _sha256(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


PEAK_HBM_BYTES_PER_S = 8.0e12       # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s achievable


def sources_sha256():
    """sha-256 over the kernel sources (csrc/*.hip, *.h): the provenance key of the committed PMC profiles."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "multimodal-dynamics_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_traffic(workload_key):
    """HBM-side bytes of one step of this workload from the committed rocprofv3 PMC passes (profiles/r4/traffic_<key>.json, else r3's:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; profiles/collect_r4.sh traffic).  PMC counters cannot be
    read from inside this process, so the values are the last profiled ones -- WITH their provenance: the profile records
    the sha-256 of the kernel sources it was taken on, and the values are reported as None (plus the reason) when the
    sources have changed since.  Returns (dict | None, note)."""
    d = None
    for rnd in ("r4", "r3"):                      # the latest round's collection first
        rel = ("profiles", rnd, f"traffic_{workload_key}.json")
        try:
            d = json.load(open(os.path.join(ROOT, *rel)))
            break
        except (OSError, ValueError):
            continue
    if d is None:
        return None, {"stale": f"no PMC profile committed for this workload (profiles/r4|r3/traffic_{workload_key}.json)"}
    note = {"profile": "/".join(rel), "collected": d.get("collected"), "sources_sha256": d.get("sources_sha256")}
    if d.get("sources_sha256") and d["sources_sha256"] == sources_sha256():
        return d, note
    note["stale"] = "kernel sources changed since the profile was collected"
    return None, note


def host_cpu_share():
    """CPU threads this process may really use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    found = False
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n, found = min(n, max(1, int(float(quota) / float(period) + 0.5))), True
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n, found = min(n, max(1, int(q / per + 0.5))), True
    except (OSError, ValueError):
        pass
    if not found:
        n = min(n, 16)          # a one-GPU box's documented CPU share
    return max(1, n)


def host_cpu_all():
    """os.cpu_count() as SURVEY.md section 8(d) specifies for the CPU baseline, bounded by the affinity mask and by a
    cgroup quota when one is visible (more runnable threads than the quota allows only thrash)."""
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(batch, threads, size=64, budget_s=30.0):
    """The CPU oracle (a port of the reference path executing the reference's own schedule: 8 encoder + 14
    decoder forwards, autograd backward, Adam) timed on this box's host cores.  Bounded: a probe step at bs=32
    decides whether the full bs=256 step fits the budget; otherwise a bs=64 sample is timed."""
    from oracle import mvae_oracle as O
    from mmdyn_hip.models.shapes import state_dict_shapes
    from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
    torch.set_num_threads(threads)

    def run(bs, steps):
        prm, buf = O.split_state(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True, size=size), 0))
        inputs, targets = seeded_batch(bs, 1234, size=size)
        eps, masks = seeded_noise(bs, 256, 7, 8, 4321)
        opt = O.Adam([prm[k] for k in prm], lr=1e-3)
        times, loss = [], None
        for s in range(steps):
            t0 = time.perf_counter()
            opt.zero_grad()
            _, loss, _ = O.evaluate_mvae(prm, inputs, targets, eps, masks, KL_WEIGHT, 1000.0, True, buf)
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
            print(f"[cpu_baseline] bs={bs} step {s}: {times[-1]:.2f} s ({threads} threads)", file=sys.stderr, flush=True)
        return times, float(loss.detach())

    pb = 32 if size == 64 else 8
    probe, _ = run(pb, 2)
    est_full = probe[-1] * (batch / float(pb))
    bs = batch if est_full * 3 <= 1.5 * budget_s else max(pb, min(batch, 64 if size == 64 else 16))
    n_timed = 6 if est_full * 7 <= budget_s else 2             # ~10-30 s of CPU work in total
    times, loss = run(bs, 1 + n_timed)
    best, med = min(times[1:]), sorted(times[1:])[len(times[1:]) // 2]
    cpu = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": bs / best, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n_timed} timed steps (+1 warm-up) of the cnn-mvae+pose {size}x{size} train step at bs={bs}; "
                      f"min {best:.2f} s/step (value), median {med:.2f} s/step; {cpu}, os.cpu_count()={os.cpu_count()}",
            "loss": loss}


_RESULT_FD = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when
    its communicator is created), so the real stdout is set aside for the result and everything else -- C libraries
    included, they share file descriptor 1 -- goes to stderr."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def _emit(obj):
    sys.stdout.flush()
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (json.dumps(obj) + "\n").encode())


def _self_launch(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves, the way the driver does for N > 1
    (``python -m torch.distributed.run --nproc-per-node N``), as a CHILD process -- this process has not touched the GPU
    and never will; it relays rank 0's single JSON line and exits with the child's return code.  (The reference has no
    counterpart: it is a single process, problems.py:52, 388.)"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True)
    lines = [ln for ln in proc.stdout.read().splitlines() if ln.startswith("{")]
    rc = proc.wait()
    for ln in lines[-1:]:
        _emit(json.loads(ln))
    if rc != 0 or not lines:
        raise SystemExit(rc or 1)


def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a HIP graph")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--image-size", type=int, choices=(64, 128, 256), default=64,
                    help="64: the reference's only input size (BASELINE configs[1], default).  128 / 256: the extended stacks of "
                         "models/shapes.py (configs[3] / configs[4]; no reference architecture exists for them)")
    ap.add_argument("--problem", choices=("seq_modeling", "dyn_modeling"), default="seq_modeling",
                    help="dyn_modeling (configs[3]): the batch is --seq-length frames per sequence, targets are the next frames "
                         "(DynModeling.parse_input, problems.py:765-803); the step itself is the same computation")
    ap.add_argument("--seq-length", type=int, default=4)
    ap.add_argument("--dtype", choices=("f32", "f32x3", "bf16", "bf16s", "fp16", "fp16s"), default="f32",
                    help="f32: the BASELINE configs[1] line (default).  bf16s: bf16 activation storage + bf16 matrix "
                         "cores, fp32 accumulate / master weights = the per-GPU share of configs[2] (use --batch 128).  "
                         "bf16: bf16 matrix-core operands only (fp32 storage).  fp16: fp16 matrix-core operands, fp32 "
                         "accumulate / storage / master weights (configs[4]).  fp16s: fp16 + fp16 activation storage")
    ap.add_argument("--defer-wgrad", choices=("auto", "on", "off"), default="auto",
                    help="decoder weight-gradient GEMMs on two extra streams next to the encoder backward (auto: the engine's rule, "
                         "on in fp32 on one GPU)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="BatchNorm statistics over the global batch (N > 1; one small all-reduce per BatchNorm layer and "
                         "direction, captured into the lanes' HIP graphs, each lane on its own RCCL communicator).  Default: "
                         "local statistics")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the extra timed run in the fp32x3 arithmetic that a default (f32, one GPU) run reports next to its value")
    ap.add_argument("--infer", action="store_true",
                    help="time forward-only inference instead (model.eval(): joint visual+tactile+pose pass through the "
                         "module API, running-estimate BatchNorm); prints its own JSON line, not the BASELINE metric")
    ap.add_argument("--no-grouped-heads", action="store_true",
                    help="A/B switch: launch the heads of the three encoders one by one instead of as grouped launches")
    ap.add_argument("--single-lane", action="store_true",
                    help="no visual/tactile stream overlap: per-kernel durations in a rocprofv3 trace then match "
                         "the roofline object's live HIP-event measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MMDYN_BENCH_DRYRUN=emu (tests only): rehearse the multi-rank control flow of this script -- rendezvous, sharding,
    # barriers, max-over-ranks timing, the JSON line -- on CPU with gloo and the kernels replaced by the test suite's
    # emulation.  The numbers are meaningless and the line says so; the product path below never takes this branch.
    dry = os.environ.get("MMDYN_BENCH_DRYRUN") == "emu"
    rehearse = False
    if dry:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_backend import EmuBackend
        from mmdyn_hip import ops as _ops
        _ops.set_backend(EmuBackend())
        torch.set_num_threads(2)
        dev = torch.device("cpu")
        args.no_graph, args.no_cpu_baseline = True, True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
        # MMDYN_BENCH_REHEARSE_ONE_GPU=1 (rehearsal on a one-GPU box): every rank on cuda:0, collectives over gloo -- the real
        # kernels, HIP graphs and the data-parallel schedule (buckets between the graph rows), without RCCL.  The line says so.
        rehearse = os.environ.get("MMDYN_BENCH_REHEARSE_ONE_GPU") == "1"
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    pg = None
    if world > 1 or os.environ.get("MMDYN_BENCH_FORCE_PG") == "1":      # (the flag: 1-rank rehearsal of the RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dry or rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        pg = dist.group.WORLD
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: one process per GPU is launched by torch.distributed.run "
              f"--nproc-per-node N; reporting n_gpus={world}", file=sys.stderr)

    from mmdyn_hip.engine import MVAEStep
    from mmdyn_hip.models import setup_model, NoiseSource
    from mmdyn_hip.profiling import profile_step
    from mmdyn_hip.utils.seeded_init import seeded_batch

    torch.manual_seed(0)
    S = args.image_size
    PREC = {"f32": "fp32", "f32x3": "fp32x3", "bf16": "bf16", "bf16s": "bf16s", "fp16": "fp16", "fp16s": "fp16s"}[args.dtype]
    assert abs(algo_gflop_per_sample(64) - ALGO_GFLOP_PER_SAMPLE) < 1e-3
    gflop_per_sample = algo_gflop_per_sample(S)
    model = setup_model("cnn-mvae", cross_modal=True, condition_dim=0, input_dim=S * S, architecture="cnn",
                        conditional=False, categorical_conditions=False, latent_size=256, use_pose=True).to(dev).train()
    if args.infer:
        inputs, _ = seeded_batch(args.batch, 1234 + rank, size=S)
        v, t, p = [z.to(dev) for z in inputs]
        from mmdyn_hip.engine import MVAEInference
        model.eval()
        eng = MVAEInference(model, precision=PREC,
                            use_graph=not args.no_graph,
                            seed=1234 + rank)
        for _ in range(args.warmup):
            eng([v, t], pose=p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eng([v, t], pose=p)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _emit({"metric": "visuotactile samples/sec (inference, joint v+t+p forward, eval mode)",
                          "value": args.batch * args.steps / dt, "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
                          "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": f"cnn-mvae joint inference forward, bs={args.batch}, engine.MVAEInference "
                                                 f"(prepacked weights, two streams, {'eager' if args.no_graph else 'HIP graph'})",
                                     "algorithmic_gflop_per_sample": 0.274,
                                     "tflops": args.batch * args.steps / dt * 0.274e9 / 1e12}})
        return
    step = MVAEStep(model, lr=1e-3, pose_multiplier=1000.0, noise=NoiseSource(1234 + rank), process_group=pg,
                    world_size=world, two_lanes=not args.single_lane,
                    precision=PREC, defer_wgrad={"auto": None, "on": True, "off": False}[args.defer_wgrad],
                    sync_bn=args.sync_bn and pg is not None, group_heads=not args.no_grouped_heads)
    inputs, targets = seeded_batch(args.batch, 1234 + rank, size=S)
    inputs, targets = [x.to(dev) for x in inputs], [x.to(dev) for x in targets]
    if args.problem == "dyn_modeling":
        # the frames of --seq-length-long sequences are the samples; the target of a frame is the next frame of its
        # sequence, the last frame's target is the sequence's final target (DynModeling.parse_input, problems.py:765-803:
        # roll by -1 and patch the sequence ends; the pose target is rolled without the patch, as the reference does)
        Lq = args.seq_length
        if args.batch % Lq:
            raise SystemExit("--batch must be a multiple of --seq-length for dyn_modeling")
        final = targets
        targets = [torch.roll(x, -1, dims=0) for x in inputs]
        for k in (0, 1):
            targets[k][Lq - 1::Lq] = final[k][Lq - 1::Lq]

    def eager_step():
        return step.train_step(inputs, targets, KL_WEIGHT)

    def one_step():
        if args.no_graph:
            return step.train_step(inputs, targets, KL_WEIGHT)
        return step.train_step_graphed(inputs, targets, KL_WEIGHT)

    for _ in range(args.warmup):
        one_step()
    sync()
    if pg is not None:
        dist.barrier()
        sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    host_enqueue = time.perf_counter() - t0      # host side done enqueueing; the GPU is still running if it is ahead
    sync()
    if pg is not None:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t0
    if pg is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    # per-kernel timing of one extra (untimed) step, HIP events on the launch stream
    # (one lane, eager launches: per-kernel durations are then not inflated by the other lane's kernels)
    lanes_on, step.lanes.on = step.lanes.on, False
    kern = {} if dry else profile_step(eager_step)
    step.lanes.on = lanes_on
    if pg is not None:
        dist.barrier()
    if rank != 0:
        if pg is not None:
            dist.destroy_process_group()
        return

    global_batch = args.batch * world
    sps = global_batch * args.steps / elapsed
    ig = kern.get("igemm_nt", {"ms": 0.0, "flops": 0.0, "calls": 0, "bytes": 0.0})
    wg = kern.get("wgrad_tn", {"ms": 0.0, "flops": 0.0, "calls": 0, "bytes": 0.0})
    dom = ig if ig["ms"] >= wg["ms"] else wg
    # (the implicit-GEMM entry points are served by igemm_ws_kernel -- the LDS-DMA ring, 36 of the 40 launches of the fp32
    #  bs-256 step -- and by igemm_nt_kernel, the register-staged form, for the rest)
    dom_name = "igemm_ws_kernel" if dom is ig else "wgrad_tn_kernel"
    if args.dtype == "f32x3" and dom is ig:
        dom_name = "igemm_nt_kernel (X3 instances)"
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    total_ms = sum(d["ms"] for d in kern.values())
    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else (PEAK_F16_MFMA_TFLOPS if args.dtype in ("fp16", "fp16s") else PEAK_BF16_MFMA_TFLOPS)
    if args.dtype == "f32x3":
        # fp32 products as SIX bf16 products of the exact three-term operand split: an algorithmic (fp32) flop costs six flops of
        # the bf16 matrix pipe, so the ceiling for algorithmic flops is the dense bf16 peak / 6 (2.65 x the native fp32 peak)
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
    # byte side: the committed PMC passes of THIS workload (whole step + the implicit-GEMM launches)
    wkey = f"s{S}_{args.dtype}_b{args.batch}_{args.problem}"
    prof, traffic_note = pmc_traffic(wkey)
    traffic = prof["hbm_bytes_per_launch"] if (prof and dom is ig) else None
    step_bytes = (prof["whole_step"]["fetch_bytes"] + prof["whole_step"]["write_bytes"]) if prof else None
    step_s = elapsed / args.steps
    step_hbm_frac = step_bytes / step_s / PEAK_HBM_BYTES_PER_S if step_bytes else None
    step_mfma_frac = sps / world * gflop_per_sample * 1e9 / 1e12 / peak
    kern_hbm_frac = (traffic / (dom["ms"] / max(dom["calls"], 1) * 1e-3) / PEAK_HBM_BYTES_PER_S) if traffic and dom["ms"] > 0 else None
    kern_mfma_frac = achieved / peak

    def bound_of(mfma, hbm):
        # the resource with the larger share of its peak; "launch" when neither reaches a fifth of its peak: the time then
        # goes to the chain of dependent launches, not to a roofline resource
        if hbm is None:           # no byte-side evidence (no committed PMC profile of these sources): claim the MFMA bound only
            return "mfma" if mfma >= 0.2 else "unknown"      # where the matrix side alone supports it
        if max(mfma, hbm) < 0.2:
            return "launch"
        return "hbm" if hbm > mfma else "mfma"
    arith = {"f32": "fp32", "f32x3": "fp32 storage and results; GEMMs on the bf16 matrix cores through the exact three-term split of "
                                     "their fp32 operands (six products, fp32 accumulate; error vs fp64 <= native fp32 MFMA)",
             "bf16": "bf16 matrix-core operands (fp32 accumulate, storage and master weights)",
             "bf16s": "bf16 activation storage + bf16 matrix-core operands (fp32 accumulate and master weights)",
             "fp16": "fp16 matrix-core operands (fp32 accumulate, storage and master weights)",
             "fp16s": "fp16 activation storage + fp16 matrix-core operands (fp32 accumulate and master weights, loss scale 4B)"
             }[args.dtype]
    which = ("BASELINE configs[1]" if (S == 64 and args.dtype in ("f32", "f32x3") and args.problem == "seq_modeling") else
             "per-GPU share of BASELINE configs[2]" if (S == 64 and args.dtype in ("bf16", "bf16s")) else
             "per-GPU share of BASELINE configs[3] (extension: no reference architecture at this size)" if S == 128 else
             "per-GPU share of BASELINE configs[4] (extension: no reference architecture at this size)" if S == 256 else
             "variant")
    workload = (f"cnn-mvae visuotactile+pose {S}x{S}, bs={args.batch} per GPU, {arith}, {args.problem} train step "
                f"(7 subset ELBOs + backward + Adam), {which}")
    out = {
        "metric": "visuotactile samples/sec (train) + ELBO vs CPU ref, cnn-mvae 64x64 bs256 @1/2/4/8 GPU",
        "value": sps, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": global_batch, "parallelism": f"dp{world}", "bn": "sync" if (args.sync_bn and pg is not None) else "local",
                   "launch": "eager" if (args.no_graph or (args.sync_bn and pg is not None and dry)) else "hip_graph",
                   "final_loss": final_loss, "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps},
        "roofline": {"bound": bound_of(kern_mfma_frac, kern_hbm_frac),
                     "kernel": dom_name + (" (+ igemm_nt_kernel: every implicit-GEMM launch of the step is counted)"
                                           if dom is ig else ""),
                     "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "traffic_provenance": traffic_note,
                     "hbm_frac": kern_hbm_frac,
                     "algorithmic_flops_per_launch": dom["flops"] / max(dom["calls"], 1),
                     "operand_bytes_per_launch": dom["bytes"] / max(dom["calls"], 1),
                     "launches_per_step": dom["calls"], "avg_launch_ms": dom["ms"] / max(dom["calls"], 1),
                     "kernel_share_of_step": dom["ms"] / total_ms if total_ms else None,
                     "algorithmic_gflop_per_sample": gflop_per_sample,
                     "step_algorithmic_tflops": sps / world * gflop_per_sample * 1e9 / 1e12,
                     "step_frac_of_peak": step_mfma_frac,
                     "step_hbm_bytes": step_bytes, "step_hbm_frac_of_8TBps": step_hbm_frac,
                     "step_launches": prof["whole_step"]["launches"] if prof else None,
                     "step_bound": bound_of(step_mfma_frac, step_hbm_frac)},
    }
    if args.dtype == "f32x3":
        out["roofline"]["peak_note"] = ("dense bf16 MFMA peak / 6: every fp32 product is six bf16 products of the three-term split; "
                                        "launches the split does not serve run on the fp32 matrix cores (peak %.1f)" % PEAK_FP32_MFMA_TFLOPS)
        out["roofline"]["frac_of_native_fp32_mfma_peak"] = achieved / PEAK_FP32_MFMA_TFLOPS
    if dom is ig and args.dtype in ("f32", "bf16s", "fp16s"):
        # What the ring kernels' operand path was measured to do (round 4, cache counters per launch shape:
        # profiles/r4/cache_by_launch_f32.txt; LAB_NOTES E).  Round 3 called 7.5 TB/s "the Infinity-Cache LDS-fill rate" and
        # priced the kernels against it; the counters say otherwise: the fp32 64x64 launches fill at ~7.3 TB/s with 0.87-0.93
        # of their L1 misses HITTING L2 at 170-250 cycles average latency (~2.6 KB in flight per CU), and cutting the L2 misses
        # of a launch by 18 % (tap order, profiles/r4/ab_taporder*.txt) moves its time by 1 %.  The rate is this kernel
        # structure's LDS-DMA issue rate (one 1-KiB piece per ~58 cycles per CU with six loader waves), NOT a property of the
        # cache level that serves the fills -- the hardware guide's L2-served fill rate is 16.8-18.8 TB/s.  Reported as an
        # observation, not as a roofline.
        fpb = 16.0 if args.dtype == "f32" else 32.0
        out["roofline"]["lds_fill_path"] = {
            "tile": "64x64 one-tile-per-block ring kernel (most implicit-GEMM launches of the step; the six largest run 128x128 persistent tiles at 32 flop per filled byte)", "flop_per_filled_byte": fpb,
            "observed_fill_rate_TBps": 7.3, "l2_hit_rate_of_fills": [0.87, 0.93],
            "avg_l1_to_l2_read_latency_cycles": [170, 250],
            "tflops_at_observed_fill_rate": fpb * 7.3, "achieved_over_that": achieved / (fpb * 7.3),
            "is_a_hardware_ceiling": False,
            "source": "profiles/r4/cache_by_launch_f32.txt (TCC_HIT/MISS, TCP_TCC_READ_REQ(_LATENCY) per launch shape); "
                      "profiles/r4/ab_taporder.txt + ab_taporder_l2_hit_rates.txt; MI355X_MICROARCH.md 'Indexed rows: gather into LDS'"}
    if rehearse:
        out["rehearsal"] = "all ranks on ONE GPU, collectives over gloo (MMDYN_BENCH_REHEARSE_ONE_GPU=1): schedule check, not a scaling number"
    if dry:
        out["dry_run"] = "CPU rehearsal with emulated kernels (MMDYN_BENCH_DRYRUN=emu): control flow only, numbers meaningless"
    if args.breakdown:
        for k, d in sorted(kern.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 and d["flops"] else 0.0
            gb = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
            print(f"{k:24s} calls {d['calls']:4d}  {d['ms']:8.3f} ms  {tf:7.2f} TFLOP/s  {gb:8.1f} GB/s(args)", file=sys.stderr)
        print(f"sum of kernel time {total_ms:.3f} ms vs step {1e3 * elapsed / args.steps:.3f} ms", file=sys.stderr)
        print("per-shape MFMA launches: name, (mode,G,Bg,Hi,Wi,Cin,Ho,Wo,N,ldc,stride,offset,act,splitk) | "
              "(mode,Bt,Hr,Wr,Cd,Hi,Wi,Cg,stride,offset,chunks)", file=sys.stderr)
        for k, d in sorted(profile_step.by_shape.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            print(f"  {k[0]:9s} {str(k[1:]):70s} x{d['calls']:2d} {d['ms']:7.3f} ms {tf:6.1f} TF/s", file=sys.stderr)
    if world == 1 and args.dtype == "f32" and not args.no_alt and not dry and not args.no_graph:
        # The same workload, same seeds, in the engine's "fp32x3" arithmetic (fp32 storage and results; the GEMMs on the bf16 matrix
        # cores through the exact three-term operand split, csrc/igemm_nt.hip X3) -- timed like `value` (W warm-up + K graph-replayed
        # steps between synchronisations), reported NEXT to it: `value` above is the native fp32 matrix-core arithmetic.
        del step
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        model3 = setup_model("cnn-mvae", cross_modal=True, condition_dim=0, input_dim=S * S, architecture="cnn",
                             conditional=False, categorical_conditions=False, latent_size=256, use_pose=True).to(dev).train()
        step3 = MVAEStep(model3, lr=1e-3, pose_multiplier=1000.0, noise=NoiseSource(1234 + rank), two_lanes=not args.single_lane,
                         precision="fp32x3", defer_wgrad={"auto": None, "on": True, "off": False}[args.defer_wgrad],
                         group_heads=not args.no_grouped_heads)
        for _ in range(args.warmup):
            step3.train_step_graphed(inputs, targets, KL_WEIGHT)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss3 = step3.train_step_graphed(inputs, targets, KL_WEIGHT)
        sync()
        dt3 = time.perf_counter() - t0
        out["alt_arithmetic"] = {
            "name": "fp32x3", "value": args.batch * args.steps / dt3, "unit": "samples/s", "ms_per_step": 1e3 * dt3 / args.steps,
            "final_loss": float(loss3), "final_loss_native_fp32": final_loss,
            "what": "fp32 operands split exactly into three bf16 terms (round-to-nearest), six of the nine cross products on "
                    "v_mfma_f32_32x32x16_bf16, fp32 accumulate; dropped terms < 2^-23 |a||b| per product (less than one fp32 rounding); "
                    "error against fp64 no larger than the native fp32 matrix cores' (profiles/r4/ab_x3_*.txt); same oracle "
                    "tolerances (tests/test_model_gpu.py::test_fused_engine_vs_oracle[*-fp32x3]); python bench.py --dtype f32x3 "
                    "makes it the measured arithmetic"}
        del step3, model3
    if world == 1 and not args.no_cpu_baseline:
        # torch.set_num_threads(os.cpu_count()).  On a box whose CPU share is smaller than the
        # machine (the one-GPU box: 16 of 256 hardware threads, no visible quota) the figure at the documented share is
        # reported next to it; `value` / `cores` are the faster of the two, i.e. the strongest CPU baseline measured.
        share, allc = host_cpu_share(), host_cpu_all()
        base = cpu_baseline(args.batch, share, S, 20.0)
        if allc != share:
            full = cpu_baseline(args.batch, allc, S, 12.0)
            best, other = (full, base) if full["value"] > base["value"] else (base, full)
            base = dict(best)
            base["other_thread_count"] = {k: other[k] for k in ("value", "cores", "sample")}
        out["cpu_baseline"] = base
    _emit(out)
    if pg is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
