"""End-to-end parity on a real MI355X through the C ABI: module API (reference schedule), fused engine and
cnn-vae config 1 against the golden vectors produced by the reference; plus a larger-batch comparison of the
fused engine against the CPU oracle (B=32) and size-independent properties at the BASELINE batch (B=256)."""
import numpy as np
import pytest
import torch

from oracle import mvae_oracle as O
from mmdyn_hip.engine import MVAEStep
from mmdyn_hip.models import InjectedNoise, NoiseSource
from mmdyn_hip.models.shapes import state_dict_shapes
from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
import test_model_emu as T

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_mvae_forward_subsets(golden_dir):
    T.check_forward_subsets(golden_dir, DEV)


@pytest.mark.parametrize("fname,use_pose", [("mvae_pose_B4.npz", True), ("mvae_nopose_B4.npz", False)])
def test_reference_schedule_step(golden_dir, fname, use_pose):
    T.check_reference_schedule_step(golden_dir, DEV, fname, use_pose)


@pytest.mark.parametrize("fname,use_pose", [("mvae_pose_B4.npz", True), ("mvae_nopose_B4.npz", False)])
def test_fused_engine_matches_reference(golden_dir, fname, use_pose):
    T.check_fused_engine(golden_dir, DEV, fname, use_pose)


@pytest.mark.parametrize("fname,use_pose", [("mvae_pose_B4.npz", True), ("mvae_nopose_B4.npz", False)])
def test_fused_engine_exact_running_stats(golden_dir, fname, use_pose):
    T.check_fused_engine(golden_dir, DEV, fname, use_pose, exact=True)


@pytest.mark.parametrize("mask_channels", [1, 3])
def test_fused_engine_mask_loss(mask_channels):
    T.check_fused_engine_mask_loss(DEV, mask_channels)


def test_fused_engine_mask_loss_graph_replay():
    """The masked step replayed from HIP graphs (device-side draws): the mask is a static input of the capture, so each
    replay sees the mask of its call; masked / unmasked sums of the visual passes against torch on the step's own logits."""
    import torch.nn.functional as F
    B = 4
    step = MVAEStep(T.build("cnn-mvae", True, False, DEV), noise=NoiseSource(3), keep_logits=True)   # (every pass's logits, not only the published pass's)
    inputs, targets = seeded_batch(B, 77, with_pose=False)
    inputs, targets = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    gen = torch.Generator().manual_seed(9)
    masks = [(torch.rand(B, 1, 64, 64, generator=gen) > 0.5).float().to(DEV), torch.zeros(B, 1, 64, 64, device=DEV),
             torch.ones(B, 1, 64, 64, device=DEV), (torch.rand(B, 1, 64, 64, generator=gen) > 0.2).float().to(DEV)]
    step.train_step_graphed(inputs, targets, 0.3, loss_mask=masks[2])     # eager warm-up step + capture
    for i, mk in enumerate(masks):
        loss = float(step.train_step_graphed(inputs, targets, 0.3, loss_mask=mk))
        assert np.isfinite(loss)
        acc = step.acc.cpu()
        lv, lt = step.last["logits_v"].view(2, B, 3, 64, 64), step.last["logits_t"].view(2, B, 3, 64, 64)
        bce = lambda lg, tg, m: float(F.binary_cross_entropy_with_logits(lg * m, tg * m, reduction="sum"))
        one = torch.ones_like(mk)
        # pass slots: 0 = joint (visual + tactile terms), 1 = visual only, 2 = tactile only
        for slot, terms in ((0, [(lv[0], targets[0]), (lt[0], targets[1])]), (1, [(lv[1], targets[0])]), (2, [(lt[1], targets[1])])):
            assert float(acc[0, slot]) == pytest.approx(sum(bce(a, b, mk) for a, b in terms), rel=1e-5), (i, slot)
            assert float(acc[3, slot]) == pytest.approx(sum(bce(a, b, one) for a, b in terms), rel=1e-5), (i, slot)
    assert step._graph is not None and step._graph[0][-1] == (B, 1, 64, 64)
    # a step without mask re-captures (different static inputs) and fills the plain slots only
    step.train_step_graphed(inputs, targets, 0.3)
    assert float(step.acc[3].abs().sum()) == 0.0


def test_vae_config1(golden_dir):
    T.check_vae_config1(golden_dir, DEV)


def test_conditional_mvae(golden_dir):
    T.check_conditional(golden_dir, DEV)


def test_fused_engine_conditional(golden_dir):
    """--conditional through the fused engine against the reference's vectors; then the same step captured into HIP graphs
    (the condition is a static input of the capture): a replay with another condition changes the loss."""
    T.check_fused_engine_conditional(golden_dir, DEV)
    m = T.setup_model("cnn-mvae", cross_modal=True, **dict(T.MODEL_KW, conditional=True, condition_dim=3, use_pose=True))
    m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
    m.to(DEV).train()
    step = MVAEStep(m, noise=NoiseSource(2), lr=0.0)
    inputs, targets = seeded_batch(4, 321)
    inputs, targets = [t.to(DEV) for t in inputs], [t.to(DEV) for t in targets]
    c0, c1 = torch.zeros(4, 3, device=DEV), torch.full((4, 3), 2.0, device=DEV)
    step.train_step_graphed(inputs, targets, 1.0, condition=c0)
    la = float(step.train_step_graphed(inputs, targets, 1.0, condition=c0))
    lb = float(step.train_step_graphed(inputs, targets, 1.0, condition=c1))
    assert step._graph is not None and np.isfinite(la) and np.isfinite(lb) and abs(la - lb) > 1e-3 * abs(la)


def test_dyn_modeling_and_cli(tmp_path):
    T.check_dyn_modeling_and_cli(tmp_path, no_cuda=False)


def test_training_loop_uses_graph_replay_and_learns(tmp_path):
    """Problem.train() on the GPU replays the fused step from HIP graphs (one capture per epoch: the annealed KL
    weight is baked in); the loss on a fixed synthetic stream falls over the epochs."""
    from mmdyn_hip.problems.problems import SeqModeling, SyntheticVisuoTactile
    prob = SeqModeling(T.args(num_epochs=3, batchsize=8, lr=1e-3), log_dir=str(tmp_path),
                       train_loader=SyntheticVisuoTactile(6, 8), test_loader=SyntheticVisuoTactile(2, 8, seed=7))
    prob.train()
    assert prob._step._graph is not None
    tr = prob._logger_dict['Loss/train_epoch']
    assert len(tr) == 3 and tr[-1] < tr[0]


def test_eval_mode(golden_dir):
    T.check_eval_mode(golden_dir, DEV)


def test_inference_engine(golden_dir):
    T.check_inference_engine(golden_dir, DEV)


def test_mlp_vae(golden_dir):
    T.check_mlp_vae(golden_dir, DEV)


def test_regressor(golden_dir, tmp_path):
    T.check_regressor(golden_dir, DEV, tmp_path)


def test_conditional_training_loops(tmp_path):
    T.check_conditional_loops(tmp_path, no_cuda=False)


class _ReluKnifeEdges:
    """A ReLU input within fp32 rounding of zero has no defined gradient mask: the native and the split arithmetic (and the
    oracle) may land on either side, and ONE flipped unit of the pose decoder moves the upstream gradients by 2e-3 (the pose term
    is weighted 1000x; profiles/r4/x3_relu_knife_edge_b130.txt: B = 130, {tactile, pose} subset, sample 37, unit 507 of
    deconv_net.2: +3.5e-7 natively, <= 0 in the split).  Either subgradient is right.  While active, the oracle's pose decoder takes
    the ENGINE's mask for the units -- and only those -- whose oracle pre-activation is below 1e-5 of the layer's rms, and counts how
    many it had to flip (`flips`), which the test bounds: a real error in the pose decoder flips thousands, a knife edge one or two.
    Used ONLY as the second opinion of an fp32x3 case whose gradients upstream of the pose decoder's ReLUs missed the bound against
    the untouched oracle (test_fused_engine_vs_oracle); the native fp32 cases never see it."""
    THR = 1e-5

    def __init__(self, dp, live, B):
        self.dp, self.live, self.B, self.flips, self.calls = dp, list(live), B, 0, 0

    def __enter__(self):
        self._orig = O.pose_decoder
        O.pose_decoder = self._pose_decoder
        return self

    def __exit__(self, *exc):
        O.pose_decoder = self._orig

    def _relu(self, u, eng_h):
        mask = u > 0
        if eng_h is not None:
            knife = u.detach().abs() < self.THR * float(u.detach().pow(2).mean().sqrt())
            eng = eng_h > 0
            self.flips += int((knife & (eng != mask)).sum())
            mask = torch.where(knife, eng, mask)
        return u * mask

    def _pose_decoder(self, z, prm, pre="pose_decoder"):
        k = self.calls % 7                      # _evaluate_mvae's pass index (problems.py:473-546: one pose decoding per pass)
        self.calls += 1
        h1 = h2 = None
        if k in self.live:
            g = self.live.index(k)
            h1, h2 = self.dp["h1"][g * self.B:(g + 1) * self.B], self.dp["h2"][g * self.B:(g + 1) * self.B]
        h = self._relu(torch.nn.functional.linear(z, prm[pre + ".deconv_net.0.weight"], prm[pre + ".deconv_net.0.bias"]), h1)
        h = self._relu(torch.nn.functional.linear(h, prm[pre + ".deconv_net.2.weight"], prm[pre + ".deconv_net.2.bias"]), h2)
        return torch.nn.functional.linear(h, prm[pre + ".deconv_net.4.weight"], prm[pre + ".deconv_net.4.bias"])


# parameters whose gradient does not pass through a ReLU of the pose decoder on its way back: the image decoders and the pose
# decoder's output layer (its gradient needs the forward activations only)
_NOT_BEHIND_POSE_RELU = ("visual_decoder.", "tactile_decoder.", "pose_decoder.deconv_net.4.")


@pytest.mark.parametrize("B,n_steps,precision", [(32, 3, "fp32"), (256, 1, "fp32"), (1, 1, "fp32"), (5, 2, "fp32"), (37, 1, "fp32"),
                                                 (130, 1, "fp32"), (32, 3, "fp32x3"), (256, 1, "fp32x3"), (130, 1, "fp32x3"),
                                                 (131, 1, "fp32x3"), (200, 1, "fp32x3")])
def test_fused_engine_vs_oracle(B, n_steps, precision):
    """(precision "fp32x3", the product's default: the convolution-level GEMMs run on the bf16 matrix cores through the exact
    three-term split of their fp32 operands -- held to the SAME tolerances as the native fp32 arithmetic, at the same batch sizes
    incl. the ragged B = 130.)
    ELBO (total and each of the 7 partials) within 1e-4 relative of the CPU oracle, gradients within 1e-3
    relative L2 per tensor, loss still within 1e-4 after 3 Adam steps; B=256 is the
    BASELINE batch (one oracle step takes a few seconds on the host cores); 1, 5, 37 and 130 are ragged sizes: no
    row count is a multiple of any tile, split-K / chunk / band sizes all hit their remainders.
    The oracle runs UNTOUCHED and every case is compared with that run.  Only if an fp32x3 case then misses the gradient bound, and
    only on tensors behind a ReLU of the pose decoder (every other tensor must have met it against the untouched oracle), the
    oracle is run a second time with the engine's mask on that decoder's knife-edge units (_ReluKnifeEdges: between one and three
    units below 1e-5 of the layer rms) and the bound must hold against that run."""
    klw = 1.0 / 50
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    prm, buf = O.split_state(sd)
    inputs, targets = seeded_batch(B, 1234)
    eps, masks = seeded_noise(B, 256, 7 * n_steps, 8 * n_steps, 4321)
    m = T.build("cnn-mvae", True, True, DEV)
    step = MVAEStep(m, noise=InjectedNoise(eps, masks), precision=precision)
    names = list(prm.keys())
    opt = O.Adam([prm[k] for k in names], lr=1e-3)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]

    def rel_errors():
        named = dict(m.named_parameters())
        return {k: float((named[k].grad.double().cpu() - prm[k].grad.double()).norm() / (prm[k].grad.double().norm() + 1e-30))
                for k in names}

    for s in range(n_steps):
        opt.zero_grad()
        loss = step.forward(gi, gt, klw)
        buf0 = {k: v.clone() for k, v in buf.items()}
        _, loss_o, partials_o = O.evaluate_mvae(prm, inputs, targets, eps[7 * s:7 * s + 7], masks[8 * s:8 * s + 8], klw,
                                                1000.0, True, buf)
        loss_o.backward()
        assert float(loss) == pytest.approx(float(loss_o.detach()), rel=1e-4), s
        # every partial within 1e-4 on identical weights (step 0).  After Adam steps the weights themselves differ by
        # rounding-level gradient differences that Adam amplifies where |g| ~ 0 (update = lr * g / (|g| + eps)), so the
        # small pose-only partial drifts at the 1e-4 level whichever summation order the kernels use
        # (tests/microbench/drift_probe.py: 5e-5 .. 1.2e-4 at step 2); the total stays within 1e-4
        np.testing.assert_allclose(step.partials[:7].cpu().numpy(), [float(x.detach()) for x in partials_o],
                                   rtol=1e-4 if s == 0 else 5e-4)
        dp = {k: step.ctx["dp"][k].detach().cpu().clone() for k in ("h1", "h2")}
        h = step.backward()
        if s == 0:
            bad = {k: e for k, e in rel_errors().items() if not e < 1e-3}
            if bad:
                # the native arithmetic is held to the untouched oracle, full stop
                assert precision == "fp32x3", bad
                assert not any(k.startswith(_NOT_BEHIND_POSE_RELU) for k in bad), bad
                opt.zero_grad()
                buf.update(buf0)                   # (the second run starts from the same running statistics)
                with _ReluKnifeEdges(dp, step.pass_p, B) as knife:
                    _, loss_k, _ = O.evaluate_mvae(prm, inputs, targets, eps[7 * s:7 * s + 7], masks[8 * s:8 * s + 8], klw,
                                                   1000.0, True, buf)
                assert 1 <= knife.flips <= 3, knife.flips          # (B = 130, fp32x3, r4 kernels: 1)
                loss_k.backward()
                assert float(loss_k.detach()) == pytest.approx(float(loss_o.detach()), rel=1e-6)
                for k, e in rel_errors().items():
                    assert e < 1e-3, (k, e)
        step.optimizer_step(h)
        opt.step()


@pytest.mark.parametrize("variant", ["no_pose", "conditional", "mask_loss"])
def test_fp32x3_plane_operands_in_the_engine_variants(variant):
    """The engine variants the oracle test above does not build -- no pose term, --conditional, --mask-loss -- at a batch where the
    plane kernels serve the convolution-level launches (B = 96): the fp32x3 step against the native fp32 step on the same weights,
    inputs and injected noise.  Both are fp32-grade arithmetics of the same computation: loss within 2e-6 relative, every gradient
    within 2e-4 relative L2 (rounding-level differences through BatchNorm's 1/sigma and the knife edges of ReLU / dropout masks)."""
    B, klw = 96, 0.05
    use_pose = variant == "conditional"            # (--mask-loss is defined for models without the pose term: problems.py:445-447)
    results = []
    for precision in ("fp32", "fp32x3"):
        kw = dict(T.MODEL_KW, use_pose=use_pose)
        if variant == "conditional":
            kw.update(conditional=True, condition_dim=3)
        m = T.setup_model("cnn-mvae", cross_modal=True, **kw)
        m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
        m.to(DEV).train()
        eps, masks = seeded_noise(B, 256, 7, 8, 4321)
        step = MVAEStep(m, noise=InjectedNoise(eps, masks), precision=precision)
        inputs, targets = seeded_batch(B, 99, with_pose=use_pose)
        gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
        extra = {}
        if variant == "conditional":
            extra["condition"] = torch.linspace(-1, 1, B * 3).view(B, 3).to(DEV)
        if variant == "mask_loss":
            g = torch.Generator().manual_seed(5)
            extra["loss_mask"] = (torch.rand(B, 1, 64, 64, generator=g) > 0.3).float().to(DEV)
        loss = float(step.forward(gi, gt, klw, **extra))
        step.backward()
        results.append((loss, {k: v.grad.double().cpu().clone() for k, v in m.named_parameters()}))
    (l0, g0), (l1, g1) = results
    assert l1 == pytest.approx(l0, rel=2e-6)
    for k in g0:
        assert float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)) < 2e-4, k
    from mmdyn_hip import ops
    assert ops.B.lib.mmdyn_igemm_planes_served(ops.CONV, 2, B, 16, 16, 64, 8, 8, 128) == 1     # (the decoders' launches took plane operands)


def test_fc_level_launches_on_the_plane_ring_match_the_default_step(monkeypatch):
    """layers.FC_PLANES (round 6, off by default: measured 0.6 % slower on the two-lane step): the decoders' Linear forward -- z and its
    activated output as planes, written by the product-of-experts launch and the GEMM epilogue -- and the encoder FC layer's input
    gradient on the DENSE mode of the plane-ring kernel, B = 256, against the default fp32x3 step on the same weights, inputs and
    injected noise: loss within 2e-6 relative, every gradient within 2e-4 relative L2; eager and replayed from HIP graphs."""
    from mmdyn_hip import layers
    B, klw = 256, 0.05
    results = []
    for fc in (False, True):
        monkeypatch.setattr(layers, "FC_PLANES", fc)
        m = T.build("cnn-mvae", True, True, DEV)
        eps, masks = seeded_noise(B, 256, 7, 8, 4321)
        step = MVAEStep(m, noise=InjectedNoise(eps, masks))
        inputs, targets = seeded_batch(B, 99)
        gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
        loss = float(step.forward(gi, gt, klw))
        if fc:
            assert step.ctx["zzplv"] is not None and step.ctx["zzplt"] is not None       # (z arrived split at both image decoders)
            from mmdyn_hip import ops
            assert isinstance(step.ctx["dv"]["h0"], ops.Planes)
        step.backward()
        results.append((loss, {k: v.grad.double().cpu().clone() for k, v in m.named_parameters()}))
        step.close()
    (l0, g0), (l1, g1) = results
    assert l1 == pytest.approx(l0, rel=2e-6)
    for k in g0:
        assert float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)) < 2e-4, k
    # the same path captured into HIP graphs and replayed
    monkeypatch.setattr(layers, "FC_PLANES", True)
    step = MVAEStep(T.build("cnn-mvae", True, True, DEV), noise=NoiseSource(5))
    inputs, targets = seeded_batch(B, 99)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    losses = [float(step.train_step_graphed(gi, gt, klw)) for _ in range(4)]
    assert step._graph is not None and all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]
    step.close()


@pytest.mark.parametrize("precision", ["fp32x3", "bf16s"])
def test_single_launch_finalize_of_small_tables_leaves_the_step_bit_identical(precision):
    """ops.B.ticket_max_work (round 6: BatchNorm finalize launches with at most 1024 partial-sum rows take the single-launch
    "last block finishes" form): three replayed steps at B = 64 with the rule on (the default), off (rounds 3-5) and on for EVERY
    finalize give the same loss step by step and the same weights, bit for bit, afterwards."""
    from mmdyn_hip import ops
    B, klw = 64, 0.05
    prev = ops.B.ticket_max_work
    assert prev == 1024
    results = []
    try:
        for w in (1024, 0, 1 << 30):
            ops.B.ticket_max_work = w
            m = T.build("cnn-mvae", True, True, DEV)
            step = MVAEStep(m, noise=NoiseSource(11), precision=precision)
            inputs, targets = seeded_batch(B, 5)
            gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
            losses = [float(step.train_step_graphed(gi, gt, klw)) for _ in range(3)]
            assert step._graph is not None
            torch.cuda.synchronize()
            results.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
            step.close()
    finally:
        ops.B.ticket_max_work = prev
    (l0, s0) = results[0]
    for l1, s1 in results[1:]:
        assert l1 == pytest.approx(l0, rel=1e-7)        # (the loss sums are fp64 atomics: their order is free)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k


def test_bf16_engine_vs_oracle():
    """BASELINE configs[2] arithmetic (bf16 matrix-core operands, fp32 accumulate / storage / master weights) against
    the fp32 CPU oracle, B=32, injected noise.  Stated tolerance for this mode: ELBO and each partial within 5e-3
    relative (measured 8e-5 on the total), gradients within 1e-1 relative L2 per tensor (measured: median 1.9e-2,
    worst 7.5e-2 on the encoder conv weights, whose gradients come through the longest bf16 chains)."""
    B, klw = 32, 1.0 / 50
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    prm, buf = O.split_state(sd)
    inputs, targets = seeded_batch(B, 1234)
    eps, masks = seeded_noise(B, 256, 7, 8, 4321)
    m = T.build("cnn-mvae", True, True, DEV)
    step = MVAEStep(m, noise=InjectedNoise(eps, masks), precision="bf16")
    _, loss_o, partials_o = O.evaluate_mvae(prm, inputs, targets, eps, masks, klw, 1000.0, True, buf)
    loss_o.backward()
    loss = step.forward([x.to(DEV) for x in inputs], [x.to(DEV) for x in targets], klw)
    rel_loss = abs(float(loss) - float(loss_o.detach())) / abs(float(loss_o.detach()))
    assert rel_loss < 5e-4, rel_loss          # measured 8e-5
    np.testing.assert_allclose(step.partials[:7].cpu().numpy(), [float(x.detach()) for x in partials_o], rtol=5e-3)
    step.backward()
    named = dict(m.named_parameters())
    errs = sorted(((float((named[k].grad.double().cpu() - prm[k].grad.double()).norm()
                           / (prm[k].grad.double().norm() + 1e-30)), k) for k in prm), reverse=True)
    print("bf16 loss rel err", rel_loss, "worst gradient rel-L2:", errs[:6], "median", errs[len(errs) // 2])
    worst = errs[0][0]
    assert worst < 1e-1, errs[:6]
    assert rel_loss > 1e-7          # and it really is a different arithmetic from the fp32 path
    from mmdyn_hip import ops
    assert ops.B.precision == "fp32"          # the engine restores the default after every call


@pytest.fixture(scope="module")
def nccl_group():
    """ONE one-rank RCCL group for the whole module (creating a second group after destroying the first in the same
    process aborts inside the runtime on this stack)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    import gc
    gc.collect()                       # engines (and their captured graphs) of the tests above are gone before the group
    torch.cuda.synchronize()
    dist.destroy_process_group()


def test_sync_bn_single_rank_equals_local(tmp_path, nccl_group):
    """sync_bn with a one-rank RCCL group: the all-reduces are identities, so loss and gradients must equal the local
    BatchNorm step bit for bit up to the fp64 -> fp32 rounding of the statistics (exercises the RCCL + kernel path)."""
    B, klw = 8, 0.02
    inputs, targets = seeded_batch(B, 77)
    eps, masks = seeded_noise(B, 256, 7, 8, 78)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    out = []
    for sync in (False, True):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=InjectedNoise(eps, masks), process_group=nccl_group, world_size=1, sync_bn=sync)
        loss = float(step.forward(gi, gt, klw))
        for h in step.backward():
            h.wait()
        out.append((loss, step.params.grad.clone()))
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-6)
    assert float((out[0][1] - out[1][1]).norm() / out[0][1].norm()) < 1e-5


def test_graphed_step_with_process_group_equals_single(tmp_path, nccl_group):
    """The data-parallel form of the graph-replayed step (encoder backward cut after the FC layer, three gradient
    buckets all-reduced between graph launches, Adam outside the graphs) with a one-rank RCCL group -- the all-reduces
    are identities -- must train exactly like the single-process graphs: same losses over four steps, same weights."""
    B, klw, steps = 8, 0.02, 4
    inputs, targets = seeded_batch(B, 91)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    out = []
    for pg in (None, nccl_group):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=NoiseSource(92), process_group=pg, world_size=1)    # device-side Philox: capturable
        losses = [float(step.train_step_graphed(gi, gt, klw)) for _ in range(steps)]
        torch.cuda.synchronize()
        out.append((losses, step.params.flat.clone()))
        # the bucket bounds cover the flat buffer in the order the gradients become ready
        assert step.params.bucket_bounds[-1] == step.params.total and len(step.params.bucket_bounds) == 3
        step.close()
        del step
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-6)
    assert float((out[0][1] - out[1][1]).norm() / out[0][1].norm()) < 1e-6


def test_fp16_graph_replay_with_process_group_keeps_its_loss_scale(nccl_group):
    """in the fp16 modes the loss scale is 4 * B of the LATEST _begin(); the data-parallel graph replay
    runs Adam eagerly, so an eval_step on another batch size between two replays must not change the scale Adam divides
    by.  Train (graphed, B = 8), evaluate B = 4, train again == train, train."""
    B, klw = 8, 0.02
    inputs, targets = seeded_batch(B, 95)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    out = []
    for with_eval in (False, True):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=NoiseSource(96), process_group=nccl_group, world_size=1, precision="fp16")
        step.train_step_graphed(gi, gt, klw)               # warm-up + capture
        step.train_step_graphed(gi, gt, klw)               # first replay
        if with_eval:
            step.eval_step([x[:4] for x in gi], [x[:4] for x in gt], klw)
            assert step.loss_scale == 4.0 * 4 and step._graph[2] == 4.0 * B
        loss = float(step.train_step_graphed(gi, gt, klw))
        torch.cuda.synchronize()
        out.append((loss, step.params.flat.clone(), step.adam_m.clone()))
        assert step.skipped_steps == 0
        step.close()
        del step
    # (the evaluation draws from the device-side noise stream, so the step after it sees other noise: compare the moments'
    #  magnitude, which a gradient scaled by 2 would double, and the parameters, which must have moved equally far)
    assert float(out[1][2].norm() / out[0][2].norm()) == pytest.approx(1.0, rel=0.15)     # (stale scale: 1.37)
    assert out[0][0] == pytest.approx(out[1][0], rel=2e-2)


def test_sync_bn_inside_the_phase_graphs(tmp_path, nccl_group):
    """SyncBN statistics all-reduces captured INTO the lanes' HIP graphs (each lane on its own RCCL communicator, so the
    two concurrently replayed graphs never interleave collectives of one communicator): graph replay == eager launches,
    step for step, with a one-rank group (the only multi-process layout a one-GPU box offers; the two-rank arithmetic is
    covered on gloo by tests/test_ddp_gloo.py)."""
    B, klw, steps = 8, 0.02, 4
    inputs, targets = seeded_batch(B, 93)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    out = []
    for graphed in (False, True):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=NoiseSource(94), process_group=nccl_group, world_size=1, sync_bn=True)
        assert step._sync_graph_ok and len(step._sync.lane_groups) == 2
        run = step.train_step_graphed if graphed else step.train_step
        losses = [float(run(gi, gt, klw)) for _ in range(steps)]
        torch.cuda.synchronize()
        if graphed:
            assert step._graph is not None          # really replayed from graphs (no silent eager fallback)
        out.append((losses, step.params.flat.clone()))
        step.close()                                # graphs with captured collectives go before their communicators
        del step
    assert out[0][0] == pytest.approx(out[1][0], rel=1e-6)
    assert float((out[0][1] - out[1][1]).norm() / out[0][1].norm()) < 1e-6


@pytest.mark.parametrize("B", [32, 5, 37])
def test_bf16_storage_engine_vs_oracle(B):
    """precision="bf16s" (BASELINE configs[2]: bf16 activation storage + bf16 matrix cores, fp32 accumulate / master
    weights) against the fp32 CPU oracle.  Stated tolerance: total ELBO within 5e-4 relative (measured 0.8-1.1e-4), partials
    within 1e-2, gradients within 1.5e-1 relative L2 per tensor = twice the worst measured (7.7e-2, the encoder conv weights,
    whose gradients come through the longest bf16 chains; median 1.9e-2; values are printed).  B = 5 and 37 are ragged: no row count is a multiple
    of a tile, so the all-bf16 GEMM kernels (64-channel K-steps, transposing LDS reads, shared quad walks) see their
    masked remainders."""
    klw = 1.0 / 50
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    prm, buf = O.split_state(sd)
    inputs, targets = seeded_batch(B, 1234)
    eps, masks = seeded_noise(B, 256, 7 * 4, 8 * 4, 4321)
    m = T.build("cnn-mvae", True, True, DEV)
    step = MVAEStep(m, noise=InjectedNoise(eps, masks), precision="bf16s")
    _, loss_o, partials_o = O.evaluate_mvae(prm, inputs, targets, eps[:7], masks[:8], klw, 1000.0, True, buf)
    loss_o.backward()
    loss = step.forward([x.to(DEV) for x in inputs], [x.to(DEV) for x in targets], klw)
    assert step.ctx["dv"]["stages"][2]["y"].dtype == torch.bfloat16
    loss0 = float(loss)                     # (the engine's loss tensor is overwritten by later steps)
    rel_loss = abs(loss0 - float(loss_o.detach())) / abs(float(loss_o.detach()))
    np.testing.assert_allclose(step.partials[:7].cpu().numpy(), [float(x.detach()) for x in partials_o], rtol=1e-2)
    step.backward()
    named = dict(m.named_parameters())
    errs = sorted(((float((named[k].grad.double().cpu() - prm[k].grad.double()).norm()
                           / (prm[k].grad.double().norm() + 1e-30)), k) for k in prm), reverse=True)
    print("bf16s loss rel err", rel_loss, "worst gradient rel-L2:", errs[:4], "median", errs[len(errs) // 2])
    assert rel_loss < 5e-4 and errs[0][0] < 1.5e-1, (rel_loss, errs[:4])   # measured: 1.1e-4 / 7.7e-2 worst over B = 5, 32, 37
    for s in range(3):                      # and it trains: a few Adam steps on the fixed batch lower the loss
        l = step.train_step([x.to(DEV) for x in inputs], [x.to(DEV) for x in targets], klw)
    assert float(l) < loss0


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
def test_full_size_properties_b256(precision):
    """BASELINE batch (256): properties that need no CPU run of the same size --
    (i) the total equals the sum of the 7 partial ELBOs; (ii) replaying the same step from the same state and
    noise is bit-reproducible (no atomics on the data path except the fp64 loss sums); (iii) gradients are
    finite and the loss decreases over a few Adam steps on a fixed batch; (iv) a batch made of the B=32 oracle
    batch repeated 8x gives the same per-sample ELBO for the pose-only pass (no BatchNorm on that path)."""
    B, klw = 256, 1.0 / 50
    inputs, targets = seeded_batch(B, 99)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    losses = []
    for rep in range(2):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=NoiseSource(7), precision=precision)
        run = []
        for s in range(4):
            loss = step.train_step(gi, gt, klw)
            assert abs(float(step.partials[:7].sum()) - float(loss)) <= 1e-4 * abs(float(loss))
            assert torch.isfinite(step.params.grad).all()
            run.append(float(loss))
        losses.append(run)
    assert losses[0] == losses[1]
    assert losses[0][-1] < losses[0][0]


@pytest.mark.parametrize("B,precision", [(16, "fp32"), (256, "fp32x3"), (128, "fp32x3")])
def test_graph_replay_equals_eager_steps(B, precision):
    """The HIP-graph replay of the fused step is the same computation as the eager launches: identical losses over
    several optimiser steps from the same state and the same Philox stream, and fresh noise on every replay.
    (fp32x3 at the BASELINE batch: the plane kernels, the split kernels -- persistent, register-staged and weight-gradient -- inside
    captured graphs on three streams, the deferred weight-gradient queues behind the main stream's work; at bs 128 the encoder's
    launches fall under the plane kernels' work thresholds and the two operand formats mix.)"""
    klw = 0.02
    inputs, targets = seeded_batch(B, 5)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    runs = []
    for graphed in (False, True):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=NoiseSource(11), precision=precision)
        losses = []
        for s in range(5):
            loss = step.train_step_graphed(gi, gt, klw) if graphed else step.train_step(gi, gt, klw)
            losses.append(float(loss))
        runs.append(losses)
    assert runs[0] == pytest.approx(runs[1], rel=1e-6)
    assert len(set(runs[1])) == 5


# ---- BASELINE configs[3] / configs[4]: the 128 / 256 pixel extensions and the fp16 matrix-core mode ------------------
@pytest.mark.parametrize("size,B,use_pose", [(128, 8, True), (128, 5, False), (256, 3, True)])
def test_extended_image_sizes_vs_oracle(size, B, use_pose):
    """No reference architecture exists for these sizes (its FC is fixed at 256*5*5); the oracle restates the stack
    defined in models/shapes.py.  fp32: ELBO and partials 1e-4, gradients 1e-3, as for the reference's own size."""
    T.check_extended_size_vs_oracle(DEV, size, B, use_pose, n_steps=2)


@pytest.mark.parametrize("size,B", [(128, 48), (256, 16)])
def test_fp32x3_extended_sizes_vs_oracle(size, B):
    """The 128 / 256 pixel stacks in the "fp32x3" arithmetic (their 128 x 128 and 256 x 256 layers are where the split's launch rule
    bites at these batch sizes), at the fp32 mode's tolerances."""
    T.check_extended_size_vs_oracle(DEV, size, B, True, n_steps=1, precision="fp32x3")


@pytest.mark.parametrize("size,B", [(64, 32), (256, 4), (64, 256)])
def test_fp16_engine_vs_oracle(size, B):
    """fp16 matrix-core operands (v_mfma_f32_32x32x16_f16 / 32x32x8_f16), fp32 accumulate, storage and master weights
    -- BASELINE configs[4]'s arithmetic -- against the fp32 CPU oracle.  Stated tolerance: ELBO and partials 2e-3
    relative, gradients 5e-2 relative L2 per tensor at B = 32 (measured 2.4e-2; fp16 has 3 more mantissa bits than bf16,
    whose bound is 1.5e-1), 1e-1 at B = 4 on 256x256 (measured 4.1e-2).  B = 256 (the BASELINE batch): the unscaled loss
    gradients (sigmoid - t) / 256 and kl_w / 256 sit at the edge of fp16's normal range; the engine's loss scale 4 * B
    (MVAEStep.loss_scale) keeps them in it -- same bound as at B = 32."""
    worst = T.check_extended_size_vs_oracle(DEV, size, B, True, n_steps=1, precision="fp16", loss_tol=2e-3,
                                            grad_tol=5e-2 if B >= 32 else 1e-1)
    print(f"fp16 size {size} B {B}: worst gradient rel-L2 {worst:.2e}")


@pytest.mark.parametrize("size,B", [(64, 32), (64, 37), (128, 8), (256, 4), (256, 96)])
def test_fp16_storage_engine_vs_oracle(size, B):
    """precision="fp16s": fp16 matrix cores AND fp16 storage of the convolution-level activations, their gradients and the
    packed weights (the bytes of "bf16s" with three more mantissa bits; loss scale 4 * B as in "fp16") against the fp32 CPU
    oracle.  Stated tolerance: ELBO and partials 2e-3 relative, gradients 1e-1 relative L2 per tensor (between "fp16", whose
    storage is fp32, and "bf16s": 1.5e-1).  B = 37: ragged against every tile.  (256, 96): the 256-pixel stack at a
    batch whose oracle step (7 passes with autograd on 16 host threads: ~0.18 s per sample, measured 8.5 s at B = 48) and oracle
    memory (~160 MB of saved activations per sample) stay well inside a minute and the box's RAM."""
    worst = T.check_extended_size_vs_oracle(DEV, size, B, True, n_steps=1, precision="fp16s", loss_tol=2e-3, grad_tol=1e-1)
    print(f"fp16s size {size} B {B}: worst gradient rel-L2 {worst:.2e}")


@pytest.mark.parametrize("size,B,precision", [(128, 128, "fp32"), (256, 32, "fp16"), (64, 128, "bf16s"), (256, 256, "fp16"),
                                              (64, 128, "fp16s"), (256, 256, "fp16s")])
def test_extended_sizes_full_batch_properties(size, B, precision):
    """Per-GPU shares of BASELINE configs[3] (bs 512 / 4 GPUs at 128x128), configs[2] (bs 1024 / 8 GPUs, bf16 storage) and
    configs[4] (256x256, fp16: bs 2048 / 8 GPUs = 256 per GPU, and a 32-sample slice):
    size-independent properties -- the total equals the sum of the partials, gradients finite, bit-reproducible from the
    same state and noise, loss decreasing over Adam steps, HIP-graph replay equal to eager launches."""
    klw = 1.0 / 50
    inputs, targets = seeded_batch(B, 99, size=size)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    runs = []
    for graphed in (False, False, True):
        m = T.build("cnn-mvae", True, True, DEV, size=size)
        step = MVAEStep(m, noise=NoiseSource(7), precision=precision)
        run = []
        # (the fp16 modes at 256x256: round 2's loss scale overflowed in the fourth step with bench.py's batch -- more steps,
        #  and the overflow guard must not have fired)
        for s in range(10 if (size == 256 and precision.startswith("fp16")) else 4):
            loss = step.train_step_graphed(gi, gt, klw) if graphed else step.train_step(gi, gt, klw)
            assert abs(float(step.partials[:7].sum()) - float(loss)) <= 1e-4 * abs(float(loss))
            assert torch.isfinite(step.params.grad).all()
            run.append(float(loss))
        assert step.skipped_steps == 0
        runs.append(run)
    assert runs[0] == runs[1]
    assert runs[2] == pytest.approx(runs[0], rel=1e-6)
    assert runs[0][-1] < runs[0][0]


def test_dyn_modeling_128(tmp_path):
    """configs[3]: dyn_modeling (one-step predictor, problems.py:765-803) on 128x128 frames through the Problem layer."""
    from mmdyn_hip.problems.problems import DynModeling, SyntheticVisuoTactile
    prob = DynModeling(T.args(problem_type="dyn_modeling", num_epochs=2, batchsize=4, image_size=128),
                       log_dir=str(tmp_path / "dyn128"), seq_length=3,
                       train_loader=SyntheticVisuoTactile(3, 4, seq_length=3, size=128),
                       test_loader=SyntheticVisuoTactile(1, 4, 3, seed=7, size=128))
    assert prob._step is not None
    prob.train()
    assert prob._step.last["recon_x"][0].shape == (12, 3, 128, 128)
    tr = prob._logger_dict["Loss/train_epoch"]
    assert all(np.isfinite(tr)) and len(tr) == 2


def test_one_graph_capture_serves_the_kl_annealing_schedule():
    """The annealed KL weight (problems.py:212-216) is read from device memory by the loss assembly and the latent
    backward: replays with a new weight need no re-capture and equal the eager steps."""
    B = 8
    inputs, targets = seeded_batch(B, 5)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    kls = [0.02, 0.02, 0.04, 0.5, 1.0]
    runs, graphs = [], None
    for graphed in (False, True):
        m = T.build("cnn-mvae", True, True, DEV)
        step = MVAEStep(m, noise=NoiseSource(11))
        losses = []
        for i, kl in enumerate(kls):
            loss = step.train_step_graphed(gi, gt, kl) if graphed else step.train_step(gi, gt, kl)
            losses.append(float(loss))
            if graphed and i == 1:
                graphs = step._graph[1]
        if graphed:
            assert step._graph[1] is graphs                  # captured once, at the first replayed step
        runs.append(losses)
    assert runs[0] == pytest.approx(runs[1], rel=1e-6)


def test_inference_engine_extended_size():
    """engine.MVAEInference (prepacked weights, HIP-graph replay, eval-mode BatchNorm) on the 128-pixel stack against the
    oracle in eval mode on the same running statistics."""
    from mmdyn_hip.engine import MVAEInference
    from mmdyn_hip.utils.seeded_init import seeded_running_stats
    size, B = 128, 3
    sd = seeded_running_stats(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True, size=size), 0), 7)
    m = T.build("cnn-mvae", True, True, DEV, size=size)
    m.load_state_dict(sd)
    m.eval()
    inputs, _ = seeded_batch(B, 21, size=size)
    eng = MVAEInference(m, seed=3)
    for _ in range(2):                                   # second call: graph replay
        v, t, p, mu, lv = eng([inputs[0].to(DEV), inputs[1].to(DEV)], pose=inputs[2].to(DEV))
    prm, buf = O.split_state(sd, requires_grad=False)
    with O.eval_mode():
        mo, lo = O.image_encoder(inputs[0], prm, "visual_encoder", None, buf)
    assert tuple(v.shape) == (B, 3, size, size) and tuple(t.shape) == (B, 3, size, size) and tuple(p.shape) == (B, 7)
    # means of the joint posterior need all three experts; the visual expert alone is checked through a visual-only call
    v1, t1, p1, mu1, lv1 = eng([inputs[0].to(DEV), None], pose=None)
    pm, plv = O.product_of_experts(torch.stack([torch.zeros_like(mo), mo]), torch.stack([torch.zeros_like(lo), lo]))
    torch.testing.assert_close(mu1.cpu(), pm, rtol=1e-4, atol=3e-5)
    torch.testing.assert_close(lv1.cpu(), plv, rtol=1e-4, atol=3e-5)
    s = eng.inference(4)
    assert tuple(s[0].shape) == (4, 3, size, size)


def test_inference_engine_on_plane_operands_at_the_serving_batch():
    """engine.MVAEInference at B = 256 in the default arithmetic (round 6: its convolution-level launches take plane operands -- the
    weights' plane twins written once by refresh(), the activations written split by the eval-mode BatchNorm pass, whose mean / rstd
    are precomputed): against the CPU oracle in eval mode (posterior means / log-variances of the visual expert alone and a
    deterministic decode of z = means), and against the native-fp32 engine; then new weights + refresh(): the captured graphs stay
    valid and give what a fresh engine gives."""
    from mmdyn_hip.engine import MVAEInference
    from mmdyn_hip.utils.seeded_init import seeded_running_stats
    from mmdyn_hip import layers
    B = 256
    shapes = state_dict_shapes("cnn-mvae", use_pose=True)
    sd = seeded_running_stats(seeded_state_dict(shapes, 0), 7)
    m = T.build("cnn-mvae", True, True, DEV)
    m.load_state_dict(sd)
    m.eval()
    inputs, _ = seeded_batch(B, 33)
    gi = [x.to(DEV) for x in inputs]
    eng = MVAEInference(m, seed=3)
    assert eng.precision == "fp32x3" and len(eng._plan.twins) > 0                          # conv weights split once, by the pack plan
    assert all(k.endswith(".eval_stats") is False or isinstance(v, tuple) for k, v in eng.buf["ve"].items())
    assert any(k.endswith(".eval_stats") for k in eng.buf["ve"]) and any(k.endswith(".eval_stats") for k in eng.buf["vd"])
    for _ in range(2):                                                                     # second call: graph replay
        v1, t1, p1, mu1, lv1 = [None if x is None else x.clone() for x in eng([gi[0], None], pose=None)]
    prm, buf = O.split_state(sd, requires_grad=False)
    with O.eval_mode():
        mo, lo = O.image_encoder(inputs[0], prm, "visual_encoder", None, buf)
    pm, plv = O.product_of_experts(torch.stack([torch.zeros_like(mo), mo]), torch.stack([torch.zeros_like(lo), lo]))
    torch.testing.assert_close(mu1.cpu(), pm, rtol=1e-4, atol=3e-5)
    torch.testing.assert_close(lv1.cpu(), plv, rtol=1e-4, atol=3e-5)
    ref = MVAEInference(m, precision="fp32", seed=3)                                       # same Philox stream: same z
    for _ in range(2):
        v0, t0, p0, mu0, lv0 = [None if x is None else x.clone() for x in ref([gi[0], None], pose=None)]
    torch.testing.assert_close(mu1, mu0, rtol=1e-4, atol=3e-5)
    assert float((v1 - v0).norm() / v0.norm()) < 1e-4 and float((t1 - t0).norm() / t0.norm()) < 1e-4
    # new weights and running statistics, refresh(): same graphs, new results -- equal to a fresh engine's
    sd2 = seeded_running_stats(seeded_state_dict(shapes, 5), 11)
    ptrs = [x.data_ptr() for x in eng.buf["ve"][next(k for k in eng.buf["ve"] if k.endswith(".eval_stats"))]]
    m.load_state_dict(sd2)
    eng.refresh()
    assert ptrs == [x.data_ptr() for x in eng.buf["ve"][next(k for k in eng.buf["ve"] if k.endswith(".eval_stats"))]]
    n_graphs = len(eng._graphs)
    a = [None if x is None else x.clone() for x in eng([gi[0], gi[1]], pose=gi[2])]
    a = [None if x is None else x.clone() for x in eng([gi[0], None], pose=None)]
    assert len(eng._graphs) == n_graphs + 1                                                # (the joint shape was new; the visual-only graph was reused)
    fresh = MVAEInference(m, seed=3)
    for _ in range(2):
        b = [None if x is None else x.clone() for x in fresh([gi[0], None], pose=None)]
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])                             # means / log_var: no randomness
    assert not torch.equal(a[3], mu1)
    n_twins = len(layers.PLANE_TWIN)
    eng.close(), ref.close(), fresh.close()
    assert len(layers.PLANE_TWIN) < n_twins


@pytest.mark.parametrize("B", [1, 77, 130, 300])
def test_inference_engine_ragged_batches_default_arithmetic_equals_native(B):
    """engine.MVAEInference at batch sizes that are no multiple of a GEMM tile (and one below every plane kernel's work threshold): the
    default arithmetic against the native-fp32 engine on the same Philox stream, every modality subset; replays included."""
    from mmdyn_hip.engine import MVAEInference
    from mmdyn_hip.utils.seeded_init import seeded_running_stats
    sd = seeded_running_stats(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0), 7)
    m = T.build("cnn-mvae", True, True, DEV)
    m.load_state_dict(sd)
    m.eval()
    inputs, _ = seeded_batch(B, 44)
    gi = [x.to(DEV) for x in inputs]
    engs = [MVAEInference(m, seed=9), MVAEInference(m, precision="fp32", seed=9)]
    try:
        for x, pose in (([gi[0], gi[1]], gi[2]), ([None, gi[1]], gi[2]), ([gi[0], None], None)):
            outs = []
            for e in engs:
                for _ in range(2):                     # (eager capture pass, then a replay; both engines draw the same z each time)
                    o = [None if t is None else t.clone() for t in e(x, pose=pose)]
                outs.append(o)
            a, b = outs
            torch.testing.assert_close(a[3], b[3], rtol=1e-4, atol=3e-5)
            torch.testing.assert_close(a[4], b[4], rtol=1e-4, atol=3e-5)
            for i in (0, 1, 2):
                if a[i] is not None:
                    assert tuple(a[i].shape) == tuple(b[i].shape)
                    assert float((a[i] - b[i]).norm() / (b[i].norm() + 1e-30)) < 2e-4, (B, i)
    finally:
        for e in engs:
            e.close()


def test_data_parallel_schedule_two_ranks_on_one_gpu():
    """``bench.py --gpus 2`` with both ranks on this one GPU and the collectives over gloo (MMDYN_BENCH_REHEARSE_ONE_GPU=1): the real
    kernels and HIP graphs under the data-parallel schedule -- gradient buckets reduced between the graph rows, rank-0 broadcast,
    capture-key agreement, barriers -- run to the end and train (a finite, falling loss); no RCCL, no scaling claim."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MMDYN_BENCH_REHEARSE_ONE_GPU="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "16",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["launch"] == "hip_graph" and "rehearsal" in d
    assert np.isfinite(d["config"]["final_loss"]) and d["config"]["final_loss"] < 90000.0


def test_single_lane_engine_replays_graphs_too():
    """MVAEStep(two_lanes=False) -- the diagnostic form bench.py --single-lane builds -- captures and replays its step like the
    two-lane engine (it used to fail in the capture: no side streams) and trains identically."""
    B, klw = 8, 0.02
    inputs, targets = seeded_batch(B, 5)
    gi, gt = [x.to(DEV) for x in inputs], [x.to(DEV) for x in targets]
    runs = []
    for two in (True, False):
        step = MVAEStep(T.build("cnn-mvae", True, True, DEV), noise=NoiseSource(11), two_lanes=two)
        runs.append([float(step.train_step_graphed(gi, gt, klw)) for _ in range(4)])
    assert runs[0] == pytest.approx(runs[1], rel=1e-6) and runs[0][-1] < runs[0][0]
