#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE on CPU.

Run only in the development container (needs /root/reference; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

What it does (SURVEY.md section 8c recipe):
  * registers empty stand-in modules for the reference's *non-numeric* imports
    (torchvision, cv2, pyquaternion, tensorboard) so that
    mmdyn.pytorch.problems.problems can be imported; no arithmetic is stubbed;
  * builds the reference's own MVAE / VAE through its ``setup_model`` and loads
    name-keyed deterministic weights (mmdyn_hip.utils.seeded_init) into them;
  * injects the random draws (``torch.randn`` noise, dropout keep-masks) so the
    outputs are reproducible functions of committed data;
  * calls the reference's ``Reconstruction._evaluate_mvae`` / ``_mvae_elbo_loss`` /
    ``_elbo_loss`` / ``SeqModeling._evaluate_model`` / ``*.parse_input`` /
    ``ProductOfExperts`` / ``MVAE.forward`` / ``inference`` and stores inputs and
    outputs as ``.npz`` (data only; no reference source is stored).
"""
import io
import pickle
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mmdyn_hip.utils.seeded_init import (  # noqa: E402
    seeded_state_dict, seeded_batch, seeded_noise, seeded_running_stats)


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.datasets = stub("torchvision.datasets", VisionDataset=object)
    tv.transforms = stub("torchvision.transforms")
    tv.utils = stub("torchvision.utils")
    stub("cv2")
    stub("pyquaternion", Quaternion=object)
    stub("torch.utils.tensorboard", SummaryWriter=object)
    real_popen = os.popen
    os.popen = lambda *a, **k: io.StringIO("24 80")
    try:
        from mmdyn.pytorch.problems import problems as P
        from mmdyn.pytorch.models import models as M, vae as V
    finally:
        os.popen = real_popen
    return P, M, V


P, M, V = import_reference()
torch.set_num_threads(8)


class Injector:
    """Replaces torch.randn and F.dropout by queues of pre-drawn values."""

    def __init__(self, eps, masks, p=0.1):
        self.eps, self.masks, self.p = list(eps), list(masks), p
        self.used_eps, self.used_masks = 0, 0

    def __enter__(self):
        self._randn, self._dropout = torch.randn, F.dropout

        def randn(*size, **kw):
            e = self.eps[self.used_eps]
            self.used_eps += 1
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            assert tuple(e.shape) == shape, (e.shape, shape)
            return e.clone()

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training:                      # module.eval(): nn.Dropout is the identity
                return x
            assert abs(p - self.p) < 1e-12
            m = self.masks[self.used_masks]
            self.used_masks += 1
            assert m.shape == x.shape
            return x * (m.to(x.dtype) / (1.0 - p))

        torch.randn = randn
        F.dropout = dropout
        return self

    def __exit__(self, *exc):
        torch.randn, F.dropout = self._randn, self._dropout


def summarize(t, k=48):
    t = t.detach().to(torch.float64).reshape(-1)
    n = t.numel()
    idx = torch.linspace(0, n - 1, steps=min(k, n)).round().long()
    return np.concatenate([[float(t.sum()), float(t.norm()), float(n)], t[idx].numpy()]).astype(np.float64)


def make_self(model, use_pose, model_name, kl_weight, pose_multiplier=1000.0, input_type="visuotactile",
              conditional=False, seq_length=1):
    s = types.SimpleNamespace()
    s._model = model
    s._kl_weight = kl_weight
    s._pose_multiplier = pose_multiplier
    s._conditional = conditional
    s._seq_length = seq_length
    s._device = torch.device("cpu")
    s.parameters = {"use_pose": use_pose, "model_name": model_name, "mask_loss": False,
                    "input_type": input_type}
    s.partials = []

    def elbo(self, *a, **k):
        r = P.Reconstruction._mvae_elbo_loss(self, *a, **k)
        self.partials.append(float(r.detach()))
        return r

    s._mvae_elbo_loss = types.MethodType(elbo, s)
    s._elbo_loss = types.MethodType(P.Reconstruction._elbo_loss, s)
    s._evaluate_mvae = types.MethodType(P.Reconstruction._evaluate_mvae, s)
    return s


MODEL_KW = dict(condition_dim=0, input_dim=4096, architecture="cnn", conditional=False,
                categorical_conditions=False, latent_size=256)


def build(model_name, cross_modal, use_pose=None, seed=0):
    kw = dict(MODEL_KW)
    if use_pose is not None:
        kw["use_pose"] = use_pose
    model = M.setup_model(model_name, cross_modal=cross_modal, **kw)
    model.load_state_dict(seeded_state_dict(model.state_dict(), seed))
    model.train()
    return model


def gen_mvae_step(use_pose, batch, fname, n_steps=3):
    model = build("cnn-mvae", True, use_pose)
    n_pass, n_mask = (7, 8) if use_pose else (3, 4)
    inputs, targets = seeded_batch(batch, 1234, with_pose=use_pose)
    eps, masks = seeded_noise(batch, 256, n_pass * n_steps, n_mask * n_steps, 4321)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    slf = make_self(model, use_pose, "cnn-mvae", kl_weight=1.0 / 50)
    out = {"batch": batch, "use_pose": int(use_pose), "kl_weight": 1.0 / 50, "pose_multiplier": 1000.0,
           "torch_version": torch.__version__, "n_steps": n_steps, "lr": 1e-3}
    for i, t in enumerate(inputs):
        out[f"in{i}"] = t.numpy()
    for i, t in enumerate(targets):
        out[f"tg{i}"] = t.numpy()
    # layer probes of the first (joint v,t) pass
    probes = {}

    def hook(name):
        def f(mod, inp, outp):
            if name not in probes and torch.is_tensor(outp):
                probes[name] = summarize(outp)
        return f

    handles = []
    for name, mod in model.named_modules():
        if len(list(mod.children())) == 0 and name:
            handles.append(mod.register_forward_hook(hook(name)))
    with Injector(eps, masks) as inj:
        for step in range(n_steps):
            slf.partials = []
            opt.zero_grad()
            outputs, loss = slf._evaluate_mvae(x=list(inputs), targets=list(targets))
            loss.backward()
            if step == 0:
                for h in handles:
                    h.remove()
                out["loss_partials"] = np.array(slf.partials, dtype=np.float64)
                out["means"] = outputs["means"].detach().numpy()
                out["log_var"] = outputs["log_var"].detach().numpy()
                for i, r in enumerate(outputs["recon_x"]):
                    out[f"recon{i}"] = r.detach().numpy() if r.dim() == 2 else summarize(r, 256)
                pm = outputs["perf_measure"]
                out["perf_measure"] = np.array([pm["visual"], pm["tactile"], pm.get("pose", 0.0)], dtype=np.float64)
                for n, p_ in model.named_parameters():
                    out["grad/" + n] = summarize(p_.grad)
                for n, v in probes.items():
                    out["probe/" + n] = v
            opt.step()
            out[f"loss_step{step}"] = np.float64(loss.item())
            if step in (0, n_steps - 1):
                for n, p_ in model.named_parameters():
                    out[f"param_step{step}/" + n] = summarize(p_)
                for n, b in model.named_buffers():
                    out[f"buffer_step{step}/" + n] = b.detach().numpy().astype(np.float64)
        assert inj.used_eps == n_pass * n_steps and inj.used_masks == n_mask * n_steps
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss per step", [float(out[f"loss_step{s}"]) for s in range(n_steps)],
          "partials", out["loss_partials"])


def gen_mvae_forward(batch, fname):
    """MVAE.forward for each of the seven modality subsets + inference()."""
    model = build("cnn-mvae", True, True)
    inputs, _ = seeded_batch(batch, 77, with_pose=True)
    v, t, p = inputs
    subsets = [(1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1), (1, 0, 1), (0, 1, 1), (0, 0, 1)]
    eps, masks = seeded_noise(batch, 256, len(subsets) + 1, 8, 99)
    out = {"batch": batch, "subsets": np.array(subsets)}
    out["in0"], out["in1"], out["in2"] = v.numpy(), t.numpy(), p.numpy()
    with Injector(eps, masks), torch.no_grad():
        for i, (a, b, c) in enumerate(subsets):
            vr, tr, pr, mu, lv = model([v if a else None, t if b else None], pose=p if c else None)
            out[f"s{i}/visual"] = summarize(vr, 256)
            out[f"s{i}/tactile"] = summarize(tr, 256)
            out[f"s{i}/pose"] = pr.numpy()
            out[f"s{i}/means"] = mu.numpy()
            out[f"s{i}/log_var"] = lv.numpy()
        vr, tr = model.inference(n=batch)
        out["inference/visual"] = summarize(vr, 256)
        out["inference/tactile"] = summarize(tr, 256)
        out["inference/visual_full0"] = vr[0].numpy()
    for n, b in model.named_buffers():
        out["buffer/" + n] = b.detach().numpy().astype(np.float64)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "ok")


def gen_vae_step(batch, fname):
    """BASELINE config 1: cnn-vae --input-type visual seq_modeling (problems.py:702-716)."""
    model = build("cnn-vae", False)
    g = torch.Generator().manual_seed(555)
    x = torch.rand(batch, 3, 64, 64, generator=g)
    y = torch.rand(batch, 3, 64, 64, generator=g)
    eps, masks = seeded_noise(batch, 256, 2, 2, 31)
    slf = make_self(model, False, "cnn-vae", kl_weight=1.0 / 50, input_type="visual")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    out = {"batch": batch, "kl_weight": 1.0 / 50, "x": x.numpy(), "y": y.numpy()}
    with Injector(eps, masks):
        for step in range(2):
            opt.zero_grad()
            outputs, loss = P.SeqModeling._evaluate_model(
                slf, {"model_input": x, "shock": None}, {"target_output": y, "loss_mask": None})
            loss.backward()
            if step == 0:
                out["means"] = outputs["means"].detach().numpy()
                out["log_var"] = outputs["log_var"].detach().numpy()
                out["recon"] = summarize(outputs["recon_x"], 256)
                out["perf_measure"] = np.float64(outputs["perf_measure"]["visual"])
                for n, p_ in model.named_parameters():
                    out["grad/" + n] = summarize(p_.grad)
            opt.step()
            out[f"loss_step{step}"] = np.float64(loss.item())
        for n, p_ in model.named_parameters():
            out["param_step1/" + n] = summarize(p_)
        for n, b in model.named_buffers():
            out["buffer_step1/" + n] = b.detach().numpy().astype(np.float64)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", float(out["loss_step0"]), float(out["loss_step1"]))


def gen_conditional(batch, fname):
    """--conditional (shock-conditioned) cnn-mvae: vae.py:196, 231-237, 257, 286-291; problems.py:692-695."""
    kw = dict(MODEL_KW)
    kw.update(conditional=True, condition_dim=3, use_pose=True)
    model = M.setup_model("cnn-mvae", cross_modal=True, **kw)
    model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
    model.train()
    inputs, targets = seeded_batch(batch, 321, with_pose=True)
    g = torch.Generator().manual_seed(9)
    cond = torch.rand(batch, 3, generator=g)
    eps, masks = seeded_noise(batch, 256, 7, 8, 77)
    slf = make_self(model, True, "cnn-mvae", kl_weight=0.02, conditional=True)
    out = {"batch": batch, "cond": cond.numpy(), "kl_weight": 0.02}
    with Injector(eps, masks):
        outputs, loss = slf._evaluate_mvae(x=list(inputs), targets=list(targets), condition=cond)
        loss.backward()
    out["loss"] = np.float64(loss.item())
    out["loss_partials"] = np.array(slf.partials, dtype=np.float64)
    out["means"] = outputs["means"].detach().numpy()
    out["recon2"] = outputs["recon_x"][2].detach().numpy()
    out["recon0"] = summarize(outputs["recon_x"][0], 256)
    for n, p_ in model.named_parameters():
        out["grad/" + n] = summarize(p_.grad)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", float(out["loss"]))


def gen_regressor(batch, fname):
    """Regressor baseline (models.py:28-77) with the Regression problem's MSE-sum criterion (problems.py:323-335),
    plain and shock-conditioned (conditional=True, num_classes=3)."""
    out = {"batch": batch}
    g = torch.Generator().manual_seed(808)
    x = torch.rand(batch, 3, 64, 64, generator=g)
    pose = torch.rand(batch, 7, generator=g)
    cond = torch.rand(batch, 3, generator=g)
    _, masks = seeded_noise(batch, 256, 1, 2, 55)
    out.update(x=x.numpy(), pose=pose.numpy(), cond=cond.numpy())
    for tag, kw in (("plain", dict(num_classes=3)), ("cond", dict(conditional=True, num_classes=3))):
        model = M.Regressor(out_dim=7, **kw)
        model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
        model.train()
        with Injector([], masks[:1] if tag == "plain" else masks[1:]):
            y = model(x, cond) if tag == "cond" else model(x)
            loss = torch.nn.MSELoss(reduction='sum')(y.view(pose.size()), pose)
            loss.backward()
        out[f"{tag}/out"] = y.detach().numpy()
        out[f"{tag}/loss"] = np.float64(loss.item())
        out[f"{tag}/keys"] = np.array(list(model.state_dict().keys()))
        for n, p_ in model.named_parameters():
            out[f"{tag}/grad/" + n] = summarize(p_.grad)
        for n, b in model.named_buffers():
            out[f"{tag}/buffer/" + n] = b.detach().numpy().astype(np.float64)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", float(out["plain/loss"]), float(out["cond/loss"]))


def gen_dataset(fname):
    """The reference's own tree compiler (datasets.py:159-267) on tests/synthetic_tree.build_tree, and PIL's
    Resize(64)+ToTensor arithmetic (datasets.py:23-31) on sample frames.  torchvision is not installed here; what its
    Resize does to a PIL image is ``img.resize((w, h), BILINEAR)`` with the short side scaled to 64."""
    import random
    import tempfile
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synthetic_tree as ST
    from mmdyn.pytorch.utils import datasets as RD
    out = {}
    for tag, shock in (("shock", True), ("plain", False)):
        with tempfile.TemporaryDirectory() as tmp:
            ST.build_tree(tmp, shock=shock)
            random.seed(7)
            ds = RD.VisuoTactileDataset(train=True, transform=None, dataset_path=tmp)
            with open(ds.dataset_path, "rb") as f:
                comp = pickle.load(f)
            desc = ST.describe(comp)
            out[f"tree/{tag}/keys"] = np.array(list(desc.keys()))
            out[f"tree/{tag}/sha"] = np.array(list(desc.values()))
            out[f"tree/{tag}/n_train"] = len(ds)
            out[f"tree/{tag}/seq_length"] = ds.seq_length
            out[f"tree/{tag}/n_test"] = len(RD.VisuoTactileDataset(train=False, transform=None, dataset_path=tmp))
            if shock:
                out["tree/sample_pose"] = np.asarray(comp["data"][0][1][2])
                out["tree/sample_shock"] = np.asarray(comp["data"][0][1][4])
                out["tree/sample_visual"] = comp["data"][0][0][0]
                img = comp["data"][0][0][0]
                out["tree/sample_visual_64"] = np.array(Image.fromarray(img).resize((64, 64), Image.BILINEAR))
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    out["resize/noise"] = noise
    out["resize/noise_64"] = np.array(Image.fromarray(noise).resize((64, 64), Image.BILINEAR))
    out["resize/noise_128"] = np.array(Image.fromarray(noise).resize((128, 128), Image.BILINEAR))
    rect = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    out["resize/rect"] = rect
    out["resize/rect_64"] = np.array(Image.fromarray(rect).resize((106, 64), Image.BILINEAR))     # Resize(64): 64 x int(64*200/120)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "sequences", int(out["tree/shock/n_train"]), int(out["tree/shock/n_test"]))


def gen_mlp_vae(batch, fname):
    """mlp-vae (config.MODELS[0]): VAE with the mlp Encoder / Decoder (vae.py:14-19, 218-222, 281-283) on flat inputs.
    The reference's Decoder takes ``output_dim`` (default 784), not ``input_dim``, so only input_dim = 784 is
    self-consistent; that is the case pinned here, with problems._elbo_loss as the criterion."""
    model = M.setup_model("mlp-vae", input_dim=784, architecture="mlp", latent_size=32, condition_dim=0,
                          conditional=False, categorical_conditions=False)
    model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
    model.train()
    g = torch.Generator().manual_seed(99)
    x = torch.rand(batch, 1, 28, 28, generator=g)          # > 2-D input: VAE.forward flattens it (vae.py:82-83)
    eps = [torch.randn(batch, 32, generator=g)]
    slf = make_self(model, False, "mlp-vae", kl_weight=0.1, input_type="visual")
    with Injector(eps, []):
        recon, means, log_var = model(x)
        loss = slf._elbo_loss(recon, x, means, log_var)
        loss.backward()
    out = {"batch": batch, "x": x.numpy(), "eps": eps[0].numpy(), "recon": recon.detach().numpy(),
           "means": means.detach().numpy(), "log_var": log_var.detach().numpy(), "loss": np.float64(loss.item()),
           "keys": np.array(list(model.state_dict().keys()))}
    for n, p_ in model.named_parameters():
        out["grad/" + n] = summarize(p_.grad)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", float(out["loss"]))


def gen_eval_mode(batch, fname):
    """model.eval(): BatchNorm2d on the running estimates, Dropout off (deployment-style inference; the reference's
    own loops never leave train mode).  cnn-mvae forward + inference, cnn-vae forward, Regressor."""
    out = {"batch": batch}
    inputs, _ = seeded_batch(batch, 4242, with_pose=True)
    eps = [torch.randn(batch, 256, generator=torch.Generator().manual_seed(11 + i)) for i in range(4)]
    model = build("cnn-mvae", True, use_pose=True)
    model.load_state_dict(seeded_running_stats(model.state_dict()))
    model.eval()
    with torch.no_grad(), Injector(eps[:2], []):
        v, t, p, mu, lv = model([inputs[0], inputs[1]], pose=inputs[2])
        iv, it = model.inference(n=batch)
    out.update({"mvae/visual": summarize(v, 256), "mvae/tactile": summarize(t, 256), "mvae/visual0": v[0].numpy(),
                "mvae/pose": p.numpy(), "mvae/means": mu.numpy(), "mvae/log_var": lv.numpy(),
                "mvae/inference_visual0": iv[0].numpy(), "mvae/inference_tactile": summarize(it, 256)})
    for k, b in model.named_buffers():
        out["mvae/buffer/" + k] = b.numpy().astype(np.float64)          # must be untouched by eval forwards
    vae = build("cnn-vae", False)
    vae.load_state_dict(seeded_running_stats(vae.state_dict()))
    vae.eval()
    with torch.no_grad(), Injector(eps[2:3], []):
        r, mu, lv = vae(inputs[1])
    out.update({"vae/recon": summarize(r, 256), "vae/recon0": r[0].numpy(), "vae/means": mu.numpy()})
    reg = M.Regressor(out_dim=7, conditional=False, num_classes=0)
    sd = seeded_running_stats(seeded_state_dict(reg.state_dict(), 0))
    reg.load_state_dict(sd)
    reg.eval()
    with torch.no_grad():
        out["regressor/out"] = reg(inputs[0]).numpy()
    for i in range(4):
        out[f"eps{i}"] = eps[i].numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "ok")


def gen_small_ops(fname):
    g = torch.Generator().manual_seed(2024)
    out = {}
    # ProductOfExperts (vae.py:311-318), including extreme log-variances
    mu = torch.randn(4, 5, 16, generator=g)
    lv = torch.randn(4, 5, 16, generator=g) * 3.0
    lv[0] = 0.0
    mu[0] = 0.0
    lv[1, 0, :4] = torch.tensor([-30.0, -18.0, 20.0, 40.0])
    pm, plv = V.ProductOfExperts()(mu, lv)
    out["poe/mu"], out["poe/logvar"] = mu.numpy(), lv.numpy()
    out["poe/out_mu"], out["poe/out_logvar"] = pm.numpy(), plv.numpy()
    # _mvae_elbo_loss (problems.py:421-458) and _elbo_loss (:401-419)
    slf = types.SimpleNamespace(_kl_weight=0.37, _pose_multiplier=1000.0)
    rv = torch.randn(3, 3, 8, 8, generator=g) * 4
    rt = torch.randn(3, 3, 8, 8, generator=g) * 4
    rp = torch.randn(3, 7, generator=g)
    xv = torch.rand(3, 3, 8, 8, generator=g)
    xt = torch.rand(3, 3, 8, 8, generator=g)
    xp = torch.rand(3, 7, generator=g)
    means = torch.randn(3, 16, generator=g)
    log_var = torch.randn(3, 16, generator=g)
    for k, val in dict(rv=rv, rt=rt, rp=rp, xv=xv, xt=xt, xp=xp, means=means, log_var=log_var).items():
        out["elbo/" + k] = val.numpy()
    out["elbo/kl_weight"] = np.float64(0.37)
    out["elbo/mvae_vtp"] = np.float64(P.Reconstruction._mvae_elbo_loss(slf, [rv, rt, rp], [xv, xt, xp], means, log_var))
    out["elbo/mvae_v"] = np.float64(P.Reconstruction._mvae_elbo_loss(slf, [rv], [xv], means, log_var))
    out["elbo/mvae_p"] = np.float64(P.Reconstruction._mvae_elbo_loss(slf, [rp], [xp], means, log_var))
    out["elbo/vae"] = np.float64(P.Reconstruction._elbo_loss(slf, rv, xv, means, log_var))
    mask = (torch.rand(3, 1, 8, 8, generator=g) > 0.5).float()
    out["elbo/mask"] = mask.numpy()
    out["elbo/vae_masked"] = np.float64(P.Reconstruction._elbo_loss(slf, rv, xv, means, log_var, loss_mask=mask))
    out["elbo/mvae_vt_masked"] = np.float64(
        P.Reconstruction._mvae_elbo_loss(slf, [rv, rt], [xv, xt], means, log_var, loss_mask=mask))
    # KL annealing schedule (problems.py:212-216)
    s2 = types.SimpleNamespace(parameters={"annealing_epochs": 50}, _kl_weight=None)
    sched = []
    for e in range(60):
        P.Problem._anneal_KL(s2, e)
        sched.append(s2._kl_weight)
    out["anneal/kl"] = np.array(sched, dtype=np.float64)
    # parse_input: SeqModeling (problems.py:634-673), DynModeling (:765-803); flat [B*L, ...] frames
    L, nseq = 3, 4
    n = L * nseq
    data = [torch.rand(n, 3, 4, 4, generator=g), torch.rand(n, 3, 4, 4, generator=g), torch.rand(n, 7, generator=g),
            torch.rand(n, 2, generator=g), torch.rand(n, 3, generator=g)]
    target = [torch.rand(n, 3, 4, 4, generator=g), torch.rand(n, 3, 4, 4, generator=g), torch.rand(n, 7, generator=g),
              torch.rand(n, 1, 4, 4, generator=g)]
    for i, d in enumerate(data):
        out[f"parse/data{i}"] = d.numpy()
    for i, d in enumerate(target):
        out[f"parse/target{i}"] = d.numpy()
    out["parse/seq_length"] = L
    for cls, tag in ((P.SeqModeling, "seq"), (P.DynModeling, "dyn")):
        for it in ("visual", "tactile", "visuotactile"):
            s3 = types.SimpleNamespace(_seq_length=L, _device=torch.device("cpu"), parameters={"input_type": it})
            xi, tg = cls.parse_input(s3, [d.clone() for d in data], [d.clone() for d in target])
            mi, to = xi["model_input"], tg["target_output"]
            if not isinstance(mi, list):
                mi, to = [mi], [to]
            for j in range(len(mi)):
                out[f"parse/{tag}/{it}/model_input{j}"] = mi[j].numpy()
                out[f"parse/{tag}/{it}/target_output{j}"] = to[j].numpy()
            out[f"parse/{tag}/{it}/input_pose"] = xi["input_object_pose"][0].numpy()
            out[f"parse/{tag}/{it}/target_pose"] = tg["target_object_pose"][0].numpy()
            out[f"parse/{tag}/{it}/avail"] = xi["input_available_modals"].numpy()
            out[f"parse/{tag}/{it}/shock"] = xi["shock"].numpy()
            out[f"parse/{tag}/{it}/loss_mask"] = tg["loss_mask"].numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "ok")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "conditional":
        gen_conditional(2, "mvae_conditional_B2.npz")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "eval":
        gen_eval_mode(3, "eval_mode_B3.npz")
        sys.exit(0)
    gen_eval_mode(3, "eval_mode_B3.npz")
    if len(sys.argv) > 1 and sys.argv[1] == "mlp":
        gen_mlp_vae(6, "mlp_vae_B6.npz")
        sys.exit(0)
    gen_mlp_vae(6, "mlp_vae_B6.npz")
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        gen_dataset("dataset_tree.npz")
        sys.exit(0)
    gen_dataset("dataset_tree.npz")
    if len(sys.argv) > 1 and sys.argv[1] == "regressor":
        gen_regressor(4, "regressor_B4.npz")
        sys.exit(0)
    gen_regressor(4, "regressor_B4.npz")
    gen_conditional(2, "mvae_conditional_B2.npz")
    gen_small_ops("small_ops.npz")
    gen_mvae_forward(3, "mvae_forward_B3.npz")
    gen_vae_step(16, "vae_visual_B16.npz")
    gen_mvae_step(False, 4, "mvae_nopose_B4.npz", n_steps=2)
    gen_mvae_step(True, 4, "mvae_pose_B4.npz", n_steps=3)
