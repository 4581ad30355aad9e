"""Pins the CPU oracle (oracle/mvae_oracle.py) to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import mvae_oracle as O
from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise
from mmdyn_hip.models.shapes import state_dict_shapes

torch.set_num_threads(min(8, os.cpu_count() or 1))


def summarize(t, k=48):
    t = t.detach().to(torch.float64).reshape(-1)
    n = t.numel()
    idx = torch.linspace(0, n - 1, steps=min(k, n)).round().long()
    return np.concatenate([[float(t.sum()), float(t.norm()), float(n)], t[idx].numpy()])


def close_summary(got, want, rtol, what):
    got, want = np.asarray(got), np.asarray(want)
    scale = max(abs(want[1]) / max(np.sqrt(want[2]), 1.0), 1e-30)   # rms of the tensor
    assert abs(got[1] - want[1]) <= rtol * max(abs(want[1]), 1e-30) + 1e-12, (what, "norm", got[1], want[1])
    np.testing.assert_allclose(got[3:], want[3:], rtol=0, atol=rtol * 50 * scale + 1e-12, err_msg=what)


def close_params(got, want, lr, n_steps, what):
    """Post-Adam parameters.  Adam's early updates are ~lr*sign(g), so an element whose gradient is at
    rounding-noise level may legitimately move by up to 2*lr per step in the other direction; require
    the bulk to agree tightly and every element to stay inside that envelope."""
    got, want = np.asarray(got), np.asarray(want)
    assert abs(got[1] - want[1]) <= 1e-4 * abs(want[1]) + 1e-9, (what, "norm", got[1], want[1])
    err = np.abs(got[3:] - want[3:])
    assert np.all(err <= 2.0 * lr * n_steps + 1e-9), (what, err.max())
    assert np.mean(err <= 0.02 * lr * n_steps + 1e-4 * np.abs(want[3:])) >= 0.9, (what, err)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_small_ops(golden_dir):
    g = load(golden_dir, "small_ops.npz")
    mu, lv = O.product_of_experts(torch.tensor(g["poe/mu"]), torch.tensor(g["poe/logvar"]))
    np.testing.assert_allclose(mu.numpy(), g["poe/out_mu"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(lv.numpy(), g["poe/out_logvar"], rtol=1e-6, atol=1e-6)
    T = lambda k: torch.tensor(g["elbo/" + k])
    klw = float(g["elbo/kl_weight"])
    vtp = O.mvae_elbo_loss([T("rv"), T("rt"), T("rp")], [T("xv"), T("xt"), T("xp")], T("means"), T("log_var"), klw)
    assert float(vtp) == pytest.approx(float(g["elbo/mvae_vtp"]), rel=1e-6)
    assert float(O.mvae_elbo_loss([T("rv")], [T("xv")], T("means"), T("log_var"), klw)) == \
        pytest.approx(float(g["elbo/mvae_v"]), rel=1e-6)
    assert float(O.mvae_elbo_loss([T("rp")], [T("xp")], T("means"), T("log_var"), klw)) == \
        pytest.approx(float(g["elbo/mvae_p"]), rel=1e-6)
    assert float(O.elbo_loss(T("rv"), T("xv"), T("means"), T("log_var"), klw)) == \
        pytest.approx(float(g["elbo/vae"]), rel=1e-6)
    assert float(O.elbo_loss(T("rv"), T("xv"), T("means"), T("log_var"), klw, loss_mask=T("mask"))) == \
        pytest.approx(float(g["elbo/vae_masked"]), rel=1e-6)
    assert float(O.mvae_elbo_loss([T("rv"), T("rt")], [T("xv"), T("xt")], T("means"), T("log_var"), klw,
                                  loss_mask=T("mask"))) == pytest.approx(float(g["elbo/mvae_vt_masked"]), rel=1e-6)
    np.testing.assert_allclose([O.anneal_kl(e, 50) for e in range(60)], g["anneal/kl"], rtol=0, atol=0)


@pytest.mark.parametrize("tag", ["seq", "dyn"])
@pytest.mark.parametrize("input_type", ["visual", "tactile", "visuotactile"])
def test_parse_input(golden_dir, tag, input_type):
    g = load(golden_dir, "small_ops.npz")
    data = [torch.tensor(g[f"parse/data{i}"]) for i in range(5)]
    target = [torch.tensor(g[f"parse/target{i}"]) for i in range(4)]
    fn = O.seq_parse_input if tag == "seq" else O.dyn_parse_input
    x, t = fn(data, target, int(g["parse/seq_length"]), input_type)
    mi, to = x["model_input"], t["target_output"]
    if not isinstance(mi, list):
        mi, to = [mi], [to]
    pre = f"parse/{tag}/{input_type}/"
    for j in range(len(mi)):
        np.testing.assert_array_equal(mi[j].numpy(), g[pre + f"model_input{j}"])
        np.testing.assert_array_equal(to[j].numpy(), g[pre + f"target_output{j}"])
    np.testing.assert_array_equal(x["input_object_pose"][0].numpy(), g[pre + "input_pose"])
    np.testing.assert_array_equal(t["target_object_pose"][0].numpy(), g[pre + "target_pose"])
    np.testing.assert_array_equal(x["input_available_modals"].numpy(), g[pre + "avail"])
    np.testing.assert_array_equal(x["shock"].numpy(), g[pre + "shock"])
    np.testing.assert_array_equal(t["loss_mask"].numpy(), g[pre + "loss_mask"])


def test_mvae_forward_subsets(golden_dir):
    g = load(golden_dir, "mvae_forward_B3.npz")
    B = int(g["batch"])
    prm, buf = O.split_state(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0), requires_grad=False)
    v, t, p = (torch.tensor(g[f"in{i}"]) for i in range(3))
    eps, masks = seeded_noise(B, 256, 8, 8, 99)
    mit = iter(masks)
    for i, (a, b, c) in enumerate(g["subsets"]):
        vr, tr, pr, mu, lv = O.mvae_forward(prm, v if a else None, t if b else None, p if c else None,
                                            eps[i], mit, True, buf)
        np.testing.assert_allclose(mu.numpy(), g[f"s{i}/means"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(lv.numpy(), g[f"s{i}/log_var"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(pr.numpy(), g[f"s{i}/pose"], rtol=1e-4, atol=2e-5)
        close_summary(summarize(vr, 256), g[f"s{i}/visual"], 2e-5, f"s{i} visual")
        close_summary(summarize(tr, 256), g[f"s{i}/tactile"], 2e-5, f"s{i} tactile")
    vr, tr = O.mvae_inference(prm, eps[7], buf)
    close_summary(summarize(vr, 256), g["inference/visual"], 2e-5, "inference visual")
    np.testing.assert_allclose(vr[0].numpy(), g["inference/visual_full0"], rtol=1e-4, atol=2e-5)
    for k in buf:
        np.testing.assert_allclose(buf[k].double().numpy(), g["buffer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("fname,use_pose", [("mvae_pose_B4.npz", True), ("mvae_nopose_B4.npz", False)])
def test_mvae_train_steps(golden_dir, fname, use_pose):
    g = load(golden_dir, fname)
    B, n_steps = int(g["batch"]), int(g["n_steps"])
    prm, buf = O.split_state(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=use_pose), 0))
    inputs, targets = seeded_batch(B, 1234, with_pose=use_pose)
    np.testing.assert_array_equal(inputs[0].numpy(), g["in0"])
    np.testing.assert_array_equal(targets[1].numpy(), g["tg1"])
    n_pass, n_mask = (7, 8) if use_pose else (3, 4)
    eps, masks = seeded_noise(B, 256, n_pass * n_steps, n_mask * n_steps, 4321)
    names = list(prm.keys())
    opt = O.Adam([prm[k] for k in names], lr=float(g["lr"]))
    for step in range(n_steps):
        opt.zero_grad()
        outputs, loss, partials = O.evaluate_mvae(
            prm, inputs, targets, eps[step * n_pass:(step + 1) * n_pass], masks[step * n_mask:(step + 1) * n_mask],
            float(g["kl_weight"]), float(g["pose_multiplier"]), use_pose, buf)
        loss.backward()
        assert float(loss.detach()) == pytest.approx(float(g[f"loss_step{step}"]), rel=2e-5), step
        if step == 0:
            np.testing.assert_allclose([float(x) for x in partials], g["loss_partials"], rtol=1e-5)
            np.testing.assert_allclose(outputs["means"].detach().numpy(), g["means"], rtol=1e-4, atol=2e-5)
            np.testing.assert_allclose(outputs["log_var"].detach().numpy(), g["log_var"], rtol=1e-4, atol=2e-5)
            pm = outputs["perf_measure"]
            np.testing.assert_allclose([pm["visual"], pm["tactile"]], g["perf_measure"][:2], rtol=1e-5)
            if use_pose:
                assert pm["pose"] == pytest.approx(float(g["perf_measure"][2]), rel=1e-5)
                np.testing.assert_allclose(outputs["recon_x"][2].detach().numpy(), g["recon2"], rtol=1e-4, atol=2e-5)
            close_summary(summarize(outputs["recon_x"][0], 256), g["recon0"], 2e-5, "recon0")
            for k in names:
                close_summary(summarize(prm[k].grad), g["grad/" + k], 2e-4, "grad " + k)
        opt.step()
        if step in (0, n_steps - 1):
            for k in names:
                close_params(summarize(prm[k]), g[f"param_step{step}/" + k], float(g["lr"]), step + 1, f"param {k} step {step}")
            for k in buf:
                # after >1 Adam steps the weights themselves have drifted by O(1e-2 * lr) (see close_params)
                    np.testing.assert_allclose(buf[k].double().numpy(), g[f"buffer_step{step}/" + k], rtol=2e-5,
                                               atol=1e-6 if step == 0 else 2e-3, err_msg=k)


def test_vae_config1(golden_dir):
    """BASELINE config 1: cnn-vae, visual, bs16."""
    g = load(golden_dir, "vae_visual_B16.npz")
    B = int(g["batch"])
    prm, buf = O.split_state(seeded_state_dict(state_dict_shapes("cnn-vae"), 0))
    x, y = torch.tensor(g["x"]), torch.tensor(g["y"])
    eps, masks = seeded_noise(B, 256, 2, 2, 31)
    names = list(prm.keys())
    opt = O.Adam([prm[k] for k in names], lr=1e-3)
    for step in range(2):
        opt.zero_grad()
        out, loss = O.evaluate_vae(prm, x, y, eps[step], masks[step], float(g["kl_weight"]), buf)
        loss.backward()
        assert float(loss.detach()) == pytest.approx(float(g[f"loss_step{step}"]), rel=2e-5)
        if step == 0:
            np.testing.assert_allclose(out["means"].detach().numpy(), g["means"], rtol=1e-4, atol=2e-5)
            assert out["perf_measure"] == pytest.approx(float(g["perf_measure"]), rel=1e-5)
            for k in names:
                close_summary(summarize(prm[k].grad), g["grad/" + k], 2e-4, "grad " + k)
        opt.step()
    for k in names:
        close_params(summarize(prm[k]), g["param_step1/" + k], 1e-3, 2, "param " + k)
    for k in buf:
        np.testing.assert_allclose(buf[k].double().numpy(), g["buffer_step1/" + k], rtol=2e-5, atol=2e-3)


def test_mvae_conditional(golden_dir):
    """--conditional cnn-mvae (condition_dim 3 = the shock force): vae.py:231-237 / 286-291 concatenations."""
    g = load(golden_dir, "mvae_conditional_B2.npz")
    B = int(g["batch"])
    prm, buf = O.split_state(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True, cond=3), 0))
    assert prm["visual_encoder.linear_means.weight"].shape == (256, 515)
    inputs, targets = seeded_batch(B, 321)
    eps, masks = seeded_noise(B, 256, 7, 8, 77)
    cond = torch.tensor(g["cond"])
    outputs, loss, partials = O.evaluate_mvae(prm, inputs, targets, eps, masks, float(g["kl_weight"]), 1000.0, True, buf,
                                              condition=cond)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-5)
    np.testing.assert_allclose([float(x) for x in partials], g["loss_partials"], rtol=1e-5)
    np.testing.assert_allclose(outputs["means"].detach().numpy(), g["means"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(outputs["recon_x"][2].detach().numpy(), g["recon2"], rtol=1e-4, atol=2e-5)
    close_summary(summarize(outputs["recon_x"][0], 256), g["recon0"], 2e-5, "recon0")
    for k in prm:
        close_summary(summarize(prm[k].grad), g["grad/" + k], 2e-4, "grad " + k)


@pytest.mark.parametrize("tag", ["plain", "cond"])
def test_regressor(golden_dir, tag):
    """Regressor baseline (models.py:28-77) + MSE-sum criterion (problems.py:323-335)."""
    g = load(golden_dir, "regressor_B4.npz")
    B = int(g["batch"])
    shapes = state_dict_shapes("regressor", cond=3 if tag == "cond" else 0)
    assert list(shapes.keys()) == [str(k) for k in g[tag + "/keys"]]
    prm, buf = O.split_state(seeded_state_dict(shapes, 0))
    _, masks = seeded_noise(B, 256, 1, 2, 55)
    x, pose, cond = (torch.tensor(g[k]) for k in ("x", "pose", "cond"))
    y = O.regressor_forward(prm, x, masks[0] if tag == "plain" else masks[1], cond if tag == "cond" else None, buf)
    loss = ((y - pose) ** 2).sum()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g[tag + "/loss"]), rel=2e-5)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + "/out"], rtol=1e-4, atol=2e-5)
    for k in prm:
        close_summary(summarize(prm[k].grad), g[f"{tag}/grad/" + k], 2e-4, "grad " + k)
    for k in buf:
        np.testing.assert_allclose(buf[k].double().numpy(), g[f"{tag}/buffer/" + k], rtol=1e-5, atol=1e-6, err_msg=k)


def test_mlp_vae(golden_dir):
    """config.MODELS[0] 'mlp-vae' on flat 784 inputs (the only input size the reference's mlp Decoder is consistent
    with) + problems._elbo_loss."""
    g = load(golden_dir, "mlp_vae_B6.npz")
    shapes = state_dict_shapes("mlp-vae", latent=32)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    prm, _ = O.split_state(seeded_state_dict(shapes, 0))
    x = torch.tensor(g["x"])
    recon, mu, lv = O.mlp_vae_forward(prm, x, torch.tensor(g["eps"]))
    loss = O.elbo_loss(recon, x, mu, lv, 0.1)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=2e-5)
    np.testing.assert_allclose(recon.detach().numpy(), g["recon"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(mu.detach().numpy(), g["means"], rtol=1e-4, atol=2e-5)
    for k in prm:
        close_summary(summarize(prm[k].grad), g["grad/" + k], 2e-4, "grad " + k)


def test_eval_mode(golden_dir):
    """model.eval(): BatchNorm2d with the running estimates, no dropout -- cnn-mvae forward + inference, cnn-vae,
    Regressor, against the reference in eval mode."""
    from mmdyn_hip.utils.seeded_init import seeded_running_stats
    g = load(golden_dir, "eval_mode_B3.npz")
    B = int(g["batch"])
    inputs, _ = seeded_batch(B, 4242, with_pose=True)
    eps = [torch.tensor(g[f"eps{i}"]) for i in range(4)]
    with O.eval_mode(), torch.no_grad():
        sd = seeded_running_stats(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0))
        prm, buf = O.split_state(sd)
        before = {k: v.clone() for k, v in buf.items()}
        v, t, p, mu, lv = O.mvae_forward(prm, inputs[0], inputs[1], inputs[2], eps[0], iter([None, None]), True, buf)
        np.testing.assert_allclose(v[0].numpy(), g["mvae/visual0"], rtol=1e-4, atol=2e-5)
        close_summary(summarize(t, 256), g["mvae/tactile"], 2e-5, "tactile")
        np.testing.assert_allclose(p.numpy(), g["mvae/pose"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(mu.numpy(), g["mvae/means"], rtol=1e-4, atol=2e-5)
        iv, it = O.mvae_inference(prm, eps[1], buf)
        np.testing.assert_allclose(iv[0].numpy(), g["mvae/inference_visual0"], rtol=1e-4, atol=2e-5)
        for k in buf:
            assert torch.equal(buf[k], before[k]), k
            np.testing.assert_allclose(buf[k].double().numpy(), g["mvae/buffer/" + k], rtol=1e-6)
        sd = seeded_running_stats(seeded_state_dict(state_dict_shapes("cnn-vae"), 0))
        prm, buf = O.split_state(sd)
        r, mu, _ = O.vae_forward(prm, inputs[1], eps[2], None, buf)
        np.testing.assert_allclose(r[0].numpy(), g["vae/recon0"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(mu.numpy(), g["vae/means"], rtol=1e-4, atol=2e-5)
        sd = seeded_running_stats(seeded_state_dict(state_dict_shapes("regressor"), 0))
        prm, buf = O.split_state(sd)
        out = O.regressor_forward(prm, inputs[0], torch.ones(B, 512) * (1 - O.DROPOUT_P), None, buf)
        np.testing.assert_allclose(out.numpy(), g["regressor/out"], rtol=1e-4, atol=2e-5)
