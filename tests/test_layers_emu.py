"""Host-side schedule (mmdyn_hip/layers.py) checked on CPU: the kernels are replaced by the contract
emulation in tests/emu_backend.py, the expected values come from the oracle + torch autograd."""
import pytest
import torch

from oracle import mvae_oracle as O
from mmdyn_hip import layers, ops
from mmdyn_hip.models.shapes import state_dict_shapes
from mmdyn_hip.utils.seeded_init import seeded_state_dict
from emu_backend import EmuBackend


@pytest.fixture(autouse=True)
def emu():
    old = ops.set_backend(EmuBackend())
    yield
    ops.set_backend(old)


def sub(d, pre):
    return {k[len(pre) + 1:]: v for k, v in d.items() if k.startswith(pre + ".")}


def state():
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    return O.split_state(sd)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_encoder_trunk_forward_backward():
    prm, buf = state()
    torch.manual_seed(1)
    x = torch.rand(4, 3, 64, 64)
    P, Bf = sub(prm, "visual_encoder"), {k: v.clone() for k, v in sub(buf, "visual_encoder").items()}
    Pd = {k: v.detach() for k, v in P.items()}
    h, ctx = layers.encoder_trunk_forward(Pd, Bf, x, G=1)
    buf_o = {k: v.clone() for k, v in buf.items()}
    h_ref = O.image_encoder_trunk(x, prm, "visual_encoder", buf_o)
    assert rel(h, h_ref.detach()) < 1e-5
    for k in Bf:
        torch.testing.assert_close(Bf[k].double(), buf_o["visual_encoder." + k].double(), rtol=1e-5, atol=1e-6)
    dh = torch.randn_like(h)
    grads = {k: torch.zeros_like(Pd[k]) for k in layers.ENC_KEYS}
    layers.encoder_trunk_backward(Pd, ctx, dh, grads)
    (h_ref * dh).sum().backward()
    for k in layers.ENC_KEYS:
        assert rel(grads[k], prm["visual_encoder." + k].grad) < 2e-4, k


def test_encoder_trunk_groups_match_separate_calls():
    prm, buf = state()
    torch.manual_seed(2)
    x = torch.rand(4, 3, 64, 64)
    Pd = {k: v.detach() for k, v in sub(prm, "tactile_encoder").items()}
    h, _ = layers.encoder_trunk_forward(Pd, None, x, G=2)
    h0 = O.image_encoder_trunk(x[:2], prm, "tactile_encoder").detach()
    h1 = O.image_encoder_trunk(x[2:], prm, "tactile_encoder").detach()
    assert rel(h, torch.cat([h0, h1])) < 1e-5


@pytest.mark.parametrize("G", [1, 2])
def test_decoder_forward_backward(G):
    prm, buf = state()
    torch.manual_seed(3)
    z = torch.randn(4, 256, requires_grad=True)
    Pd = {k: v.detach() for k, v in sub(prm, "visual_decoder").items()}
    Bf = {k: v.clone() for k, v in sub(buf, "visual_decoder").items()}
    out, ctx = layers.decoder_forward(Pd, Bf, z.detach(), G=G)
    buf_o = {k: v.clone() for k, v in buf.items()}
    Bg = 4 // G
    ref = torch.cat([O.image_decoder(z[g * Bg:(g + 1) * Bg], prm, "visual_decoder", buf_o) for g in range(G)])
    assert out.shape == (4, 3, 64, 64)
    assert rel(out, ref.detach()) < 1e-5
    for k in Bf:
        torch.testing.assert_close(Bf[k].double(), buf_o["visual_decoder." + k].double(), rtol=1e-5, atol=1e-6)
    dl = torch.randn_like(out)
    grads = {k: torch.zeros_like(Pd[k]) for k in layers.DEC_KEYS}
    dz = layers.decoder_backward(Pd, ctx, dl, grads)
    (ref * dl).sum().backward()
    assert rel(dz, z.grad) < 2e-4
    for k in layers.DEC_KEYS:
        assert rel(grads[k], prm["visual_decoder." + k].grad) < 2e-4, k


def test_heads_and_pose_mlps():
    prm, _ = state()
    torch.manual_seed(4)
    hd = torch.randn(6, 512, requires_grad=True)
    Pd = {k: v.detach() for k, v in sub(prm, "visual_encoder").items()}
    out, c = layers.heads_forward(Pd, hd.detach())
    mu, lv = O.encoder_heads(hd, prm, "visual_encoder")
    assert rel(out[:, :256], mu.detach()) < 1e-5 and rel(out[:, 256:], lv.detach()) < 1e-5
    dout = torch.randn_like(out)
    grads = {k: torch.zeros_like(Pd[k]) for k in layers.HEAD_KEYS}
    dx = layers.heads_backward(c, dout, grads)
    ((mu * dout[:, :256]).sum() + (lv * dout[:, 256:]).sum()).backward()
    assert rel(dx, hd.grad) < 1e-4
    for k in layers.HEAD_KEYS:
        assert rel(grads[k], prm["visual_encoder." + k].grad) < 1e-4, k

    pose = torch.rand(6, 7)
    Pe = {k: v.detach() for k, v in sub(prm, "pose_encoder").items()}
    h2, c = layers.pose_encoder_trunk_forward(Pe, pose)
    h1 = torch.relu(torch.nn.functional.linear(pose, prm["pose_encoder.fc_net.0.weight"], prm["pose_encoder.fc_net.0.bias"]))
    ref = torch.nn.functional.linear(h1, prm["pose_encoder.fc_net.2.weight"], prm["pose_encoder.fc_net.2.bias"])
    assert rel(h2, ref.detach()) < 1e-5
    d = torch.randn_like(h2)
    grads = {k: torch.zeros_like(Pe[k]) for k in layers.POSE_ENC_KEYS}
    layers.pose_encoder_trunk_backward(Pe, c, d, grads)
    (ref * d).sum().backward()
    for k in layers.POSE_ENC_KEYS:
        assert rel(grads[k], prm["pose_encoder." + k].grad) < 1e-4, k

    z = torch.randn(6, 256, requires_grad=True)
    Pp = {k: v.detach() for k, v in sub(prm, "pose_decoder").items()}
    out, c = layers.pose_decoder_forward(Pp, z.detach())
    ref = O.pose_decoder(z, prm)
    assert rel(out, ref.detach()) < 1e-5
    d = torch.randn_like(out)
    grads = {k: torch.zeros_like(Pp[k]) for k in layers.POSE_DEC_KEYS}
    dz = layers.pose_decoder_backward(Pp, c, d, grads)
    (ref * d).sum().backward()
    assert rel(dz, z.grad) < 1e-4
    for k in layers.POSE_DEC_KEYS:
        assert rel(grads[k], prm["pose_decoder." + k].grad) < 1e-4, k
