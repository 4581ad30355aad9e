"""Layer stacks on a real MI355X (HIP kernels) against the CPU oracle + autograd, fp32.
Tolerances: forward 1e-5 relative L2, gradients 1e-3 relative L2 per tensor."""
import pytest
import torch

from oracle import mvae_oracle as O
from mmdyn_hip import layers
from mmdyn_hip.models.shapes import state_dict_shapes
from mmdyn_hip.utils.seeded_init import seeded_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def sub(d, pre, dev=None):
    out = {k[len(pre) + 1:]: v for k, v in d.items() if k.startswith(pre + ".")}
    if dev:
        out = {k: v.detach().to(dev) for k, v in out.items()}
    return out


def state():
    return O.split_state(seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("B,G", [(4, 1), (6, 2), (32, 1)])
def test_encoder_trunk(B, G):
    prm, buf = state()
    x = torch.rand(B, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    P, Bf = sub(prm, "visual_encoder", DEV), sub(buf, "visual_encoder", DEV)
    h, ctx = layers.encoder_trunk_forward(P, Bf, x.to(DEV), G=G)
    buf_o = {k: v.clone() for k, v in buf.items()}
    Bg = B // G
    h_ref = torch.cat([O.image_encoder_trunk(x[g * Bg:(g + 1) * Bg], prm, "visual_encoder", buf_o) for g in range(G)])
    assert rel(h, h_ref) < 2e-5
    for k in Bf:
        torch.testing.assert_close(Bf[k].cpu().double(), buf_o["visual_encoder." + k].double(), rtol=2e-5, atol=2e-6)
    dh = torch.randn(B, 512, generator=torch.Generator().manual_seed(2))
    grads = {k: torch.zeros_like(P[k]) for k in layers.ENC_KEYS}
    layers.encoder_trunk_backward(P, ctx, dh.to(DEV), grads)
    (h_ref * dh).sum().backward()
    for k in layers.ENC_KEYS:
        assert rel(grads[k], prm["visual_encoder." + k].grad) < 1e-3, k


@pytest.mark.parametrize("B,G", [(4, 1), (8, 4), (32, 2)])
def test_decoder(B, G):
    prm, buf = state()
    z = torch.randn(B, 256, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    P, Bf = sub(prm, "tactile_decoder", DEV), sub(buf, "tactile_decoder", DEV)
    out, ctx = layers.decoder_forward(P, Bf, z.detach().to(DEV), G=G)
    buf_o = {k: v.clone() for k, v in buf.items()}
    Bg = B // G
    ref = torch.cat([O.image_decoder(z[g * Bg:(g + 1) * Bg], prm, "tactile_decoder", buf_o) for g in range(G)])
    assert rel(out, ref) < 2e-5
    for k in Bf:
        torch.testing.assert_close(Bf[k].cpu().double(), buf_o["tactile_decoder." + k].double(), rtol=2e-5, atol=2e-6)
    dl = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    grads = {k: torch.zeros_like(P[k]) for k in layers.DEC_KEYS}
    dz = layers.decoder_backward(P, ctx, dl.to(DEV), grads)
    (ref * dl).sum().backward()
    assert rel(dz, z.grad) < 1e-3
    for k in layers.DEC_KEYS:
        assert rel(grads[k], prm["tactile_decoder." + k].grad) < 1e-3, k


def test_heads_and_pose_mlps():
    prm, _ = state()
    hd = torch.randn(70, 512, generator=torch.Generator().manual_seed(5)).requires_grad_(True)
    P = sub(prm, "visual_encoder", DEV)
    out, c = layers.heads_forward(P, hd.detach().to(DEV))
    mu, lv = O.encoder_heads(hd, prm, "visual_encoder")
    assert rel(out[:, :256], mu) < 1e-5 and rel(out[:, 256:], lv) < 1e-5
    dout = torch.randn(70, 512, generator=torch.Generator().manual_seed(6))
    grads = {k: torch.zeros_like(P[k]) for k in layers.HEAD_KEYS}
    dx = layers.heads_backward(c, dout.to(DEV), grads)
    ((mu * dout[:, :256]).sum() + (lv * dout[:, 256:]).sum()).backward()
    assert rel(dx, hd.grad) < 1e-4
    for k in layers.HEAD_KEYS:
        assert rel(grads[k], prm["visual_encoder." + k].grad) < 1e-4, k

    pose = torch.rand(70, 7, generator=torch.Generator().manual_seed(7))
    Pe = sub(prm, "pose_encoder", DEV)
    h2, c = layers.pose_encoder_trunk_forward(Pe, pose.to(DEV))
    h1 = torch.relu(torch.nn.functional.linear(pose, prm["pose_encoder.fc_net.0.weight"], prm["pose_encoder.fc_net.0.bias"]))
    ref = torch.nn.functional.linear(h1, prm["pose_encoder.fc_net.2.weight"], prm["pose_encoder.fc_net.2.bias"])
    assert rel(h2, ref) < 1e-5
    d = torch.randn(70, 512, generator=torch.Generator().manual_seed(8))
    grads = {k: torch.zeros_like(Pe[k]) for k in layers.POSE_ENC_KEYS}
    layers.pose_encoder_trunk_backward(Pe, c, d.to(DEV), grads)
    (ref * d).sum().backward()
    for k in layers.POSE_ENC_KEYS:
        assert rel(grads[k], prm["pose_encoder." + k].grad) < 1e-4, k

    z = torch.randn(70, 256, generator=torch.Generator().manual_seed(9)).requires_grad_(True)
    Pp = sub(prm, "pose_decoder", DEV)
    out, c = layers.pose_decoder_forward(Pp, z.detach().to(DEV))
    ref = O.pose_decoder(z, prm)
    assert rel(out, ref) < 1e-5
    d = torch.randn(70, 7, generator=torch.Generator().manual_seed(10))
    grads = {k: torch.zeros_like(Pp[k]) for k in layers.POSE_DEC_KEYS}
    dz = layers.pose_decoder_backward(Pp, c, d.to(DEV), grads)
    (ref * d).sum().backward()
    assert rel(dz, z.grad) < 1e-4
    for k in layers.POSE_DEC_KEYS:
        assert rel(grads[k], prm["pose_decoder." + k].grad) < 1e-4, k
