"""Deterministic, name-keyed weight filler.

The reference relies on PyTorch's default (unseeded) initialisers, so a 56 MB
``state_dict`` would have to be committed to pin parity.  Instead every tensor
is regenerated from ``(seed, key name, shape)`` alone: the golden-vector script
loads the result into the *reference* modules through ``load_state_dict`` and
the tests load the very same values into this package's modules, so only the
small inputs/outputs have to be stored under ``tests/golden/``.

State-dict key names / shapes follow the reference's modules
(/root/reference/mmdyn/pytorch/models/vae.py:193-216, 261-283).
"""
import zlib

import torch


def _gen(seed, name):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
    return g


def seeded_tensor(name, shape, seed=0, dtype=torch.float32):
    """Value for one state-dict entry, a pure function of (seed, name, shape)."""
    shape = tuple(shape)
    g = _gen(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "running_var":
        return torch.ones(shape, dtype=dtype)
    if len(shape) == 1:
        # BatchNorm affine or a bias vector.  BN weights sit next to running_* keys,
        # but a name alone cannot tell; both get a non-trivial, well-scaled fill.
        u = torch.rand(shape, generator=g, dtype=torch.float64)
        if leaf == "weight":
            return (0.6 + 0.8 * u).to(dtype)          # BN gamma in [0.6, 1.4)
        return (0.2 * (u - 0.5)).to(dtype)            # beta / bias in [-0.1, 0.1)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if len(shape) == 4:
        # ConvTranspose2d weights are [Cin, Cout, kh, kw]: fan-in is dim0 * k*k there,
        # but the bound only sets a scale, so one rule serves both.
        fan_in = max(shape[0], shape[1]) * shape[2] * shape[3]
    bound = (3.0 / fan_in) ** 0.5
    u = torch.rand(shape, generator=g, dtype=torch.float64)
    return ((2.0 * u - 1.0) * bound).to(dtype)


def seeded_state_dict(template, seed=0):
    """Build a full state dict from ``{name: tensor-or-shape}``."""
    out = {}
    for name, v in template.items():
        shape = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        out[name] = seeded_tensor(name, shape, seed)
    return out


def seeded_batch(batch, seed=1234, with_pose=True, size=64):
    """Synthetic visuotactile(+pose) batch, SURVEY.md section 8(d): U[0,1) images and poses.

    Returns (inputs, targets): each ``[visual, tactile, pose]`` (pose omitted when
    ``with_pose`` is False).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def draw():
        v = torch.rand(batch, 3, size, size, generator=g)
        t = torch.rand(batch, 3, size, size, generator=g)
        out = [v, t]
        if with_pose:
            out.append(torch.rand(batch, 7, generator=g))
        return out

    return draw(), draw()


def seeded_noise(batch, latent, n_eps, n_masks, seed=4321, hidden=512, p=0.1):
    """Injected randomness for parity runs: ``n_eps`` N(0,1) draws [batch, latent] and
    ``n_masks`` Bernoulli(1-p) keep-masks [batch, hidden] (uint8)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    eps = [torch.randn(batch, latent, generator=g) for _ in range(n_eps)]
    masks = [(torch.rand(batch, hidden, generator=g) >= p).to(torch.uint8) for _ in range(n_masks)]
    return eps, masks


def seeded_running_stats(state, seed=7):
    """Non-trivial BatchNorm running estimates for eval-mode vectors: running_mean ~ U(-0.2, 0.2), running_var ~
    U(0.5, 1.5), num_batches_tracked = 3 -- a pure function of (seed, key name), applied in place to a state dict."""
    for name, v in state.items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_mean":
            state[name] = (0.4 * torch.rand(v.shape, generator=_gen(seed, name), dtype=torch.float64) - 0.2).to(v.dtype)
        elif leaf == "running_var":
            state[name] = (0.5 + torch.rand(v.shape, generator=_gen(seed, name), dtype=torch.float64)).to(v.dtype)
        elif leaf == "num_batches_tracked":
            state[name] = torch.full_like(v, 3)
    return state
