"""VAE / MVAE modules with the reference's public surface, computed by the HIP kernel library.

Drop-in for /root/reference/mmdyn/pytorch/models/vae.py: same class names, constructor keyword
arguments, ``forward`` / ``inference`` signatures and return tuples, same ``state_dict`` keys and tensor
layouts, so checkpoints written by either implementation load into the other.
The sub-modules only *hold* parameters under the reference's names; all arithmetic is issued by
``Encoder.forward`` / ``Decoder.forward`` / ``MVAE.forward`` as fused kernel sequences
(:mod:`mmdyn_hip.layers`), never by ``torch.nn.functional``.

Randomness: the reference draws the reparametrisation noise with ``torch.randn`` on the CPU and the
dropout masks on the device, both unseeded (vae.py:58, 213).  Here both come from ``model.noise`` (a
:class:`NoiseSource`): by default a counter-based on-device Philox stream, or injected tensors for
parity runs.
"""
import math

import torch
import torch.nn as nn

from .. import config, ops
from . import functional as Fn
from .. import layers
from .shapes import FEAT, HID, DROPOUT_P, IMG_SIZES, extra_stages, encoder_channels, decoder_channels


# ------------------------------------------------------------------------------------------------
# parameter holders (named and shaped like the reference's nn layers)
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only holds parameters; the enclosing Encoder/Decoder "
                           "issues the fused HIP kernels")


class Conv2dParams(_Holder):
    """weight [Cout, Cin, k, k] (transposed=False) or [Cin, Cout, k, k] (transposed=True), no bias."""

    def __init__(self, cin, cout, k, stride, pad, transposed=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad, self.transposed = cin, cout, k, stride, pad, transposed
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def extra_repr(self):
        kind = "ConvTranspose2d" if self.transposed else "Conv2d"
        return f"{kind}({self.cin}, {self.cout}, kernel_size={self.k}, stride={self.stride}, padding={self.pad}, bias=False)"


class BatchNorm2dParams(_Holder):
    def __init__(self, c):
        super().__init__()
        self.num_features = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f"BatchNorm2d({self.num_features}, eps=1e-05, momentum=0.1, train-mode batch statistics)"


class LinearParams(_Holder):
    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(fin)
        nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f"Linear(in_features={self.in_features}, out_features={self.out_features}, bias=True)"


class Swish(nn.Module):
    """x * sigmoid(x) (vae.py:331-334).  Inside Encoder/Decoder it is fused into the neighbouring kernels;
    called on its own it runs the element-wise kernel."""

    def forward(self, x):
        return Fn.SwishFn.apply(x)


class _Marker(_Holder):
    """Parameter-free placeholder keeping the reference's Sequential indices (ReLU / Identity / Dropout)."""

    def __init__(self, text):
        super().__init__()
        self.text = text

    def extra_repr(self):
        return self.text


def mlp(sizes, activation="ReLU", output_activation="Identity"):
    """Holder Sequential with the reference's indexing: Linear at even positions (vae.py:14-19)."""
    mods = []
    for j in range(len(sizes) - 1):
        mods += [LinearParams(sizes[j], sizes[j + 1]), _Marker(activation if j < len(sizes) - 2 else output_activation)]
    return nn.Sequential(*mods)


# ------------------------------------------------------------------------------------------------
# noise sources
# ------------------------------------------------------------------------------------------------
class NoiseSource:
    """On-device Philox stream (mmdyn_random_normal / mmdyn_random_masks).

    The stream position is ``offset`` (host, advanced per draw) + ``base`` (a device counter).  Eager code only
    moves ``offset``.  A captured HIP graph bakes its host offsets in; :meth:`commit` then enqueues a device-side
    bump of ``base`` by everything drawn since the last commit, so every replay draws fresh numbers."""

    def __init__(self, seed=0):
        self.seed, self.offset, self.base, self._mark = int(seed), 0, None, 0

    def _base(self, device):
        if self.base is None or self.base.device != device:
            self.base = torch.zeros(1, dtype=torch.int64, device=device)
        return self.base

    def eps(self, shape, device):
        out = torch.empty(shape, device=device, dtype=torch.float32)
        ops.B.random_normal(out, self.seed, self.offset, self._base(device))
        self.offset += (out.numel() + 3) // 4
        return out

    def keep_mask(self, shape, device):
        out = torch.empty(shape, device=device, dtype=torch.uint8)
        ops.B.random_masks(out, DROPOUT_P, self.seed ^ 0x5DEECE66D, self.offset, self._base(device))
        self.offset += (out.numel() + 3) // 4
        return out

    def eps_block(self, n, B, L, device):
        """n independent [B, L] draws as one [n, B, L] tensor, one launch."""
        return self.eps((n, B, L), device)

    def mask_block(self, n, B, H, device):
        return self.keep_mask((n, B, H), device)

    def commit(self):
        """Move the draws made since the previous commit from the host offset into the device counter."""
        if self.base is not None and self.offset > self._mark:
            ops.B.counter_add(self.base, self.offset - self._mark)
            self.offset = self._mark


class InjectedNoise:
    """Replays given tensors in call order (parity tests: the same draws the oracle consumes)."""

    def __init__(self, eps_list, mask_list):
        self._eps, self._masks = list(eps_list), list(mask_list)

    def eps(self, shape, device):
        e = self._eps.pop(0)
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return e.to(device=device, dtype=torch.float32).contiguous()

    def keep_mask(self, shape, device):
        m = self._masks.pop(0)
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m.to(device=device, dtype=torch.uint8).contiguous()

    def eps_block(self, n, B, L, device):
        return torch.stack([self.eps((B, L), device) for _ in range(n)])

    def mask_block(self, n, B, H, device):
        return torch.stack([self.keep_mask((B, H), device) for _ in range(n)])

    def commit(self):
        pass


def _noise_of(module):
    n = getattr(module, "noise", None)
    if n is None:
        n = module.noise = NoiseSource(0)
    return n


# ------------------------------------------------------------------------------------------------
# reference-shaped modules
# ------------------------------------------------------------------------------------------------
class Autoencoder(nn.Module):
    """Base class (vae.py:26-67)."""

    def __init__(self, input_dim=784, encoder_hid=[256, 256], latent_size=8, decoder_hid=[256, 256],
                 condition_dim=None, architecture='mlp', conditional=False, categorical_conditions=False):
        super().__init__()
        assert type(encoder_hid) == list
        assert type(latent_size) == int
        assert type(decoder_hid) == list
        assert architecture in config.ARCHITECTURES
        self.latent_size = latent_size
        self.input_dim = input_dim
        self.condition_dim = condition_dim
        self.architecture = architecture
        self.conditional = conditional
        self.categorical_conditions = categorical_conditions
        self.noise = None

    def reparametrize(self, means, log_var):
        eps = _noise_of(self).eps((means.size(0), self.latent_size), means.device)
        return Fn.ReparamFn.apply(means, log_var, eps)

    def forward(self, x):
        raise NotImplementedError

    def inference(self, n=1):
        raise NotImplementedError


def _check_supported(architecture, conditional, categorical_conditions=False):
    if conditional and categorical_conditions:
        raise NotImplementedError("mmdyn_hip: categorical (one-hot) conditions are not built; the seq/dyn modeling "
                                  "problems use real-valued shock conditions (problems.py:675-681)")


def cnn_image_size(input_dim):
    """Side of the square input image of a cnn Encoder / Decoder.  The reference passes ``input_dim = 64 * 64``
    (problems.py:370) and never reads it in the cnn branch (its stack is fixed at 64 x 64); here it selects the stack:
    4096 (or the constructor default 784) -> the reference's, 128*128 / 256*256 -> the extended stacks of
    models/shapes.py."""
    for s in IMG_SIZES:
        if input_dim == s * s:
            return s
    return 64


def _condition(c, conditional):
    """The reference's treatment of the condition tensor (vae.py:231-237): 1-D -> column, cast to float."""
    if not conditional:
        return None
    if c is None:
        raise ValueError("conditional model called without a condition")
    if c.dim() == 1:
        c = c.unsqueeze(1)
    return c.to(torch.float32).contiguous()


class Encoder(nn.Module):
    """vae.py:179-242.  cnn: conv_net/fc_net/linear_means/linear_log_var; mlp: fc_net + the two heads."""

    def __init__(self, input_dim=784, layer_sizes=[256, 256], latent_size=8, architecture='mlp', conditional=False,
                 categorical_conditions=False, condition_dim=None, **kwargs):
        super().__init__()
        self.architecture = architecture
        self.conditional = conditional
        self.categorical_conditions = categorical_conditions
        self.condition_dim = condition_dim
        self.latent_size = latent_size
        if categorical_conditions:
            assert condition_dim is not None, "Num conditions is not specified for categorical conditions."
        _check_supported(architecture, conditional, categorical_conditions)
        cond_w = (condition_dim or 0) if conditional else 0
        if architecture == 'cnn':
            self.image_size = cnn_image_size(input_dim)
            self.extra = extra_stages(self.image_size)
            chans = encoder_channels(self.extra)
            mods = [Conv2dParams(3, 32, 4, 2, 1), Swish()]
            for j, (cin, cout) in enumerate(chans):
                last = j == len(chans) - 1
                mods += [Conv2dParams(cin, cout, 4, 1 if last else 2, 0 if last else 1), BatchNorm2dParams(cout), Swish()]
            self.conv_net = nn.Sequential(*mods)
            self.fc_net = nn.Sequential(LinearParams(FEAT, HID), Swish(), _Marker("Dropout(p=0.1)"))
            self.linear_means = LinearParams(HID + cond_w, latent_size)
            self.linear_log_var = LinearParams(HID + cond_w, latent_size)
        else:
            if conditional:
                raise NotImplementedError("mmdyn_hip: a conditional mlp Encoder cannot run in the reference either (its "
                                          "heads are sized without the condition, vae.py:220-222 vs 237)")
            layer_sizes = [input_dim] + layer_sizes
            self.fc_net = mlp(layer_sizes)
            self.linear_means = LinearParams(layer_sizes[-1], latent_size)
            self.linear_log_var = LinearParams(layer_sizes[-1], latent_size)
        self.noise = None

    def bn_buffers(self):
        return {f"conv_net.{i}.{n}": getattr(self.conv_net[i], n) for i in range(3, 3 * (4 + self.extra), 3)
                for n in ("running_mean", "running_var", "num_batches_tracked")}

    def param_keys(self):
        return layers.enc_keys(self.extra)

    def trunk(self, x):
        """Everything before the dropout: [B,512] features.  ``model.eval()``: BatchNorm uses the running estimates
        (forward only: the result carries no autograd graph)."""
        if self.architecture == 'cnn':
            sd = dict(self.named_parameters())
            if not self.training:
                with torch.no_grad():
                    P = {k: sd[k].detach() for k in self.param_keys()}
                    return layers.run(layers.encoder_trunk_forward_steps(P, self.bn_buffers(), x.detach().contiguous(),
                                                                         training=False))[0]
            return Fn.ImageEncoderTrunkFn.apply(x, self, *[sd[k] for k in self.param_keys()])
        lin = [m for m in self.fc_net if isinstance(m, LinearParams)]
        return Fn.MLPFn.apply(x, *[p for m in lin for p in (m.weight, m.bias)])

    def heads(self, h, c=None):
        out = Fn.HeadsFn.apply(h, self.linear_means.weight, self.linear_means.bias, self.linear_log_var.weight,
                               self.linear_log_var.bias, _condition(c, self.conditional))
        return out

    def forward_fused(self, x, noise, c=None):
        """Returns the fused heads output [B, 2L] (means | log_vars)."""
        h = self.trunk(x)
        if self.architecture == 'cnn' and self.training:        # nn.Dropout is the identity in eval mode
            h = Fn.DropoutFn.apply(h, noise.keep_mask(tuple(h.shape), h.device))
        return self.heads(h, c)

    def forward(self, x, c=None):
        out = self.forward_fused(x, _noise_of(self), c)
        L = self.latent_size
        return out[:, :L], out[:, L:]


class Decoder(nn.Module):
    """vae.py:245-301.  cnn: upsample + hallucinate (returns LOGITS, no sigmoid); mlp: deconv_net."""

    def __init__(self, output_dim=784, layer_sizes=[256, 256], latent_size=2, architecture='mlp', conditional=False,
                 categorical_conditions=False, condition_dim=None, **kwargs):
        super().__init__()
        self.architecture = architecture
        self.conditional = conditional
        self.categorical_conditions = categorical_conditions
        self.condition_dim = condition_dim
        if categorical_conditions:
            assert condition_dim is not None, "Num conditions is not specified for categorical conditions."
        _check_supported(architecture, conditional, categorical_conditions)
        cond_w = (condition_dim or 0) if conditional else 0
        self._cond = None
        if architecture == 'cnn':
            if latent_size % 32:
                raise NotImplementedError("mmdyn_hip: latent_size must be a multiple of 32 (MFMA K-step)")
            self.upsample = nn.Sequential(LinearParams(latent_size + cond_w, FEAT), Swish())
            # (the reference's Decoder receives the flat pixel count as ``input_dim`` through **kwargs and ignores it)
            self.image_size = cnn_image_size(kwargs.get("input_dim", output_dim))
            self.extra = extra_stages(self.image_size)
            mods = []
            for j, (cin, cout) in enumerate(decoder_channels(self.extra)):
                mods += [Conv2dParams(cin, cout, 4, 1 if j == 0 else 2, 0 if j == 0 else 1, True), BatchNorm2dParams(cout),
                         Swish()]
            self.hallucinate = nn.Sequential(*mods, Conv2dParams(32, 3, 4, 2, 1, True))
        else:
            layer_sizes = [latent_size + cond_w] + layer_sizes + [output_dim]
            self.deconv_net = mlp(layer_sizes)

    def bn_buffers(self):
        return {f"hallucinate.{i}.{n}": getattr(self.hallucinate[i], n) for i in range(1, 3 * (3 + self.extra), 3)
                for n in ("running_mean", "running_var", "num_batches_tracked")}

    def param_keys(self):
        return layers.dec_keys(self.extra)

    def forward(self, z, c=None):
        sd = dict(self.named_parameters())
        if self.architecture == 'cnn' and not self.training:   # eval: running-estimate BatchNorm, forward only
            with torch.no_grad():
                P = {k: sd[k].detach() for k in self.param_keys()}
                return layers.run(layers.decoder_forward_steps(P, self.bn_buffers(), z.detach().contiguous(),
                                                               cond=_condition(c, self.conditional), training=False))[0]
        if self.architecture == 'cnn':
            self._cond = _condition(c, self.conditional)      # read by ImageDecoderFn.forward through `holder`
            try:
                return Fn.ImageDecoderFn.apply(z, self, *[sd[k] for k in self.param_keys()])
            finally:
                self._cond = None
        cc = _condition(c, self.conditional)
        if cc is not None:
            z = torch.cat((z, cc), dim=-1)                   # vae.py:286-291 (a copy; no arithmetic)
        lin = [m for m in self.deconv_net if isinstance(m, LinearParams)]
        return Fn.MLPFn.apply(z, *[p for m in lin for p in (m.weight, m.bias)])


class ProductOfExperts(nn.Module):
    """Product of independent Gaussian experts (vae.py:304-318): mu, logvar are [M, B, D]."""

    def forward(self, mu, logvar, eps=1e-8):
        if eps != 1e-8:
            raise NotImplementedError("mmdyn_hip: the PoE kernel has the reference's eps=1e-8 built in")
        return Fn.ProductOfExpertsFn.apply(mu, logvar)


def prior_expert(size, device=torch.device('cpu')):
    """Universal N(0, 1) prior expert (vae.py:321-328): zero mean, zero log-variance."""
    return torch.zeros(size, device=device), torch.zeros(size, device=device)


class VAE(Autoencoder):
    """Single-modality VAE (vae.py:70-98)."""

    def __init__(self, use_pose=False, **kwargs):
        super().__init__(**kwargs)
        self.encoder = Encoder(**kwargs)
        self.decoder = Decoder(**kwargs)

    def forward(self, x, c=None):
        noise = _noise_of(self)
        if x.dim() > 2 and self.architecture == 'mlp':
            x = x.view(-1, self.input_dim)
        out = self.encoder.forward_fused(x, noise, c)
        L = self.latent_size
        means, log_var = out[:, :L], out[:, L:]
        eps = noise.eps((x.size(0), L), x.device)
        z = Fn.ReparamFn.apply(means, log_var, eps)
        return self.decoder(z, c), means, log_var

    def inference(self, n=1, c=None):
        dev = next(self.parameters()).device
        z = _noise_of(self).eps((n, self.latent_size), dev)
        return self.decoder(z, c)


class MVAE(Autoencoder):
    """Multimodal VAE with a product-of-experts posterior (vae.py:101-176)."""

    def __init__(self, use_pose=False, **kwargs):
        super().__init__(**kwargs)
        assert kwargs['architecture'] != 'mlp', "MVAE is not implemented with MLP"
        self._use_pose = use_pose
        self.visual_encoder = Encoder(**kwargs)
        self.visual_decoder = Decoder(**kwargs)
        self.tactile_encoder = Encoder(**kwargs)
        self.tactile_decoder = Decoder(**kwargs)
        if self._use_pose:
            self.pose_encoder = Encoder(input_dim=7, layer_sizes=[512, 512], latent_size=kwargs["latent_size"],
                                        condition_dim=0, architecture="mlp")
            self.pose_decoder = Decoder(output_dim=7, layer_sizes=[512, 512], latent_size=kwargs["latent_size"],
                                        condition_dim=0, architecture="mlp")
        self.experts = ProductOfExperts()

    def forward(self, x, pose=None, condition=None):
        assert isinstance(x, list) or isinstance(x, tuple)
        visual, tactile = x
        ref = visual if visual is not None else (tactile if tactile is not None else pose)
        batch_size = ref.size(0)
        noise = _noise_of(self)
        L = self.latent_size
        heads = [None, None, None]
        if visual is not None:
            heads[0] = self.visual_encoder.forward_fused(visual, noise, condition)
        if tactile is not None:
            heads[1] = self.tactile_encoder.forward_fused(tactile, noise, condition)
        if pose is not None and self._use_pose:
            heads[2] = self.pose_encoder.forward_fused(pose, noise, condition)
        eps = noise.eps((batch_size, L), ref.device)
        means, log_var, z = Fn.PoEReparamFn.apply(eps, L, *heads)
        visual_recon = self.visual_decoder(z, c=condition)
        tactile_recon = self.tactile_decoder(z, c=condition)
        pose_recon = self.pose_decoder(z, c=condition) if self._use_pose else None
        return visual_recon, tactile_recon, pose_recon, means, log_var

    def inference(self, n=1, c=None):
        dev = next(self.parameters()).device
        z = _noise_of(self).eps((n, self.latent_size), dev)
        return self.visual_decoder(z, c), self.tactile_decoder(z, c)
