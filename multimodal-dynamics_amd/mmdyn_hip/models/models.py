"""Model registry: the ``--model-name`` plug-in surface of /root/reference/mmdyn/pytorch/models/models.py:13-25."""
from .. import config
from torch import nn

from .. import layers
from . import functional as Fn
from .vae import (VAE, MVAE, Swish, Conv2dParams, BatchNorm2dParams, LinearParams, _Marker, _noise_of,  # noqa: F401
                  _condition)
from .shapes import FEAT, HID


def count_parameters(model):
    """Number of trainable scalars (printed by the problem classes, like the reference does)."""
    return sum(int(p.numel()) for p in model.parameters() if p.requires_grad)


def _family(model_name):
    """Registry name -> model family.  The reference dispatches on substrings of the name; the order matters
    ('cnn-mvae' contains 'vae' too)."""
    for key in ("mvae", "vae", "regressor"):
        if key in model_name:
            return key
    return None


def setup_model(model_name, cross_modal=False, **kwargs):
    """``--model-name`` plug-in point.  Behaviour kept from the reference: an unknown name fails the registry assertion;
    a multimodal VAE is only built for cross-modal input (a 'cnn-mvae' asked for a single modality falls through to the
    plain VAE, which then refuses cross-modal input); anything else terminates the process with a message."""
    assert (model_name in config.MODELS), "Model is not implement yet"
    family = _family(model_name)
    if family == "mvae" and cross_modal:
        return MVAE(**kwargs)
    if family in ("mvae", "vae"):
        assert not cross_modal, "VAE does not work with cross modal inputs."
        return VAE(**kwargs)
    if family == "regressor":
        return Regressor(**kwargs)
    exit("The model and modality combination is not valid.")


class Regressor(nn.Module):
    """Pose-regression baseline (models.py:28-77): the image-encoder conv/fc trunk, dropout, then a three-layer
    ReLU MLP to ``out_dim``; ``conditional`` concatenates the (real-valued) condition to the 512 features.
    Same state_dict keys as the reference (conv_net.*, fc_net.0.*, out_net.{0,2,4}.*)."""

    def __init__(self, out_dim=7, conditional=False, num_classes=None):
        super().__init__()
        self.conditional = conditional
        self.num_classes = num_classes
        cnn_features_comp = HID + self.conditional * self.num_classes      # TypeError for None, like the reference
        self.conv_net = nn.Sequential(
            Conv2dParams(3, 32, 4, 2, 1), Swish(),
            Conv2dParams(32, 64, 4, 2, 1), BatchNorm2dParams(64), Swish(),
            Conv2dParams(64, 128, 4, 2, 1), BatchNorm2dParams(128), Swish(),
            Conv2dParams(128, 256, 4, 1, 0), BatchNorm2dParams(256), Swish())
        self.fc_net = nn.Sequential(LinearParams(FEAT, HID), Swish(), _Marker("Dropout(p=0.1)"))
        self.out_net = nn.Sequential(LinearParams(cnn_features_comp, 256), _Marker("ReLU()"), LinearParams(256, 256),
                                     _Marker("ReLU()"), LinearParams(256, out_dim))
        self.noise = None

    def bn_buffers(self):
        return {f"conv_net.{i}.{n}": getattr(self.conv_net[i], n) for i in (3, 6, 9)
                for n in ("running_mean", "running_var", "num_batches_tracked")}

    def param_keys(self):
        return layers.ENC_KEYS

    def forward(self, x, c=None):
        import torch
        sd = dict(self.named_parameters())
        if self.training:
            h = Fn.ImageEncoderTrunkFn.apply(x, self, *[sd[k] for k in layers.ENC_KEYS])
            h = Fn.DropoutFn.apply(h, _noise_of(self).keep_mask(tuple(h.shape), h.device))
        else:                       # eval: running-estimate BatchNorm, no dropout, forward only
            with torch.no_grad():
                P = {k: sd[k].detach() for k in layers.ENC_KEYS}
                h = layers.run(layers.encoder_trunk_forward_steps(P, self.bn_buffers(), x.detach().contiguous(),
                                                                  training=False))[0]
        if self.conditional:
            h = torch.cat((h, _condition(c, True)), dim=-1)
        return Fn.PoseDecoderFn.apply(h, *[sd[f"out_net.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias")])
