"""Architecture constants and the state_dict wire format of the cnn VAE / MVAE.

Key names and tensor layouts are the reference's checkpoint format
(/root/reference/mmdyn/pytorch/models/vae.py:193-216 encoder, :261-283 decoder,
:117-123 pose MLPs): Conv2d weights ``[Cout, Cin, 4, 4]``,
ConvTranspose2d weights ``[Cin, Cout, 4, 4]``, Linear weights ``[out, in]``.
"""
from collections import OrderedDict

IMG_CH = 3
IMG_SIZE = 64                        # the reference's only input size (vae.py:195, 264, 295; problems.py:111-112)
# Larger inputs (BASELINE configs[3] 128x128, configs[4] 256x256 "deeper conv stack") are EXTENSIONS with no reference
# architecture: the reference's FC layer is fixed at 256*5*5.  They are defined here as the reference's stack with
# log2(size / 64) extra stride-2 stages of 32 -> 32 channels right behind the first encoder convolution / in front of the
# last decoder transposed convolution, so the trunk still ends at 256 x 5 x 5 and every other layer is the reference's:
#   encoder conv_net : Conv(3,32,4,2,1) Swish | extra x [Conv(32,32,4,2,1) BN Swish] | Conv(32,64) BN Swish | Conv(64,128) BN
#                      Swish | Conv(128,256,4,1,0) BN Swish            (Sequential indices 0, 2, 5, 8, 11, ...)
#   decoder hallucinate: ConvT(256,128,4,1,0) BN Swish | ConvT(128,64) BN Swish | ConvT(64,32) BN Swish |
#                      extra x [ConvT(32,32,4,2,1) BN Swish] | ConvT(32,3,4,2,1)   (indices 0, 3, 6, 9, 12, ...)
IMG_SIZES = (64, 128, 256)
ENC_CH = (32, 64, 128, 256)          # conv_net output channels
DEC_CH = (256, 128, 64, 32)          # hallucinate input channels
FEAT_HW = 5                          # trunk ends at 256 x 5 x 5
FEAT = 256 * FEAT_HW * FEAT_HW       # 6400
HID = 512
POSE_DIM = 7
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
DROPOUT_P = 0.1
POE_EPS = 1e-8


def _bn(d, pre, c):
    d[pre + ".weight"] = (c,)
    d[pre + ".bias"] = (c,)
    d[pre + ".running_mean"] = (c,)
    d[pre + ".running_var"] = (c,)
    d[pre + ".num_batches_tracked"] = ()


def extra_stages(size):
    """Number of additional stride-2 stages for an input of ``size`` x ``size`` pixels (0 for the reference's 64)."""
    if size not in IMG_SIZES:
        raise ValueError(f"image size {size} not supported (one of {IMG_SIZES})")
    return IMG_SIZES.index(size)


def encoder_channels(extra=0):
    """(Cin, Cout) of the convolutions that follow the first one."""
    return [(32, 32)] * extra + [(32, 64), (64, 128), (128, 256)]


def decoder_channels(extra=0):
    """(Cin, Cout) of the transposed convolutions in front of the last one."""
    return [(256, 128), (128, 64), (64, 32)] + [(32, 32)] * extra


def image_encoder_shapes(pre, latent=256, cond=0, size=IMG_SIZE):
    d = OrderedDict()
    d[pre + ".conv_net.0.weight"] = (32, IMG_CH, 4, 4)
    for j, (cin, cout) in enumerate(encoder_channels(extra_stages(size))):
        d[pre + f".conv_net.{2 + 3 * j}.weight"] = (cout, cin, 4, 4)
        _bn(d, pre + f".conv_net.{3 + 3 * j}", cout)
    d[pre + ".fc_net.0.weight"] = (HID, FEAT)
    d[pre + ".fc_net.0.bias"] = (HID,)
    d[pre + ".linear_means.weight"] = (latent, HID + cond)
    d[pre + ".linear_means.bias"] = (latent,)
    d[pre + ".linear_log_var.weight"] = (latent, HID + cond)
    d[pre + ".linear_log_var.bias"] = (latent,)
    return d


def image_decoder_shapes(pre, latent=256, cond=0, size=IMG_SIZE):
    d = OrderedDict()
    d[pre + ".upsample.0.weight"] = (FEAT, latent + cond)
    d[pre + ".upsample.0.bias"] = (FEAT,)
    chans = decoder_channels(extra_stages(size))
    for j, (cin, cout) in enumerate(chans):
        d[pre + f".hallucinate.{3 * j}.weight"] = (cin, cout, 4, 4)
        _bn(d, pre + f".hallucinate.{3 * j + 1}", cout)
    d[pre + f".hallucinate.{3 * len(chans)}.weight"] = (32, IMG_CH, 4, 4)
    return d


def pose_encoder_shapes(pre="pose_encoder", latent=256):
    d = OrderedDict()
    d[pre + ".fc_net.0.weight"] = (HID, POSE_DIM)
    d[pre + ".fc_net.0.bias"] = (HID,)
    d[pre + ".fc_net.2.weight"] = (HID, HID)
    d[pre + ".fc_net.2.bias"] = (HID,)
    d[pre + ".linear_means.weight"] = (latent, HID)
    d[pre + ".linear_means.bias"] = (latent,)
    d[pre + ".linear_log_var.weight"] = (latent, HID)
    d[pre + ".linear_log_var.bias"] = (latent,)
    return d


def pose_decoder_shapes(pre="pose_decoder", latent=256):
    d = OrderedDict()
    d[pre + ".deconv_net.0.weight"] = (HID, latent)
    d[pre + ".deconv_net.0.bias"] = (HID,)
    d[pre + ".deconv_net.2.weight"] = (HID, HID)
    d[pre + ".deconv_net.2.bias"] = (HID,)
    d[pre + ".deconv_net.4.weight"] = (POSE_DIM, HID)
    d[pre + ".deconv_net.4.bias"] = (POSE_DIM,)
    return d


def regressor_shapes(out_dim=7, cond=0, size=IMG_SIZE):
    """Regressor baseline (models.py:28-64)."""
    d = image_encoder_shapes("r", size=size)
    for k in [k for k in d if "linear_" in k]:
        del d[k]
    d = OrderedDict((k[2:], v) for k, v in d.items())
    for i, (fin, fout) in zip((0, 2, 4), ((HID + cond, 256), (256, 256), (256, out_dim))):
        d[f"out_net.{i}.weight"] = (fout, fin)
        d[f"out_net.{i}.bias"] = (fout,)
    return d


def mlp_vae_shapes(input_dim=784, hidden=(256, 256), latent=32, output_dim=784):
    """mlp-vae: Encoder fc_net = mlp([input_dim] + hidden) + heads, Decoder deconv_net = mlp([latent] + hidden +
    [output_dim]) (vae.py:218-222, 281-283); Linear layers sit at even Sequential indices."""
    d = OrderedDict()
    sizes = [input_dim] + list(hidden)
    for j in range(len(sizes) - 1):
        d[f"encoder.fc_net.{2 * j}.weight"] = (sizes[j + 1], sizes[j])
        d[f"encoder.fc_net.{2 * j}.bias"] = (sizes[j + 1],)
    for h in ("linear_means", "linear_log_var"):
        d[f"encoder.{h}.weight"] = (latent, sizes[-1])
        d[f"encoder.{h}.bias"] = (latent,)
    sizes = [latent] + list(hidden) + [output_dim]
    for j in range(len(sizes) - 1):
        d[f"decoder.deconv_net.{2 * j}.weight"] = (sizes[j + 1], sizes[j])
        d[f"decoder.deconv_net.{2 * j}.bias"] = (sizes[j + 1],)
    return d


def state_dict_shapes(model_name, use_pose=False, latent=256, cond=0, size=IMG_SIZE):
    """``{key: shape}`` in the reference's registration order (``size`` > 64: the extension described at the top)."""
    d = OrderedDict()
    if "mlp" in model_name:
        return mlp_vae_shapes(latent=latent)
    if "regressor" in model_name:
        return regressor_shapes(cond=cond, size=size)
    if "mvae" in model_name:
        d.update(image_encoder_shapes("visual_encoder", latent, cond, size))
        d.update(image_decoder_shapes("visual_decoder", latent, cond, size))
        d.update(image_encoder_shapes("tactile_encoder", latent, cond, size))
        d.update(image_decoder_shapes("tactile_decoder", latent, cond, size))
        if use_pose:
            d.update(pose_encoder_shapes("pose_encoder", latent))
            d.update(pose_decoder_shapes("pose_decoder", latent))
    else:
        d.update(image_encoder_shapes("encoder", latent, cond, size))
        d.update(image_decoder_shapes("decoder", latent, cond, size))
    return d
