"""Architecture constants and the state_dict wire format of the cnn VAE / MVAE.

Key names and tensor layouts are the reference's checkpoint format
(/root/reference/mmdyn/pytorch/models/vae.py:193-216 encoder, :261-283 decoder,
:117-123 pose MLPs): Conv2d weights ``[Cout, Cin, 4, 4]``,
ConvTranspose2d weights ``[Cin, Cout, 4, 4]``, Linear weights ``[out, in]``.
"""
from collections import OrderedDict

IMG_CH = 3
IMG_SIZE = 64
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


def image_encoder_shapes(pre, latent=256, cond=0):
    d = OrderedDict()
    d[pre + ".conv_net.0.weight"] = (32, IMG_CH, 4, 4)
    d[pre + ".conv_net.2.weight"] = (64, 32, 4, 4)
    _bn(d, pre + ".conv_net.3", 64)
    d[pre + ".conv_net.5.weight"] = (128, 64, 4, 4)
    _bn(d, pre + ".conv_net.6", 128)
    d[pre + ".conv_net.8.weight"] = (256, 128, 4, 4)
    _bn(d, pre + ".conv_net.9", 256)
    d[pre + ".fc_net.0.weight"] = (HID, FEAT)
    d[pre + ".fc_net.0.bias"] = (HID,)
    d[pre + ".linear_means.weight"] = (latent, HID + cond)
    d[pre + ".linear_means.bias"] = (latent,)
    d[pre + ".linear_log_var.weight"] = (latent, HID + cond)
    d[pre + ".linear_log_var.bias"] = (latent,)
    return d


def image_decoder_shapes(pre, latent=256, cond=0):
    d = OrderedDict()
    d[pre + ".upsample.0.weight"] = (FEAT, latent + cond)
    d[pre + ".upsample.0.bias"] = (FEAT,)
    d[pre + ".hallucinate.0.weight"] = (256, 128, 4, 4)
    _bn(d, pre + ".hallucinate.1", 128)
    d[pre + ".hallucinate.3.weight"] = (128, 64, 4, 4)
    _bn(d, pre + ".hallucinate.4", 64)
    d[pre + ".hallucinate.6.weight"] = (64, 32, 4, 4)
    _bn(d, pre + ".hallucinate.7", 32)
    d[pre + ".hallucinate.9.weight"] = (32, IMG_CH, 4, 4)
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


def regressor_shapes(out_dim=7, cond=0):
    """Regressor baseline (models.py:28-64)."""
    d = image_encoder_shapes("r")
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


def state_dict_shapes(model_name, use_pose=False, latent=256, cond=0):
    """``{key: shape}`` in the reference's registration order."""
    d = OrderedDict()
    if "mlp" in model_name:
        return mlp_vae_shapes(latent=latent)
    if "regressor" in model_name:
        return regressor_shapes(cond=cond)
    if "mvae" in model_name:
        d.update(image_encoder_shapes("visual_encoder", latent, cond))
        d.update(image_decoder_shapes("visual_decoder", latent, cond))
        d.update(image_encoder_shapes("tactile_encoder", latent, cond))
        d.update(image_decoder_shapes("tactile_decoder", latent, cond))
        if use_pose:
            d.update(pose_encoder_shapes("pose_encoder", latent))
            d.update(pose_decoder_shapes("pose_decoder", latent))
    else:
        d.update(image_encoder_shapes("encoder", latent, cond))
        d.update(image_decoder_shapes("decoder", latent, cond))
    return d
