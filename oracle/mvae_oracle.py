"""CPU ORACLE -- test infrastructure only.  NOT part of the product path.

A plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's cnn-mvae / cnn-vae
hot path, written as free functions over a flat ``{state_dict key: tensor}`` mapping.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``multimodal-dynamics_amd/``) never does and fails
loudly when its HIP library is missing.

Parity pin: every function here is checked against golden vectors produced by running
the reference itself in the development container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).

Reference locations restated (paths relative to /root/reference/mmdyn/pytorch):
  models/vae.py:193-242   Encoder (conv trunk, FC, dropout, two heads; mlp variant for pose)
  models/vae.py:261-301   Decoder (FC upsample, transposed-conv stack -> logits; mlp variant)
  models/vae.py:311-318   ProductOfExperts (eps added twice to the variance)
  models/vae.py:52-61     reparametrize
  models/vae.py:126-176   MVAE.forward / inference,  :81-98 VAE.forward / inference
  models/vae.py:331-334   Swish
  problems/problems.py:401-458   _elbo_loss / _mvae_elbo_loss
  problems/problems.py:473-546   _evaluate_mvae (3 or 7 modality subsets)
  problems/problems.py:702-716   SeqModeling._evaluate_model, VAE branch
  problems/problems.py:130-138, 148-156   Adam(lr) and the step body
  problems/problems.py:212-216   _anneal_KL
  problems/problems.py:634-673, 765-803   SeqModeling / DynModeling.parse_input
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
POE_EPS = 1e-8
DROPOUT_P = 0.1

# (visual?, tactile?, pose?) for the passes of _evaluate_mvae, in call order
SUBSETS_POSE = [(1, 1, 0), (1, 0, 0), (0, 1, 0), (1, 1, 1), (1, 0, 1), (0, 1, 1), (0, 0, 1)]
SUBSETS_NOPOSE = SUBSETS_POSE[:3]


def swish(x):
    return x * torch.sigmoid(x)                                             # vae.py:331-334


EVAL = False          # module.eval(): BatchNorm2d normalises with the running estimates and leaves them alone


class eval_mode:
    """``with O.eval_mode():`` -- the restatement of ``model.eval()`` (dropout is off when keep_mask is None)."""

    def __enter__(self):
        global EVAL
        self._prev, EVAL = EVAL, True

    def __exit__(self, *exc):
        global EVAL
        EVAL = self._prev


def batchnorm_train(x, prm, prefix, buffers=None):
    """Train-mode BatchNorm2d: batch statistics, biased variance for normalisation,
    unbiased for the running estimate (momentum 0.1).  Under :class:`eval_mode`: F.batch_norm(training=False)."""
    if EVAL:
        rm, rv = buffers[prefix + ".running_mean"], buffers[prefix + ".running_var"]
        y = (x - rm[None, :, None, None]) * torch.rsqrt(rv + BN_EPS)[None, :, None, None]
        return y * prm[prefix + ".weight"][None, :, None, None] + prm[prefix + ".bias"][None, :, None, None]
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    y = (x - mean[None, :, None, None]) * torch.rsqrt(var + BN_EPS)[None, :, None, None]
    y = y * prm[prefix + ".weight"][None, :, None, None] + prm[prefix + ".bias"][None, :, None, None]
    if buffers is not None:
        with torch.no_grad():
            rm, rv = buffers[prefix + ".running_mean"], buffers[prefix + ".running_var"]
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach().to(rm.dtype))
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (var.detach() * n / max(n - 1, 1)).to(rv.dtype))
            buffers[prefix + ".num_batches_tracked"] += 1
    return y


def _conv_indices(prm, pre):
    """Sequential indices of the 4-D weights under ``pre`` (conv_net / hallucinate), ascending."""
    return sorted(int(k[len(pre) + 1:].split(".")[0]) for k in prm
                  if k.startswith(pre + ".") and k.endswith(".weight") and prm[k].dim() == 4)


def image_encoder_trunk(x, prm, pre, buffers=None):
    """conv_net + fc_net up to (not including) the dropout: vae.py:197-212, 224-229.  Four convolutions for the
    reference's 64 x 64 input; the 128 / 256 pixel EXTENSIONS (no reference architecture: its FC is fixed at 256*5*5)
    have one / two more stride-2 Conv+BN+Swish stages behind the first convolution, found here by their keys."""
    idx = _conv_indices(prm, pre + ".conv_net")
    h = swish(F.conv2d(x, prm[pre + ".conv_net.0.weight"], stride=2, padding=1))
    for i in idx[1:]:
        last = i == idx[-1]                                                  # Conv2d(128, 256, 4, 1, 0): 8 -> 5
        h = F.conv2d(h, prm[pre + f".conv_net.{i}.weight"], stride=1 if last else 2, padding=0 if last else 1)
        h = swish(batchnorm_train(h, prm, pre + f".conv_net.{i + 1}", buffers))
    h = h.reshape(h.shape[0], -1)
    return swish(F.linear(h, prm[pre + ".fc_net.0.weight"], prm[pre + ".fc_net.0.bias"]))


def encoder_heads(h, prm, pre, keep_mask=None, c=None):
    """dropout (injected keep-mask, scale 1/(1-p)) then the two linear heads: vae.py:213-216, 239-240."""
    if keep_mask is not None:
        h = h * (keep_mask.to(h.dtype) / (1.0 - DROPOUT_P))
    if c is not None:
        if c.dim() == 1:
            c = c.unsqueeze(1)
        h = torch.cat((h, c.to(h.dtype)), dim=-1)                            # vae.py:231-237
    mu = F.linear(h, prm[pre + ".linear_means.weight"], prm[pre + ".linear_means.bias"])
    lv = F.linear(h, prm[pre + ".linear_log_var.weight"], prm[pre + ".linear_log_var.bias"])
    return mu, lv


def image_encoder(x, prm, pre, keep_mask, buffers=None, c=None):
    return encoder_heads(image_encoder_trunk(x, prm, pre, buffers), prm, pre, keep_mask, c)


def pose_encoder(p, prm, pre="pose_encoder"):
    """mlp Encoder [7, 512, 512]: ReLU between, Identity after; no dropout (vae.py:14-19, 218-222)."""
    h = torch.relu(F.linear(p, prm[pre + ".fc_net.0.weight"], prm[pre + ".fc_net.0.bias"]))
    h = F.linear(h, prm[pre + ".fc_net.2.weight"], prm[pre + ".fc_net.2.bias"])
    return encoder_heads(h, prm, pre, None)


def image_decoder(z, prm, pre, buffers=None, c=None):
    """vae.py:263-279, 285-301: FC+Swish -> [B,256,5,5] -> 3x(convT+BN+Swish) -> convT; returns LOGITS."""
    if c is not None:
        if c.dim() == 1:
            c = c.unsqueeze(1)
        z = torch.cat((z, c.to(z.dtype)), dim=-1)
    h = swish(F.linear(z, prm[pre + ".upsample.0.weight"], prm[pre + ".upsample.0.bias"]))
    h = h.reshape(-1, 256, 5, 5)
    idx = _conv_indices(prm, pre + ".hallucinate")                           # (more than four: the size extensions)
    for i in idx[:-1]:
        first = i == idx[0]                                                  # ConvTranspose2d(256, 128, 4, 1, 0): 5 -> 8
        h = F.conv_transpose2d(h, prm[pre + f".hallucinate.{i}.weight"], stride=1 if first else 2, padding=0 if first else 1)
        h = swish(batchnorm_train(h, prm, pre + f".hallucinate.{i + 1}", buffers))
    return F.conv_transpose2d(h, prm[pre + f".hallucinate.{idx[-1]}.weight"], stride=2, padding=1)


def pose_decoder(z, prm, pre="pose_decoder"):
    h = torch.relu(F.linear(z, prm[pre + ".deconv_net.0.weight"], prm[pre + ".deconv_net.0.bias"]))
    h = torch.relu(F.linear(h, prm[pre + ".deconv_net.2.weight"], prm[pre + ".deconv_net.2.bias"]))
    return F.linear(h, prm[pre + ".deconv_net.4.weight"], prm[pre + ".deconv_net.4.bias"])


def mlp_stack(x, prm, pre, n_layers):
    """``mlp(sizes, nn.ReLU, nn.Identity)`` (vae.py:14-19): Linear at even indices, ReLU between, none at the end."""
    for j in range(n_layers):
        x = F.linear(x, prm[f"{pre}.{2 * j}.weight"], prm[f"{pre}.{2 * j}.bias"])
        if j < n_layers - 1:
            x = torch.relu(x)
    return x


def mlp_vae_forward(prm, x, eps_noise, input_dim=784):
    """VAE.forward with the mlp architecture (vae.py:81-88): flatten, fc_net, heads, reparametrize, deconv_net."""
    if x.dim() > 2:
        x = x.reshape(-1, input_dim)
    n_enc = sum(1 for k in prm if k.startswith("encoder.fc_net.") and k.endswith(".weight"))
    n_dec = sum(1 for k in prm if k.startswith("decoder.deconv_net.") and k.endswith(".weight"))
    h = mlp_stack(x, prm, "encoder.fc_net", n_enc)
    mu = F.linear(h, prm["encoder.linear_means.weight"], prm["encoder.linear_means.bias"])
    lv = F.linear(h, prm["encoder.linear_log_var.weight"], prm["encoder.linear_log_var.bias"])
    z = reparametrize(mu, lv, eps_noise)
    return mlp_stack(z, prm, "decoder.deconv_net", n_dec), mu, lv


def regressor_forward(prm, x, keep_mask, c=None, buffers=None):
    """Regressor.forward (models.py:66-77): the image-encoder trunk, dropout, optional condition concat, then
    out_net = Linear(512[+cd],256) ReLU Linear(256,256) ReLU Linear(256,out_dim) (models.py:57-63)."""
    pp = {"r." + k: v for k, v in prm.items()}
    bb = {"r." + k: v for k, v in buffers.items()} if buffers is not None else None
    h = image_encoder_trunk(x, pp, "r", bb)
    h = h * (keep_mask.to(h.dtype) / (1.0 - DROPOUT_P))
    if c is not None:
        if c.dim() == 1:
            c = c.unsqueeze(1)
        h = torch.cat((h, c.to(h.dtype)), dim=-1)
    h = torch.relu(F.linear(h, prm["out_net.0.weight"], prm["out_net.0.bias"]))
    h = torch.relu(F.linear(h, prm["out_net.2.weight"], prm["out_net.2.bias"]))
    return F.linear(h, prm["out_net.4.weight"], prm["out_net.4.bias"])


def product_of_experts(mu, logvar, eps=POE_EPS):
    """vae.py:311-318.  mu/logvar: [M, B, D]; the variance gets ``eps`` twice."""
    var = torch.exp(logvar) + eps
    T = 1.0 / (var + eps)
    pd_mu = (mu * T).sum(0) / T.sum(0)
    pd_var = 1.0 / T.sum(0)
    return pd_mu, torch.log(pd_var + eps)


def reparametrize(mu, logvar, eps_noise):
    return eps_noise * torch.exp(0.5 * logvar) + mu                          # vae.py:57-59


def mvae_forward(prm, visual, tactile, pose, eps_noise, masks, use_pose=True, buffers=None, condition=None):
    """MVAE.forward (vae.py:126-165).  ``masks``: iterator over dropout keep-masks, consumed in the
    order visual_encoder, tactile_encoder.  Returns the reference's 5-tuple."""
    ref = visual if visual is not None else (tactile if tactile is not None else pose)
    B, L = ref.shape[0], prm["visual_encoder.linear_means.bias"].shape[0]
    mus = [torch.zeros(B, L, dtype=ref.dtype)]
    lvs = [torch.zeros(B, L, dtype=ref.dtype)]                                # prior expert, vae.py:139, 321-328
    if visual is not None:
        m, l = image_encoder(visual, prm, "visual_encoder", next(masks), buffers, condition)
        mus.append(m), lvs.append(l)
    if tactile is not None:
        m, l = image_encoder(tactile, prm, "tactile_encoder", next(masks), buffers, condition)
        mus.append(m), lvs.append(l)
    if pose is not None and use_pose:
        m, l = pose_encoder(pose, prm)
        mus.append(m), lvs.append(l)
    mu, lv = product_of_experts(torch.stack(mus), torch.stack(lvs))
    z = reparametrize(mu, lv, eps_noise)
    vr = image_decoder(z, prm, "visual_decoder", buffers, condition)
    tr = image_decoder(z, prm, "tactile_decoder", buffers, condition)
    pr = pose_decoder(z, prm) if use_pose else None
    return vr, tr, pr, mu, lv


def mvae_inference(prm, z, buffers=None):
    """MVAE.inference (vae.py:167-176) with the latent draw injected."""
    return image_decoder(z, prm, "visual_decoder", buffers), image_decoder(z, prm, "tactile_decoder", buffers)


def kl_divergence(mu, lv):
    return -0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp())                   # problems.py:406, 429


def mvae_elbo_loss(recons, targets, mu, lv, kl_weight, pose_multiplier=1000.0, loss_mask=None):
    """problems.py:421-458 with reduce=None, reduction='sum'."""
    B = targets[0].shape[0]
    rec = 0
    for r, x in zip(recons, targets):
        if r.dim() > 2:
            r = r.reshape(x.shape)
            if loss_mask is not None:
                r, x = r * loss_mask, x * loss_mask
            rec = rec + F.binary_cross_entropy_with_logits(r, x, reduction="sum")
        else:
            if loss_mask is not None:
                r, x = r * loss_mask, x * loss_mask
            rec = rec + pose_multiplier * F.mse_loss(r, x, reduction="sum")
    return (rec + kl_weight * kl_divergence(mu, lv)) / B


def elbo_loss(recon, x, mu, lv, kl_weight, loss_mask=None):
    """problems.py:401-419 with reduce=None."""
    r = recon.reshape(x.shape)
    if loss_mask is not None:
        r, x = r * loss_mask, x * loss_mask
    bce = F.binary_cross_entropy_with_logits(r, x, reduction="sum")
    return (bce + kl_weight * kl_divergence(mu, lv)) / x.shape[0]


def evaluate_mvae(prm, inputs, targets, eps_list, mask_list, kl_weight, pose_multiplier=1000.0,
                  use_pose=True, buffers=None, loss_mask=None, condition=None):
    """Reconstruction._evaluate_mvae (problems.py:473-546): the reference's own schedule --
    one full MVAE.forward (all decoders, live or not) per modality subset, losses summed."""
    eps_it, mask_it = iter(eps_list), iter(mask_list)
    subsets = SUBSETS_POSE if use_pose else SUBSETS_NOPOSE
    v, t = inputs[0], inputs[1]
    p = inputs[2] if use_pose else None
    loss, partials, keep = 0, [], {}
    for i, (a, b, c) in enumerate(subsets):
        vr, tr, pr, mu, lv = mvae_forward(prm, v if a else None, t if b else None, p if c else None,
                                          next(eps_it), mask_it, use_pose, buffers, condition)
        rec, tg = [], []
        if a:
            rec.append(vr), tg.append(targets[0])
        if b:
            rec.append(tr), tg.append(targets[1])
        if c:
            rec.append(pr), tg.append(targets[2])
        li = mvae_elbo_loss(rec, tg, mu, lv, kl_weight, pose_multiplier, loss_mask)
        partials.append(li)
        loss = loss + li
        keep[i] = (vr, tr, pr, mu, lv)
    with torch.no_grad():                                                    # problems.py:499-503, 534-535
        perf = {"visual": float(F.binary_cross_entropy_with_logits(keep[1][0], targets[0])),
                "tactile": float(F.binary_cross_entropy_with_logits(keep[2][1], targets[1]))}
        if use_pose:
            perf["pose"] = float(F.mse_loss(keep[6][2], targets[2]))
    joint = keep[3] if use_pose else keep[0]
    last = keep[len(subsets) - 1]
    outputs = {"recon_x": [joint[0], joint[1]] + ([joint[2]] if use_pose else []),
               "means": last[3], "log_var": last[4], "perf_measure": perf}
    return outputs, loss, partials


def vae_forward(prm, x, eps_noise, keep_mask, buffers=None):
    """VAE.forward (vae.py:81-88), cnn architecture."""
    mu, lv = image_encoder(x, prm, "encoder", keep_mask, buffers)
    z = reparametrize(mu, lv, eps_noise)
    return image_decoder(z, prm, "decoder", buffers), mu, lv


def evaluate_vae(prm, x, target, eps_noise, keep_mask, kl_weight, buffers=None, loss_mask=None):
    """SeqModeling._evaluate_model, VAE branch (problems.py:705-716)."""
    recon, mu, lv = vae_forward(prm, x, eps_noise, keep_mask, buffers)
    loss = elbo_loss(recon, target, mu, lv, kl_weight, loss_mask)
    with torch.no_grad():
        perf = float(F.binary_cross_entropy_with_logits(recon, target))
    return {"recon_x": recon, "means": mu, "log_var": lv, "perf_measure": perf}, loss


def anneal_kl(epoch, annealing_epochs):
    return (epoch + 1) / annealing_epochs if epoch < annealing_epochs else 1  # problems.py:212-216


class Adam:
    """torch.optim.Adam(lr) defaults restated: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params, self.lr, self.betas, self.eps, self.t = list(params), lr, betas, eps, 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


def split_state(state, dtype=torch.float32, requires_grad=True):
    """state_dict -> (params with grad, buffers) as independent CPU copies."""
    prm, buf = {}, {}
    for k, v in state.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("running_mean", "running_var"):
            buf[k] = v.detach().clone().to("cpu", dtype)
        elif leaf == "num_batches_tracked":
            buf[k] = v.detach().clone().to("cpu")
        else:
            prm[k] = v.detach().clone().to("cpu", dtype).requires_grad_(requires_grad)
    return prm, buf


def seq_parse_input(data, target, seq_length, input_type):
    """SeqModeling.parse_input (problems.py:634-673): keep frame 0 of every sequence ([::l])."""
    l = seq_length
    idx = {"visual": [0], "tactile": [1], "visuotactile": [0, 1]}[input_type]
    mi = [data[i][::l] for i in idx]
    to = [target[i][::l] for i in idx]
    if len(idx) == 1:
        mi, to = mi[0], to[0]
    x = {"model_input": mi, "input_object_pose": [data[2][::l]], "input_available_modals": data[3][::l],
         "shock": data[4][::l] if len(data) > 4 else None}
    t = {"target_output": to, "target_object_pose": [target[2][::l]], "loss_mask": target[3][::l]}
    return x, t


def dyn_parse_input(data, target, seq_length, input_type):
    """DynModeling.parse_input (problems.py:765-803) on flat [B*L, ...] frames: one-step-ahead
    targets by a roll of -1; the last frame of each sequence takes the dataset's final target for
    images, while the pose target keeps the plain roll (reference behaviour, including the wrap)."""
    l = seq_length
    idx = {"visual": [0], "tactile": [1], "visuotactile": [0, 1]}[input_type]
    mi, to = [], []
    for i in idx:
        mi.append(data[i])
        tgt = torch.roll(data[i], -1, dims=0).clone()
        tgt[l - 1::l] = target[i][l - 1::l]
        to.append(tgt)
    if len(idx) == 1:
        mi, to = mi[0], to[0]
    x = {"model_input": mi, "input_object_pose": [data[2]], "input_available_modals": data[3],
         "shock": data[4] if len(data) > 4 else None}
    t = {"target_output": to, "target_object_pose": [torch.roll(data[2], -1, dims=0)], "loss_mask": target[3]}
    return x, t
