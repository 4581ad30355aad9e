#!/usr/bin/env python3
"""Diagnostic: per-step loss and the first non-finite gradient tensor of the 256x256 bs 256 step in the fp16 modes."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip.engine import MVAEStep  # noqa: E402
from mmdyn_hip.models import setup_model, NoiseSource  # noqa: E402
from mmdyn_hip.utils.seeded_init import seeded_batch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = "cuda"
torch.manual_seed(1234)
model = setup_model("cnn-mvae", cross_modal=True, condition_dim=0, input_dim=S * S, architecture="cnn", conditional=False,
                    categorical_conditions=False, latent_size=256, use_pose=True).to(dev).train()
step = MVAEStep(model, lr=1e-3, pose_multiplier=1000.0, noise=NoiseSource(1234), precision=prec)
inputs, targets = seeded_batch(B, 1234, size=S)
inputs, targets = [x.to(dev) for x in inputs], [x.to(dev) for x in targets]
named = dict(model.named_parameters())
for s in range(30):
    loss = step.forward(inputs, targets, 0.02)
    h = step.backward()
    g = step.params.grad
    bad = [k for k, p in named.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
    gmax = float(g[torch.isfinite(g)].abs().max()) if torch.isfinite(g).any() else float("nan")
    print(f"step {s}: loss {float(loss):.1f} partials {[round(float(x)) for x in step.partials[:7]]} scaled |g|max {gmax:.3e} "
          f"loss_scale {step.loss_scale} non-finite: {bad[:6]}", flush=True)
    if bad or not torch.isfinite(loss):
        break
    step.optimizer_step(h)
