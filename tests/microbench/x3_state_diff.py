#!/usr/bin/env python3
"""Diagnostic: run the fused step's forward + backward twice -- native fp32 and a chosen split configuration (LAB switches) --
and list every tensor reachable from the engine object whose contents differ by more than 1e-4 relative after each phase."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mmdyn_hip.engine import MVAEStep  # noqa: E402
from mmdyn_hip.models import InjectedNoise  # noqa: E402
from mmdyn_hip.utils.seeded_init import seeded_batch, seeded_noise  # noqa: E402
import test_model_emu as T  # noqa: E402


def walk(obj, path, out, seen, depth=0):
    if id(obj) in seen or depth > 6:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_floating_point() and 0 < obj.numel() < 80_000_000:
            out[path] = obj.detach().float().clone()
        return
    if isinstance(obj, dict):
        for k, v in obj.items():
            walk(v, f"{path}[{k!r}]", out, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            walk(v, f"{path}[{i}]", out, seen, depth + 1)
    elif hasattr(obj, "__dict__") and not isinstance(obj, (torch.nn.Module, torch.cuda.Stream, torch.cuda.Event)):
        for k, v in vars(obj).items():
            walk(v, f"{path}.{k}", out, seen, depth + 1)
    elif hasattr(obj, "__slots__"):
        for k in obj.__slots__:
            if hasattr(obj, k):
                walk(getattr(obj, k), f"{path}.{k}", out, seen, depth + 1)


def snapshot(step):
    out = {}
    walk(step, "step", out, set())
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 130
    env = dict(kv.split("=") for kv in sys.argv[2].split(",")) if len(sys.argv) > 2 else {"MMDYN_X3_MODES": "2", "MMDYN_X3_ONLY_G": "1", "MMDYN_X3_ONLY_N": "64"}
    inputs, targets = seeded_batch(B, 1234)
    eps, masks = seeded_noise(B, 256, 7, 8, 4321)
    snaps = {}
    for prec in ("fp32", "fp32x3"):
        for k in env:
            os.environ.pop(k, None)
        if prec == "fp32x3":
            os.environ.update(env)
        m = T.build("cnn-mvae", True, True, "cuda")
        step = MVAEStep(m, noise=InjectedNoise(eps, masks), precision=prec, two_lanes=False)
        gi, gt = [x.cuda() for x in inputs], [x.cuda() for x in targets]
        step.forward(gi, gt, 0.02)
        torch.cuda.synchronize()
        a = snapshot(step)
        step.backward()
        torch.cuda.synchronize()
        snaps[prec] = (a, snapshot(step))
    for phase, name in ((0, "after forward"), (1, "after backward")):
        a, b = snaps["fp32"][phase], snaps["fp32x3"][phase]
        rows = []
        for k in a:
            if k in b and a[k].shape == b[k].shape:
                d = float((a[k] - b[k]).norm() / (a[k].norm() + 1e-30))
                if d > 1e-4:
                    rows.append((d, k, tuple(a[k].shape)))
        print(f"{name}: {len(a)} tensors compared, {len(rows)} differ by > 1e-4")
        if phase == 0:          # ReLU outputs stand in for the pre-activation's sign in the backward pass: flipped masks?
            for k in a:
                if k in b and a[k].shape == b[k].shape and ("'h1'" in k or "'h2'" in k):
                    flips = (a[k] > 0) != (b[k] > 0)
                    if int(flips.sum()):
                        idx = flips.nonzero()[:4].tolist()
                        vals = [(float(a[k][tuple(i)]), float(b[k][tuple(i)])) for i in idx]
                        print(f"   ReLU mask flips in {k} {tuple(a[k].shape)}: {int(flips.sum())} element(s), e.g. {idx} values (native, split) {vals}")
        for d, k, sh in sorted(rows, reverse=True)[:25]:
            print(f"   {d:.3e}  {k}  {sh}")


if __name__ == "__main__":
    main()
