#!/usr/bin/env python3
"""Diagnostic: storage types of the operands of every MFMA launch of one train step in a 16-bit precision mode (an fp32
tensor in front of bf16 matrix cores costs twice the bytes and the narrower K-step variant of the kernel).
usage: dtype_audit.py [precision] [batch]"""
import collections
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402
from mmdyn_hip.engine import MVAEStep  # noqa: E402
from mmdyn_hip.models import setup_model, NoiseSource  # noqa: E402
from mmdyn_hip.utils.seeded_init import seeded_batch  # noqa: E402

SHORT = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16", torch.float64: "f64", torch.uint8: "u8"}


class Audit:
    def __init__(self, inner):
        self._inner, self.name, self.rows = inner, inner.name, collections.OrderedDict()

    @property
    def precision(self):
        return self._inner.precision

    @precision.setter
    def precision(self, v):
        self._inner.precision = v

    def __getattr__(self, attr):
        fn = getattr(self._inner, attr)
        if attr not in ("igemm_nt", "igemm_nt_dgrad_bn", "wgrad_tn"):
            return fn

        def wrapped(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            key = (attr, tuple(x for x in a if isinstance(x, (int, bool))),
                   tuple(SHORT.get(x.dtype, "?") if torch.is_tensor(x) else "-" for x in a if torch.is_tensor(x) or x is None))
            self.rows.setdefault(key, []).append((s, e))
            return r
        return wrapped


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16s"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = setup_model("cnn-mvae", cross_modal=True, condition_dim=0, input_dim=4096, architecture="cnn", conditional=False,
                        categorical_conditions=False, latent_size=256, use_pose=True).to(dev).train()
    inputs, targets = seeded_batch(B, 1234)
    inputs, targets = [x.to(dev) for x in inputs], [x.to(dev) for x in targets]
    step = MVAEStep(model, noise=NoiseSource(1), precision=prec, two_lanes=False)
    for _ in range(2):
        step.train_step(inputs, targets, 0.02)
    aud = Audit(ops.B)
    ops.set_backend(aud)
    step.train_step(inputs, targets, 0.02)
    torch.cuda.synchronize()
    ops.set_backend(aud._inner)
    rows = sorted(aud.rows.items(), key=lambda kv: -sum(s.elapsed_time(e) for s, e in kv[1]))
    for (name, sig, dts), ev in rows:
        ms = sum(s.elapsed_time(e) for s, e in ev)
        flag = "  <-- fp32 operand" if ("f32" in dts[:2] or (name == "wgrad_tn" and "f32" in dts[:2])) else ""
        print(f"{name:18s} x{len(ev):2d} {ms * 1e3:7.1f} us  {str(sig):64s} {','.join(dts)}{flag}")


if __name__ == "__main__":
    main()
