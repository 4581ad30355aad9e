#!/usr/bin/env python3
"""Timing diagnostic (results are garbage by construction): how long is the graph-replayed two-lane train step when a
family of kernels is simply not launched?  Answers "what would fusing X away buy at step level" before building it.
usage: skip_kernels_step.py [precision] [batch]"""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402
from mmdyn_hip.engine import MVAEStep  # noqa: E402
from mmdyn_hip.models import setup_model, NoiseSource  # noqa: E402
from mmdyn_hip.utils.seeded_init import seeded_batch  # noqa: E402

FAMILIES = {
    "none": (),
    "bn_apply_fwd": ("bn_swish_fwd",),
    "bn_apply_bwd": ("bn_swish_bwd_apply",),
    "bn_finalize": ("bn_finalize", "bn_bwd_finalize"),
    "bn_all_elementwise": ("bn_swish_fwd", "bn_swish_bwd_apply", "bn_finalize", "bn_bwd_finalize", "bn_swish_bwd_reduce"),
    "reductions": ("wgrad_reduce", "splitk_reduce", "colsum"),
    "small_glue": ("repack2d", "act_bwd", "col2im_k4", "dropout_expand", "dropout_reduce", "linear_small_fwd",
                   "linear_small_bwd"),
    "all_memory_bound": ("bn_swish_fwd", "bn_swish_bwd_apply", "bn_finalize", "bn_bwd_finalize", "bn_swish_bwd_reduce",
                         "wgrad_reduce", "splitk_reduce", "colsum", "repack2d", "act_bwd", "col2im_k4", "tconv_out3_fwd",
                         "bce_logits_groups", "dropout_expand", "dropout_reduce"),
    "all_mfma": ("igemm_nt", "igemm_nt_dgrad_bn", "wgrad_tn"),
}


class Skipping:
    def __init__(self, inner, names):
        self._inner, self._names = inner, set(names)
        self.name = inner.name

    @property
    def precision(self):
        return self._inner.precision

    @precision.setter
    def precision(self, v):
        self._inner.precision = v

    def __getattr__(self, attr):
        fn = getattr(self._inner, attr)
        if attr in self._names:
            return lambda *a, **k: None
        return fn


def main():
    dev = torch.device("cuda")
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    inputs, targets = seeded_batch(B, 1234)
    inputs, targets = [x.to(dev) for x in inputs], [x.to(dev) for x in targets]
    base = ops.B
    for fam, names in FAMILIES.items():
        torch.manual_seed(0)
        model = setup_model("cnn-mvae", cross_modal=True, condition_dim=0, input_dim=4096, architecture="cnn",
                            conditional=False, categorical_conditions=False, latent_size=256, use_pose=True).to(dev).train()
        ops.set_backend(Skipping(base, names))
        step = MVAEStep(model, noise=NoiseSource(1), precision=prec)
        for _ in range(4):
            step.train_step_graphed(inputs, targets, 0.02)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            step.train_step_graphed(inputs, targets, 0.02)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{prec} bs{B}: without {fam:20s} {ms:7.3f} ms/step", flush=True)
        ops.set_backend(base)
        step.close()
        del step, model


if __name__ == "__main__":
    main()
