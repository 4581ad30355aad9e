#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error of the fused engine against the CPU oracle (the comparison of
tests/test_model_gpu.py::test_fused_engine_vs_oracle) for a batch size, in every fp32 arithmetic: native / three-term split.
usage: x3_grad_errors.py [B]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mvae_oracle as O  # noqa: E402
from mmdyn_hip.engine import MVAEStep  # noqa: E402
from mmdyn_hip.models import InjectedNoise  # noqa: E402
from mmdyn_hip.models.shapes import state_dict_shapes  # noqa: E402
from mmdyn_hip.utils.seeded_init import seeded_state_dict, seeded_batch, seeded_noise  # noqa: E402
import test_model_emu as T  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 130
    klw = 1.0 / 50
    sd = seeded_state_dict(state_dict_shapes("cnn-mvae", use_pose=True), 0)
    prm, buf = O.split_state(sd)
    inputs, targets = seeded_batch(B, 1234)
    eps, masks = seeded_noise(B, 256, 7, 8, 4321)
    names = list(prm.keys())
    _, loss_o, _ = O.evaluate_mvae(prm, inputs, targets, eps[:7], masks[:8], klw, 1000.0, True, buf)
    loss_o.backward()
    grads = {}
    # (LAB library, MMDYN_HIP_LIB=...lab.so: "fp32x3:igemm" / "fp32x3:wgrad" switch one kernel family's split off; ":1lane" runs the
    #  step on one stream)
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "fp32x3"]
    for var in variants:
        prec = var.split(":")[0]
        os.environ.pop("MMDYN_X3", None)
        os.environ.pop("MMDYN_X3_WGRAD", None)
        for k in ("MMDYN_X3_MODES", "MMDYN_X3_ONLY_G", "MMDYN_X3_ONLY_N", "MMDYN_X3_TILE"):
            os.environ.pop(k, None)
        os.environ.pop("MMDYN_WS_TAPORDER", None)
        os.environ.pop("MMDYN_IGEMM_WS", None)
        if ":raster" in var:          # native arithmetic, another summation order in the stride-2 convolutions
            os.environ["MMDYN_WS_TAPORDER"] = "0"
        if ":nows" in var:            # native arithmetic on the register-staged kernels (other tiles, other summation order)
            os.environ["MMDYN_IGEMM_WS"] = "0"
        for part in var.split(":"):
            if part.startswith("modes"):
                os.environ["MMDYN_X3_MODES"] = part[5:]
            if part.startswith("G"):
                os.environ["MMDYN_X3_ONLY_G"] = part[1:]
            if part.startswith("N"):
                os.environ["MMDYN_X3_ONLY_N"] = part[1:]
            if part.startswith("tile"):
                os.environ["MMDYN_X3_TILE"] = part[4:].replace("x", ",")
        if ":igemm" in var:
            os.environ["MMDYN_X3_WGRAD"] = "0"
        if ":wgrad" in var:
            os.environ["MMDYN_X3"] = "0"
        m = T.build("cnn-mvae", True, True, "cuda")
        step = MVAEStep(m, noise=InjectedNoise(eps, masks), precision=prec, two_lanes=":1lane" not in var)
        prec = var
        gi, gt = [x.cuda() for x in inputs], [x.cuda() for x in targets]
        if ":fwdonly" in var:          # the split in the forward launches only (LAB: the switch is read per launch)
            os.environ.pop("MMDYN_X3", None)
        if ":bwdonly" in var:
            os.environ["MMDYN_X3"] = "0"
        loss = step.forward(gi, gt, klw)
        torch.cuda.synchronize()
        if ":fwdonly" in var:
            os.environ["MMDYN_X3"] = "0"
        if ":bwdonly" in var:
            os.environ.pop("MMDYN_X3", None)
        step.backward()
        torch.cuda.synchronize()
        named = dict(m.named_parameters())
        grads[prec] = {k: named[k].grad.double().cpu() for k in names}
        errs = sorted(((float((grads[prec][k] - prm[k].grad.double()).norm() / (prm[k].grad.double().norm() + 1e-30)), k) for k in names),
                      reverse=True)
        print(f"{prec}: loss {float(loss):.4f} (oracle {float(loss_o):.4f}); worst gradient tensors vs the oracle:")
        for e, k in errs[:4]:
            print(f"    {e:.3e}  {k}")


if __name__ == "__main__":
    main()
