"""Diagnostic: the (direction, G, T, C) of every BatchNorm finalize call of one eager train step (fp32x3 bs 256, bf16s bs 128) and the size
of its partial-sum table -- 100 KB to 2 MB each, one row of sums per 64- or 128-row GEMM tile: what the finalize launches reduce, and why
the apply kernels cannot each re-read it (docs/LAB_NOTES.md H.g)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("multimodal-dynamics_amd", "tests", ""):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from mmdyn_hip import ops
from mmdyn_hip.engine import MVAEStep
from mmdyn_hip.models import NoiseSource
from mmdyn_hip.utils.seeded_init import seeded_batch
import test_model_emu as T
log = []
of, ob = ops.B.bn_finalize, ops.B.bn_bwd_finalize
def f(partial, mean, rstd, rm, rv, nbt, scratch, G, T_, C, rpg, eps, mom, rep):
    log.append(("fwd", G, T_, C, rpg, rep)); return of(partial, mean, rstd, rm, rv, nbt, scratch, G, T_, C, rpg, eps, mom, rep)
def b(partial, sums, dg, db, scratch, G, T_, C, beta_acc):
    log.append(("bwd", G, T_, C, None, beta_acc)); return ob(partial, sums, dg, db, scratch, G, T_, C, beta_acc)
ops.B.bn_finalize, ops.B.bn_bwd_finalize = f, b
for prec, B in (("fp32x3", 256), ("bf16s", 128)):
    del log[:]
    m = T.build("cnn-mvae", True, True, "cuda")
    step = MVAEStep(m, noise=NoiseSource(1), precision=prec)
    inputs, targets = seeded_batch(B, 5)
    gi, gt = [x.cuda() for x in inputs], [x.cuda() for x in targets]
    step.train_step(gi, gt, 0.05)
    torch.cuda.synchronize()
    print(prec, B)
    for l in log: print("  ", l, "rows", l[1]*l[2], "tableKB", l[1]*l[2]*2*l[3]*4//1024)
    step.close()
