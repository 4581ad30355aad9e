"""CLI with the reference's flags and defaults (/root/reference/mmdyn/pytorch/main.py:13-54).

With ``--dataset-path`` pointing at a dataset (the PNG/json tree or its compiled pickle) it trains on it like the
reference, decoding the frames on the GPU (utils/datasets.py).  ``--synthetic-batches N`` (this build only)
replaces the dataset by N random batches per epoch -- the BASELINE workload:

    python -m mmdyn_hip.main --problem-type seq_modeling --input-type visuotactile --model-name cnn-mvae \
        --use-pose --batchsize 256 --num-epochs 2 --synthetic-batches 20
"""
import argparse
import os
import pickle

from . import config
from .problems.problems import Reconstruction, Regression, SeqModeling, DynModeling, SyntheticVisuoTactile


# The reference's flags (same names and defaults: main.py:13-54 there) as data.  (flag, default, type or None for a switch,
# what it selects here).  Defaults are API: a reference command line must mean the same thing with this build.
_REFERENCE_FLAGS = (
    ("problem-type", "seq_modeling", str, "which problem class runs: " + " | ".join(config.PROBLEM_TYPES)),
    ("model-name", "cnn-mvae", str, "registry name of the model: " + " | ".join(config.MODELS)),
    ("input-type", "visual", str, "modality fed to the model: visual | tactile | visuotactile"),
    ("use-pose", False, None, "cnn-mvae only: the 7-DoF pose as a third modality"),
    ("lr", 0.001, float, "learning rate of the fused Adam / SGD step"),
    ("dataset-path", "~/dataset", str, "PNG + json tree or its compiled pickle (frames are decoded on the GPU)"),
    ("batchsize", 128, int, "samples per step"),
    ("criterion", "crossentropy", str, "accepted for compatibility; the VAE problems use the ELBO"),
    ("optimizer", "Adam", str, " | ".join(config.OPTIMIZERS)),
    ("num-epochs", 100, int, "training epochs"),
    ("mask-loss", False, None, "restrict the reconstruction loss to the object segment"),
    ("vis-pose", False, None, "accepted for compatibility (no plotting in this build)"),
    ("pose-multiplier", 1000, float, "weight of the pose MSE inside the ELBO"),
    ("save-name", "run", str, "name of the log directory"),
    ("no-cuda", False, None, "rejected with an error: this build has no CPU path"),
    ("kl-weight", 1.0, float, "KL weight before the annealing schedule overwrites it"),
    ("latent-size", 256, int, "latent dimension"),
    ("annealing-epochs", 50, int, "epochs over which the KL weight is ramped"),
    ("conditional", False, None, "condition encoders / decoders on the shock force"),
)
# switches of this build only
_BUILD_FLAGS = (
    ("synthetic-batches", 0, int, "train on this many random batches per epoch instead of --dataset-path"),
    ("synthetic-seq-length", 1, int, "frames per synthetic sequence"),
    ("reference-schedule", False, None, "run the reference's 7-forward autograd schedule instead of the fused step"),
    ("image-size", 64, int, "side of the input images: 64 (the reference's only size) or the 128 / 256 pixel extended stacks "
                            "(models/shapes.py; BASELINE configs[3] / configs[4], no reference architecture)"),
    ("precision", "fp32x3", str, "matrix-core arithmetic of the fused step: fp32x3 (default: fp32 storage and results, exact three-term operand split on the bf16 matrix cores) | fp32 (the native fp32 matrix cores) | bf16 | bf16s | fp16 | fp16s"),
    ("exact-running-stats", False, None, "fused step: also run the image decoders on the subset passes whose reconstruction the "
                                         "reference computes and discards, so that the decoders' BatchNorm running statistics "
                                         "get the reference's 7 (3) updates per step instead of 4 (2); ~1.4x the step time"),
)


def build_parser():
    parser = argparse.ArgumentParser(description="cnn-VAE / cnn-MVAE training on MI355X (mmdyn_hip)")
    for name, default, kind, text in _REFERENCE_FLAGS + _BUILD_FLAGS:
        if kind is None:
            parser.add_argument("--" + name, action="store_true", default=default, help=text)
        else:
            parser.add_argument("--" + name, type=kind, default=default, help=f"{text} (default: {default})")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.problem_type in config.PROBLEM_TYPES, "Invalid problem type."
    L = args.synthetic_seq_length
    shock = 3 if args.conditional else 0
    if args.synthetic_batches > 0:
        loaders = dict(train_loader=SyntheticVisuoTactile(args.synthetic_batches, args.batchsize, L, seed=1234,
                                                          shock_dim=shock, size=args.image_size),
                       test_loader=SyntheticVisuoTactile(max(1, args.synthetic_batches // 4), args.batchsize, L,
                                                         seed=4321, shock_dim=shock, size=args.image_size),
                       seq_length=L, fused=not args.reference_schedule)
    else:
        loaders = dict(fused=not args.reference_schedule)      # Problem.set_dataset reads --dataset-path
    problem_args = argparse.Namespace(**{k: v for k, v in vars(args).items()
                                         if not k.startswith('synthetic') and k != 'reference_schedule'})
    if args.problem_type == 'regression':
        problem = Regression(problem_args, **loaders)
    elif args.problem_type == 'reconstruction':
        problem = Reconstruction(problem_args, **loaders)
    elif args.problem_type == 'dyn_modeling':
        problem = DynModeling(problem_args, **loaders)
    else:
        problem = SeqModeling(problem_args, **loaders)
    with open(os.path.join(problem.log_dir, 'problem.pkl'), 'wb') as f:
        pickle.dump(problem_args, f)
    problem.train()
    return problem


if __name__ == "__main__":
    main()
