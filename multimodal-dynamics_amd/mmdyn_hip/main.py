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


def build_parser():
    parser = argparse.ArgumentParser(description='MI355X-native cnn-VAE / cnn-MVAE training')
    # Problem (same names, defaults and help as the reference)
    parser.add_argument('--problem-type', default='seq_modeling', type=str, help='Problem type (default: seq_modeling)')
    parser.add_argument('--model-name', default='cnn-mvae', type=str, help='Model architecture name')
    parser.add_argument('--input-type', default='visual', type=str,
                        help='The input modality (valid: visual, tactile, visuotactile)')
    parser.add_argument('--use-pose', action='store_true', default=False,
                        help="Use pose as additional modality, only works for MVAE) (default: False)")
    parser.add_argument('--lr', default=0.001, type=float, help='learning rate (default: 0.001)')
    parser.add_argument('--dataset-path', default="~/dataset", type=str, help='Absolute path to the dataset.')
    parser.add_argument('--batchsize', default=128, type=int, help='Batchsize (default: 128)')
    parser.add_argument('--criterion', default="crossentropy", type=str, help='Training loss (default: crossentropy)')
    parser.add_argument('--optimizer', default="Adam", type=str, help='Optimizer name (default: Adam)')
    parser.add_argument('--num-epochs', default=100, type=int, help='Number of training epochs (default: 100)')
    parser.add_argument('--mask-loss', action='store_true', default=False,
                        help="Mask the reconstruction loss to the object segment (default: False)")
    parser.add_argument('--vis-pose', action='store_true', default=False, help="Visualize pose (ignored here)")
    parser.add_argument('--pose-multiplier', default=1000, type=float, help="Multiplier for pose loss (default: 1000)")
    # Misc
    parser.add_argument('--save-name', default='run', type=str, help='Name used for the log directory (default: run)')
    parser.add_argument('--no-cuda', action='store_true', default=False,
                        help="Rejected: this build has no CPU path (use the reference for CPU runs)")
    # VAE specific
    parser.add_argument('--kl-weight', type=float, default=1.0, help="KL weight (overwritten by the annealing schedule)")
    parser.add_argument('--latent-size', type=int, default=256, help="Latent dimension (default: 256)")
    parser.add_argument('--annealing-epochs', type=int, default=50, help="Number of epochs to anneal KL for (default: 50)")
    parser.add_argument('--conditional', action='store_true', default=False, help="Conditional VAE (conditioned on the shock force)")
    # this build only
    parser.add_argument('--synthetic-batches', type=int, default=0,
                        help="train on this many random batches per epoch instead of --dataset-path")
    parser.add_argument('--synthetic-seq-length', type=int, default=1, help="frames per synthetic sequence")
    parser.add_argument('--reference-schedule', action='store_true', default=False,
                        help="run the reference's 7-forward autograd schedule instead of the fused step")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.problem_type in config.PROBLEM_TYPES, "Invalid problem type."
    L = args.synthetic_seq_length
    shock = 3 if args.conditional else 0
    if args.synthetic_batches > 0:
        loaders = dict(train_loader=SyntheticVisuoTactile(args.synthetic_batches, args.batchsize, L, seed=1234,
                                                          shock_dim=shock),
                       test_loader=SyntheticVisuoTactile(max(1, args.synthetic_batches // 4), args.batchsize, L,
                                                         seed=4321, shock_dim=shock),
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
