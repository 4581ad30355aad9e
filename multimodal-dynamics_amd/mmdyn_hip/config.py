"""Names the command line and the model / problem registries accept.

The reference validates its flags against module-level lists (mmdyn/pytorch/config.py); the same public names exist here
so that ``config.MODELS`` etc. keep working for callers, but they are derived from one table that also says where each
choice is implemented in this package.
"""

# flag value -> where it is handled here
_PROBLEMS = {
    "regression": "problems.Regression (Regressor on the conv trunk kernels)",
    "reconstruction": "problems.Reconstruction",
    "seq_modeling": "problems.SeqModeling (fused engine for cnn-mvae)",
    "dyn_modeling": "problems.DynModeling (fused engine for cnn-mvae)",
}
_MODELS = {
    "mlp-vae": ("mlp", "models.vae.VAE with MLP encoder / decoder"),
    "cnn-vae": ("cnn", "models.vae.VAE on the conv kernels"),
    "cnn-mvae": ("cnn", "models.vae.MVAE: the hot path (engine.MVAEStep)"),
    "regressor": ("cnn", "models.models.Regressor"),
}
_OPTIMIZERS = {"SGD": "problems.FusedSGD (mmdyn_sgd_step)", "Adam": "problems.FusedAdam (mmdyn_adam_step)"}
_MODALITIES = ("visual", "tactile", "pose", "visuotactile")

PROBLEM_TYPES = list(_PROBLEMS)
MODELS = list(_MODELS)
ARCHITECTURES = sorted({arch for arch, _ in _MODELS.values()}, reverse=True)      # ['mlp', 'cnn']
OPTIMIZERS = list(_OPTIMIZERS)
INPUT_TYPES = [None, *_MODALITIES]          # None: the flag's default, rejected later by the problem classes
CRITERIONS = ["crossentropy"]               # accepted by the reference's CLI, used by none of the ported problems
