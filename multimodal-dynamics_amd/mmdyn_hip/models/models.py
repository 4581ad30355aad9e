"""Model registry: the ``--model-name`` plug-in surface of /root/reference/mmdyn/pytorch/models/models.py:13-25."""
from .. import config
from .vae import VAE, MVAE, Swish  # noqa: F401  (Swish re-exported like the reference does)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def setup_model(model_name, cross_modal=False, **kwargs):
    """Same dispatch rules and assertions as the reference: 'mvae' needs cross-modal input, 'vae' must not
    get it.  'regressor' (a different problem type) is not built."""
    assert (model_name in config.MODELS), "Model is not implement yet"
    if 'mvae' in model_name and cross_modal:
        model = MVAE(**kwargs)
    elif 'vae' in model_name:
        assert not cross_modal, "VAE does not work with cross modal inputs."
        model = VAE(**kwargs)
    elif 'regressor' in model_name:
        raise NotImplementedError("mmdyn_hip: the Regressor baseline is outside the cnn-mvae hot path")
    else:
        exit("The model and modality combination is not valid.")
    return model
