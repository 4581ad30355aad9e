from .models import setup_model, count_parameters  # noqa: F401
from .vae import VAE, MVAE, Encoder, Decoder, ProductOfExperts, Swish, prior_expert, NoiseSource, InjectedNoise  # noqa: F401
