"""mmdyn_hip: MI355X-native (gfx950) cnn-VAE / cnn-MVAE hot path behind the reference's module API.

Layout: ``csrc/`` hand-written HIP kernels + C ABI (``include/mmdyn_hip.h``); ``_lib``/``ops`` ctypes binding;
``layers`` kernel schedules per layer stack; ``models`` drop-in modules; ``engine`` fused train step;
``problems``/``main`` the caller-side mirror.
"""
__version__ = "0.1"
