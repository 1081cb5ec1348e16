"""gmmvi_amd -- MI355X-native implementation of the per-iteration hot path of OlegArenz/gmmvi.

Host side: plain Python + NumPy mirroring the reference's module paths and class names
(``gmmvi_amd.gmmvi_runner.GmmviRunner``, ``gmmvi_amd.optimization.gmmvi.GMMVI`` ...).  Arithmetic: hand-written HIP
kernels for gfx950 in ``libgmmvi_hip.so`` reached through a C ABI (``include/gmmvi_hip.h``).  There is no CPU
fallback; importing the compute modules without the built library or without a GPU raises.
"""
__version__ = "0.1.0"
