"""
vidp_amd -- MI355X-native block-tri-diagonal Gauss-Markov path for the VDP / CVI-DP ELBO.

Only the hot path of AaltoML/vi-diffusion-processes behind the reference's own
interface names; compute happens in hand-written HIP kernels (csrc/) reached through a C ABI.
"""
from . import _lib  # noqa: F401
from .packed import Plan  # noqa: F401
from ._lib import FULL, SYM, TRI, VEC  # noqa: F401
