from . import shestakov_nonlinear_diffusion  # noqa: F401
