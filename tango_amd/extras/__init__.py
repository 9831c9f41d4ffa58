from . import shestakov_nonlinear_diffusion  # noqa: F401
from . import threefield  # noqa: F401
from . import bufferzone, noisyflux  # noqa: F401
