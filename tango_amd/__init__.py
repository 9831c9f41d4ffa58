"""tango_amd -- MI355X-native implementation of Tango's implicit transport inner loop.

Drop-in for the hot-path modules of sundials-codes/tango (`import tango_amd as tango`): tridiag,
HToMatrixFD, JK_to_matrix_fd, lodestro_method, derivatives, smoother, multifield, fieldgroups, solver,
extras.shestakov_nonlinear_diffusion, restart, tango_logging.  All numerics run in hand-written HIP kernels
(tango_amd/csrc, built into tango_amd/lib/libtango_hip.so and bound with ctypes); there is no CPU
fallback: importing the package without the built library fails."""
from . import _hip  # noqa: F401  (raises ImportError when libtango_hip.so is missing)
from ._hip import DeviceArray, HipError, device_count, set_stream, synchronize  # noqa: F401
from . import tango_logging  # noqa: F401
from . import tridiag, derivatives, smoother  # noqa: F401
from . import HToMatrixFD, JK_to_matrix_fd  # noqa: F401
from . import multifield, lodestro_method, fieldgroups  # noqa: F401
from . import solver  # noqa: F401
from . import extras  # noqa: F401
from . import ensemble  # noqa: F401
from . import restart  # noqa: F401
from . import interfacegrids_gene  # noqa: F401

__version__ = '0.1.0'
