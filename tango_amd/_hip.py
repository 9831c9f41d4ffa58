"""ctypes binding of libtango_hip.so (declared in include/tango_hip.h) and a minimal device array.

There is no CPU fallback: importing this module without the built library raises, and every
compute call on a machine without a GPU raises HipError from the first HIP call."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TANGO_HIP_LIB') or os.path.join(_HERE, 'lib', 'libtango_hip.so')

TG_OK, TG_ERR_ARG, TG_ERR_HIP, TG_ERR_UNSUPPORTED = 0, 1, 2, 3
POINT_FASTEST, BATCH_FASTEST, BATCH_FASTEST_TILED = 0, 1, 2
ST_CONVERGED, ST_INVALID, ST_SINGULAR, ST_SEEDED, ST_TREAD_LIMIT = 1, 2, 4, 8, 16
FLUX_SHESTAKOV, FLUX_EXTERNAL, FLUX_POWER = 0, 1, 2
STATIC_WORK_BYTES = 24 * 1024 * 1024


class HipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: build it with `python tango_amd/build.py` (hipcc, gfx950). '
        'tango_amd has no CPU fallback.')

lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_i64 = C.c_int64
_dbl = C.c_double
_int = C.c_int
_sz = C.c_size_t


class HCoeffsC(C.Structure):
    _fields_ = [(k, _p) for k in ('H1', 'H2', 'H3', 'H4', 'H6', 'H7', 'H8')] + \
               [(k, _i64) for k in ('s1', 's2', 's3', 's4', 's6', 's7', 's8')]


class State1F(C.Structure):
    _fields_ = [('nb', _i64), ('n', _i64), ('u', _p), ('um1', _p), ('uE', _p), ('DE', _p), ('cE', _p),
                ('status', _p), ('iters', _p), ('err', _p), ('err_stride', _i64)]


class Params1F(C.Structure):
    _fields_ = [('dt', _dbl), ('dx', _dbl), ('a_turb', _dbl), ('a_prof', _dbl), ('Dmin', _dbl), ('Dmax', _dbl),
                ('dpdx_thr', _dbl), ('tol', _dbl), ('UL', _p), ('dt_b', _p), ('H1', _p), ('H7', _p),
                ('h7_stride', _i64), ('S0', _p), ('src_cut', _p), ('Vprime', _p), ('gxx', _p),
                ('gradx', _p), ('gxx_turb', _p), ('gradx_turb', _p), ('flux', _p), ('flux_model', C.c_int32),
                ('smooth_window', C.c_int32), ('fixed_count', C.c_int32), ('frozen_turb', C.c_int32),
                ('iters_per_launch', C.c_int32), ('pseudo_time', C.c_int32), ('tread_lightly', C.c_int32),
                ('tread_inner_count', C.c_int32), ('tread_initial_steps', C.c_int32), ('reserved_', C.c_int32),
                ('tread_max_change', _dbl), ('tread_initial_dtau', _dbl), ('work', _p), ('work_bytes', _sz),
                ('iterates', _p), ('iterates_rows', _i64), ('flux_mult', _p), ('flux_mult_stride', _i64), ('flux_power', _dbl)]


class State3F(C.Structure):
    _fields_ = [('nb', _i64), ('n', _i64), ('u', _p * 3), ('um1', _p * 3), ('uE', _p * 3), ('DE', _p * 3),
                ('cE', _p * 3), ('status', _p), ('iters', _p), ('err', _p), ('err_stride', _i64)]


class Params3F(C.Structure):
    _fields_ = [('dt', _dbl), ('dx', _dbl), ('a_turb', _dbl), ('a_prof', _dbl), ('Dmin', _dbl), ('Dmax', _dbl),
                ('dpdx_thr', _dbl), ('tol', _dbl), ('UL', _p * 3), ('H7', _p * 3), ('nu0', _p), ('fixed_count', C.c_int32),
                ('frozen_turb', C.c_int32), ('outer_count', C.c_int32), ('reserved_', C.c_int32), ('pseudo_time', C.c_int32),
                ('tread_lightly', C.c_int32), ('tread_inner_count', C.c_int32), ('tread_initial_steps', C.c_int32),
                ('tread_max_change', _dbl), ('tread_initial_dtau', _dbl), ('flux_model', C.c_int32), ('reserved2_', C.c_int32),
                ('flux', _p * 3), ('H6', _p * 3), ('H8', _p * 3), ('h6_stride', _i64 * 3), ('h7_stride', _i64 * 3),
                ('h8_stride', _i64 * 3)]


# every exported symbol of include/tango_hip.h with its prototype
PROTOTYPES = {
    'tg_version': (_int, []),
    'tg_last_error': (C.c_char_p, []),
    'tg_device_count': (_int, [C.POINTER(_int)]),
    'tg_set_device': (_int, [_int]),
    'tg_malloc': (_int, [C.POINTER(_p), _sz]),
    'tg_free': (_int, [_p]),
    'tg_memset': (_int, [_p, _int, _sz, _p]),
    'tg_copy_h2d': (_int, [_p, _p, _sz, _p]),
    'tg_copy_d2h': (_int, [_p, _p, _sz, _p]),
    'tg_copy_d2d': (_int, [_p, _p, _sz, _p]),
    'tg_copy_d2d_2d': (_int, [_p, _sz, _p, _sz, _sz, _sz, _p]),
    'tg_stream_create': (_int, [C.POINTER(_p)]),
    'tg_stream_destroy': (_int, [_p]),
    'tg_stream_sync': (_int, [_p]),
    'tg_stream_probe': (_int, [_p * 5, _p * 4, _i64, _p]),
    'tg_tridiag_work_bytes': (_sz, [_i64, _i64, _int]),
    'tg_picard_1f_work_bytes': (_sz, [_i64, _i64]),
    'tg_picard_1f_work_bytes_tread': (_sz, [_i64, _i64]),
    'tg_spline_work_bytes': (_sz, [_i64, _i64]),
    'tg_spline_resample': (_int, [_p, _p, _i64, _i64, _p, _i64, _i64, _i64, _int, _p, _p, _sz, _p]),
    'tg_tridiag_solve': (_int, [_p, _p, _p, _p, _p, _i64, _i64, _int, _p, _p, _sz, _p]),
    'tg_h_to_matrix': (_int, [_dbl, _p, _dbl, _p, _p, C.POINTER(HCoeffsC), _p, _p, _p, _p, _p, _i64, _i64, _p]),
    'tg_residual': (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    'tg_ewma': (_int, [_p, _p, _dbl, _int, _i64, _p]),
    'tg_ewma_profiles': (_int, [_p, _p, _dbl, _p, _i64, _i64, _p]),
    'tg_host_alloc': (_int, [C.POINTER(_p), _sz]),
    'tg_host_free': (_int, [_p]),
    'tg_flux_split': (_int, [_p, _p, _dbl, _dbl, _dbl, _dbl, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    'tg_ddx': (_int, [_p, _dbl, _int, _p, _i64, _i64, _p]),
    'tg_moving_average': (_int, [_p, _int, _p, _i64, _i64, _p]),
    'tg_zone_apply': (_int, [_p, _i64, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _int, _p]),
    'tg_tukey_window': (_int, [_p, _i64, _dbl, _p]),
    'tg_ar1_noise': (_int, [C.c_uint64, C.c_uint64, _dbl, _dbl, C.c_int32, _p, _p, _i64, _i64, _p]),
    'tg_shestakov_flux': (_int, [_p, _dbl, _p, _i64, _i64, _p]),
    'tg_pair_solve': (_int, [_p] * 12 + [_i64, _i64, _p, _p]),
    'tg_block3_solve': (_int, [_p, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    'tg_picard_1f_step': (_int, [C.POINTER(State1F), C.POINTER(Params1F), _p]),
    'tg_picard_1f_step_static': (_int, [C.POINTER(State1F), C.POINTER(Params1F), _p]),
    'tg_status_reduce': (_int, [_p, _i64, _p, _p]),
    'tg_status_or': (_int, [_p, _i64, C.c_int32, _p]),
    'tg_residual_reduce': (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p]),
    'tg_picard_3f_step': (_int, [C.POINTER(State3F), C.POINTER(Params3F), _p]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def check(code):
    if code != TG_OK:
        msg = lib.tg_last_error().decode(errors='replace')
        if code == TG_ERR_ARG:
            raise ValueError(msg)
        if code == TG_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise HipError(msg)


def device_count():
    n = _int(0)
    code = lib.tg_device_count(C.byref(n))
    return n.value if code == TG_OK else 0


_stream = None   # current stream handle (None = default stream)


def set_stream(handle):
    """Use the given hipStream_t (an int, e.g. torch.cuda.current_stream().cuda_stream) for all calls."""
    global _stream
    _stream = handle


def stream():
    return _stream


def synchronize():
    check(lib.tg_stream_sync(_stream))


class DeviceArray:
    """A float64 / int32 / int64 array in device memory owned by this object."""

    def __init__(self, shape, dtype=np.float64, zero=False, at=None, owner=None):
        """at=None: a new allocation owned by this object.  at=<device address>: a window into memory that `owner`
        keeps alive (a pool): nothing is allocated or freed here."""
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.owner = owner
        if at is None:
            ptr = _p()
            check(lib.tg_malloc(C.byref(ptr), self.nbytes))
            self.ptr = ptr.value
            self._owns = True
        else:
            self.ptr = int(at)
            self._owns = False
        if zero:
            check(lib.tg_memset(self.ptr, 0, self.nbytes, _stream))

    @classmethod
    def from_host(cls, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype or (a.dtype if isinstance(a, np.ndarray) else np.float64))
        d = cls(a.shape, a.dtype)
        d.set(a)
        return d

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != self.shape:
            a = np.ascontiguousarray(np.broadcast_to(a, self.shape))
        check(lib.tg_copy_h2d(self.ptr, a.ctypes.data, self.nbytes, _stream))
        check(lib.tg_stream_sync(_stream))   # the host buffer may go away after return

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        check(lib.tg_copy_d2h(out.ctypes.data, self.ptr, self.nbytes, _stream))
        check(lib.tg_stream_sync(_stream))
        return out

    def columns_to_host(self, k):
        """The first k columns of a 2-D array as a C-contiguous host array (packed on the device first)."""
        rows, cols = self.shape
        k = max(0, min(int(k), cols))
        if k == cols:
            return self.to_host()
        if k == 0:
            return np.empty((rows, 0), self.dtype)
        item = np.dtype(self.dtype).itemsize
        packed = DeviceArray((rows, k), self.dtype)
        check(lib.tg_copy_d2d_2d(packed.ptr, k * item, self.ptr, cols * item, k * item, rows, _stream))
        return packed.to_host()

    def copy_from(self, other):
        assert other.nbytes == self.nbytes
        check(lib.tg_copy_d2d(self.ptr, ptr_of(other), self.nbytes, _stream))

    def data_ptr(self):
        return self.ptr

    @property
    def __cuda_array_interface__(self):
        """Lets torch.as_tensor(array, device='cuda') (and CuPy-style consumers) view the memory without a copy."""
        return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (int(self.ptr or 0), False), 'version': 2, 'strides': None}

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __del__(self):
        p, self.ptr = getattr(self, 'ptr', None), None
        if p and getattr(self, '_owns', True):
            try:
                lib.tg_free(p)
            except Exception:
                pass


def ptr_of(a):
    """Device address of a DeviceArray, a CUDA/HIP torch tensor, or None."""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    if hasattr(a, 'ptr') and hasattr(a, 'owner'):      # ensemble.DeviceView
        return a.ptr
    if hasattr(a, 'data_ptr'):
        if hasattr(a, 'is_cuda') and not a.is_cuda:
            raise ValueError('tensor is not on the GPU')
        return int(a.data_ptr())
    raise TypeError(f'not a device array: {type(a)!r}')


def is_device(a):
    return isinstance(a, DeviceArray) or (hasattr(a, 'data_ptr') and getattr(a, 'is_cuda', False))


class PinnedArray:
    """A host array in page-locked memory (copies to and from the device run at full PCIe rate and asynchronously);
    `.array` is the NumPy view."""

    def __init__(self, shape, dtype=np.float64):
        self.shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        ptr = _p()
        check(lib.tg_host_alloc(C.byref(ptr), max(self.nbytes, 8)))
        self.ptr = ptr.value
        buf = (C.c_char * max(self.nbytes, 8)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape, dtype=np.int64))).reshape(self.shape)

    def upload(self, dev):
        """Asynchronous copy into a device array (the caller keeps this buffer unchanged until the stream has passed it)."""
        check(lib.tg_copy_h2d(ptr_of(dev), self.ptr, self.nbytes, _stream))

    def download(self, dev):
        check(lib.tg_copy_d2h(self.ptr, ptr_of(dev), self.nbytes, _stream))
        check(lib.tg_stream_sync(_stream))
        return self.array

    def __del__(self):
        p, self.ptr = getattr(self, 'ptr', None), None
        self.array = None
        if p:
            try:
                lib.tg_host_free(p)
            except Exception:
                pass
