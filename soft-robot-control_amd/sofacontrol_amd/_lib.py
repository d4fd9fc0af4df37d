"""ctypes binding of libsofacontrol_hip.so (C ABI: include/sofacontrol_hip.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRH_LIB_PATH: another build of the same library (e.g. a -DSRH_PROFILE build for tools/probes/*); default = the in-tree one
LIB_PATH = os.environ.get('SRH_LIB_PATH') or os.path.join(_HERE, 'libsofacontrol_hip.so')

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class HipError(RuntimeError):
    pass


class SLocpProblem(C.Structure):
    _fields_ = [('N', C.c_int), ('n_x', C.c_int), ('n_u', C.c_int), ('n_z', C.c_int),
                ('H', c_double_p), ('Qz', c_double_p), ('R', c_double_p), ('Qzf', c_double_p),
                ('x_scale', c_double_p),
                ('nU', C.c_int), ('UA', c_double_p), ('Ub', c_double_p),
                ('nX', C.c_int), ('XA', c_double_p), ('Xb', c_double_p),
                ('nXf', C.c_int), ('XfA', c_double_p), ('Xfb', c_double_p),
                ('ndU', C.c_int), ('dUA', c_double_p), ('dUb', c_double_p),
                ('tr_active', C.c_int)]


class SGustoParams(C.Structure):
    _fields_ = [('delta0', C.c_double), ('omega0', C.c_double), ('rho', C.c_double),
                ('beta_fail', C.c_double), ('gamma_fail', C.c_double), ('epsilon', C.c_double),
                ('omega_max', C.c_double), ('convg_thresh', C.c_double), ('max_gusto_iters', C.c_int)]


class SrhKernelInfo(C.Structure):
    """include/sofacontrol_hip.h: srh_kernel_info -- which kernels a LOCP / GuSTO plan launches."""
    _fields_ = [('family', C.c_int32), ('lean_args', C.c_int32 * 6), ('fused_args', C.c_int32 * 3),
                ('handed_over', C.c_int32), ('lds_bytes_lean', C.c_int32), ('lds_bytes_fused', C.c_int32),
                ('threads', C.c_int32)]

    def as_dict(self):
        """{'family': 'lean' | 'fused', 'kernel': the name rocprof shows, 'lean': (n_u, n_x, GX, N, j0, state rows) | None,
        'fused': (split, n_u, n_x), 'handed_over': problems of the last solve the lean kernel passed on (-1: none yet)}."""
        lean = tuple(int(v) for v in self.lean_args) if self.family == 1 else None
        fused = (bool(self.fused_args[0]), int(self.fused_args[1]), int(self.fused_args[2]))
        kern = ('lean<%d, %d, %d, %d, %d, %d>' % lean) if lean else ('fused<%s, %d, %d>' % (str(fused[0]).lower(), fused[1], fused[2]))
        return {'family': 'lean' if self.family == 1 else 'fused', 'kernel': kern, 'lean': lean, 'fused': fused,
                'handed_over': int(self.handed_over), 'lds_bytes_lean': int(self.lds_bytes_lean),
                'lds_bytes_fused': int(self.lds_bytes_fused), 'threads': int(self.threads)}


class SIlqrParams(C.Structure):
    _fields_ = [('max_iter', C.c_int), ('epsilon', C.c_double), ('alpha0', C.c_double),
                ('alpha_scaling', C.c_double), ('improv_lb', C.c_double), ('improv_ub', C.c_double),
                ('alpha_min', C.c_double), ('counter_limit', C.c_int), ('rho0', C.c_double),
                ('drho0', C.c_double), ('rho_scaling', C.c_double), ('rho_increase_fp', C.c_double),
                ('rho_max', C.c_double), ('rho_min', C.c_double), ('include_input_var_constraint', C.c_int),
                ('do_linesearch', C.c_int), ('regularize', C.c_int), ('state_regularization', C.c_int)]


_lib = None


def lib():
    """Load the shared library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise HipError('libsofacontrol_hip.so is missing (%s): run `python -c "import '
                           '__graft_entry__ as g; g.build()"` or `make -C soft-robot-control_amd/csrc`; '
                           'there is no CPU fallback' % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.srh_last_error.restype = C.c_char_p
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().srh_last_error().decode(errors='replace')
        if rc == -1:
            raise RuntimeError('%s: %s' % (what, msg))   # the reference raises RuntimeError on bad args
        raise HipError('%s failed (code %d): %s' % (what, rc, msg))


def dptr(a):
    """double* of a C-contiguous float64 array (or NULL for None)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(c_int32_p)


def f64(a):
    """C-contiguous float64 copy/view of a (None passes through)."""
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    rc = lib().srh_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(i):
    check(lib().srh_set_device(C.c_int(i)), 'srh_set_device')


class DeviceBuffer:
    """A raw HBM allocation made through the C ABI (for callers that want resident inputs)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().srh_malloc(C.byref(self.ptr), C.c_size_t(self.nbytes)), 'srh_malloc')

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(lib().srh_memcpy_h2d(b.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)), 'h2d')
        return b

    def to_array(self, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().srh_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(out.nbytes)), 'd2h')
        return out

    def free(self):
        if self.ptr:
            lib().srh_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync():
    check(lib().srh_sync(), 'srh_sync')
