"""U^T M U on a resident dense M (n_f x n_f): GB/s of the one HBM pass over M (191 MB at n_f = 4884)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd')); sys.path.insert(0, ROOT)
import workloads as wl
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
w = wl.diamond_c2()
L = _lib.lib()
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
n_f, r = w['U'].shape
M = np.random.default_rng(0).standard_normal((n_f, n_f))
dM = _lib.DeviceBuffer.from_array(M); dO = _lib.DeviceBuffer(r * r * 8)
e0, e1 = C.c_void_p(), C.c_void_p()
L.srh_event_create(C.byref(e0)); L.srh_event_create(C.byref(e1))
for _ in range(3):
    _lib.check(L.srom_reduce_matrix_dev(rom.handle, dM.ptr, C.c_int64(n_f), 1, 1, dO.ptr, None), 'reduce')
_lib.sync()
L.srh_event_record(e0, None)
reps = 10
for _ in range(reps):
    _lib.check(L.srom_reduce_matrix_dev(rom.handle, dM.ptr, C.c_int64(n_f), 1, 1, dO.ptr, None), 'reduce')
L.srh_event_record(e1, None)
_lib.sync()
ms = C.c_float(); L.srh_event_elapsed_ms(e0, e1, C.byref(ms))
t = ms.value / reps * 1e-3
got = dO.to_array((r, r))
ref = w['U'].T @ M @ w['U']
print('U^T M U: %.1f us per matrix, %.0f GB/s of M, max err %.2e' % (t * 1e6, M.nbytes / t / 1e9, np.abs(got - ref).max()))

# four matrices per launch pair (srom_reduce_matrices_dev): K, D, M, S of one TPWL point (tpwl/tpwl_utils.py:96-103)
Ms = [_lib.DeviceBuffer.from_array(np.random.default_rng(1 + i).standard_normal((n_f, n_f))) for i in range(4)]
Os = [_lib.DeviceBuffer(r * r * 8) for _ in range(4)]
PP = C.c_void_p * 4
mp, op = PP(*[b.ptr for b in Ms]), PP(*[b.ptr for b in Os])
for _ in range(3):
    _lib.check(L.srom_reduce_matrices_dev(rom.handle, mp, 4, op, None), 'reduce4')
_lib.sync()
L.srh_event_record(e0, None)
for _ in range(reps):
    _lib.check(L.srom_reduce_matrices_dev(rom.handle, mp, 4, op, None), 'reduce4')
L.srh_event_record(e1, None)
_lib.sync()
L.srh_event_elapsed_ms(e0, e1, C.byref(ms))
t4 = ms.value / reps * 1e-3
alg = 4 * (n_f * n_f + 2 * n_f * r + r * r) * 8
print('U^T M U x 4 in one launch pair: %.1f us (%.1f us per matrix), %.0f GB/s algorithmic = %.3f of 8 TB/s' %
      (t4 * 1e6, t4 * 1e6 / 4, alg / t4 / 1e9, alg / t4 / 8e12))
