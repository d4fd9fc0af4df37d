"""Latency probe of the per-step projection: python wrapper vs bare C-ABI call vs device-resident call."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.')
import workloads as wl
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
w = wl.diamond_c2()
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
n_f, r = w['U'].shape
x = np.ascontiguousarray(np.concatenate((w['v_ref'], w['q_ref'])) + 1.0)
out = np.empty(2 * r)
L = _lib.lib()
def t(fn, reps=500):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
print('python compute_RO_state(xf)  %.1f us' % t(lambda: rom.compute_RO_state(xf=x)))
xp, op = _lib.dptr(x), _lib.dptr(out)
print('bare srom_project (host ptr) %.1f us' % t(lambda: L.srom_project(rom.handle, 2, xp, C.c_int64(1), op)))
dx = _lib.DeviceBuffer.from_array(x); do = _lib.DeviceBuffer(2 * r * 8)
def dev():
    L.srom_project_dev(rom.handle, 2, dx.ptr, C.c_int64(1), C.c_int64(2 * n_f), do.ptr, C.c_int64(2 * r), None)
    L.srh_sync()
print('srom_project_dev + sync      %.1f us' % t(dev))
def dev_nosync():
    L.srom_project_dev(rom.handle, 2, dx.ptr, C.c_int64(1), C.c_int64(2 * n_f), do.ptr, C.c_int64(2 * r), None)
print('srom_project_dev (enqueue)   %.1f us' % t(dev_nosync)); L.srh_sync()
xq = np.ascontiguousarray(w['q_ref'] + 1.0); oq = np.empty(r)
print('bare srom_project q only     %.1f us' % t(lambda: L.srom_project(rom.handle, 0, _lib.dptr(xq), C.c_int64(1), _lib.dptr(oq))))
# EKF step
import scipy.sparse as sp, io, contextlib
sys.path.insert(0, 'tests')
from bench import build_model
tp, gm = build_model(w)
nodes = np.arange(0, 10 * 150, 150)
Cf = sp.lil_matrix((30, 2 * n_f))
for i, nd in enumerate(nodes):
    for a in range(3):
        Cf[3 * i + a, n_f + 3 * nd + a] = 1.0
tp.set_measurement_model(Cf.tocsr())
from sofacontrol_amd.tpwl.observer import DiscreteEKFObserver
ekf = DiscreteEKFObserver(tp)
u = np.full(w['m'], 100.0); y = tp.y_ref + 0.01
print('python ekf.update            %.1f us' % t(lambda: ekf.update(u, y, w['dt']), 200))
up, yp, xo = _lib.dptr(u), _lib.dptr(y), np.empty(2 * r)
print('bare sekf_step               %.1f us' % t(lambda: L.sekf_step(ekf._h, up, yp, None, None, None, _lib.dptr(xo)), 200))
