"""Launches the MFMA kernels of the path (POD projection, one-pass U^T M U, snapshot Gramian) at their bench shapes for
rocprofv3 --pmc passes (profiles/r01_mfma_util.*, profiles/r02_*_pmc.*)."""
import ctypes as C, os, sys
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
B = 65536
dX = _lib.DeviceBuffer.from_array(wl.snapshots(w['q_ref'], B, seed=2)); dXr = _lib.DeviceBuffer(B * r * 8)
for _ in range(6):
    _lib.check(L.srom_project_dev(rom.handle, 0, dX.ptr, C.c_int64(B), C.c_int64(n_f), dXr.ptr, C.c_int64(r), None), 'project')
_lib.sync()
dM = _lib.DeviceBuffer.from_array(np.random.default_rng(0).standard_normal((n_f, n_f))); dP = _lib.DeviceBuffer(r * r * 8)
for _ in range(6):
    _lib.check(L.srom_reduce_matrix_dev(rom.handle, dM.ptr, C.c_int64(n_f), 1, 1, dP.ptr, None), 'reduce')
_lib.sync()
# four matrices in one launch pair (srom_reduce_matrices_dev: K, D, M, S of one TPWL point)
Ms = [dM] + [_lib.DeviceBuffer.from_array(np.random.default_rng(1 + i).standard_normal((n_f, n_f))) for i in range(3)]
Ps = [_lib.DeviceBuffer(r * r * 8) for _ in range(4)]
PP = C.c_void_p * 4
mp, op = PP(*[b.ptr for b in Ms]), PP(*[b.ptr for b in Ps])
for _ in range(6):
    _lib.check(L.srom_reduce_matrices_dev(rom.handle, mp, 4, op, None), 'reduce4')
_lib.sync()
for b in Ms[1:]:
    b.free()
n_s, nf2 = 10000, 6250
dS = _lib.DeviceBuffer.from_array(np.random.default_rng(7).standard_normal((n_s, nf2))); dG = _lib.DeviceBuffer(n_s * n_s * 8)
for _ in range(4):
    _lib.check(L.srom_gramian_dev(dS.ptr, C.c_int64(n_s), C.c_int64(nf2), C.c_int64(nf2), dG.ptr, None), 'gramian')
_lib.sync()
print('done')
