"""The r = 36 projection (the reference's SHIPPED Diamond basis size) for rocprofv3 --pmc passes: which pipe is busy beside the HBM stream.
Usage (GPU box): rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace ... -- python3 tools/pmc_proj_r36.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd')); sys.path.insert(0, ROOT)
import workloads as wl
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
L = _lib.lib()
B = 65536
for r in (30, 36):
    w = wl.diamond_c2(r=r)
    rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
    n_f = w['U'].shape[0]
    dX = _lib.DeviceBuffer.from_array(wl.snapshots(w['q_ref'], B, seed=2)); dXr = _lib.DeviceBuffer(B * r * 8)
    for _ in range(8):
        _lib.check(L.srom_project_dev(rom.handle, 0, dX.ptr, C.c_int64(B), C.c_int64(n_f), dXr.ptr, C.c_int64(r), None), 'project')
    _lib.sync()
    dX.free(); dXr.free()
print('done')
