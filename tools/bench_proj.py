"""POD projection of a resident batch (B x n_f, Diamond shape): GB/s of the one HBM pass over X, per call, with HIP
events around back-to-back launches.  300 launches by default: a 20-launch run reads ~15-20 % low while the clocks ramp.
Usage (GPU box): python tools/bench_proj.py [B [launches]]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd')); sys.path.insert(0, ROOT)
import workloads as wl
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD, SROM_Q, SROM_RAW
from oracle import pod as opod

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = wl.diamond_c2()
L = _lib.lib()
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
n_f, r = w['U'].shape
X = wl.snapshots(w['q_ref'], B, seed=2)
dX = _lib.DeviceBuffer.from_array(X); dO = _lib.DeviceBuffer(B * r * 8)
e0, e1 = C.c_void_p(), C.c_void_p()
L.srh_event_create(C.byref(e0)); L.srh_event_create(C.byref(e1))
for which, name in ((SROM_Q, 'q (reference subtracted)'), (SROM_RAW, 'raw')):
    call = lambda: _lib.check(L.srom_project_dev(rom.handle, C.c_int(which), dX.ptr, C.c_int64(B), C.c_int64(n_f), dO.ptr,
                                                  C.c_int64(r), None), 'project')
    for _ in range(3):
        call()
    _lib.sync()
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    L.srh_event_record(e0, None)
    for _ in range(reps):
        call()
    L.srh_event_record(e1, None)
    _lib.sync()
    ms = C.c_float(); L.srh_event_elapsed_ms(e0, e1, C.byref(ms))
    t = ms.value / reps * 1e-3
    got = dO.to_array((B, r))
    ref = (X[:512] - (w['q_ref'] if which == SROM_Q else 0.0)) @ w['U']
    print('project %-26s B = %d: %.1f us, %.0f GB/s of X, max err (512 rows) %.2e'
          % (name, B, t * 1e6, X.nbytes / t / 1e9, np.abs(got[:512] - ref).max()))
