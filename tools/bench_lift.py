"""POD lift (compute_FO_state batch) and the r = 36 projection on resident buffers: GB/s of the algorithmic bytes."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.')
import workloads as wl
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
L = _lib.lib()
B = 65536
n_f = int(sys.argv[1]) if len(sys.argv) > 1 else 4884
e0, e1 = C.c_void_p(), C.c_void_p()
L.srh_event_create(C.byref(e0)); L.srh_event_create(C.byref(e1))
def timed(fn, reps=300):     # long enough for the clocks to ramp (20 launches read 20 % low)
    for _ in range(50): fn()
    _lib.sync(); L.srh_event_record(e0, None)
    for _ in range(reps): fn()
    L.srh_event_record(e1, None); _lib.sync()
    ms = C.c_float(); L.srh_event_elapsed_ms(e0, e1, C.byref(ms)); return ms.value / reps * 1e-3
for r in (30, 36):
    U, q_ref, v_ref = wl.pod_basis(n_f, r, seed=0)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    X = wl.snapshots(q_ref, B, seed=2)
    dX = _lib.DeviceBuffer.from_array(X); dXr = _lib.DeviceBuffer(B * r * 8)
    t = timed(lambda: _lib.check(L.srom_project_dev(rom.handle, 0, dX.ptr, C.c_int64(B), C.c_int64(n_f), dXr.ptr, C.c_int64(r), None), 'p'))
    byt = 8 * (B * n_f + n_f * r + n_f + B * r)
    print('project r=%d: %.3f ms, %.0f GB/s (%.1f %% of 8 TB/s)' % (r, t * 1e3, byt / t / 1e9, byt / t / 8e10))
    t = timed(lambda: _lib.check(L.srom_lift_dev(rom.handle, 0, dXr.ptr, C.c_int64(B), C.c_int64(r), dX.ptr, C.c_int64(n_f), None), 'l'))
    print('lift    r=%d: %.3f ms, %.0f GB/s (%.1f %% of 8 TB/s)' % (r, t * 1e3, byt / t / 1e9, byt / t / 8e10))
    Xr = dXr.to_array((4, r)); Xf = dX.to_array((4, n_f))
    print('   lift check', float(np.abs(Xf - (Xr @ U.T + q_ref)).max()))
    dX.free(); dXr.free()
