"""Micro-benchmark of the batched POD projection kernel (HBM roofline)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'soft-robot-control_amd'))
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD

r = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n_f = int(sys.argv[3]) if len(sys.argv) > 3 else 4884
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
U, _ = np.linalg.qr(rng.standard_normal((n_f, r)))
q_ref = rng.uniform(-108, 107, n_f)
rom = POD(dict(U=U, q_ref=q_ref, v_ref=np.zeros(n_f)))
X = (q_ref + 5 * rng.standard_normal((B, n_f)))
dX = _lib.DeviceBuffer.from_array(X)
dO = _lib.DeviceBuffer(B * r * 8)
L = _lib.lib()
e0, e1 = C.c_void_p(), C.c_void_p()
L.srh_event_create(C.byref(e0)); L.srh_event_create(C.byref(e1))
def run(n):
    for _ in range(n):
        _lib.check(L.srom_project_dev(rom.handle, 0, dX.ptr, C.c_int64(B), C.c_int64(n_f), dO.ptr, C.c_int64(r), None), 'proj')
run(3); _lib.sync()
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
L.srh_event_record(e0, None); run(iters); L.srh_event_record(e1, None)
ms = C.c_float(); L.srh_event_elapsed_ms(e0, e1, C.byref(ms))
t = ms.value / iters * 1e-3
bytes_alg = B * n_f * 8 + n_f * r * 8 + n_f * 8 + B * r * 8
print('B=%d r=%d: %.3f ms/launch, %.1f GB/s algorithmic (%.1f%% of 8 TB/s), %.1f TFLOP/s' %
      (B, r, t * 1e3, bytes_alg / t / 1e9, bytes_alg / t / 8e12 * 100, 2.0 * B * n_f * r / t / 1e12))
out = dO.to_array((B, r))
ref = (X[:256] - q_ref) @ U
print('max err', np.abs(out[:256] - ref).max())
