"""C4 per-GPU shard: G = S S^T (10 000 x 6250 f64) through srom_gramian_dev, executed TFLOP/s (upper triangle of tiles)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.')
from sofacontrol_amd import _lib
L = _lib.lib()
n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_f = int(sys.argv[2]) if len(sys.argv) > 2 else 6250
rng = np.random.default_rng(0)
S = rng.standard_normal((n_s, 64)) @ rng.standard_normal((64, n_f)) + 1e-3 * rng.standard_normal((n_s, n_f))
dS = _lib.DeviceBuffer.from_array(S); dG = _lib.DeviceBuffer(n_s * n_s * 8)
e0, e1 = C.c_void_p(), C.c_void_p()
L.srh_event_create(C.byref(e0)); L.srh_event_create(C.byref(e1))
def run():
    _lib.check(L.srom_gramian_dev(dS.ptr, C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), dG.ptr, None), 'gramian')
for _ in range(5): run()
_lib.sync(); L.srh_event_record(e0, None)
reps = 20
for _ in range(reps): run()
L.srh_event_record(e1, None); _lib.sync()
ms = C.c_float(); L.srh_event_elapsed_ms(e0, e1, C.byref(ms))
t = ms.value / reps * 1e-3
flop = float(n_s) * (n_s + 128) * n_f
print('n_s=%d n_f=%d: %.2f ms, %.1f TFLOP/s executed (%.1f %% of 78.6)' % (n_s, n_f, t * 1e3, flop / t / 1e12, flop / t / 78.6e10))
G = dG.to_array((n_s, n_s))[:64, :200]
print('check', float(np.abs(G - S[:64] @ S[:200].T).max() / np.abs(G).max()))
