import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd')]
import torch; torch.cuda.init()
import bench, workloads as wl
from scipy.interpolate import interp1d
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron
_lib.set_device(0)
w = wl.diamond_c2(); r = w['r']
N, m, dt = w['N'], w['m'], w['dt']
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
tp, gm = bench.build_model(w)
xc, fc = gm.get_characteristic_vals()
X = wl.snapshots(w['q_ref'], 4, seed=2)
x0 = np.concatenate((np.zeros((4, r)), rom.compute_RO_state(qf=X)), axis=1)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
b = 0
u_init = np.zeros((1, N, m)); x_init, _ = tp.rollout(x0[b:b + 1], u_init, dt)
z = np.stack([zi(dt * np.arange(N + 1))])
t0 = time.perf_counter()
g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[b:b + 1], u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']),
          X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, batch=1, max_trace=8, max_gusto_iters=5)
print('constructor (incl. first solve): %.1f ms' % ((time.perf_counter() - t0) * 1e3))
time.sleep(float(os.environ.get("PRESLEEP", "0")))
for k in range(12):
    t0 = time.perf_counter(); g.solve_batch(x0[b:b + 1], u_init, x_init, z=z); t = time.perf_counter() - t0
    print('solve %d: %.2f ms' % (k, t * 1e3))
