"""SCP kernel under load: identical C2 rollouts (same x0, same target: same iteration count in every workgroup) for a
growing number of concurrent workgroups -- ms per SCP iteration of one workgroup.  Separates what a rollout costs
alone from what it costs with all 256 CUs busy (clock, L2 footprint of the per-rollout workspaces)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd'))
import workloads as wl
import bench
from scipy.interpolate import interp1d
from sofacontrol_amd import _lib
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron

w = wl.diamond_c2()
N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
tp, gm = bench.build_model(w)
xc, fc = gm.get_characteristic_vals()
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
for R in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 256, 512, 1024]:
    x0 = np.zeros((R, 2 * r)); x0[:, r:] = 0.05
    u_init = np.zeros((R, N, m))
    x_init, _ = tp.rollout(x0, u_init, dt)
    z = np.stack([zi(1.0 + dt * np.arange(N + 1))] * R)
    g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']),
              X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, batch=R, max_trace=0, max_gusto_iters=5)
    g.max_gusto_iters = 5
    ts = []
    for _ in range(3):
        _lib.sync(); t0 = time.perf_counter()
        g.solve_batch(x0, u_init, x_init, z=z)
        ts.append(time.perf_counter() - t0)
    it = int(g.iters[0])
    waves = -(-R // 256)
    print('R=%5d: %8.2f ms per solve, %d SCP iterations per rollout, %6.2f ms per iteration per workgroup (%d round(s))' % (
        R, min(ts) * 1e3, it, min(ts) * 1e3 / it / waves, waves))
