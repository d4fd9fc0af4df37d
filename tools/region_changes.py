import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'soft-robot-control_amd')
import workloads as wl, bench
from scipy.interpolate import interp1d
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron
w = wl.diamond_c2(); N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
tp, gm = bench.build_model(w); xc, fc = gm.get_characteristic_vals()
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
R = 64
rom_q = np.random.default_rng(2).standard_normal((R, r)) * 3
x0 = np.concatenate((np.zeros((R, r)), rom_q), axis=1)
u_init = np.zeros((R, N, m)); x_init, _ = tp.rollout(x0, u_init, dt)
z = np.stack([zi(b * 10.0 / R + dt * np.arange(N + 1)) for b in range(R)])
g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3, batch=R, max_trace=0, max_gusto_iters=5)
ch = []
for b in range(R):
    idx = np.asarray(tp.calc_nearest_point(g.xopt[b][:N]))
    ch.append(int((np.diff(idx) != 0).sum()))
print('region changes per 50-stage horizon: mean %.1f min %d max %d' % (np.mean(ch), min(ch), max(ch)), 'iters', g.iters[:8])
