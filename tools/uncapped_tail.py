"""Where the uncapped (500 SCP iterations, reference default) constructor solve of the C2 bench batch spends its wall time:
the whole batch, then its slowest rollouts one at a time (delta / omega trace of each: how many QPs ran with a small trust
region), and the same rollouts with SRH_GUSTO_NO_LEAN=1 semantics reported by the caller running the script twice.
Usage (GPU box, repo root): python tools/uncapped_tail.py [--rollouts 4096] [--top 6]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
torch.cuda.init()
import bench
import workloads as wl
from scipy.interpolate import interp1d
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


R_, top = arg('--rollouts', 4096), arg('--top', 6)
w = wl.diamond_c2()
N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
tp, gm = bench.build_model(w, 1354)
xc, fc = gm.get_characteristic_vals()
X = wl.snapshots(w['q_ref'], R_, seed=2)
x0 = np.concatenate((np.zeros((R_, r)), rom.compute_RO_state(qf=X)), axis=1)
u_init = np.zeros((R_, N, m))
x_init, _ = tp.rollout(x0, u_init, dt)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
z = np.stack([zi(b * 10.0 / R_ + dt * np.arange(N + 1)) for b in range(R_)])
kw = dict(U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), x_char=xc, f_char=fc, convg_thresh=1e-3)
print('library', _lib.LIB_PATH)
_lib.sync()
t0 = time.perf_counter()
g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, batch=R_, max_trace=0, **kw)
t_ctor = time.perf_counter() - t0
t0 = time.perf_counter()
g.solve_batch(x0, u_init, x_init, z=z)
t_again = time.perf_counter() - t0
it = g.iters.copy()
print('batch of %d: constructor %.3f s, second solve %.3f s; %d SCP iterations, max %d, status!=0 %d, %s' %
      (R_, t_ctor, t_again, it.sum(), it.max(), (g.status != 0).sum(), g.kernel_info))
order = np.argsort(-it)[:top]
del g
g1 = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[0], np.zeros((N, m)), x_init[0], z=z[0], max_trace=512, **kw)
for b in order:
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        g1.solve(x0[b], np.zeros((N, m)), x_init[b], z=z[b])
        ts.append(time.perf_counter() - t0)
    tr = g1.trace[0, :int(g1.iters[0])]
    small = int((tr[:, 1] < 100.0).sum())
    print('rollout %4d: %3d SCP iterations (%d in the batch), status %d, %.1f ms alone = %.2f ms per iteration; %d QPs with delta < 100 '
          '(min delta %.3g, max omega %.3g); handed over %s' % (b, int(g1.iters[0]), it[b], int(g1.status[0]), min(ts) * 1e3,
                                                                  min(ts) * 1e3 / int(g1.iters[0]), small, tr[:, 1].min(), tr[:, 2].max(),
                                                                  g1.kernel_info['handed_over']))
    if os.environ.get('SRH_GUSTO_TRACE_QIT'):
        print('   per SCP iteration [interior-point iterations + 1000 (pass + 1); lean iterations show rho]:', ' '.join('%g' % v for v in tr[:, 3]))
        print('   delta:', ' '.join('%.3g' % v for v in tr[:, 1]))
        print('   omega:', ' '.join('%.3g' % v for v in tr[:, 2]))
