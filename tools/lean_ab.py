"""A/B measurement of one build of the lean GuSTO kernel (SRH_LIB_PATH selects the library): BASELINE C2 (or C5 with `c5`),
  * 4096 (C5: 256) rollouts, capped solves, device-resident (sgusto_plan_solve_dev): SCP iterations/s, best of 3;
  * one rollout at a time: ms per SCP iteration (median of 8 rollouts);
  * a fingerprint of the results (sum of SCP iterations, status counts, norms of the trajectories) -- two builds that differ
    in scheduling only must print the same fingerprint to ~1e-9;
  * optionally (`--check`) the first 3 rollouts against the numpy oracle.
Usage (GPU box, repo root): python tools/lean_ab.py [c2|c5] [--rollouts N] [--check] [--skip-big]"""
import ctypes as C
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

which = 'c5' if 'c5' in sys.argv[1:] else 'c2'
R_ = 4096 if which == 'c2' else 256
if '--rollouts' in sys.argv:
    R_ = int(sys.argv[sys.argv.index('--rollouts') + 1])
w = wl.diamond_c2() if which == 'c2' else wl.trunk_c5()
tip = 1354 if which == 'c2' else w['tip_node']
N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
n, nz = 2 * r, 6
L = _lib.lib()
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
tp, gm = bench.build_model(w, tip)
xc, fc = gm.get_characteristic_vals()
X = wl.snapshots(w['q_ref'], R_, seed=2 if which == 'c2' else 9)
x0 = np.concatenate((np.zeros((R_, r)), rom.compute_RO_state(qf=X)), axis=1)
u_init = np.zeros((R_, N, m))
x_init, _ = tp.rollout(x0, u_init, dt)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
z = np.stack([zi(b * 10.0 / R_ + dt * np.arange(N + 1)) for b in range(R_)])
Xp = Polyhedron(w['XA'], w['Xb']) if w['XA'] is not None else None
print('library', _lib.LIB_PATH)
if '--skip-big' not in sys.argv:
    # the constructor solve capped as well (the uncapped one costs a second and is not what is compared here)
    g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Xp, x_char=xc, f_char=fc,
              convg_thresh=1e-3, batch=R_, max_trace=0, max_gusto_iters=5)
    d = {k: _lib.DeviceBuffer.from_array(v) for k, v in dict(x0=x0, u_init=u_init, x_init=x_init, z=z).items()}
    o = dict(xopt=_lib.DeviceBuffer(R_ * (N + 1) * n * 8), uopt=_lib.DeviceBuffer(R_ * N * m * 8), zopt=_lib.DeviceBuffer(R_ * (N + 1) * nz * 8),
             iters=_lib.DeviceBuffer(R_ * 4), status=_lib.DeviceBuffer(R_ * 4))
    ts = []
    for rep in range(4):
        _lib.sync()
        t0 = time.perf_counter()
        _lib.check(L.sgusto_plan_solve_dev(g.plan, d['x0'].ptr, d['u_init'].ptr, d['x_init'].ptr, d['z'].ptr, None, None, o['xopt'].ptr,
                                           o['uopt'].ptr, o['zopt'].ptr, o['iters'].ptr, o['status'].ptr, None, None), 'solve')
        _lib.sync()
        ts.append(time.perf_counter() - t0)
    it = o['iters'].to_array((R_,), dtype=np.int32)
    st = o['status'].to_array((R_,), dtype=np.int32)
    xo = o['xopt'].to_array((R_, N + 1, n)); uo = o['uopt'].to_array((R_, N, m))
    print('%s %d rollouts: %.2f ms (all: %s) -> %.1f k SCP iterations/s; %s' %
          (which, R_, min(ts[1:]) * 1e3, ' '.join('%.2f' % (t * 1e3) for t in ts), it.sum() / min(ts[1:]) / 1e3, g.kernel_info))
    print('fingerprint: iters %d status!=0 %d |x| %.12e |u| %.12e' % (it.sum(), (st != 0).sum(), np.linalg.norm(xo), np.linalg.norm(uo)))
    if '--check' in sys.argv:
        from oracle import gusto as ogusto
        model = dict(w['tab'], w_q=1.0, w_v=0.0)
        for b in range(3):
            xe, ue, ze, tr = ogusto.solve(model, w['Ad'], w['Bd'], w['dd'], w['H'], N, dt, w['Qz'], w['R'], x0[b], u_init[b], x_init[b], z=z[b],
                                          U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']) if w['XA'] is not None else None, x_char=xc, f_char=fc,
                                          convg_thresh=1e-3, qp_solver='riccati_ipm', max_gusto_iters=5)
            print('oracle rollout %d: iters %d vs %d, rel x %.2e u %.2e' % (b, len(tr), it[b], np.abs(xo[b] - xe).max() / np.abs(xe).max(),
                                                                          np.abs(uo[b] - ue).max() / np.abs(ue).max()))
    del g
g1 = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[0], np.zeros((N, m)), x_init[0], z=z[0], U=Polyhedron(w['UA'], w['Ub']), X=Xp, x_char=xc, f_char=fc,
           convg_thresh=1e-3, max_trace=0, max_gusto_iters=5)
per = []
for b in range(8):
    bb = b * (R_ // 8)
    t0 = time.perf_counter()
    g1.solve(x0[bb], np.zeros((N, m)), x_init[bb], z=z[bb])
    t = time.perf_counter() - t0
    per.append(t / int(g1.iters[0]))
per.sort()
print('one rollout at a time: %.3f ms per SCP iteration (median of 8; min %.3f max %.3f)' % (per[4] * 1e3, per[0] * 1e3, per[-1] * 1e3))
