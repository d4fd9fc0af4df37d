"""The rollouts of bench.py's scp_reference_horizons that the lean kernel hands to the fused kernel (N = 5: problem 6, N = 3: problem 2
of 8): which status, what the fused kernel then does (SCP trace), how long, and the oracle's trace of the same problem."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
torch.cuda.init()
import bench
import workloads as wl
from scipy.interpolate import interp1d
from oracle import gusto as ogusto
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron
for N, dt, with_X, cap, b in ((3, 0.1, False, 5, 2), (5, 0.05, True, 500, 6)):
    w = wl.diamond_c2(N=N, dt=dt, with_X=with_X)
    m, r = w['m'], w['r']
    rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
    tp, gm = bench.build_model(w, 1354)
    xc, fc = gm.get_characteristic_vals()
    reps = 8
    X = wl.snapshots(w['q_ref'], reps, seed=2)
    x0 = np.concatenate((np.zeros((reps, r)), rom.compute_RO_state(qf=X)), axis=1)
    u0 = np.zeros((N, m))
    x_init, _ = tp.rollout(x0, np.zeros((reps, N, m)), dt)
    zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
    z = np.stack([zi(bb * 10.0 / reps + dt * np.arange(N + 1)) for bb in range(reps)])
    Xp = Polyhedron(w['XA'], w['Xb']) if with_X else None
    for env in ({}, {'SRH_LEAN_NO_WAVE': '1'}, {'SRH_GUSTO_NO_LEAN': '1'}):
        for k in ('SRH_LEAN_NO_WAVE', 'SRH_GUSTO_NO_LEAN'):
            os.environ.pop(k, None)
        os.environ.update(env)
        g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[0], u0, x_init[0], z=z[0], U=Polyhedron(w['UA'], w['Ub']), X=Xp, x_char=xc, f_char=fc,
                  convg_thresh=1e-3, max_trace=64, max_gusto_iters=cap)
        g.solve(x0[b], u0, x_init[b], z=z[b])
        t0 = time.perf_counter()
        g.solve(x0[b], u0, x_init[b], z=z[b])
        t = (time.perf_counter() - t0) * 1e3
        it = int(g.iters[0])
        print('N = %d problem %d %s: %.2f ms, iters %d status %d, %s' % (N, b, env or 'default', t, it, int(g.status[0]), g.kernel_info))
        print('   trace (J, delta, omega, rho):', np.array2string(np.asarray(g.trace)[0, :it], precision=4, max_line_width=200).replace('\n', ' '))
    model = dict(w['tab'], w_q=1.0, w_v=0.0)
    xe, ue, ze, tr = ogusto.solve(model, w['Ad'], w['Bd'], w['dd'], w['H'], N, dt, w['Qz'], w['R'], x0[b], u0, x_init[b], z=z[b], U=(w['UA'], w['Ub']),
                                  X=(w['XA'], w['Xb']) if with_X else None, x_char=xc, f_char=fc, convg_thresh=1e-3, qp_solver='riccati_ipm',
                                  max_gusto_iters=cap)
    print('   oracle: iters %d' % len(tr), [tuple(round(float(v), 4) for v in t4[:4]) + (t4[4], t4[5]) for t4 in tr])
    print('   rel x %.2e u %.2e' % (np.abs(g.xopt[0] - xe).max() / np.abs(xe).max(), np.abs(g.uopt[0] - ue).max() / max(1e-12, np.abs(ue).max())))
