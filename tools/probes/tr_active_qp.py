"""One C2 QP with an ACTIVE trust region (delta small enough that the condensed minimiser leaves it): the stage-wise
Riccati path that 4 % of the bench QPs take.  Time and interior-point iterations next to the same QP with delta = 1e4."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import workloads as wl
from oracle import gusto as ogusto, tpwl as otpwl, pod as opod
from scipy.interpolate import interp1d
from helpers import Poly
from sofacontrol_amd.scp.locp import LOCP
w = wl.diamond_c2()
N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
model = dict(w['tab'], w_q=1.0, w_v=0.0)
X = wl.snapshots(w['q_ref'], 6, seed=2)
x0 = np.concatenate((np.zeros((6, r)), opod.project(w['U'], w['q_ref'], X)), axis=1)
xc, fc = otpwl.characteristic_vals(model)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
b = 0
z = zi(dt * np.arange(N + 1))
xk = otpwl.rollout(model, w['Ad'], w['Bd'], w['dd'], x0[b], np.zeros((N, m)))
A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
locp = LOCP(N, w['H'], w['Qz'], w['R'], U=Poly(w['UA'], w['Ub']), X=Poly(w['XA'], w['Xb']), x_char=xc)
for delta in (1e4, 1.0, 0.1):
    locp.update(list(A_k), list(B_k), list(d_k), x0[b], xk, delta, 1.0, z=z)
    locp.solve()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); J, ok, st = locp.solve(); ts.append(time.perf_counter() - t0)
    xs, us, _ = locp.get_solution()
    md = np.abs((1.0 / np.abs(xc)) * (xs - xk)).max()
    print('delta %-8g J %.6e  ipm iterations %d  %.2f ms   max scaled move %.3g' % (delta, J, st.num_iters, min(ts) * 1e3, md))
