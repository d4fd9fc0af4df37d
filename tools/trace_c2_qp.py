"""Interior-point trace (SRH_LOCP_TRACE=1) of the first QP of a BASELINE C2 / C5 rollout through the LOCP class."""
import sys, os
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import workloads as wl
from oracle import gusto as ogusto, tpwl as otpwl, pod as opod
from scipy.interpolate import interp1d
from helpers import Poly
from sofacontrol_amd.scp.locp import LOCP
which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
b = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = wl.diamond_c2() if which == 'c2' else wl.trunk_c5()
N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
model = dict(w['tab'], w_q=1.0, w_v=0.0)
X = wl.snapshots(w['q_ref'], 6, seed=2)
x0 = np.concatenate((np.zeros((6, r)), opod.project(w['U'], w['q_ref'], X)), axis=1)
xc, fc = otpwl.characteristic_vals(model)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
z = zi(b * 10.0 / 6 + dt * np.arange(N + 1))
xk = otpwl.rollout(model, w['Ad'], w['Bd'], w['dd'], x0[b], np.zeros((N, m)))
A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
locp = LOCP(N, w['H'], w['Qz'], w['R'], U=Poly(w['UA'], w['Ub']), X=Poly(w['XA'], w['Xb']) if w['XA'] is not None else None, x_char=xc)
locp.update(list(A_k), list(B_k), list(d_k), x0[b], xk, 1e4, 1.0, z=z)
J, ok, st = locp.solve()
print('J', J, ok, st.num_iters)
