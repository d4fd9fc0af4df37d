import sys, os
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
X = wl.snapshots(w['q_ref'], 4, seed=2)
x0 = np.concatenate((np.zeros((4, r)), opod.project(w['U'], w['q_ref'], X)), axis=1)
xc, fc = otpwl.characteristic_vals(model)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
z = zi(b * 2.5 + dt * np.arange(N + 1))
# run the oracle GuSTO loop to get the sequence of QPs, trace each through the device LOCP
xk = otpwl.rollout(model, w['Ad'], w['Bd'], w['dd'], x0[b], np.zeros((N, m)))
uk = np.zeros((N, m))
locp = LOCP(N, w['H'], w['Qz'], w['R'], U=Poly(w['UA'], w['Ub']), X=Poly(w['XA'], w['Xb']), x_char=xc)
for it in range(3):
    A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
    locp.update(list(A_k), list(B_k), list(d_k), x0[b], xk, 1e4, 1.0, z=z)
    J, ok, st = locp.solve()
    print('SCP iteration', it, 'J', J, ok, 'ipm iterations', st.num_iters, flush=True)
    xk, uk, _ = locp.get_solution()
