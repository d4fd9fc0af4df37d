import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import workloads as wl
import test_lean_gpu as T
from oracle import gusto as ogusto
w = wl.trunk_c5()
model = dict(w['tab'], w_q=1.0, w_v=0.0)
qp = T.first_qp(w, b=0, B=8, seed=9)
rf = T.locp_solve(w, qp, 1e4, False)
xk = rf[3]
A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
qp = dict(qp, A=A_k, B=B_k, d=d_k, xk=xk)
os.environ['SRH_LOCP_TRACE'] = '1'
print('---- lean', flush=True)
rl = T.locp_solve(w, qp, 1e4, True); print("lean J %.10e" % rl[0])
print('---- fused', flush=True)
rf = T.locp_solve(w, qp, 1e4, False)
