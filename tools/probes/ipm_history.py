"""Interior-point history (mu, dual / primal residual, step length) of the first QPs of BASELINE C2 / C5 through the lean kernel:
how many iterations does the last decade of the stopping rule cost?  python tools/probes/ipm_history.py [c2|c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import workloads as wl
import test_lean_gpu as T
from oracle import gusto as ogusto
which = 'c5' if 'c5' in sys.argv[1:] else 'c2'
w = wl.trunk_c5() if which == 'c5' else wl.diamond_c2()
model = dict(w['tab'], w_q=1.0, w_v=0.0)
os.environ['SRH_LOCP_TRACE'] = '1'
for b in range(3):
    qp = T.first_qp(w, b=b, B=8, seed=9)
    print('---- rollout %d, QP 1 (linearised about the zero-input rollout)' % b, flush=True)
    r1 = T.locp_solve(w, qp, 1e4, True)
    print('J %.10e iters %d' % (r1[0], r1[2]), flush=True)
    xk = r1[3]
    A_k, B_k, d_k, idx = ogusto.traj_dynamics(model, w['Ad'], w['Bd'], w['dd'], xk)
    qp2 = dict(qp, A=A_k, B=B_k, d=d_k, xk=xk)
    print('---- rollout %d, QP 2 (about the minimiser of QP 1; cold start here)' % b, flush=True)
    r2 = T.locp_solve(w, qp2, 1e4, True)
    print('J %.10e iters %d' % (r2[0], r2[2]), flush=True)
