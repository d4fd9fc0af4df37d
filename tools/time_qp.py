"""Single-QP wall time of the LOCP kernel for the Diamond (n_u = 4) and Trunk (n_u = 8) stage shapes."""
import sys, time
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qp_cases import make_case
from helpers import Poly
from sofacontrol_amd.scp.locp import LOCP
for name, kw in (('diamond m=4', dict(r=30, m=4, P=32, N=50, seed=11, q_scale=0.02, use_X=True, u_max=1500.0, amp=0.1, x_box=4.0)),
                 ('trunk   m=8', dict(r=30, m=8, P=32, N=50, seed=11, q_scale=0.02, use_X=False, u_max=800.0, amp=0.1))):
    case, _ = make_case(**kw)
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*case['U']),
                X=Poly(*case['X']) if case['X'] is not None else None, x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'])
    locp.solve()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); J, ok, st = locp.solve(); ts.append(time.perf_counter() - t0)
    print('%s: %.1f ms per QP, %d IPM iterations, %.2f ms per iteration' % (name, min(ts) * 1e3, st.num_iters, min(ts) * 1e3 / st.num_iters))
