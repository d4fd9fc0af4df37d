import sys, time
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qp_cases import make_case
from helpers import Poly
from sofacontrol_amd.scp.locp import LOCP
for r in (25, 28, 30, 32):
    case, _ = make_case(r=r, m=4, P=32, N=50, seed=11, q_scale=0.02, use_X=True, u_max=1500.0, amp=0.1, x_box=4.0)
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']), x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'])
    locp.solve()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); J, ok, st = locp.solve(); ts.append(time.perf_counter() - t0)
    print('r=%d (n_x=%d): %.1f ms per QP, %d IPM iterations, %.2f ms per iteration' % (r, 2 * r, min(ts) * 1e3, st.num_iters, min(ts) * 1e3 / st.num_iters))
