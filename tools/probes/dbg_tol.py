import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import test_locp_gpu as T
from qp_cases import make_case
from oracle import riccati_ipm as ripm, condensed_ipm as cipm
def rel(a, b): return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
for name, kw in (('trunk', dict(r=30, m=8, P=32, N=50, seed=11, q_scale=0.02, use_X=False, u_max=800.0, amp=0.1, x_box=4.0)),
                 ('trunkX', dict(r=30, m=8, P=32, N=50, seed=11, q_scale=0.02, use_X=True, u_max=800.0, amp=0.1, x_box=4.0)),
                 ('r36m4X', dict(r=36, m=4, P=16, N=30, seed=17, q_scale=0.02, use_X=True, u_max=800.0, amp=0.1, x_box=4.0)),
                 ('r36m4', dict(r=36, m=4, P=16, N=30, seed=17, q_scale=0.02, use_X=False, u_max=800.0, amp=0.1, x_box=4.0)),
                 ('r36m8', dict(r=36, m=8, P=16, N=30, seed=17, q_scale=0.02, use_X=False, u_max=800.0, amp=0.1, x_box=4.0))):
    case, _ = make_case(**kw)
    (xe, ue, se), Je = T.oracle_solution(case)
    locp = T.product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'])
    J, ok, st = locp.solve()
    x, u, s = locp.get_solution()
    p = ripm.Problem(**dict(case, tr_active=False))
    xr, ur, sr, Jr, info = ripm.solve(p)
    inside = np.abs(case['x_scale'] * (xr[1:] - case['xk'][1:])).max() <= case['delta']
    print('%-7s vs exact x %.2e u %.2e | vs relaxed-path oracle x %.2e u %.2e (inside %s, its %d vs %d) | exact vs relaxed-oracle x %.2e' %
          (name, rel(x, xe), rel(u, ue), rel(x, xr), rel(u, ur), inside, st.num_iters, info['iters'], rel(xr, xe)))
