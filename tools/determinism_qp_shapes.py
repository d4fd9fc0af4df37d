import sys, numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from qp_cases import make_case
from helpers import Poly
from sofacontrol_amd.scp.locp import LOCP
for r, m, useX, umax in ((30, 8, False, 800.0), (36, 4, True, 1500.0), (36, 8, False, 800.0), (30, 4, True, 1500.0), (20, 4, True, 1500.0)):
    case, _ = make_case(r=r, m=m, P=32, N=50, seed=11, q_scale=0.02, use_X=useX, u_max=umax, amp=0.1, **({'x_box': 4.0} if useX else {}))
    locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']) if case['X'] is not None else None, x_char=1. / case['x_scale'])
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'])
    locp.solve(); ref = (locp.x.copy() if hasattr(locp, 'x') else None)
    outs = []
    for _ in range(8):
        J, ok, st = locp.solve()
        x = np.asarray(locp.get_solution()[0]) if hasattr(locp, 'get_solution') else None
        outs.append((float(J), x))
    same = all(o[0] == outs[0][0] and (o[1] is None or np.array_equal(o[1], outs[0][1])) for o in outs)
    print('r=%d m=%d: 8 repeated solves bit-identical: %s (J = %.12g)' % (r, m, same, outs[0][0]))
