import sys, os; sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/soft-robot-control_amd')
import numpy as np, qp_cases
from helpers import Poly
from oracle import locp as olocp
from sofacontrol_amd.scp.locp import LOCP
g = np.load('/root/repo/tests/golden/g21_locp_nullspace.npz')
for name in qp_cases.NULLSPACE_CASES:
    case, ns = qp_cases.nullspace_case(name)
    kw = dict(case)
    qp = olocp.build_qp(kw.pop('N'), kw.pop('H'), kw.pop('Qz'), kw.pop('R'), kw.pop('Ad'), kw.pop('Bd'), kw.pop('dd'), kw.pop('x0'), kw.pop('xk'), kw.pop('delta'), kw.pop('omega'), **kw)
    xe, ue, se = olocp.split(qp, g[name + '_wopt'])
    Je = float(g[name + '_Jopt'])
    for tol in ('1e-9','1e-10','1e-11','1e-12'):
        os.environ['SRH_NULLSPACE_TOL']=tol
        locp = LOCP(case['N'], case['H'], case['Qz'], case['R'], U=Poly(*case['U']), X=Poly(*case['X']), x_char=1. / case['x_scale'], input_nullspace=ns)
        locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'], u=case['u_des'])
        J, ok, stats = locp.solve()
        x,u,s = locp.get_solution(); st=locp.nullspace_stats
        print(name, tol, ok, 'dJ', J-Je, 'relx', np.abs(x-xe).max()/np.abs(xe).max(), 'relu', np.abs(u-ue).max()/np.abs(ue).max(), st['qp_solves'], st['gap'], st['mu'], 'ms', stats.solve_time*1e3)
