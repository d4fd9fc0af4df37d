import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd'))
import numpy as np, ctypes as C
from qp_cases import CASES, make_case
from test_locp_gpu import oracle_solution, product_locp, rel
from oracle import riccati_ipm as ri
from sofacontrol_amd import _lib
C2 = dict(r=30, m=4, P=64, N=50, seed=7, q_scale=0.02, use_X=True, u_max=1500.0, amp=0.1)
for name in (sys.argv[1:] or list(CASES)):
    case, _ = make_case(**(C2 if name == 'c2' else CASES[name]))
    (xe, ue, se), Je = oracle_solution(case)
    xp, up, sp, Jp, ip = ri.solve(ri.Problem(**case), tol=1e-12)
    locp = product_locp(case)
    locp.update(list(case['Ad']), list(case['Bd']), list(case['dd']), case['x0'], case['xk'], case['delta'], case['omega'], z=case['z'], zf=case.get('zf'))
    d = locp._data; N, n, m = locp.N, locp.n_x, locp.n_u
    x = np.empty((N + 1, n)); u = np.empty((N, m)); s = np.empty(N + 1)
    J = np.empty(1); status = np.empty(1, dtype=np.int32); iters = np.empty(1, dtype=np.int32)
    _lib.check(_lib.lib().slocp_solve(C.byref(locp._prob), C.c_int64(1), _lib.dptr(d['Ad']), _lib.dptr(d['Bd']), _lib.dptr(d['dd']), _lib.dptr(d['x0']), _lib.dptr(d['xk']), _lib.dptr(locp._delta), _lib.dptr(locp._omega), _lib.dptr(d['z']), _lib.dptr(d['zf']), _lib.dptr(d['u']), _lib.dptr(x), _lib.dptr(u), _lib.dptr(s), _lib.dptr(J), _lib.iptr(status), _lib.iptr(iters)), 'slocp')
    print('port vs dev relx %.2e relu %.2e Jp-J %.3e' % (rel(x, xp), rel(u, up), Jp - J[0]))
    print('%-28s status %d iters %d (port %d %s) J %.10g Je %.10g relx %.2e relu %.2e' % (name, status[0], iters[0], ip['iters'], ip['status'], J[0], Je, rel(x, xe), rel(u, ue)))
