"""Debug helper: where do the device iLQR and the oracle loop part on the C3 configuration (dt = 0.05, backward Euler)?
Runs both with max_iter = 0, 1, 2, ... and prints the differences of cost, x, u; also the 'be' Jacobians at random points."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import workloads as wl
from oracle import ssm as ossm, lqr as olqr
from test_ssm_gpu import product_ssm
from sofacontrol_amd.lqr.ilqr import iLQR
from sofacontrol_amd.utils import QuadraticCost

c3 = wl.ssm_c3(256)
n, m, N, dt, discr = c3['n'], c3['m'], c3['N'], c3['dt'], c3['discr']
if len(sys.argv) > 1:
    dt = float(sys.argv[1])
if len(sys.argv) > 2:
    discr = sys.argv[2]
model = ossm.synthetic(n, m, 3, 2, seed=95)
s = product_ssm(model, discr=discr)
s.H = model['W'][:, :n].copy()
rng = np.random.default_rng(0)
X = 0.1 * rng.standard_normal((5, n)); U = rng.standard_normal((5, m))
A, B, d = s.get_jacobians(X, U, dt)
for i in range(5):
    Ao, Bo, do = ossm.jacobians(model, X[i], U[i], dt, discr)
    print('jacobians %d: dA %.2e dB %.2e dd %.2e' % (i, np.abs(A[i] - Ao).max(), np.abs(B[i] - Bo).max(), np.abs(d[i] - do).max()))
b = 0
x0, zt = c3['x0'][b], c3['zt'][b]
for mi in (0, 1, 2, 3, 5, 100):
    il = iLQR(dt, s, QuadraticCost(Q=c3['Qz'], R=c3['R'], Qf=c3['Qf']), N)
    il.params.max_iter = mi
    il.set_target(zt)
    x, u, K = il.ilqr_computation(x0)
    o = olqr.ILQRGeneric(lambda xx, uu: ossm.jacobians(model, xx, uu, dt, discr), lambda xx: ossm.observe(model, xx) + model['z_ref'], s.H, n, m,
                         c3['Qz'], c3['R'], c3['Qf'], N)
    o.p.max_iter = mi
    xo, uo, Ko = o.solve(x0, zt)
    print('max_iter %3d: iters %d / %d cost %.12e / %.12e  dx %.2e du %.2e dK %.2e (|x| %.2e |u| %.2e) alphas %s' %
          (mi, int(il.iters[0]), len(o.trace) - 1, float(il.cost[0]), o.trace[-1][1], np.abs(x - xo).max(), np.abs(u - uo).max(), np.abs(K - Ko).max(),
           np.abs(xo).max(), np.abs(uo).max(), [t[2] for t in o.trace][:6]))
    if mi == 0:
        dKt = np.abs(K - Ko).reshape(N, -1).max(axis=1)
        print('   dK per stage (last 6):', dKt[-6:], ' first 3:', dKt[:3], ' |K| last', np.abs(Ko[-1]).max())
        dxt = np.abs(x - xo).max(axis=1)
        print('   dx per stage first 4:', dxt[:4], 'du first 3', np.abs(u - uo).max(axis=1)[:3])
