"""iLQR on the Diamond shape (r = 30 -> n_x = 60, n_u = 4, horizon 50): one problem and a batch, GPU kernel vs
the numpy oracle port of the reference loop."""
import io, contextlib, sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'soft-robot-control_amd'), ROOT, os.path.join(ROOT, 'tests')]
from oracle import lqr as olqr
from helpers import golden_problem, product_tpwl
from sofacontrol_amd.lqr.ilqr import iLQR
from sofacontrol_amd.utils import QuadraticCost
r = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, P, N, dt = 4, 32, 50, 0.05
model, U, q_ref, v_ref, Hf = golden_problem(r, m, P, 40, 55, q_scale=0.2)
tp = product_tpwl(model, U, q_ref, v_ref, Hf)
with contextlib.redirect_stdout(io.StringIO()):
    tp.pre_discretize(dt)
Ad, Bd, dd = np.stack(tp.A_d), np.stack(tp.B_d), np.stack(tp.d_d)
H, z_ref = np.asarray(tp.H), np.asarray(tp.z_ref)
Qz = np.diag([0., 0., 0., 100., 100., 10.]); R = 1e-3 * np.eye(m)
th = np.linspace(0, 3.0, N + 1)
zt = np.zeros((N + 1, 6)); zt[:, 3] = -0.02 * np.sin(th); zt[:, 4] = 0.01 * np.sin(2 * th)
zt = zt + z_ref
rng = np.random.default_rng(1)
x0 = 1e-3 * rng.standard_normal(2 * r)
il = iLQR(dt, tp, QuadraticCost(Q=Qz, R=R, Qf=10 * Qz), N)
il.set_target(zt)
il.ilqr_computation(x0)
t0 = time.perf_counter(); x, u, K = il.ilqr_computation(x0); t1 = time.perf_counter() - t0
it1 = int(il.iters[0])
o = olqr.ILQR(model, Ad, Bd, dd, H, z_ref, Qz, R, 10 * Qz, N)
t0 = time.perf_counter(); xo, uo, Ko = o.solve(x0, zt); tc = time.perf_counter() - t0
print('single: GPU %.2f ms (%d iterations, %.2f ms/iter)  CPU port %.1f ms (%d iterations)  speedup %.1fx  max|dx| %.1e'
      % (t1 * 1e3, it1, t1 * 1e3 / max(it1, 1), tc * 1e3, len(o.trace) - 1, tc / t1, np.abs(x - xo).max()))
Bn = 1024
X0 = 1e-3 * rng.standard_normal((Bn, 2 * r))
il.ilqr_computation(X0)
t0 = time.perf_counter(); il.ilqr_computation(X0); tb = time.perf_counter() - t0
print('batch %d: %.1f ms, %.0f iLQR iterations/s' % (Bn, tb * 1e3, il.iters.sum() / tb))
for mi in (-1, 0):
    il.params.max_iter = mi
    il.ilqr_computation(x0)
    t0 = time.perf_counter()
    for _ in range(5): il.ilqr_computation(x0)
    print('max_iter %d: %.2f ms' % (mi, (time.perf_counter() - t0) / 5 * 1e3))
