"""Debug: per-stage difference of the lean kernel's states against the fused kernel's on one C5 QP."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import workloads as wl
import test_lean_gpu as T
from oracle import tpwl as otpwl
w = wl.trunk_c5()
qp = T.first_qp(w, b=0)
rl = T.locp_solve(w, qp, 1e4, True)
rf = T.locp_solve(w, qp, 1e4, False)
xl, xf = np.asarray(rl[3]), np.asarray(rf[3])
print('J', rl[0], rf[0])
err = np.abs(xl - xf).max(axis=1)
print('per-stage max |dx|:', np.array2string(err, precision=2, max_line_width=200))
# the states the solution's inputs give with the stage matrices, and with zero inputs
A, B, d = qp['A'], qp['B'], qp['d']
x = [qp['x0']]; x0r = [qp['x0']]
for k in range(w['N']):
    x.append(A[k] @ x[-1] + B[k] @ np.asarray(rl[4])[k] + d[k]); x0r.append(A[k] @ x0r[-1] + d[k])
print('lean x vs rollout of lean u:', np.abs(xl - np.array(x)).max(), ' vs free response:', np.abs(xl - np.array(x0r)).max())
k = int(np.argmax(err > 1e-6))
print('first bad stage', k, 'components bad', np.nonzero(np.abs(xl[k] - xf[k]) > 1e-6)[0][:20], 'values', xl[k][:4], xf[k][:4])
u = np.asarray(rl[4]); x0 = qp['x0']
cands = {'A x0 + d': A[0] @ x0 + d[0], 'B u0 + d': B[0] @ u[0] + d[0], 'd': d[0], 'A x0 + B u0': A[0] @ x0 + B[0] @ u[0], 'true': A[0] @ x0 + B[0] @ u[0] + d[0],
         'B u0': B[0] @ u[0], 'A x0': A[0] @ x0}
for kname, v in cands.items():
    print('%-12s max|x1 - cand| = %.3e   cand[:4] = %s' % (kname, np.abs(xl[1] - v).max(), v[:4]))
print('x1 lean', xl[1][:8]); print('x0', x0[:4], x0[30:34]); print('u0', u[0])
# which stages agree with a re-rollout started from the lean kernel's OWN previous state (one-step consistency)?
one = np.array([np.abs(xl[k + 1] - (A[k] @ xl[k] + B[k] @ u[k] + d[k])).max() for k in range(w['N'])])
print('one-step inconsistency per stage:', np.array2string(one, precision=2, max_line_width=200))
np.savez(os.path.join(ROOT, 'gpurun_out', 'dbg_lean_x.npz'), xl=xl, xf=xf, u=u, A=np.asarray(A), B=np.asarray(B), d=np.asarray(d), x0=x0, xk=qp['xk'])
print('A[N-1][0][:2]', A[-1][0][:2], 'A[N-1][1][0]', A[-1][1][0], 'xf[N][0]', xf[-1][0], 'xf[N-1][0]', xf[-2][0], 'u[0][2]', u[0][2], 'A[0][0][:2]', A[0][0][:2])
