"""The reference's hardware loop (examples/hardware/diamond_SSM.py:353-361: SSM n_x = 6, n_u = 4, N = 3, dt = 0.02,
max_gusto_iters = 0) through GuSTO.solve: median / p95 wall time per call, on the device path (csrc/gusto_ssm.hip) and -- with
SRH_GUSTO_SSM_HOST_LOOP=1 -- on the host loop around the device QP.  SRH_GUSTO_SSM_NO_LEAN=1: the device path without the lean
one-wave interior point.  Usage (GPU box, repo root): python tools/time_ssm_rti.py [--batch B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd')]
import numpy as np
import workloads as wl
from sofacontrol_amd.SSM.ssm import SSMDynamics
from sofacontrol_amd.scp.models.ssm import SSMGuSTO
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import HyperRectangle


def mat(v):
    a = np.empty((1, 1), dtype=object)
    a[0, 0] = np.asarray(v)
    return a


def sc(v):
    return mat(np.array([[v]]))


B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 1
n6, m4, N3, dt2 = 6, 4, 3, 0.02
mdl = wl.ssm_model(n6, m4, 3, 2, seed=96)
s6 = SSMDynamics(mdl['z_ref'].copy(), discrete=False, discr_method='be',
                 model=dict(Ts=sc(dt2), w_coeff=mat(mdl['W']), v_coeff=mat(mdl['V']), r_coeff=mat(mdl['R']), B=mat(mdl['B']),
                            rd_coeff=mat(mdl['Rd']), Bd=mat(mdl['Bd'])),
                 params=dict(state_dim=sc(n6), input_dim=sc(m4), output_dim=sc(n6), SSM_order=sc(2), ROM_order=sc(3)))
gm6 = SSMGuSTO(s6)
Qz6 = np.zeros((n6, n6)); Qz6[0, 0] = Qz6[1, 1] = 100.0
if '--three' in sys.argv:          # the driver's own cost: x, y, z of the end effector (examples/hardware/diamond_SSM.py:322-326)
    Qz6[2, 2] = 100.0
R6 = 0.003 * np.eye(m4)
U = HyperRectangle([1500.0] * m4, [0.0] * m4)
x06 = np.zeros(n6)
u6 = np.zeros((N3, m4))
xi6, _ = s6.rollout(x06, u6, dt2)
z6 = np.tile(np.array([0.02, -0.01, 0.015 if '--three' in sys.argv else 0.0, 0, 0, 0.0]), (N3 + 1, 1))
KEEP = '--keep' in sys.argv          # keep_solver_state=True: the reference's warm_start semantics across calls
if B == 1:
    g6 = GuSTO(gm6, N3, dt2, Qz6, R6, x06, u6, xi6, z=z6, U=U, verbose=0, max_gusto_iters=0, convg_thresh=1e-3, keep_solver_state=KEEP)
    call = lambda: g6.solve(x06, u6, xi6, z6, None, None)
else:
    xb, ub, xib, zb = np.tile(x06, (B, 1)), np.tile(u6, (B, 1, 1)), np.tile(xi6, (B, 1, 1)), np.tile(z6, (B, 1, 1))
    g6 = GuSTO(gm6, N3, dt2, Qz6, R6, xb, ub, xib, z=zb, U=U, verbose=0, max_gusto_iters=0, convg_thresh=1e-3, batch=B, first_solve_cap=5)
    call = lambda: g6.solve_batch(xb, ub, xib, zb)
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    call()
    ts.append(time.perf_counter() - t0)
ts = np.sort(np.array(ts[20:])) * 1e3
print('SSM GuSTO real-time iteration%s, batch %d, path %s: median %.3f ms, p95 %.3f ms, min %.3f ms per call; iters %s; kernel %s' %
      (' (solver state kept)' if KEEP else '', B, 'device' if getattr(g6, '_ssm', False) else 'host loop', np.median(ts), ts[int(len(ts) * 0.95)], ts[0], g6.iters[:4],
       g6.kernel_info))
if os.environ.get('SRH_GUSTO_TRACE_QIT') and g6.trace is not None:
    print('   shader clocks of the last call: linearise %.0f, QP %.0f, tests %.0f; interior-point iterations + 1000 (pass + 1): %.0f' % tuple(g6.trace[0, 0]))
    if np.isfinite(g6.trace[0, 1, 0]) and g6.trace[0, 1, 0] > 0:      # a library built with -DQDU_CLOCKS=1 (locp_dense_u.h)
        print('   dense one-wave QP clocks: set-up (free response, sensitivities) %.0f, rows + Hessian %.0f, interior point outside the solves %.0f, '
              'factorisations %.0f, solves %.0f, tail (rollout, objective) %.0f' % tuple(np.concatenate((g6.trace[0, 1], g6.trace[0, 2, :2]))))
        print('   ... the interior point outside the solves, by part: row arithmetic to the fence %.0f, mu / r_p reductions %.0f, gradient + C^T products + r_d %.0f, '
              'step lengths (after the solve) %.0f, rest %.0f' % tuple(list(np.concatenate((g6.trace[0, 2, 2:], g6.trace[0, 3, :2]))) + [g6.trace[0, 1, 2]]))
