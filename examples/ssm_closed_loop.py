"""Closed loop on an SSM reduced model without SOFA: the reference's SSM driver shape (examples/hardware/diamond_SSM.py --
SSMGuSTO + GuSTOSolverNode + the `scp` controller of SSM/controllers.py) on a synthetic polynomial model, with the
model itself as the plant.  Every simulation step goes through `controller.evaluate(sim_time, y, x, u_prev)`:
measurement -> (re-projection onto the admissible set) -> SSMObserver (W_map on the device) -> replan every N_replan
steps with GuSTO (device QP with the per-stage observer linearisation, asynchronous client) -> input.

    python examples/ssm_closed_loop.py [--steps 100]

Needs an MI355X (no CPU fallback)."""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=100, help='simulation steps of 0.02 s')
    args = ap.parse_args()

    from oracle import ssm as ossm                      # only the synthetic model generator and the plant's algebra
    from test_ssm_gpu import product_ssm
    import sofacontrol_amd.SSM.controllers as sctl
    from sofacontrol_amd.scp.models.ssm import SSMGuSTO
    from sofacontrol_amd.scp.standalone import GuSTOSolverNode
    from sofacontrol_amd.utils import HyperRectangle, Polyhedron, vq2qv

    n, m, N, dt = 4, 2, 8, 0.02
    model = ossm.synthetic(n, m, 3, 2, seed=81)
    model['W'][:] = 0.0; model['W'][:, :n] = np.eye(n)          # consistent observation / reduction maps: z = x + z_ref
    model['V'][:] = 0.0; model['V'][:, :n] = np.eye(n)
    s = product_ssm(model, discr='fe')
    zf = lambda x: ossm.observe(model, x) + model['z_ref']
    x = np.zeros(n)
    z_goal = zf(x) + np.array([0.08, -0.04, 0.0, 0.0])
    Qz, R = np.diag([10., 10., 0.1, 0.1]), 1e-2 * np.eye(m)
    U = HyperRectangle([2.0] * m, [-2.0] * m)
    with contextlib.redirect_stdout(io.StringIO()):
        node = GuSTOSolverNode(SSMGuSTO(s), N, dt, Qz, R, x, z=z_goal - model['z_ref'], U=U, verbose=0, max_gusto_iters=4,
                               convg_thresh=1e-4)
    zr = vq2qv(model['z_ref'])
    Y = Polyhedron(np.kron(np.eye(n), np.array([[1.], [-1.]])), np.ravel(np.column_stack((zr + 1.0, -(zr - 1.0)))),
                   with_reproject=True)
    ctrl = sctl.scp(s, None, dt, N_replan=2, delay=0.0, solver_node=node, wait=False, Y=Y)
    ctrl.set_sim_timestep(dt)
    u = np.zeros(m)
    t_eval = []
    for k in range(args.steps):
        y = vq2qv(zf(x))
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            u = ctrl.evaluate(k * dt, y, None, u)
        t_eval.append(time.perf_counter() - t0)
        A, B, d = ossm.jacobians(model, x, u, dt, 'fe')
        x = A @ x + B @ u + d
        if k % 10 == 0:
            print('t = %.2f s  tracked-output error %.4f  u = %s' % (k * dt, np.linalg.norm((zf(x) - z_goal)[:2]), np.round(u, 3)))
    te = np.sort(np.array(t_eval))
    print('%d replans, median solve %.2f ms; evaluate(): median %.0f us, 99th percentile %.0f us'
          % (len(ctrl.solve_times), 1e3 * float(np.median(ctrl.solve_times)), 1e6 * te[len(te) // 2], 1e6 * te[int(0.99 * len(te))]))


if __name__ == '__main__':
    main()
