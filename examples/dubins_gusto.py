"""The reference's stand-alone GuSTO demonstration (sofacontrol/scp/example.py) on the device QP: a unicycle driven to a
terminal pose under input-rate limits, N = 50, dt = 0.1, no warm start.  Prints the final pose and the SCP trace instead
of plotting.

    python examples/dubins_gusto.py

Needs an MI355X (no CPU fallback)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'soft-robot-control_amd'))


def main():
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.scp.models.dubins_car import DubinsCar
    from sofacontrol_amd.utils import HyperRectangle

    model = DubinsCar()
    dU = HyperRectangle(np.array([0.1, 0.1]), np.array([-0.1, -0.1]))
    N, dt = 50, 0.1
    Qz, R, Qzf = np.zeros((3, 3)), np.eye(2), 100 * np.eye(3)
    zf_des = np.array([4., 5., 0.])
    x0 = np.zeros(3)
    u_init = np.zeros((N, 2))
    x_init = model.rollout(x0, u_init, dt)
    gusto = GuSTO(model, N, dt, Qz, R, x0, u_init, x_init, u=u_init, zf=zf_des, Qzf=Qzf, U=None, dU=dU, verbose=1,
                  warm_start=False, x_char=np.array([1., 1., np.pi]))
    x, u, z, _ = gusto.get_solution()
    n = int(gusto.iters[0])
    print('SCP iterations: %d   (J, delta, omega, rho) per iteration:\n%s' % (n, np.array2string(gusto.trace[0, :n], precision=4)))
    print('final pose %s (target %s), largest input increment %.3f' % (np.round(x[-1], 4), zf_des, np.abs(np.diff(u, axis=0)).max()))


if __name__ == '__main__':
    main()
