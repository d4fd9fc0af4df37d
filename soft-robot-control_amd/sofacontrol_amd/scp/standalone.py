"""In-process GuSTO solver node (sofacontrol/scp/standalone.py:11-120; the same logic serves the ROS
node of scp/ros.py:48-159 minus rclpy): target interpolation, zero-input initial guess, first solve,
and the warm-started receding-horizon callback (shift of the previous solution, ros.py:109-114)."""
import numpy as np
from scipy.interpolate import interp1d

from .gusto import GuSTO


def runGuSTOSolverStandAlone(model, N, dt, Qz, R, x0, t=None, z=None, u=None, Qzf=None, zf=None, U=None, X=None,
                             Xf=None, dU=None, verbose=0, warm_start=True, **kwargs):
    problem = GuSTOSolverNode(model, N, dt, Qz, R, x0, t=t, z=z, u=u, Qzf=Qzf, zf=zf, U=U, X=X, Xf=Xf, dU=dU,
                              verbose=verbose, warm_start=warm_start, **kwargs)
    return problem.get_solution()


class GuSTOSolverNode():
    def __init__(self, model, N, dt, Qz, R, x0, t=None, z=None, u=None, Qzf=None, zf=None, U=None, X=None, Xf=None,
                 dU=None, verbose=0, warm_start=True, **kwargs):
        self.model = model
        self.N = N
        self.dt = dt
        x_char, f_char = self.model.get_characteristic_vals()
        self.Qzf = Qzf
        self.t = t
        self.z = z
        self.u = u
        if z is not None and z.ndim == 2:
            self.z_interp = interp1d(t, z, axis=0, bounds_error=False, fill_value=(z[0, :], z[-1, :]))
        if u is not None and u.ndim == 2:
            self.u_interp = interp1d(t, u, axis=0, bounds_error=False, fill_value=(u[0, :], u[-1, :]))
        u_init = np.zeros((self.N, self.model.n_u))
        x_init, _ = self.model.rollout(x0, u_init, self.dt)
        z, zf, u = self.get_target(0.0)
        self.gusto = GuSTO(model, N, dt, Qz, R, x0, u_init, x_init, z=z, u=u, Qzf=Qzf, zf=zf, U=U, X=X, Xf=Xf, dU=dU,
                           verbose=verbose, warm_start=warm_start, x_char=x_char, f_char=f_char, **kwargs)
        self.xopt, self.uopt, self.zopt, _ = self.gusto.get_solution()
        self.topt = self.dt * np.arange(self.N + 1)

    def get_solution(self):
        self.xopt, self.uopt, self.zopt, _ = self.gusto.get_solution()
        return self.xopt, self.uopt, self.zopt, self.topt

    def get_target(self, t0):
        """standalone.py:85-120 (constant 1-D targets are tiled per step)."""
        t = t0 + self.dt * np.arange(self.N + 1)
        if self.z is not None:
            z = self.z_interp(t) if self.z.ndim == 2 else np.tile(self.z.reshape(1, -1), (self.N + 1, 1))
        else:
            z = None
        zf = z[-1, :] if (self.Qzf is not None and z is not None) else None
        if self.u is not None:
            u = self.u_interp(t)[:self.N] if self.u.ndim == 2 else np.tile(self.u.reshape(1, -1), (self.N, 1))
        else:
            u = None
        return z, zf, u

    def _warm_start(self, t0):
        """Initial guess of a receding-horizon request: the previous solution shifted to start at t0, its last
        sample held over the part of the new horizon it does not cover (scp/ros.py:109-114)."""
        idx0 = np.argwhere(self.topt >= t0)[0, 0]
        u_init = self.uopt[-1, :].reshape(1, -1).repeat(self.N, axis=0)
        u_init[0:self.N - idx0] = self.uopt[idx0:, :]
        x_init = self.xopt[-1, :].reshape(1, -1).repeat(self.N + 1, axis=0)
        x_init[0:self.N + 1 - idx0] = self.xopt[idx0:, :]
        return u_init, x_init

    def gusto_callback(self, t0, x0):
        """scp/ros.py:94-127 without the ROS message types: returns (t, xopt, uopt, zopt, solve_time)."""
        z, zf, u = self.get_target(t0)
        u_init, x_init = self._warm_start(t0)
        self.gusto.solve(x0, u_init, x_init, z=z, zf=zf, u=u)
        self.xopt, self.uopt, zopt, t_solve = self.gusto.get_solution()
        self.topt = t0 + self.dt * np.arange(self.N + 1)
        return self.topt, self.xopt, self.uopt, zopt, t_solve

    # ---- the same request split into enqueue / poll / collect (the client's wait=False path)
    @property
    def supports_async(self):
        return (bool(getattr(self.gusto, '_fused', False)) and not getattr(self.gusto, '_ssm', False)
                and self.gusto.batch == 1)

    def gusto_callback_begin(self, t0, x0):
        z, zf, u = self.get_target(t0)
        u_init, x_init = self._warm_start(t0)
        self._t0_pending = t0
        self.gusto.solve_begin(x0, u_init, x_init, z=z, zf=zf, u=u)

    def gusto_callback_done(self):
        return self.gusto.solve_done()

    def gusto_callback_end(self):
        self.gusto.solve_end()
        self.xopt, self.uopt, zopt, t_solve = self.gusto.get_solution()
        self.topt = self._t0_pending + self.dt * np.arange(self.N + 1)
        return self.topt, self.xopt, self.uopt, zopt, t_solve

    def gusto_service(self, request, response=None):
        """The GuSTOsrv wire format (dependencies/ros/GuSTOsrv.srv:1-40, scp/ros.py:94-127): request fields `t0`
        (float64) and `x0` (float64[], flat); response fields `t, xopt, uopt, zopt` (float64[], row-major
        flattened, utils.np2arr) and `solve_time`.  Any object with those attributes works (a ROS2 message on
        a robot, GuSTOsrvRequest / GuSTOsrvResponse below without ROS)."""
        from ..utils import arr2np, np2arr
        if response is None:
            response = GuSTOsrvResponse()
        x0 = arr2np(request.x0, self.model.n_x, squeeze=True)
        t, xopt, uopt, zopt, t_solve = self.gusto_callback(request.t0, x0)
        response.t = np2arr(t)
        response.xopt = np2arr(xopt)
        response.uopt = np2arr(uopt)
        response.zopt = np2arr(zopt)
        response.solve_time = float(t_solve)
        return response


class GuSTOsrvRequest:
    """Request half of dependencies/ros/GuSTOsrv.srv (only t0, x0 are read by the solver node)."""

    def __init__(self, t0=0.0, x0=()):
        self.horizon = self.n_u = self.n_x = self.n_z = 0
        self.t0 = float(t0)
        self.x0 = list(np.asarray(x0, dtype=np.float64).ravel())
        self.u_init, self.x_init, self.z, self.zf, self.u = [], [], [], [], []


class GuSTOsrvResponse:
    def __init__(self):
        self.t, self.xopt, self.uopt, self.zopt = [], [], [], []
        self.solve_time = 0.0
