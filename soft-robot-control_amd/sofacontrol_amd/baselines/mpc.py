"""Linear MPC on the LOCP QP with the trust region switched off (sofacontrol/baselines/ros.py:14-136:
`runMPCSolver`, `MPCSolver`; the ROS service wrapper `MPCSolverNode` is next-tier).  Constant (A_d, B_d, d_d)
over the horizon, the same HIP QP kernel as GuSTO's LOCP (`is_tr_active=False`, locp.py:57)."""
import numpy as np
from scipy.interpolate import interp1d

from ..scp.locp import LOCP


def runMPCSolver(model, N, dt, cost_params, x0, target, U=None, X=None, Xf=None, dU=None, verbose=0, warm_start=True,
                 **kwargs):
    """baselines/ros.py:14-28 without the ROS spin: build the solver, return its first solution."""
    return MPCSolver(model, N, dt, cost_params, x0, target, U=U, X=X, Xf=Xf, dU=dU, verbose=verbose,
                     warm_start=warm_start, **kwargs).get_solution()


class MPCSolver:
    def __init__(self, model, horizon, dt, cost_params, x0, target, U=None, X=None, Xf=None, dU=None, verbose=0,
                 warm_start=True, **kwargs):
        self.model = model
        self.planning_horizon = horizon
        self.dt = dt
        self.target = target
        self.cost_params = cost_params
        if self.target.z is not None and self.target.z.ndim == 2:
            self.z_interp = interp1d(self.target.t, self.target.z, axis=0, bounds_error=False,
                                     fill_value=(self.target.z[0, :], self.target.z[-1, :]))
        if self.target.u is not None and self.target.u.ndim == 2:
            self.u_interp = interp1d(self.target.t, self.target.u, axis=0, bounds_error=False,
                                     fill_value=(self.target.u[0, :], self.target.u[-1, :]))
        self.verbose = verbose
        self.locp = LOCP(self.planning_horizon, self.model.H, self.cost_params.Q, self.cost_params.R,
                         Qzf=self.cost_params.Qf, U=U, X=X, Xf=Xf, dU=dU, verbose=(verbose == 2), warm_start=warm_start,
                         is_tr_active=False, **kwargs)
        self.A_d = [self.model.A_d for _ in range(self.planning_horizon)]
        self.B_d = [self.model.B_d for _ in range(self.planning_horizon)]
        if hasattr(self.model, 'd_d'):
            self.d_d = [self.model.d_d for _ in range(self.planning_horizon)]
        else:
            self.d_d = [np.zeros(self.model.A_d.shape[0]) for _ in range(self.planning_horizon)]
        self.X = X
        self.xopt = self.uopt = self.zopt = self.topt = None
        self.solve(0.0, np.asarray(x0, dtype=np.float64).ravel())

    def solve(self, t0, x0):
        """One receding-horizon solve (constructor body 79-99 and MPC_callback 159-184 of baselines/ros.py)."""
        z, zf, u = self.get_target(t0)
        self.locp.update(self.A_d, self.B_d, self.d_d, x0, None, 0, 0, z=z, zf=zf, u=u)
        Jstar, success, stats = self.locp.solve()
        if success:
            if self.verbose:
                print('{:.3f} s from LOCP solve'.format(stats.solve_time))
            self.xopt, self.uopt, _ = self.locp.get_solution()
            self.Jstar, self.solve_time = Jstar, stats.solve_time
        else:
            print('No solution found, extending previous solution')
            self.xopt = np.concatenate((self.xopt[1:, :], np.expand_dims(self.xopt[-1, :], axis=0)), axis=0)
            self.uopt = np.concatenate((self.uopt[1:, :], np.expand_dims(self.uopt[-1, :], axis=0)), axis=0)
        return success

    def get_solution(self):
        self.zopt = np.transpose(self.model.H @ self.xopt.T)
        self.topt = self.dt * np.arange(self.planning_horizon + 1)
        return self.xopt, self.uopt, self.zopt, self.topt

    def get_target(self, t0):
        """baselines/ros.py:101-135 (constant targets are tiled per step)."""
        N = self.planning_horizon
        t = t0 + self.dt * np.arange(N + 1)
        if self.target.z is not None:
            z = self.z_interp(t) if self.target.z.ndim == 2 else np.tile(self.target.z.reshape(1, -1), (N + 1, 1))
        else:
            z = None
        zf = z[-1, :] if (self.cost_params.Qf is not None and z is not None) else None
        if self.target.u is not None:
            u = self.u_interp(t)[:N] if self.target.u.ndim == 2 else np.tile(self.target.u.reshape(1, -1), (N, 1))
        else:
            u = None
        return z, zf, u
