"""Closed-loop controllers on the MI355X hot path -- surface of sofacontrol/tpwl/controllers.py.

`TemplateController.evaluate` (lines 85-117) keeps the reference's time gating verbatim (host glue); what it
calls runs on the device: the POD projection of the full FEM state (`rom.compute_RO_state`, line 96), the
policy computation (`iLQR.ilqr_computation`, GuSTO through a solver client) and the nearest-point lookup of
the LQR feedback (`calc_nearest_point`, line 330).  The ROS2 `GuSTOClientNode` of the reference is replaced
by `GuSTOClient`, an in-process client with the same methods around `scp.standalone.GuSTOSolverNode`; any
object with that client protocol (e.g. a ROS client) can be passed as `client=`."""
import numpy as np
from scipy.interpolate import interp1d

from .observer import FullStateObserver
from .. import closed_loop_controller
from ..lqr.ilqr import iLQR
from ..lqr.traj_tracking_lqr import TrajTrackingLQR
from ..lqr.lqr import DLQR, dare_batch


class GuSTOClient:
    """In-process stand-in for scp/ros.py:162-223 (send_request / check_if_done / force_wait / force_spin /
    get_solution) around a GuSTOSolverNode: requests are solved synchronously on the GPU."""

    def __init__(self, solver_node):
        self.node = solver_node
        self._res = None

    def send_request(self, t0, x0, wait=True):
        self._res = self.node.gusto_callback(t0, np.asarray(x0, dtype=np.float64))

    def force_spin(self):
        pass

    def check_if_done(self):
        return self._res is not None

    def force_wait(self):
        pass

    def get_solution(self, n_x, n_u):
        t, xopt, uopt, zopt, t_solve = self._res
        return np.asarray(t), np.asarray(uopt).reshape(-1, n_u), np.asarray(xopt).reshape(-1, n_x), t_solve


class TemplateController(closed_loop_controller.TemplateController):
    """tpwl/controllers.py:18-132."""

    def __init__(self, dyn_sys, cost_params, dt=0.01, observer=None, delay=2, u0=None):
        super().__init__()
        self.dyn_sys = dyn_sys
        self.dt = dt
        self.input_dim = self.dyn_sys.get_input_dim()
        self.state_dim = self.dyn_sys.get_state_dim()
        self.cost_params = cost_params
        self.observer = observer if observer is not None else FullStateObserver(self.state_dim, self.dyn_sys.H)
        self.t_delay = delay
        self.u0 = u0 if u0 is not None else np.zeros(self.input_dim)
        self.t_compute = 0.
        self.u = self.u0

    def validate_problem(self):
        raise NotImplementedError('Must be subclassed')

    def recompute_policy(self, t_step):
        return True if t_step == 0 else False

    def compute_policy(self, t_step, x_belief):
        raise NotImplementedError('Must be subclassed')

    def compute_input(self, t_step, x_belief):
        raise NotImplementedError('Must be subclassed')

    def evaluate(self, sim_time, y, x, u_prev):
        """controllers.py:85-117."""
        sim_time = round(sim_time, 4)
        x_actual = self.dyn_sys.rom.compute_RO_state(xf=x)           # POD projection kernel
        self.observer.update(u_prev, y, self.sim_dt, x=x_actual)
        if round(sim_time, 4) < round(self.t_delay, 4):
            self.u = self.u0
        else:
            if round(sim_time - self.t_delay, 4) >= round(self.t_compute, 4):
                if self.recompute_policy(self.t_compute):
                    self.compute_policy(self.t_compute, self.observer.x)
                self.u = self.compute_input(self.t_compute, self.observer.x)
                self.t_compute += self.dt
                self.t_compute = round(self.t_compute, 4)
        self.u = np.atleast_1d(self.u)
        return self.u.copy()

    def save_controller_info(self):
        info = dict()
        info['cost_params'] = self.cost_params
        if self.observer is not None:
            info['observer_params'] = self.observer.get_observer_params()
        if self.dyn_sys is not None:
            info['dyn_sys_params'] = self.dyn_sys.get_sim_params()
            info['state_dim'] = self.dyn_sys.get_state_dim()
            info['input_dim'] = self.dyn_sys.get_input_dim()
        return info


class ilqr(TemplateController):
    """controllers.py:135-206."""

    def __init__(self, dyn_sys, cost_params, target, dt=0.01, observer=None, delay=2., u0=None, **kwargs):
        super().__init__(dyn_sys=dyn_sys, cost_params=cost_params, dt=dt, observer=observer, delay=delay, u0=u0)
        self.target = target
        self.setpoint_reaching = True
        self.validate_problem()
        if self.setpoint_reaching:
            self.final_time, self.planning_horizon = self.get_problem_horizon(kwargs.get('tf'))
        else:
            self.final_time, self.planning_horizon = self.get_problem_horizon(self.target.t[-1])
        self.policy = iLQR(dt=self.dt, model=self.dyn_sys, cost_params=self.cost_params,
                           planning_horizon=self.planning_horizon)
        self.x_bar = None
        self.u_bar = None
        self.K = None

    def get_problem_horizon(self, tf):
        if tf is None:
            raise RuntimeError('Final time not set for single-shooting ilqr')
        return tf, int(tf / self.dt)

    def validate_problem(self):
        assert self.target.z is not None and self.target.Hf is not None
        assert self.target.Hf.shape[0] == self.target.z.shape[-1]
        assert self.target.z.ndim <= 2
        if self.target.z.ndim == 2:
            self.setpoint_reaching = False
        assert (np.asarray(self.target.Hf @ self.dyn_sys.rom.V) == self.dyn_sys.H).all()
        output_dim = self.dyn_sys.get_output_dim()
        if self.setpoint_reaching:
            assert self.cost_params.Qf.shape == (output_dim, output_dim)
        assert self.cost_params.Q.shape == (output_dim, output_dim)
        assert self.cost_params.R.shape == (self.input_dim, self.input_dim)

    def compute_policy(self, t_step, x_belief):
        if self.setpoint_reaching:
            self.policy.set_target(np.repeat(self.target.z[np.newaxis, :], self.planning_horizon + 1, axis=0))
        else:
            z_interp = interp1d(self.target.t, self.target.z, axis=0)
            self.policy.set_target(z_interp(np.linspace(0, self.final_time, self.planning_horizon + 1)))
        self.x_bar, self.u_bar, self.K = self.policy.ilqr_computation(x_belief)

    def compute_input(self, t_step, x_belief):
        if t_step > self.final_time:
            self.u = self.u0
        else:
            step = int(t_step / self.dt)
            self.u = self.u_bar[step] + self.K[step] @ (x_belief - self.x_bar[step])
        return self.u


class scp(TemplateController):
    """controllers.py:209-346.  `client` is any object with the GuSTOClientNode protocol (scp/ros.py:162-223);
    `solver_node` (a scp.standalone.GuSTOSolverNode) is wrapped into the in-process `GuSTOClient`."""

    def __init__(self, dyn_sys, cost, dt, N_replan=None, observer=None, delay=2, u0=None, wait=True, client=None,
                 solver_node=None, **kwargs):
        super().__init__(dyn_sys, None, dt=dt, observer=observer, delay=delay, u0=u0)
        self.N_replan = N_replan if N_replan is not None else 1
        self.t_opt = None
        self.u_opt = None
        self.x_opt = None
        self.u_bar = None
        self.x_bar = None
        self.wait = wait
        self.t_next_solve = 0
        self.initialized = False
        self.solve_times = []
        if client is None:
            if solver_node is None:
                raise RuntimeError('scp needs client= (GuSTOClientNode protocol) or solver_node= (GuSTOSolverNode)')
            client = GuSTOClient(solver_node)
        self.GuSTO = client
        self.z_opt_horizon = []
        self.t_opt_horizon = []
        self.mpc = kwargs.pop('mpc', False)
        # per-point LQR gains (controllers.py:238-246: scipy dare per point) in one batched launch
        Ad, Bd = [], []
        for i in range(self.dyn_sys.num_points):
            A_d, B_d, _ = self.dyn_sys.discretize_dynamics(self.dyn_sys.tpwl_dict['A_c'][i], self.dyn_sys.tpwl_dict['B_c'][i],
                                                            self.dyn_sys.tpwl_dict['d_c'][i], dt)
            Ad.append(A_d); Bd.append(B_d)
        Kb, _ = dare_batch(np.stack(Ad), np.stack(Bd), cost.Q, cost.R)
        self.K = [Kb[i] for i in range(Kb.shape[0])]

    def compute_policy(self, t_step, x_belief):
        """controllers.py:248-274."""
        if not self.initialized:
            self.run_GuSTO(t_step, x_belief, wait=True)
            self.update_policy(init=True)
            self.initialized = True
        else:
            self.update_policy()
        self.t_next_solve = round(self.t_opt[-1], 6)
        x0 = x_belief if self.mpc else self.x_opt[-1, :]
        self.run_GuSTO(self.t_opt[-1], x0, wait=self.wait)

    def run_GuSTO(self, t0, x0, wait):
        self.GuSTO.send_request(t0, x0, wait=wait)

    def recompute_policy(self, t_step):
        return round(t_step, 4) >= round(self.t_next_solve, 4)

    def update_policy(self, init=False):
        """controllers.py:288-324."""
        if not self.GuSTO.check_if_done():
            print('GuSTO cannot provide real-time compatibility, consider modifying problem')
            self.GuSTO.force_wait()
        t_opt_p, u_opt_p, x_opt_p, t_solve = self.GuSTO.get_solution(self.state_dim, self.input_dim)
        self.solve_times.append(t_solve)
        u_opt_intp = interp1d(t_opt_p, np.vstack((u_opt_p, u_opt_p[-1, :])), axis=0)
        x_opt_intp = interp1d(t_opt_p, x_opt_p, axis=0)
        if init:
            t_opt_new = self.dt * np.arange(self.N_replan + 1)
            self.t_opt = t_opt_new
            self.u_opt = u_opt_intp(t_opt_new)
            self.x_opt = x_opt_intp(t_opt_new)
        else:
            t_opt_new = self.t_opt[-1] + self.dt * np.arange(self.N_replan + 1)
            u_opt_new = u_opt_intp(t_opt_new)
            x_opt_new = x_opt_intp(t_opt_new)
            self.t_opt = np.concatenate((self.t_opt, t_opt_new[1:]))
            self.u_opt = np.concatenate((self.u_opt[:-1, :], u_opt_new))
            self.x_opt = np.concatenate((self.x_opt, x_opt_new[1:, :]))
        self.z_opt_horizon.append(self.dyn_sys.x_to_zfyf(x_opt_p, zf=True))
        self.t_opt_horizon.append(t_opt_p)
        self.u_bar = interp1d(self.t_opt, self.u_opt, axis=0)
        self.x_bar = interp1d(self.t_opt, self.x_opt, axis=0)

    def compute_input(self, t_step, x_belief):
        """controllers.py:326-333."""
        self.GuSTO.force_spin()
        i_near = self.dyn_sys.calc_nearest_point(self.x_bar(t_step))
        return self.u_bar(t_step) + self.K[i_near] @ (x_belief - self.x_bar(t_step))

    def save_controller_info(self):
        info = dict()
        info['t_opt'] = self.t_opt
        info['u_opt'] = self.u_opt
        info['z_opt'] = self.dyn_sys.x_to_zfyf(self.x_opt, zf=True)
        info['solve_times'] = self.solve_times
        info['rollout_time'] = self.N_replan * self.dt
        info['z_rollout'] = self.z_opt_horizon
        info['t_rollout'] = self.t_opt_horizon
        return info


class TrajTracking(TemplateController):
    """controllers.py:349-395."""

    def __init__(self, dyn_sys, cost_params, target, dt=0.01, observer=None, delay=2., u0=None, **kwargs):
        super().__init__(dyn_sys=dyn_sys, cost_params=cost_params, dt=dt, observer=observer, delay=delay, u0=u0)
        self.target = target
        self.validate_problem()
        self.final_time = self.target.t[-1]
        self.policy = TrajTrackingLQR(dt=dt, model=dyn_sys, cost_params=self.cost_params)
        self.x_bar, self.u_bar, self.K = self.policy.compute_policy(self.target)

    def validate_problem(self):
        assert self.target.x is not None and self.target.u is not None and self.target.t is not None
        assert self.target.x.ndim == 2 and self.target.u.ndim == 2
        assert self.target.u.shape[-1] == self.input_dim
        assert self.target.x.shape[-1] == self.state_dim
        assert self.cost_params.Q.shape == (self.state_dim, self.state_dim)
        assert self.cost_params.R.shape == (self.input_dim, self.input_dim)

    def compute_policy(self, t_step, x_belief):
        pass

    def compute_input(self, t_step, x_belief):
        if t_step > self.final_time:
            self.u = self.u0
        else:
            step = int(t_step / self.dt)
            self.u = np.atleast_1d(self.u_bar[step] + self.K[step] @ (x_belief - self.x_bar[step]))
        return self.u


class StateDLQR(TemplateController):
    """controllers.py:398-437."""
    LQR_type = DLQR

    def __init__(self, dyn_sys, cost_params, target, dt=0.01, observer=None, delay=2, u0=None, **kwargs):
        super().__init__(dyn_sys=dyn_sys, cost_params=cost_params, dt=dt, observer=observer, delay=delay, u0=u0)
        self.target = target
        self.validate_problem()
        self.policy = self.LQR_type(dt=dt, model=dyn_sys, cost_params=self.cost_params)
        self.x_bar, self.u_bar, self.K = self.policy.compute_policy(target=self.target)

    def validate_problem(self):
        assert self.target.A is not None and self.target.B is not None and self.target.u is not None \
            and self.target.x is not None
        assert self.target.A.shape == (self.state_dim, self.state_dim)
        assert self.target.B.shape == (self.state_dim, self.input_dim)
        assert self.cost_params.Q.shape == (self.state_dim, self.state_dim)
        assert self.cost_params.R.shape == (self.input_dim, self.input_dim)

    def compute_policy(self, t_step, x_belief):
        pass

    def compute_input(self, t_step, x_belief):
        self.u = self.u_bar + self.K @ (x_belief - self.x_bar)
        return self.u
