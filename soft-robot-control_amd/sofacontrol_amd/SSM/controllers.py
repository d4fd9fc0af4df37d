"""Closed-loop controllers for SSM reduced models (protocol of sofacontrol/SSM/controllers.py:16-310; behaviour pinned
by the golden sequence g15 recorded from the imported reference).

Differences to the TPWL controllers (`tpwl/controllers.py`, whose control clock, plan tape and in-process solver
client are reused here): the belief comes straight from the measurement -- `SSMObserver` maps the observed outputs
onto the manifold coordinates with the model's `W_map`, a device call -- measurements that left the admissible set
`Y` are first projected back onto it (`Polyhedron(with_reproject=True)`, exact projection on the device), and every
replan starts from the current belief at the current control time and is taken as soon as it is requested (no
feedback gain around the plan: the SSM `scp` applies the plan's input directly)."""
import numpy as np

from .. import closed_loop_controller
from ..tpwl.controllers import GuSTOClient, _PlanTape, _Schedule, _on_grid
from ..utils import vq2qv


class SSMObserver:
    """SSM/controllers.py:302-310: z = measurement in [q; v] order, x = W_map(z - z_ref)."""

    def __init__(self, dyn_sys):
        self.dyn_sys = dyn_sys
        self.z = None
        self.x = None

    def update(self, u, y, dt, x=None):
        self.z = vq2qv(np.asarray(y, dtype=np.float64))
        self.x = self.dyn_sys.W_map(self.dyn_sys.zfyf_to_zy(zf=self.z))


class TemplateController(closed_loop_controller.TemplateController):
    """Measurement -> (re-projection onto Y) -> observer -> control clock (SSM/controllers.py:16-138)."""

    def __init__(self, dyn_sys, cost_params, dt=0.01, delay=2, u0=None, **kwargs):
        super().__init__()
        self.dyn_sys = dyn_sys
        self.cost_params = cost_params
        self.dt = dt
        self.t_delay = delay
        self.input_dim = dyn_sys.get_input_dim()
        self.state_dim = dyn_sys.get_state_dim()
        self.observer = SSMObserver(dyn_sys)
        self.u0 = np.zeros(self.input_dim) if u0 is None else u0
        self.u = self.u0
        self.Y = kwargs.pop('Y', None)
        self._clock = _Schedule(dt, delay)

    @property
    def t_compute(self):
        return self._clock.t_next

    def set_sim_timestep(self, dt):
        self.sim_dt = dt

    def validate_problem(self):
        raise NotImplementedError('Must be subclassed')

    def recompute_policy(self, t_step):
        return t_step == 0

    def compute_policy(self, t_step, x_belief):
        raise NotImplementedError('Must be subclassed')

    def compute_input(self, t_step, x_belief):
        raise NotImplementedError('Must be subclassed')

    def evaluate(self, sim_time, y, x, u_prev):
        if self.Y is not None and not self.Y.contains(y):
            y = self.Y.project_to_polyhedron(y)
        self.observer.update(None, y, None)
        clock = self._clock
        if not clock.started(sim_time):
            self.u = self.u0
        elif clock.due(sim_time):
            t = clock.t_next
            if self.recompute_policy(t):
                self.compute_policy(t, self.observer.x)
            self.u = self.compute_input(t, self.observer.x)
            clock.advance()
        self.u = np.atleast_1d(self.u)
        return self.u.copy()

    def save_controller_info(self):
        info = {'cost_params': self.cost_params}
        if self.dyn_sys is not None:
            info.update(dyn_sys_params=self.dyn_sys.get_sim_params(), state_dim=self.dyn_sys.get_state_dim(),
                        input_dim=self.dyn_sys.get_input_dim())
        return info


class scp(TemplateController):
    """Receding-horizon SCP on an SSM model (SSM/controllers.py:140-252).  Every `N_replan` control steps a plan is
    requested from the current belief at the current control time and its first N_replan steps are appended to the
    tape; the input is the tape's.  `client`: any GuSTOClientNode-protocol object (scp/ros.py:162-223); or
    `solver_node`: a `scp.standalone.GuSTOSolverNode` over an `SSMGuSTO` model, wrapped in the in-process
    `GuSTOClient` (asynchronous with wait=False)."""

    def __init__(self, dyn_sys, cost, dt, N_replan=None, delay=2, u0=None, wait=True, client=None, solver_node=None,
                 **kwargs):
        super().__init__(dyn_sys, None, dt=dt, delay=delay, u0=u0, **kwargs)
        if client is None:
            if solver_node is None:
                raise RuntimeError('scp needs client= (GuSTOClientNode protocol) or solver_node= (GuSTOSolverNode)')
            client = GuSTOClient(solver_node)
        self.GuSTO = client
        self.cost = cost
        self.N_replan = 1 if N_replan is None else N_replan
        self.wait = wait
        self.initialized = False
        self.solve_times = []
        self.z_opt_horizon, self.t_opt_horizon = [], []
        self.x_opt_current = self.u_opt_current = None
        self._tape = _PlanTape(dt, self.N_replan, grid_samples=True)

    t_opt = property(lambda self: self._tape.t)
    u_opt = property(lambda self: self._tape.u)
    x_opt = property(lambda self: self._tape.x)

    def u_bar(self, t):
        return self._tape.u_at(t)

    def x_bar(self, t):
        return self._tape.x_at(t)

    def recompute_policy(self, t_step):
        return int(round(_on_grid(t_step) / self.dt)) % self.N_replan == 0

    def run_GuSTO(self, t0, x0, wait):
        self.GuSTO.send_request(t0, x0, wait=wait)

    def compute_policy(self, t_step, x_belief):
        print('t_sim = {:.3f}'.format(t_step))
        self.run_GuSTO(t_step, x_belief, wait=self.wait if self.initialized else True)   # the first plan is awaited
        self.initialized = True
        self.update_policy()

    def update_policy(self, init=None):
        if not self.GuSTO.check_if_done():
            print('GuSTO cannot provide real-time compatibility, consider modifying problem')
            self.GuSTO.force_wait()
        t_plan, u_plan, x_plan, t_solve = self.GuSTO.get_solution(self.state_dim, self.input_dim)
        self.solve_times.append(t_solve)
        self._tape.append(t_plan, u_plan, x_plan)
        self.z_opt_horizon.append(self.dyn_sys.x_to_zfyf(x_plan))
        self.t_opt_horizon.append(t_plan)
        self.x_opt_current, self.u_opt_current = x_plan, u_plan

    def compute_input(self, t_step, x_belief):
        self.GuSTO.force_spin()
        return self.u_bar(t_step)

    def save_controller_info(self):
        return {'t_opt': self.t_opt, 'u_opt': self.u_opt, 'z_opt': self.dyn_sys.x_to_zfyf(self.x_opt, zf=True),
                'solve_times': self.solve_times, 'rollout_time': self.N_replan * self.dt,
                'z_rollout': self.z_opt_horizon, 't_rollout': self.t_opt_horizon}
