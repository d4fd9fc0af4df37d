"""GuSTO sequential convex programming on MI355X -- surface of sofacontrol/scp/gusto.py:25-490.

For a TPWL model (scp/models/tpwl.py adapter) the whole `solve` -- nearest-point linearisation along the
trajectory, the LOCP QP (Riccati interior point), trust-region / model-accuracy / convergence tests
and re-linearisation -- runs inside ONE persistent HIP kernel (csrc/gusto.hip: gusto_kernel), one
workgroup per rollout; `batch` independent rollouts (different x0 / targets, same model) can be solved
by one launch with `GuSTO.solve_batch`.  An SSM polynomial model (scp/models/ssm.py adapter: the reference's hardware loop,
examples/hardware/diamond_SSM.py:353-361) has its own persistent kernel (csrc/gusto_ssm.hip: analytic linearisation of the
dynamics and of the output map inside the loop).  Any other TemplateModel -- a user's Python dynamics, weighting-mode TPWL,
input-rate rows -- runs the SCP rules on the host around the device QP (`LOCP`): still no CPU arithmetic for the QP."""
import ctypes as C
import os
import time

import numpy as np

from .. import _lib
from .locp import LOCP, make_problem
from .models.tpwl import TPWLGuSTO
from .models.ssm import SSMGuSTO

#### Default variables for GuSTO (gusto.py:12-22) ####
DELTA0 = 1e4
OMEGA0 = 1
RHO = 0.1
BETA_FAIL = 0.5
BETA_SUCC = 2
EPSILON = 0.01
GAMMA_FAIL = 5
OMEGA_MAX = 1e10
MAX_ITERS = 500
CONVERGE = 0.1


class GuSTO:
    def __init__(self, model, N, dt, Qz, R, x0, u_init, x_init, z=None, u=None, Qzf=None, zf=None, U=None, X=None,
                 Xf=None, dU=None, verbose=0, visual=None, warm_start=True, **kwargs):
        self.model = model
        self.n_x = x0.shape[-1]
        self.n_u = R.shape[0]
        self.n_z = Qz.shape[0]
        self.dt = dt
        self.N = N
        self.Qz, self.R, self.Qzf = Qz, R, Qzf
        self.U, self.X, self.Xf, self.dU = U, X, Xf, dU
        self.verbose = verbose
        self.visual = visual
        self.locp_solve_time = None
        # gusto.py:83-119: parameters popped from kwargs
        self.delta0 = kwargs.pop('delta0', DELTA0)
        self.omega0 = kwargs.pop('omega0', OMEGA0)
        self.rho = kwargs.pop('rho', RHO)
        self.beta_fail = kwargs.pop('beta_fail', BETA_FAIL)
        self.beta_succ = kwargs.pop('beta_succ', BETA_SUCC)
        self.gamma_fail = kwargs.pop('gamma_fail', GAMMA_FAIL)
        self.omega_max = kwargs.pop('omega_max', OMEGA_MAX)
        self.epsilon = kwargs.pop('epsilon', EPSILON)
        self.convg_thresh = kwargs.pop('convg_thresh', CONVERGE)
        self.x_char = kwargs.pop('x_char', np.ones(self.n_x))
        self.x_scale = 1. / np.abs(self.x_char)
        self.f_char = kwargs.pop('f_char', np.ones(self.n_x))
        self.f_scale = 1. / np.abs(self.f_char)
        self.jit = kwargs.pop('jit', True)      # meaningless here; accepted for signature parity
        user_max_iters = kwargs.pop('max_gusto_iters', MAX_ITERS)
        # keep_solver_state=True: the first QP of every solve starts from the previous solve's minimiser and multipliers -- what the
        # reference's warm_start=True does through its persistent cvxpy problem (locp.py:181).  Off by default: solves are independent.
        self.keep_solver_state = bool(kwargs.pop('keep_solver_state', False))
        # first_solve_cap: cap of the constructor's own solve (the reference runs it with its default of 500 SCP iterations whatever
        # max_gusto_iters says, gusto.py:142-147 -- None keeps that); a caller that only wants the plan built can ask for less
        first_solve_cap = kwargs.pop('first_solve_cap', None)
        self.batch = int(kwargs.pop('batch', 1))
        self.max_trace = int(kwargs.pop('max_trace', 64))
        self.x_k = None
        self.u_k = None
        self.nonlinear_observer = model.nonlinear_observer
        # input-rate constraints couple the stages: they go through the generic loop around the (augmented) device QP; so does the
        # input_nullspace term (locp.py:258-261: a norm over the whole input sequence, LOCP._solve_nullspace)
        nullspace = kwargs.get('input_nullspace') is not None
        self._fused = (isinstance(model, TPWLGuSTO) and not self.nonlinear_observer and dU is None and not nullspace and
                       getattr(model.dyn_sys, 'tpwl_method', 'nn') == 'nn')
        # an SSM model: the whole solve in csrc/gusto_ssm.hip (no terminal cost, no rate rows; the state polyhedron is applied to the
        # states by the reference's own test, gusto.py:185-201, so its matrix must have n_x columns)
        self._ssm = (isinstance(model, SSMGuSTO) and dU is None and Qzf is None and Xf is None and not nullspace and
                     (X is None or np.asarray(X.A).shape[1] == self.n_x) and hasattr(model.dyn_sys, 'handle') and
                     not os.environ.get('SRH_GUSTO_SSM_HOST_LOOP'))          # (that knob: the host loop, for A/B runs and tests)
        self._plan = C.c_void_p()
        self.trace = None
        self.iters = None
        self.status = None
        if self._ssm:
            self._create_ssm_plan(model, N, dt, Qz, R, U, X)
            self._fused = True
        elif self._fused:
            prob, self._keep = make_problem(N, model.H, Qz, R, Qzf, U, X, Xf, dU, None, True)
            par = self._params(MAX_ITERS)
            xc, fc = _lib.f64(self.x_char), _lib.f64(self.f_char)
            _lib.check(_lib.lib().sgusto_plan_create(C.byref(self._plan), model.dyn_sys.handle_for(dt), C.byref(prob),
                                                     C.byref(par), C.c_double(dt), C.c_int64(self.batch),
                                                     _lib.dptr(xc), _lib.dptr(fc), C.c_int(self.max_trace)),
                       'sgusto_plan_create')
        else:
            self.locp = LOCP(self.N, self.model.H, self.Qz, self.R, Qzf=self.Qzf, U=self.U, X=self.X, Xf=self.Xf,
                             dU=self.dU, verbose=(verbose == 2), warm_start=warm_start, x_char=self.x_char,
                             nonlinear_observer=self.nonlinear_observer, **kwargs)
        # gusto.py:142-147: the first solve may take up to MAX_ITERS, then the user's limit applies
        self.max_gusto_iters = MAX_ITERS if first_solve_cap is None else int(first_solve_cap)
        if x0.ndim == 1:
            self.solve(x0, u_init, x_init, z, zf, u)
        else:
            self.solve_batch(x0, u_init, x_init, z, zf, u)
        self.max_gusto_iters = user_max_iters
        if self._ssm:                                # (solve_batch sets the cap of every call)
            if self.keep_solver_state:
                _lib.check(_lib.lib().sgusto_ssm_plan_set_warm_across(self._plan, C.c_int(1)), 'set_warm_across')
        elif self._fused:
            _lib.check(_lib.lib().sgusto_plan_set_max_iters(self._plan, C.c_int(int(user_max_iters))), 'set_max_iters')
            if self.keep_solver_state:
                _lib.check(_lib.lib().sgusto_plan_set_warm_across(self._plan, C.c_int(1)), 'set_warm_across')
                if not self.solver_state_kept:
                    import warnings
                    warnings.warn('GuSTO(keep_solver_state=True): this plan\'s kernels (%s) start every solve cold -- only the lean kernels '
                                  'with box input rows keep the solver state between solves' % self.kernel_info['kernel'])

    def _create_ssm_plan(self, model, N, dt, Qz, R, U, X):
        """The resident plan of csrc/gusto_ssm.hip.  With a nonlinear output map the QP is posed in the augmented state [x ; zeta]
        (LOCP._init_augmented: H_a = [0 I], X on zeta, zero trust-region scale on zeta) -- the same problem data the host loop
        handed to the QP kernel; the kernel fills the per-stage matrices itself."""
        sys_ = model.dyn_sys
        proto = LOCP(N, model.H, Qz, R, Qzf=None, U=U, X=X, Xf=None, dU=None, x_char=self.x_char,
                     nonlinear_observer=self.nonlinear_observer)
        self._keep = (proto, proto._prob, proto._keep)
        Hm = _lib.f64(np.asarray(model.H).reshape(self.n_z, self.n_x))
        fc = _lib.f64(self.f_char)
        nX = 0 if X is None else int(np.asarray(X.A).shape[0])
        XA = None if X is None else _lib.f64(np.asarray(X.A).reshape(nX, self.n_x))
        Xb = None if X is None else _lib.f64(np.asarray(X.b).reshape(nX))
        par = self._params(MAX_ITERS)
        _lib.check(_lib.lib().sgusto_ssm_plan_create(C.byref(self._plan), sys_.handle, C.byref(proto._prob), C.byref(par),
                                                     C.c_double(dt), C.c_int(sys_._mode()), C.c_int64(self.batch), _lib.dptr(fc),
                                                     _lib.dptr(Hm), C.c_int(nX), _lib.dptr(XA), _lib.dptr(Xb),
                                                     C.c_int(self.max_trace)), 'sgusto_ssm_plan_create')

    def _params(self, max_iters):
        return _lib.SGustoParams(float(self.delta0), float(self.omega0), float(self.rho), float(self.beta_fail),
                                 float(self.gamma_fail), float(self.epsilon), float(self.omega_max),
                                 float(self.convg_thresh), int(max_iters))

    def __del__(self):
        try:
            if self._plan:
                (_lib.lib().sgusto_ssm_plan_destroy if self._ssm else _lib.lib().sgusto_plan_destroy)(self._plan)
                self._plan = C.c_void_p()
        except Exception:
            pass

    @property
    def plan(self):
        return self._plan

    @property
    def solver_state_kept(self):
        """True when keep_solver_state was requested AND the plan's kernels honour it (sgusto_plan_warm_across_active)."""
        if not self.keep_solver_state or not self._fused:
            return False
        if self._ssm:
            return True
        a = C.c_int(0)
        _lib.check(_lib.lib().sgusto_plan_warm_across_active(self._plan, C.byref(a)), 'sgusto_plan_warm_across_active')
        return bool(a.value)

    @property
    def variant(self):
        """(split panel, compile-time n_u, compile-time n_x) of the kernel instantiation this plan launches; zeros mean
        run-time extents ((False, 0, 0) = the all-sizes kernel).  None for the host-loop models."""
        if not self._fused or self._ssm:
            return None
        sp, mu, nx = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().sgusto_plan_variant(self._plan, C.byref(sp), C.byref(mu), C.byref(nx)), 'sgusto_plan_variant')
        return bool(sp.value), mu.value, nx.value

    @property
    def costs(self):
        """(batch,) optimal LOCP value of the solution every rollout of the last solve returned (sgusto_plan_costs; fused
        TPWL plans only): what a sharded batch gathers to pick its best rollout (distributed.gather_rollout_costs)."""
        if not self._fused or self._ssm:
            raise NotImplementedError('GuSTO.costs: per-rollout costs are kept by the resident TPWL plan only')
        J = np.empty(self.batch)
        _lib.check(_lib.lib().sgusto_plan_costs(self._plan, _lib.dptr(J)), 'sgusto_plan_costs')
        return J

    @property
    def kernel_info(self):
        """What the last solve of this plan launched (sgusto_plan_info): the kernel family, the template arguments of
        the instantiation -- the name a rocprof trace shows, e.g. 'lean<4, 60, 4, 50, 7, 4>' for BASELINE C2 -- and how
        many rollouts the lean kernel handed to the fused one.  The host-loop models report their LOCP plan's kernels."""
        if self._ssm:
            return {'family': 'ssm', 'kernel': 'gusto_ssm_kernel', 'lean': None, 'fused': None, 'handed_over': 0}
        if not self._fused:
            return self.locp.kernel_info
        info = _lib.SrhKernelInfo()
        _lib.check(_lib.lib().sgusto_plan_info(self._plan, C.byref(info)), 'sgusto_plan_info')
        return info.as_dict()

    # ---- helper tests with the reference's names (host arrays; used by the generic loop / by users)
    def is_converged(self, x, u):
        dx = (1. / self.n_x) * np.sum(np.linalg.norm(np.multiply(self.x_scale, x - self.x_k), axis=1))
        dsol = (1. / self.N) * dx
        return dsol, bool(dsol <= self.convg_thresh)

    def is_valid_iteration(self, itr):
        return itr <= self.max_gusto_iters

    def is_in_trust_region(self, x, delta):
        max_diff = np.max(np.linalg.norm(np.multiply(self.x_scale, x - self.x_k), np.inf, axis=1))
        if max_diff - delta > self.epsilon:
            return max_diff, False
        return 0.0, True

    def state_constraints_violated(self, x):
        max_violation = 0.0
        if self.X is not None:
            for i in range(x.shape[0]):
                max_violation = max(max_violation, self.X.get_constraint_violation(x[i, :]))
        return max_violation, not (max_violation > self.epsilon)

    def compute_accuracy(self, x, u, J):
        error = 0
        approx = 0
        if hasattr(self.model, 'get_continuous_dynamics_batch'):
            # same sums as the loop below with the two trajectories linearised in one device call each
            fk, Ak, Bk = self.model.get_continuous_dynamics_batch(self.x_k[:-1], self.u_k)
            f, _, _ = self.model.get_continuous_dynamics_batch(x[:-1], u)
            for i in range(x.shape[0] - 1):
                f_approx = fk[i] + Ak[i] @ (x[i, :] - self.x_k[i, :]) + Bk[i] @ (u[i, :] - self.u_k[i, :])
                error += self.dt * np.linalg.norm(np.multiply(self.f_scale, f[i] - f_approx), 2)
                approx += self.dt * np.linalg.norm(np.multiply(self.f_scale, f_approx), 2)
            return error / (J + approx)
        for i in range(x.shape[0] - 1):
            fk, Ak, Bk = self.model.get_continuous_dynamics(self.x_k[i, :], self.u_k[i, :])
            f, _, _ = self.model.get_continuous_dynamics(x[i, :], u[i, :])
            f_approx = fk + Ak @ (x[i, :] - self.x_k[i, :]) + Bk @ (u[i, :] - self.u_k[i, :])
            error += self.dt * np.linalg.norm(np.multiply(self.f_scale, f - f_approx), 2)
            approx += self.dt * np.linalg.norm(np.multiply(self.f_scale, f_approx), 2)
        return error / (J + approx)

    def get_traj_dynamics(self, x, u):
        if hasattr(self.model, 'get_discrete_dynamics_batch'):
            A, B, d = self.model.get_discrete_dynamics_batch(x[:-1], u, self.dt)
            return list(A), list(B), list(d)
        A_d, B_d, d_d = [], [], []
        for i in range(x.shape[0] - 1):
            A, B, d = self.model.get_discrete_dynamics(x[i, :], u[i, :], self.dt)
            A_d.append(A); B_d.append(B); d_d.append(d)
        return A_d, B_d, d_d

    def get_observer_linearizations(self, x, u):
        """gusto.py:240-251."""
        if hasattr(self.model, 'get_observer_jacobians_batch'):
            H, c = self.model.get_observer_jacobians_batch(x, self.dt)
            return list(H), list(c)
        H_d, c_d = [], []
        for i in range(x.shape[0]):
            H, c = self.model.get_observer_jacobians(x[i, :], None, self.dt)
            H_d.append(H); c_d.append(c)
        return H_d, c_d

    # ---- solve
    def solve_batch(self, x0, u_init, x_init, z=None, zf=None, u=None):
        """`batch` independent rollouts in one launch: x0 (B,n_x), u_init (B,N,n_u), x_init (B,N+1,n_x),
        z (B,N+1,n_z) ..."""
        if not self._fused:
            raise RuntimeError('solve_batch needs a TPWLGuSTO or SSMGuSTO model (a resident device plan)')
        B, N, n, m, nz = self.batch, self.N, self.n_x, self.n_u, self.n_z
        f = _lib.f64
        x0 = f(np.asarray(x0).reshape(B, n)); u_init = f(np.asarray(u_init).reshape(B, N, m))
        x_init = f(np.asarray(x_init).reshape(B, N + 1, n))
        z = None if z is None else f(np.asarray(z).reshape(B, N + 1, nz))
        zf = None if (zf is None or self.Qzf is None) else f(np.asarray(zf).reshape(B, nz))
        u = None if u is None else f(np.asarray(u).reshape(B, N, m))
        xo = np.empty((B, N + 1, n)); uo = np.empty((B, N, m)); zo = np.empty((B, N + 1, nz))
        iters = np.empty(B, dtype=np.int32); status = np.empty(B, dtype=np.int32)
        trace = np.full((B, self.max_trace, 4), np.nan) if self.max_trace > 0 else None
        if self._ssm:
            _lib.check(_lib.lib().sgusto_ssm_plan_set_max_iters(self._plan, C.c_int(int(self.max_gusto_iters))), 'set_max_iters')
            t0 = time.time()
            _lib.check(_lib.lib().sgusto_ssm_plan_solve(self._plan, _lib.dptr(x0), _lib.dptr(u_init), _lib.dptr(x_init), _lib.dptr(z),
                                                        _lib.dptr(u), _lib.dptr(xo), _lib.dptr(uo), _lib.dptr(zo), _lib.iptr(iters),
                                                        _lib.iptr(status), _lib.dptr(trace)), 'sgusto_ssm_plan_solve')
            self.locp_solve_time = time.time() - t0
            self.iters, self.status, self.trace = iters, status, trace
            self.xopt, self.uopt, self.zopt = xo, uo, zo
            return xo, uo, zo
        _lib.check(_lib.lib().sgusto_plan_set_max_iters(self._plan, C.c_int(int(self.max_gusto_iters))), 'set_max_iters')
        t0 = time.time()
        _lib.check(_lib.lib().sgusto_plan_solve(self._plan, _lib.dptr(x0), _lib.dptr(u_init), _lib.dptr(x_init),
                                                _lib.dptr(z), _lib.dptr(zf), _lib.dptr(u), _lib.dptr(xo), _lib.dptr(uo),
                                                _lib.dptr(zo), _lib.iptr(iters), _lib.iptr(status), _lib.dptr(trace)),
                   'sgusto_plan_solve')
        self.locp_solve_time = time.time() - t0
        self.iters, self.status, self.trace = iters, status, trace
        self.xopt, self.uopt, self.zopt = xo, uo, zo
        return xo, uo, zo

    # ---- asynchronous solve (fused plans): the request runs on the plan's own HIP stream
    def solve_begin(self, x0, u_init, x_init, z=None, zf=None, u=None):
        """Enqueue one solve (same arguments as `solve` / `solve_batch`) and return immediately; `solve_done()` polls,
        `solve_end()` waits and installs the result like `solve` does.  (scp/ros.py:183-223 `send_request(wait=False)`.)"""
        if not self._fused or self._ssm:
            raise RuntimeError('asynchronous solves need the fused TPWL plan')
        B, N, n, m, nz = self.batch, self.N, self.n_x, self.n_u, self.n_z
        f = _lib.f64
        x0 = f(np.asarray(x0).reshape(B, n)); u_init = f(np.asarray(u_init).reshape(B, N, m))
        x_init = f(np.asarray(x_init).reshape(B, N + 1, n))
        z = None if z is None else f(np.asarray(z).reshape(B, N + 1, nz))
        zf = None if (zf is None or self.Qzf is None) else f(np.asarray(zf).reshape(B, nz))
        u = None if u is None else f(np.asarray(u).reshape(B, N, m))
        _lib.check(_lib.lib().sgusto_plan_set_max_iters(self._plan, C.c_int(int(self.max_gusto_iters))), 'set_max_iters')
        self._t_begin = time.time()
        _lib.check(_lib.lib().sgusto_plan_solve_begin(self._plan, _lib.dptr(x0), _lib.dptr(u_init), _lib.dptr(x_init),
                                                      _lib.dptr(z), _lib.dptr(zf), _lib.dptr(u),
                                                      C.c_int(1 if self.max_trace > 0 else 0)), 'sgusto_plan_solve_begin')

    def prepare_async(self):
        _lib.check(_lib.lib().sgusto_plan_prepare_async(self._plan), 'sgusto_plan_prepare_async')

    def solve_done(self):
        done = C.c_int(0)
        _lib.check(_lib.lib().sgusto_plan_solve_done(self._plan, C.byref(done)), 'sgusto_plan_solve_done')
        return bool(done.value)

    def solve_end(self):
        B, N, n, m, nz = self.batch, self.N, self.n_x, self.n_u, self.n_z
        xo = np.empty((B, N + 1, n)); uo = np.empty((B, N, m)); zo = np.empty((B, N + 1, nz))
        iters = np.empty(B, dtype=np.int32); status = np.empty(B, dtype=np.int32)
        trace = np.full((B, self.max_trace, 4), np.nan) if self.max_trace > 0 else None
        _lib.check(_lib.lib().sgusto_plan_solve_end(self._plan, _lib.dptr(xo), _lib.dptr(uo), _lib.dptr(zo), _lib.iptr(iters),
                                                    _lib.iptr(status), _lib.dptr(trace)), 'sgusto_plan_solve_end')
        # the solver's own time (device events around the request: copies in, kernels, copies out), as the reference reports
        # it -- not the begin-to-collect wall time, which contains whatever the caller did in between
        ms = C.c_double(-1.0)
        _lib.check(_lib.lib().sgusto_plan_last_async_ms(self._plan, C.byref(ms)), 'sgusto_plan_last_async_ms')
        self.request_wall_time = time.time() - self._t_begin
        self.locp_solve_time = ms.value * 1e-3 if ms.value >= 0 else self.request_wall_time
        self.iters, self.status, self.trace = iters, status, trace
        if B == 1:
            self.xopt, self.uopt, self.zopt = xo[0], uo[0], zo[0]
            self.x_k, self.u_k = self.xopt.copy(), self.uopt.copy()
        else:
            self.xopt, self.uopt, self.zopt = xo, uo, zo
        return self.xopt, self.uopt, self.zopt

    def solve(self, x0, u_init, x_init, z=None, zf=None, u=None):
        """gusto.py:283-487."""
        if self._fused:
            if self.batch != 1:
                raise RuntimeError('GuSTO was built with batch=%d: use solve_batch' % self.batch)
            xo, uo, zo = self.solve_batch(x0[None], u_init[None], x_init[None], None if z is None else z[None],
                                          None if zf is None else zf[None], None if u is None else u[None])
            self.xopt, self.uopt, self.zopt = xo[0], uo[0], zo[0]
            self.x_k, self.u_k = self.xopt.copy(), self.uopt.copy()
            st = int(self.status[0])
            if st == 1:
                print('Iteration {} of problem cannot be solved, see solver status for more information'.format(int(self.iters[0])))
            elif st == 2:
                print('omega > omega_max, solution did not converge')
            elif st == 3:
                print('Max iterations, solution did not converge')
            elif self.verbose >= 1:
                print('Solved in {} iterations/{:.3f} seconds'.format(int(self.iters[0]), self.locp_solve_time))
            return
        self._solve_host_loop(x0, u_init, x_init, z, zf, u)

    # ---- generic models: the SCP rules on the host, the QP on the device
    def _linearise(self):
        """Stage matrices (and output maps) of the current iterate (gusto.py:225-251)."""
        dyn = self.get_traj_dynamics(self.x_k, self.u_k)
        obs = self.get_observer_linearizations(self.x_k, self.u_k) if self.nonlinear_observer else (None, None)
        return dyn, obs

    def _judge(self, st, J, x_new, u_new, itr):
        """One transition of the (J, delta, omega) state machine the kernels and oracle/gusto.py share (gusto.py:371-428, SURVEY
        appendix B).  `st` holds delta, omega and the previous accepted (J, delta, omega); returns (accepted, converged, rho_k)."""
        _, inside = self.is_in_trust_region(x_new, st['delta'])
        if not inside:                                   # step leaves the trust region: harder penalty, same QP data
            st['omega'] *= self.gamma_fail
            return False, False, -1.0
        rho_k = self.compute_accuracy(x_new, u_new, J)
        if rho_k > self.rho and itr != 1:                # model too inaccurate over this step: shrink, same QP data
            st['delta'] *= self.beta_fail
            return False, False, rho_k
        if st['prev'] == (st['delta'], st['omega']) and st['J_prev'] <= J:
            st['delta'] *= self.beta_fail
        st['prev'], st['J_prev'] = (st['delta'], st['omega']), J
        _, feasible = self.state_constraints_violated(x_new)
        if not feasible:
            st['omega'] *= self.gamma_fail
        _, settled = self.is_converged(x_new, u_new)
        return True, settled and feasible, rho_k

    def _solve_host_loop(self, x0, u_init, x_init, z, zf, u):
        """GuSTO.solve for a generic TemplateModel (gusto.py:283-487): linearise -> device QP -> `_judge` -> accept / re-linearise."""
        self.x_k, self.u_k = x_init, u_init
        (A_d, B_d, d_d), (H_d, c_d) = self._linearise()
        st = {'delta': self.delta0, 'omega': self.omega0, 'prev': (np.inf, np.inf), 'J_prev': np.inf}
        fresh, done, itr, t_locp, log = True, False, 0, 0.0, []
        while self.is_valid_iteration(itr) and not done and st['omega'] <= self.omega_max:
            self.locp.update(A_d, B_d, d_d, x0, self.x_k, st['delta'], st['omega'], z=z, zf=zf, u=u, full=fresh, Hd=H_d, cd=c_d)
            fresh = False
            J, ok, stats = self.locp.solve()
            if not ok:                                   # gusto.py:357-365: report, keep the last accepted iterate
                print('Iteration {} of problem cannot be solved, see solver status for more information'.format(itr))
                self.xopt, self.uopt = np.copy(self.x_k), np.copy(self.u_k)
                self.zopt = (self.model.dyn_sys.C_map(self.xopt.T) if self.nonlinear_observer
                             else np.transpose(self.model.H @ self.xopt.T))
                return
            t_locp += stats.solve_time
            x_new, u_new, _ = self.locp.get_solution()
            row = [J, st['delta'], st['omega'], -1.0]   # (J, delta, omega, rho) per QP, like the kernels' trace
            fresh, done, row[3] = self._judge(st, J, x_new, u_new, itr)
            log.append(row)
            itr += 1
            if fresh:
                self.x_k, self.u_k = x_new.copy(), u_new.copy()
                if self.max_gusto_iters >= 1:
                    (A_d, B_d, d_d), (H_d, c_d) = self._linearise()
        if st['omega'] > self.omega_max:
            print('omega > omega_max, solution did not converge')
        if not self.is_valid_iteration(itr - 1):
            print('Max iterations, solution did not converge')
        self.xopt, self.uopt = np.copy(self.x_k), np.copy(self.u_k)
        self.zopt = np.transpose(self.model.H @ self.xopt.T)
        self.locp_solve_time = t_locp
        self.iters = np.array([itr], dtype=np.int32)
        if self.max_trace > 0:
            self.trace = np.full((1, max(self.max_trace, len(log)), 4), np.nan)
            self.trace[0, :len(log)] = np.asarray(log, dtype=np.float64).reshape(-1, 4)

    def get_solution(self):
        return self.xopt, self.uopt, self.zopt, self.locp_solve_time
