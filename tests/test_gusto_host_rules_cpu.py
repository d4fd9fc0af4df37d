"""The SCP rules of the generic-model GuSTO loop (sofacontrol_amd/scp/gusto.py: `_judge`, the (J, delta, omega) state machine of
sofacontrol/scp/gusto.py:371-428 / SURVEY.md appendix B) without a GPU: the host loop with a scripted QP in place of the device QP
must walk the same (delta, omega, accepted) sequence as the numpy statement of the reference loop (oracle/gusto.py) does with the
same QP answers.  The device QP itself is covered by the -m gpu tests; this is the host logic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd')]


class _ScriptedLOCP:
    """Stands in for scp.locp.LOCP: hands out a prepared list of (J, x, u) answers, records what update() was told."""

    def __init__(self, answers):
        self.answers, self.calls = list(answers), []
        self._sol = None

    def update(self, A, B, d, x0, xk, delta, omega, z=None, zf=None, u=None, full=True, **kw):
        self.calls.append((float(delta), float(omega), bool(full)))

    def solve(self):
        J, x, u = self.answers.pop(0)
        self._sol = (x, u, None)
        return J, True, type('S', (), {'solve_time': 0.0})()

    def get_solution(self):
        return self._sol


class _LinearModel:
    """x+ = A x + B u, f = Ac x + Bc u (exactly linear: the model-accuracy ratio is 0 unless the test perturbs it)."""
    nonlinear_observer = False

    def __init__(self, n, m, bend=0.0):
        rng = np.random.default_rng(3)
        self.Ac = -np.eye(n) + 0.1 * rng.standard_normal((n, n))
        self.Bc = rng.standard_normal((n, m))
        self.H = np.eye(2, n)
        self.bend = bend

    def get_continuous_dynamics(self, x, u):
        f = self.Ac @ x + self.Bc @ u + self.bend * np.sin(x)
        return f, self.Ac + self.bend * np.diag(np.cos(x)), self.Bc

    def get_discrete_dynamics(self, x, u, dt):
        n = x.shape[0]
        return np.eye(n) + dt * self.Ac, dt * self.Bc, np.zeros(n)


def _gusto(model, N, n, m, answers, **kw):
    from sofacontrol_amd.scp.gusto import GuSTO
    g = object.__new__(GuSTO)              # no device: the constructor would create a plan and solve once
    g.model, g.N, g.n_x, g.n_u, g.n_z, g.dt = model, N, n, m, 2, 0.1
    g.delta0, g.omega0, g.rho, g.beta_fail, g.gamma_fail = kw.get('delta0', 1.0), 1.0, kw.get('rho', 0.1), 0.5, 5.0
    g.omega_max, g.epsilon, g.convg_thresh = kw.get('omega_max', 1e10), 0.01, kw.get('convg_thresh', 1e-3)
    g.x_scale, g.f_scale = np.ones(n), np.ones(n)
    g.X = kw.get('X')
    g.nonlinear_observer = False
    g.max_gusto_iters, g.max_trace = kw.get('max_gusto_iters', 20), 32
    g.locp = _ScriptedLOCP(answers)
    g._fused = g._ssm = False
    return g


def test_host_loop_walks_the_reference_state_machine():
    """A scripted sequence that visits every branch: a step outside the trust region (omega x 5, same QP data), an inaccurate step
    (delta / 2, same QP data), an accepted step (re-linearised), a repeated (delta, omega) with a cost that did not drop (delta / 2
    on acceptance), another inaccurate step (rejected: delta / 2, same QP data), a last accepted one.  State rows: the next test."""
    N, n, m = 4, 3, 2
    x_init, u_init = np.zeros((N + 1, n)), np.zeros((N, m))
    far = np.full((N + 1, n), 3.0)                      # leaves delta0 = 1 by more than epsilon
    near = np.full((N + 1, n), 0.5)
    big = near + 0.4                                    # inside delta = 0.5, far enough for the bent dynamics to be inaccurate
    u1 = np.full((N, m), 0.1)
    # bend > 0 makes the dynamics nonlinear: a step of size 0.5 has a large model-accuracy ratio, a tiny step a small one
    model = _LinearModel(n, m, bend=4.0)
    tiny = np.full((N + 1, n), 1e-3)
    answers = [(10.0, far, u1),                          # itr 0: outside the trust region        -> omega 1 -> 5
               (10.0, near, u1),                         # itr 1: rho_k > rho but itr == 1         -> accepted all the same (gusto.py:383)
               (9.0, near + tiny, u1),                   # itr 2: accurate small step, accepted
               (9.5, near + 2 * tiny, u1),               # itr 3: same (delta, omega), J went up   -> accepted, delta halved
               (0.01, big + 2 * tiny, u1),               # itr 4: large step, small cost: rho_k > rho -> rejected, delta halved
               (8.0, near + 3 * tiny, u1)]               # itr 5: accepted
    g = _gusto(model, N, n, m, answers, max_gusto_iters=5, convg_thresh=1e-9, rho=0.05)      # (the 0.4 step's ratio is 0.08)
    g._solve_host_loop(np.zeros(n), u_init, x_init, None, None, None)
    calls = g.locp.calls
    assert [c[:2] for c in calls] == [(1.0, 1.0), (1.0, 5.0), (1.0, 5.0), (1.0, 5.0), (0.5, 5.0), (0.25, 5.0)], calls
    # QP data replaced (full=True) exactly behind accepted steps: the first QP, then after itr 1, 2, 3 -- not after the rejected ones
    assert [c[2] for c in calls] == [True, False, True, True, True, False], calls
    assert int(g.iters[0]) == 6
    tr = g.trace[0, :6]
    assert tr[0, 3] == -1.0 and np.all(tr[1:, 3] >= 0.0)           # rho_k = -1 exactly where the trust-region test failed
    assert tr[4, 3] > g.rho and tr[2, 3] < g.rho
    np.testing.assert_allclose(g.xopt, near + 3 * tiny)


def test_host_loop_state_rows_raise_the_penalty_and_block_convergence():
    from sofacontrol_amd.utils import Polyhedron
    N, n, m = 3, 2, 1
    model = _LinearModel(n, m, bend=0.0)
    bad = np.full((N + 1, n), 0.2)                       # violates x_0 <= 0.1 by 0.1 > epsilon
    good = np.full((N + 1, n), 0.05)
    u1 = np.zeros((N, m))
    answers = [(5.0, bad, u1), (4.0, bad, u1), (3.0, good, u1), (3.0, good, u1)]
    X = Polyhedron(np.array([[1.0, 0.0]]), np.array([0.1]))
    g = _gusto(model, N, n, m, answers, X=X, max_gusto_iters=10, convg_thresh=1e-6)
    g._solve_host_loop(np.zeros(n), u1, np.zeros((N + 1, n)), None, None, None)
    # accepted but infeasible twice (the second one would count as converged: identical iterate, yet the violation blocks it), then feasible,
    # then converged on the repeated feasible iterate
    assert [c[:2] for c in g.locp.calls] == [(1.0, 1.0), (1.0, 5.0), (1.0, 25.0), (1.0, 25.0)], g.locp.calls
    assert int(g.iters[0]) == 4
    np.testing.assert_allclose(g.xopt, good)


def test_host_loop_stops_on_the_penalty_cap_and_on_a_failed_qp(capsys):
    N, n, m = 2, 2, 1
    model = _LinearModel(n, m)
    far = np.full((N + 1, n), 9.0)
    u1 = np.zeros((N, m))
    g = _gusto(model, N, n, m, [(1.0, far, u1)] * 5, omega_max=100.0, max_gusto_iters=50)
    g._solve_host_loop(np.zeros(n), u1, np.zeros((N + 1, n)), None, None, None)
    assert [c[1] for c in g.locp.calls] == [1.0, 5.0, 25.0] and int(g.iters[0]) == 3      # 125 > omega_max ends the loop
    assert 'omega > omega_max' in capsys.readouterr().out
    np.testing.assert_array_equal(g.xopt, np.zeros((N + 1, n)))                           # never accepted: the initial guess stays

    class _Fails(_ScriptedLOCP):
        def solve(self):
            return np.inf, False, None
    g2 = _gusto(model, N, n, m, [])
    g2.locp = _Fails([])
    g2._solve_host_loop(np.zeros(n), u1, np.ones((N + 1, n)), None, None, None)
    assert 'cannot be solved' in capsys.readouterr().out
    np.testing.assert_array_equal(g2.xopt, np.ones((N + 1, n)))
