"""GPU parity: closed-loop controller glue (tpwl/controllers.py) against the imported reference driven by the
same scripted (sim_time, y, x, u_prev) sequence (golden g8; the ROS client is replaced on both sides by the
same deterministic fake that returns an analytic 'solution')."""
import io
import contextlib

import numpy as np
import pytest

from helpers import golden_problem, product_tpwl, tip_selector

pytestmark = pytest.mark.gpu


class FakeGuSTOClient:
    """Mirror of tests/golden/make_golden.py:FakeGuSTOClient."""
    N, dt_g = 8, 0.05

    def __init__(self):
        self.done = False

    def send_request(self, t0, x0, wait=True):
        self.t0, self.x0 = float(t0), np.asarray(x0, dtype=float).copy()
        self.done = True

    def force_spin(self):
        pass

    def check_if_done(self):
        return self.done

    def force_wait(self):
        pass

    def get_solution(self, n_x, n_u):
        t = self.t0 + self.dt_g * np.arange(self.N + 1)
        x = np.stack([self.x0 * np.cos(3 * (tt - self.t0)) + 0.01 * np.sin(tt + np.arange(n_x)) for tt in t])
        u = np.stack([50.0 + 40.0 * np.sin(2 * tt + np.arange(n_u)) for tt in t[:-1]])
        return t, u, x, 0.0123


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_scp_controller_matches_reference(golden):
    from sofacontrol_amd.tpwl import controllers as ctl
    from sofacontrol_amd.utils import QuadraticCost
    g = golden('g8_controllers')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 40, q_scale=0.05)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    cost = QuadraticCost(Q=g['Q'], R=g['R'])
    c = quiet(ctl.scp, tp, cost, 0.01, N_replan=3, delay=0.02, client=FakeGuSTOClient())
    c.set_sim_timestep(0.01)
    np.testing.assert_allclose(np.stack(c.K), g['K'], rtol=0, atol=1e-8 * np.abs(g['K']).max())
    us = [quiet(c.evaluate, k * 0.01, None, g['x_full'][k], np.zeros(3)) for k in range(g['x_full'].shape[0])]
    np.testing.assert_allclose(np.stack(us), g['u'], rtol=0, atol=1e-7 * max(1.0, np.abs(g['u']).max()))
    info = c.save_controller_info()
    assert sorted(info) == ['rollout_time', 'solve_times', 't_opt', 't_rollout', 'u_opt', 'z_opt', 'z_rollout']
    np.testing.assert_allclose(info['t_opt'], g['t_opt'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(info['u_opt'], g['u_opt'], rtol=0, atol=1e-9 * np.abs(g['u_opt']).max())
    np.testing.assert_allclose(info['z_opt'], g['z_opt'], rtol=0, atol=1e-9 * max(1.0, np.abs(g['z_opt']).max()))
    assert len(info['solve_times']) == int(g['n_solves']) and info['rollout_time'] == float(g['rollout_time'])


def test_ilqr_controller_matches_reference(golden):
    from sofacontrol_amd.tpwl import controllers as ctl
    from sofacontrol_amd.tpwl.tpwl_utils import Target
    from sofacontrol_amd.utils import QuadraticCost
    g = golden('g8_controllers')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 40, q_scale=0.05)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    tgt = Target()
    tgt.t, tgt.z, tgt.Hf = g['il_t'], g['il_z'], Hf
    Qz = np.diag([0, 0, 0, 100., 100., 0])
    ci = quiet(ctl.ilqr, tp, QuadraticCost(Q=Qz, R=1e-3 * np.eye(3), Qf=Qz), tgt, dt=0.02, delay=0.0)
    ci.set_sim_timestep(0.01)
    us = [quiet(ci.evaluate, k * 0.01, None, g['x_full'][k], np.zeros(3)) for k in range(8)]
    np.testing.assert_allclose(ci.x_bar, g['il_xbar'], rtol=0, atol=1e-6 * max(1.0, np.abs(g['il_xbar']).max()))
    np.testing.assert_allclose(np.stack(us), g['il_u'], rtol=0, atol=1e-5 * max(1.0, np.abs(g['il_u']).max()))


def test_scp_controller_with_in_process_solver_node(golden):
    """End to end on the device: projection -> GuSTO replans (fused kernel) -> LQR feedback."""
    from sofacontrol_amd.tpwl import controllers as ctl
    from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
    from sofacontrol_amd.scp.standalone import GuSTOSolverNode
    from sofacontrol_amd.utils import QuadraticCost, HyperRectangle
    g6 = golden('g6_gusto')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    gm = TPWLGuSTO(tp)
    quiet(gm.pre_discretize, 0.05)
    node = quiet(GuSTOSolverNode, gm, 12, 0.05, g6['Qz'], g6['R'], np.zeros(8), t=g6['t'], z=g6['zt'],
                 U=HyperRectangle([800.] * 3, [0.] * 3), convg_thresh=1e-3, max_gusto_iters=5)
    H = tp.H
    cost = QuadraticCost(Q=H.T @ g6['Qz'] @ H + 1e-2 * np.eye(8), R=1e-3 * np.eye(3))
    c = quiet(ctl.scp, tp, cost, 0.05, N_replan=2, delay=0.0, solver_node=node)
    c.set_sim_timestep(0.05)
    x_ref = np.concatenate((v_ref, q_ref))
    rng = np.random.default_rng(0)
    for k in range(6):
        u = quiet(c.evaluate, k * 0.05, None, x_ref + 1e-3 * rng.standard_normal(x_ref.size), np.zeros(3))
        assert u.shape == (3,) and np.all(np.isfinite(u))
    info = c.save_controller_info()
    assert len(info['solve_times']) >= 2 and info['t_opt'][0] == 0.0


def test_traj_tracking_and_state_dlqr_controllers(golden):
    """TrajTracking and StateDLQR controllers (controllers.py:349-437) against the imported reference (g13)."""
    from sofacontrol_amd.tpwl import controllers as ctl
    from sofacontrol_amd.tpwl.tpwl_utils import Target, DynamicsTarget
    from sofacontrol_amd.utils import QuadraticCost
    g = golden('g13_controllers2')
    model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 40, q_scale=0.05)
    tp = product_tpwl(model, U, q_ref, v_ref, Hf)
    cost = QuadraticCost(Q=g['Q'], R=g['R'])
    dt = 0.02
    tgt = Target()
    tgt.t, tgt.u, tgt.x = g['tt_t'], g['tt_u_target'], g['tt_x_target']
    c = quiet(ctl.TrajTracking, tp, cost, tgt, dt=dt, delay=0.02)
    c.set_sim_timestep(dt)
    np.testing.assert_allclose(np.asarray(c.K), g['tt_K'], rtol=0, atol=1e-8 * np.abs(g['tt_K']).max())
    us = [quiet(c.evaluate, k * dt, None, g['tt_x_full'][k], np.zeros(3)) for k in range(13)]
    np.testing.assert_allclose(np.stack(us), g['tt_u'], rtol=0, atol=1e-7 * max(1.0, np.abs(g['tt_u']).max()))
    dtg = DynamicsTarget()
    dtg.A, dtg.B = model['A_c'][2], model['B_c'][2]
    dtg.x = np.concatenate((model['v'][2], model['q'][2]))
    dtg.u = model['u'][2]
    c2 = quiet(ctl.StateDLQR, tp, cost, dtg, dt=dt, delay=0.0)
    c2.set_sim_timestep(dt)
    np.testing.assert_allclose(np.asarray(c2.K), g['dl_K'], rtol=0, atol=2e-3 * np.abs(g['dl_K']).max())   # 1e-4 stopping rule
    us2 = [quiet(c2.evaluate, k * dt, None, g['dl_x_full'][k], np.zeros(3)) for k in range(4)]
    np.testing.assert_allclose(np.stack(us2), g['dl_u'], rtol=0, atol=1e-3 * max(1.0, np.abs(g['dl_u']).max()))


def _c2_node(max_iters=500):
    """A GuSTOSolverNode on BASELINE config C2 (Diamond r = 30, N = 50): one request = tens of milliseconds of GPU work."""
    import bench
    import workloads as wl
    from sofacontrol_amd.scp.standalone import GuSTOSolverNode
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2()
    tp, gm = bench.build_model(w)
    node = quiet(GuSTOSolverNode, gm, w['N'], w['dt'], w['Qz'], w['R'], np.zeros(2 * w['r']), t=w['t'], z=w['z'],
                 U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), convg_thresh=1e-3, max_gusto_iters=max_iters)
    return w, tp, node


def test_gusto_client_wait_false_is_asynchronous():
    """send_request(wait=False) (scp/ros.py:183-198) returns while the GPU is still solving; check_if_done polls,
    force_wait blocks (ros.py:199-210); the result is the one the synchronous request gives."""
    import time
    from sofacontrol_amd.tpwl import controllers as ctl
    w, tp, node_a = _c2_node()
    _, _, node_s = _c2_node()
    rng = np.random.default_rng(3)
    x0 = node_a.xopt[2] + 1e-2 * rng.standard_normal(2 * w['r'])
    t0 = 2 * w['dt']
    ca, cs = ctl.GuSTOClient(node_a), ctl.GuSTOClient(node_s)
    t_start = time.perf_counter()
    quiet(cs.send_request, t0, x0, wait=True)
    t_sync = time.perf_counter() - t_start
    assert cs.check_if_done()
    t_start = time.perf_counter()
    ca.send_request(t0, x0, wait=False)
    t_send = time.perf_counter() - t_start
    done_at_once = ca.check_if_done()
    polls = 0
    while not ca.check_if_done():
        polls += 1
        time.sleep(1e-3)
    t_total = time.perf_counter() - t_start
    ca.force_wait()
    assert t_send < 0.25 * t_sync, (t_send, t_sync)           # the request came back long before a solve can finish
    assert not done_at_once and polls >= 1, (t_send, t_total, t_sync)
    ta, ua, xa, _ = ca.get_solution(2 * w['r'], w['m'])
    ts, us, xs, _ = cs.get_solution(2 * w['r'], w['m'])
    np.testing.assert_array_equal(ta, ts)
    np.testing.assert_array_equal(ua, us)
    np.testing.assert_array_equal(xa, xs)
    # the reported solve time of an asynchronous request is the solver's own (device events), not the begin-to-collect wall
    # time: collect a second request late and compare
    ca.send_request(t0, x0, wait=False)
    time.sleep(0.25)
    ca.force_wait()
    g = node_a.gusto
    assert g.request_wall_time >= 0.25 and 1e-4 < g.locp_solve_time < 0.5 * g.request_wall_time, (g.locp_solve_time, g.request_wall_time)


def test_scp_controller_wait_false_overlaps_solve_with_simulation_steps():
    """scp(wait=False) (controllers.py:276-292): evaluate() hands the replan to the GPU and returns; the inputs are the
    ones the blocking controller computes."""
    import time
    from sofacontrol_amd.tpwl import controllers as ctl
    from sofacontrol_amd.utils import QuadraticCost
    outs, t_replan_step = {}, {}
    for wait in (True, False):
        w, tp, node = _c2_node(max_iters=5)
        n = 2 * w['r']
        cost = QuadraticCost(Q=tp.H.T @ w['Qz'] @ tp.H + 1e-2 * np.eye(n), R=1e-3 * np.eye(w['m']))
        c = quiet(ctl.scp, tp, cost, w['dt'], N_replan=2, delay=0.0, solver_node=node, wait=wait)
        c.set_sim_timestep(w['dt'])
        x_ref = np.concatenate((w['v_ref'], w['q_ref']))
        rng = np.random.default_rng(1)
        us, ts = [], []
        for k in range(7):
            xf = x_ref + 1e-3 * rng.standard_normal(x_ref.size)
            t_start = time.perf_counter()
            us.append(quiet(c.evaluate, k * w['dt'], None, xf, np.zeros(w['m'])))
            ts.append(time.perf_counter() - t_start)
            if not wait:
                time.sleep(0.05)            # the "simulation step": the replan finishes in the background
        outs[wait], t_replan_step[wait] = np.stack(us), ts[2]        # step 2 requests a replan (N_replan = 2)
    np.testing.assert_allclose(outs[False], outs[True], rtol=0, atol=1e-9 * max(1.0, np.abs(outs[True]).max()))
    assert t_replan_step[False] < 0.5 * t_replan_step[True], t_replan_step
