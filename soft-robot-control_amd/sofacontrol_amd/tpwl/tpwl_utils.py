"""TPWL model assembly: full-order linearisation points -> reduced piecewise-affine model, on the device.

Protocol of sofacontrol/tpwl/tpwl_utils.py: `Target` / `DynamicsTarget` containers and
`TPWLSnapshotData` with `save_snapshot(point, prev_point) -> bool`, `add_point(point)`, `simulation_end(filename)`,
the data dictionary `dict` (keys q, v, u, K, D, M, S, H, b, f, q+, v+, A_c, B_c, d_c, A_d, B_d, d_d, rom_info,
info, dt -- the on-disk format TPWLATV loads) and the acceptance tests `evaluate_point_dist/_dynamics`.  Behaviour
pinned by the golden g12 (recorded from the imported reference).

Every full-order quantity of a point goes through the POD handle: states by `compute_RO_state`, the n_f x n_f
matrices K, D, M, S by `compute_RO_matrix` (ONE HBM pass over each 191 MB matrix at the Diamond size), H / b / f by
the left projection.  What is left for the host is r x r algebra (`extract_AB`, `extract_AB_d`), once per point.
"""
import numpy as np

from .. import utils as scutils


class Target:
    """What a controller should reach: times `t`, outputs `z`, inputs `u`, reduced states `x`, output selector `Hf`."""

    def __init__(self):
        self.t = self.u = self.z = self.x = self.Hf = None

    def load_target_file(self, file):
        stored = scutils.load_data(file)
        for key in ('t', 'u', 'z', 'Hf'):
            setattr(self, key, stored.get(key))


class DynamicsTarget(Target):
    """An operating point (A, B, x, u) for the LQR controllers."""

    def __init__(self):
        super().__init__()
        self.A = self.B = None


# how each field of a full-order point is reduced: (dictionary key, attribute of the point, reducer name)
_REDUCTIONS = (
    ('q', 'q', 'pos'), ('v', 'v', 'vel'), ('u', 'u', 'keep'),
    ('K', 'K', 'both'), ('D', 'D', 'both'), ('M', 'M', 'both'),
    ('b', 'b', 'left'), ('f', 'f', 'left'), ('H', 'H', 'left'), ('S', 'S', 'both'),
    ('q+', 'q_next', 'pos'), ('v+', 'v_next', 'vel'),
)
_MODEL_KEYS = ('A_c', 'B_c', 'd_c', 'A_d', 'B_d', 'd_d', 'z', 'z_est')


class TPWLSnapshotData(scutils.SnapshotData):
    def __init__(self, rom, config, info=None, Hf=None):
        super().__init__(save_dynamics=True)
        self.rom, self.config, self.Hf = rom, config, Hf
        self.dict.update({k: [] for k in _MODEL_KEYS})
        self.dict['rom_info'] = rom.get_info()
        self.info = {} if info is None else info
        self.saved_tpwl_steps = []
        self.save_step = 0
        if config.eval_type == 'dynamics':          # acceptance by one-step prediction error of the model so far
            self.sim_sys_class, self.sim_sys_params = config.sim_sys, config.constants_sim
        self._reduce = {
            'pos': lambda a: rom.compute_RO_state(qf=a), 'vel': lambda a: rom.compute_RO_state(vf=a),
            'both': rom.compute_RO_matrix, 'left': lambda a: rom.compute_RO_matrix(a, left=True), 'keep': lambda a: a}

    # ---- collection protocol
    def save_snapshot(self, point, prev_point):
        return False if prev_point is None else self.evaluate_point(point, prev_point)

    def add_point(self, point):
        store = self.dict
        if store['dt'] == -1:
            store['dt'] = point.dt
        self.saved_tpwl_steps.append(point.t)
        print('Time: {}, Number of points saved: {}'.format(point.t, len(self.saved_tpwl_steps)))
        # the two-sided reductions of a point (K, D, M, S: tpwl_utils.py:96-103 of the reference, one after the other) in ONE call when
        # the ROM offers it: four n_f x n_f matrices share a launch pair (POD.compute_RO_matrices); same values as one by one
        both = [(key, attr) for key, attr, how in _REDUCTIONS if how == 'both']
        batched = {}
        if hasattr(self.rom, 'compute_RO_matrices') and len(both) > 1:
            batched = dict(zip((k for k, _ in both), self.rom.compute_RO_matrices([getattr(point, a) for _, a in both])))
        for key, attr, how in _REDUCTIONS:
            store[key].append(batched[key] if key in batched else self._reduce[how](getattr(point, attr)))
        if self.config.save_continuous_TPWL:
            self.add_continuous_TPWL()
        if self.config.save_discrete_TPWL:
            self.add_discrete_TPWL()
        if self.config.eval_type == 'dynamics':
            self.sim_sys = self.sim_sys_class(data=store, params=self.sim_sys_params)

    def simulation_end(self, filename):
        cfg = self.config
        print('Computed TPWL, resulting in %d linearization points' % len(self.saved_tpwl_steps))
        self.info.update(state_dim=str(self.rom.rom_dim), nbr_lin=str(len(self.saved_tpwl_steps)),
                         saved_step_nbrs=self.saved_tpwl_steps, tpwl_method=cfg.eval_type, tpwl_parameters=vars(cfg),
                         tpwl_type=cfg.TPWL_type, discr_type=cfg.discr_type)
        if cfg.eval_type == 'dynamics':
            del self.info['tpwl_parameters']['sim_sys']         # a class object: not part of the stored parameters
        self.dict['info'] = self.info
        print('Saving TPWL data to {}...'.format(filename))
        scutils.dict_lists_to_array(self.dict)
        scutils.save_data(filename, self.dict)
        print('Done.')

    # ---- does a candidate point add information?
    def evaluate_point(self, point, prev_point):
        if not self.dict['q']:
            return True                                          # the first point always enters the model
        if self.config.eval_type == 'distance':
            return self.evaluate_point_dist(point)
        if self.config.eval_type == 'dynamics':
            return self.evaluate_point_dynamics(point, prev_point)
        return None

    def _beyond_threshold(self, q_part, v_part):
        """Separate criteria: either part alone may trigger; otherwise their sum is compared."""
        thr = self.config.TPWL_threshold
        if self.config.TPWL_separate_calculation:
            return bool(q_part >= thr or v_part >= thr)
        return bool(q_part + v_part >= thr)

    def evaluate_point_dist(self, point):
        """Far enough (weighted reduced distance) from every stored point?"""
        w = self.config.TPWL_weighting_factors
        dq = w['q'] * np.linalg.norm(self._reduce['pos'](point.q) - np.asarray(self.dict['q']), axis=1)
        dv = w['v'] * np.linalg.norm(self._reduce['vel'](point.v) - np.asarray(self.dict['v']), axis=1)
        if self.config.TPWL_separate_calculation:
            return self._beyond_threshold(np.min(dq), np.min(dv))
        return self._beyond_threshold(np.min(dq + dv), 0.0)

    def evaluate_point_dynamics(self, point, prev_point):
        """Does the model collected so far mispredict the step prev_point -> point?  (Unactuated steps never add.)"""
        if not np.any(prev_point.u):
            return False
        cfg, rom = self.config, self.rom
        x_now = scutils.qv2x(point.q, point.v)
        x_before = scutils.qv2x(prev_point.q, prev_point.v)
        xr_before = rom.compute_RO_state(xf=x_before)
        xr_pred = self.sim_sys.update_state(xr_before, prev_point.u, prev_point.dt)
        if self.Hf is not None and cfg.output_based:
            z_pred = self.Hf @ rom.compute_FO_state(x=xr_pred)
            z_true = self.Hf @ x_now
            self.dict['z_est'].append(z_pred)
            self.dict['z'].append(z_true)
            return bool(np.linalg.norm(z_pred - z_true) >= cfg.TPWL_threshold)
        if cfg.fom_based:          # compare increments in the full-order space
            true_step, pred_step = x_now - x_before, rom.compute_FO_state(x=xr_pred) - x_before
        else:                      # ... or in the reduced space
            true_step, pred_step = rom.compute_RO_state(xf=x_now) - xr_before, xr_pred - xr_before
        (dq, dv), (dq_p, dv_p) = scutils.x2qv(true_step), scutils.x2qv(pred_step)
        w = cfg.TPWL_weighting_factors
        return self._beyond_threshold(w['q'] * np.linalg.norm(dq_p - dq), w['v'] * np.linalg.norm(dv_p - dv))

    # ---- reduced model of the newest point
    def _last(self, *keys):
        return [self.dict[k][-1] for k in keys]

    def add_continuous_TPWL(self):
        """x' = A_c x + B_c u + d_c with x = [v; q]: second-order form M v' = -D v - K (q - q_i) + f + H u."""
        K, D, M, H, f, q = self._last('K', 'D', 'M', 'H', 'f', 'q')
        A, B = scutils.extract_AB(K, D, M, H)
        accel0 = np.linalg.solve(M, f + K @ q)
        self.dict['A_c'].append(A)
        self.dict['B_c'].append(B)
        self.dict['d_c'].append(np.concatenate((accel0, np.zeros_like(accel0))))

    def add_discrete_TPWL(self):
        """x+ = A_d x + B_d u + d_d from the simulator's own step matrix S; d_d closes the recorded step exactly."""
        S, K, H, q, v, qn, vn, u = self._last('S', 'K', 'H', 'q', 'v', 'q+', 'v+', 'u')
        A_d, B_d = scutils.extract_AB_d(S, K, H, self.dict['dt'])
        x, x_next = scutils.qv2x(q, v), scutils.qv2x(qn, vn)
        self.dict['A_d'].append(A_d)
        self.dict['B_d'].append(B_d)
        self.dict['d_d'].append(x_next - A_d @ x - B_d @ np.atleast_1d(u))
