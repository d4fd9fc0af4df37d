"""Target containers (sofacontrol/tpwl/tpwl_utils.py:5-38)."""
import numpy as np

from .. import utils as scutils


class Target:
    def __init__(self):
        self.t = None
        self.u = None
        self.z = None
        self.x = None
        self.Hf = None

    def load_target_file(self, file):
        data = scutils.load_data(file)
        self.t = data.get('t')
        self.u = data.get('u')
        self.z = data.get('z')
        self.Hf = data.get('Hf')


class DynamicsTarget(Target):
    def __init__(self):
        super().__init__()
        self.A = None
        self.B = None
        self.x = None


class TPWLSnapshotData(scutils.SnapshotData):
    """sofacontrol/tpwl/tpwl_utils.py:40-290: collects the points of a TPWL model.  Every full-order quantity
    of a point (q, v, K, D, M, S, H, b, f, q+, v+) is reduced on the device (POD.compute_RO_state /
    compute_RO_matrix: one HBM pass over each n_f x n_f matrix); the r x r assembly of (A_c, B_c, d_c) and
    (A_d, B_d, d_d) that follows is one-off host algebra like in the reference."""

    def __init__(self, rom, config, info=None, Hf=None):
        super().__init__(save_dynamics=True)
        for k in ('A_c', 'B_c', 'd_c', 'A_d', 'B_d', 'd_d', 'z', 'z_est'):
            self.dict[k] = []
        self.rom = rom
        self.dict['rom_info'] = self.rom.get_info()
        self.config = config
        if self.config.eval_type == 'dynamics':
            self.sim_sys_class = self.config.sim_sys
            self.sim_sys_params = self.config.constants_sim
        self.info = dict() if info is None else info
        self.save_step = 0
        self.saved_tpwl_steps = []
        self.Hf = Hf

    def add_point(self, point):
        """tpwl_utils.py:84-117."""
        if self.dict['dt'] == -1:
            self.dict['dt'] = point.dt
        self.saved_tpwl_steps.append(point.t)
        print('Time: {}, Number of points saved: {}'.format(point.t, len(self.saved_tpwl_steps)))
        rom = self.rom
        self.dict['q'].append(rom.compute_RO_state(qf=point.q))
        self.dict['v'].append(rom.compute_RO_state(vf=point.v))
        self.dict['u'].append(point.u)
        self.dict['K'].append(rom.compute_RO_matrix(point.K))
        self.dict['D'].append(rom.compute_RO_matrix(point.D))
        self.dict['M'].append(rom.compute_RO_matrix(point.M))
        self.dict['b'].append(rom.compute_RO_matrix(point.b, left=True))
        self.dict['f'].append(rom.compute_RO_matrix(point.f, left=True))
        self.dict['H'].append(rom.compute_RO_matrix(point.H, left=True))
        self.dict['S'].append(rom.compute_RO_matrix(point.S))
        self.dict['q+'].append(rom.compute_RO_state(qf=point.q_next))
        self.dict['v+'].append(rom.compute_RO_state(vf=point.v_next))
        if self.config.save_continuous_TPWL:
            self.add_continuous_TPWL()
        if self.config.save_discrete_TPWL:
            self.add_discrete_TPWL()
        if self.config.eval_type == 'dynamics':
            self.sim_sys = self.sim_sys_class(data=self.dict, params=self.sim_sys_params)

    def save_snapshot(self, point, prev_point):
        return self.evaluate_point(point, prev_point) if prev_point is not None else False

    def simulation_end(self, filename):
        """tpwl_utils.py:130-153."""
        print('Computed TPWL, resulting in %d linearization points' % len(self.saved_tpwl_steps))
        self.info['state_dim'] = str(self.rom.rom_dim)
        self.info['nbr_lin'] = str(len(self.saved_tpwl_steps))
        self.info['saved_step_nbrs'] = self.saved_tpwl_steps
        self.info['tpwl_method'] = self.config.eval_type
        self.info['tpwl_parameters'] = vars(self.config)
        self.info['tpwl_type'] = self.config.TPWL_type
        self.info['discr_type'] = self.config.discr_type
        if self.config.eval_type == 'dynamics':
            del self.info['tpwl_parameters']['sim_sys']
        self.dict['info'] = self.info
        print('Saving TPWL data to {}...'.format(filename))
        scutils.dict_lists_to_array(self.dict)
        scutils.save_data(filename, self.dict)
        print('Done.')

    def evaluate_point(self, point, prev_point):
        if not self.dict['q']:
            return True
        if self.config.eval_type == 'distance':
            return self.evaluate_point_dist(point)
        elif self.config.eval_type == 'dynamics':
            return self.evaluate_point_dynamics(point, prev_point)

    def evaluate_point_dist(self, point):
        """tpwl_utils.py:170-196."""
        q_d = np.asarray(self.rom.compute_RO_state(qf=point.q) - np.asarray(self.dict['q']))
        v_d = np.asarray(self.rom.compute_RO_state(vf=point.v) - np.asarray(self.dict['v']))
        q_d = self.config.TPWL_weighting_factors['q'] * np.linalg.norm(q_d, axis=1)
        v_d = self.config.TPWL_weighting_factors['v'] * np.linalg.norm(v_d, axis=1)
        if self.config.TPWL_separate_calculation:
            return bool(np.min(q_d) >= self.config.TPWL_threshold or np.min(v_d) >= self.config.TPWL_threshold)
        return bool(np.min(q_d + v_d) >= self.config.TPWL_threshold)

    def evaluate_point_dynamics(self, point, prev_point):
        """tpwl_utils.py:199-261."""
        add_point = False
        if not (prev_point.u == np.zeros_like(prev_point.u)).all():
            x = scutils.qv2x(point.q, point.v)
            x_prev = scutils.qv2x(prev_point.q, prev_point.v)
            x_prev_r = self.rom.compute_RO_state(xf=x_prev)
            x_r_tpwl = self.sim_sys.update_state(x_prev_r, prev_point.u, prev_point.dt)
            w = self.config.TPWL_weighting_factors
            if self.Hf is not None and self.config.output_based:
                zf_est = self.Hf @ self.rom.compute_FO_state(x=x_r_tpwl)
                zf = self.Hf @ x
                if np.linalg.norm(zf_est - zf) >= self.config.TPWL_threshold:
                    add_point = True
                self.dict['z_est'].append(zf_est)
                self.dict['z'].append(zf)
            else:
                if not self.config.fom_based:
                    x_r = self.rom.compute_RO_state(xf=x)
                    dq, dv = scutils.x2qv(x_r - x_prev_r)
                    dq_e, dv_e = scutils.x2qv(x_r_tpwl - x_prev_r)
                else:
                    x_tpwl = self.rom.compute_FO_state(x=x_r_tpwl)
                    dq, dv = scutils.x2qv(x - x_prev)
                    dq_e, dv_e = scutils.x2qv(x_tpwl - x_prev)
                q_err = w['q'] * np.linalg.norm(dq_e - dq)
                v_err = w['v'] * np.linalg.norm(dv_e - dv)
                if self.config.TPWL_separate_calculation:
                    add_point = bool(q_err >= self.config.TPWL_threshold or v_err >= self.config.TPWL_threshold)
                else:
                    add_point = bool(q_err + v_err >= self.config.TPWL_threshold)
        return add_point

    def add_continuous_TPWL(self):
        """tpwl_utils.py:263-276."""
        d = self.dict
        A, B = scutils.extract_AB(d['K'][-1], d['D'][-1], d['M'][-1], d['H'][-1])
        b_n = np.linalg.solve(d['M'][-1], d['f'][-1] + d['K'][-1] @ d['q'][-1])
        d['A_c'].append(A)
        d['B_c'].append(B)
        d['d_c'].append(np.hstack((b_n, np.zeros(np.shape(b_n)))))

    def add_discrete_TPWL(self):
        """tpwl_utils.py:279-290."""
        d = self.dict
        A_d, B_d = scutils.extract_AB_d(d['S'][-1], d['K'][-1], d['H'][-1], d['dt'])
        x = scutils.qv2x(d['q'][-1], d['v'][-1])
        x_next = scutils.qv2x(d['q+'][-1], d['v+'][-1])
        d['A_d'].append(A_d)
        d['B_d'].append(B_d)
        d['d_d'].append(x_next - A_d @ x - B_d @ np.atleast_1d(d['u'][-1]))
