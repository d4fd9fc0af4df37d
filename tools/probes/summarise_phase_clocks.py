"""profiles/r05_lean_phase_clocks.json from the raw outputs of tools/probes/lean_phase_clocks.py: shader clocks of the lean GuSTO kernel per
phase, per SCP iteration and per interior-point iteration, for one rollout at N = 5 (X box), N = 3 (no state rows) and N = 50 (C2) --
`before`: the eight-wave interior point (ql::ipm_box) at every horizon, the state at the start of round 5; `after`: the final tree (short
horizons: ql::ipm_wave on one wave, whose laps are not split: everything between the condensation and the final rollout is `interior point`).
Usage: python tools/probes/summarise_phase_clocks.py before_raw.json after_raw.json out.json"""
import json, sys


def table(case):
    cl = case.get('clocks_last_solve')
    if not cl:
        return None
    its = cl['scp_iterations']
    g, q, nw, st, tl, sp = cl['gusto_clocks'], cl['qp_laps'], cl['newton_laps'], cl['step_laps'], cl['qp_tail'], cl['split']
    ipm_it, qps = tl['ipm-iterations'], tl['qps']
    # phases of the interior point (zero for the one-wave form, which laps the whole loop as grad+newton)
    ipm = {'rows -> weights, gradients, per-stage sums (incl. its reduction)': q['rows'],
           'Gram fill K = I + Ls^T G D^-1 G^T Ls + scaling': q['gram'],
           'factorisation of K beside the front of the Newton solve (the longer of the two) + unit tiles': q['cholesky'],
           'Newton solves: predictor back half + corrector (gradients, G / G^T products, K solves)': q['grad+newton'],
           'step rows (dl, dt, step length candidates)': st['step rows (both modes)'], 'reduction: step length': st['reduce(amax)'],
           'affine complementarity rows': st['affine mu rows'], 'reduction: affine mu': st['reduce(mu_aff)'],
           'accept step, loop bookkeeping': q['steps'], 'stage factors (general rows only)': q['stage-factors']}
    qp_other = {'set-up + zero-input rollout (free response)': q['setup+rollout'], 'condensation (G by the adjoint recursion)': q['condense'],
                'rollout of the minimiser': tl['rollout-of-minimiser'], 'objective + trust-region test': tl['objective+tr-test']}
    scp = {'initial copy + nearest points (once per solve)': g['init+nearest'], 'loop top': g['loop-top'], 'trust-region test': g['tr-test'],
           'nearest points of the new trajectory': g['nearest(new)'], 'model accuracy': g['accuracy'],
           'state-constraint violation + convergence tests': g['tests'], 'accept + copy': g['accept+nearest']}
    ipm_total = sum(ipm.values())
    total = g['qp'] + sum(scp.values())
    out = {'scp_iterations': its, 'qps': qps, 'interior_point_iterations': ipm_it, 'warm_started_qps': tl['warm-qps'],
           'kernel_total_clocks': total, 'clocks_per_scp_iteration': total / its,
           'interior_point': {'clocks': ipm_total, 'per_interior_point_iteration': ipm_total / max(1, ipm_it), 'share_of_kernel': ipm_total / total,
                              'phases_per_interior_point_iteration': {k: v / max(1, ipm_it) for k, v in ipm.items() if v},
                              'inside the Newton laps (per interior-point iteration)': {k: v / max(1, ipm_it) for k, v in nw.items() if v},
                              'split stretch (per factorisation)': {k: v / max(1, ipm_it + qps) for k, v in sp.items() if v}},
           'qp_outside_the_interior_point': {'clocks': sum(qp_other.values()), 'per_scp_iteration': {k: v / its for k, v in qp_other.items()},
                                             'share_of_kernel': sum(qp_other.values()) / total},
           'scp_loop_outside_the_qp': {'clocks': sum(scp.values()), 'per_scp_iteration': {k: v / its for k, v in scp.items()},
                                       'share_of_kernel': sum(scp.values()) / total},
           'unaccounted_inside_qp (lap granularity)': g['qp'] - ipm_total - sum(qp_other.values())}
    return out


if __name__ == '__main__':
    before, after, dst = (json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3])
    res = {'what': __doc__.split('Usage')[0].strip(), 'clock': 'shader clock of the -DSRH_PROFILE build (clock64); the ms figures are the PRODUCT build timed in the same run '
                                                              '(host wall time per SCP iteration, one rollout at a time, median of 8)', 'cases': {}}
    for key in after['cases']:
        e = {'N': after['cases'][key]['N'], 'X_rows': after['cases'][key]['X_rows']}
        for tag, src in (('before', before), ('after', after)):
            c = src['cases'].get(key)
            if c is None:
                continue
            e[tag] = {'kernel': c.get('product', {}).get('kernel'), 'product_ms_per_scp_iteration': c.get('product', {}).get('ms_per_scp_iteration_median'),
                      'clocks': table(c)}
        res['cases'][key] = e
    json.dump(res, open(dst, 'w'), indent=1)
    for key, e in res['cases'].items():
        for tag in ('before', 'after'):
            if tag in e and e[tag]['clocks']:
                t = e[tag]['clocks']
                print('%-26s %-6s %-26s %.3f ms/it  %8.0f clocks/SCP it  IPM %6.0f clocks/it (%.0f %%)  QP rest %.0f %%  SCP rest %.0f %%' %
                      (key, tag, e[tag]['kernel'], e[tag]['product_ms_per_scp_iteration'], t['clocks_per_scp_iteration'],
                       t['interior_point']['per_interior_point_iteration'], 100 * t['interior_point']['share_of_kernel'],
                       100 * t['qp_outside_the_interior_point']['share_of_kernel'], 100 * t['scp_loop_outside_the_qp']['share_of_kernel']))
