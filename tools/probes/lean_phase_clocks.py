"""Per-phase shader clocks of the lean GuSTO kernel for ONE rollout at the horizons the reference's drivers use and at BASELINE C2
(N = 5 with the X box: examples/diamond/diamond.py:309-316; N = 3 without state rows: examples/hardware/diamond.py:393-399;
N = 50: C2).  Needs a -DSRH_PROFILE build of lean.hip (tools/build_lean.sh -> gpurun_variants/libsofacontrol_hip_prof.so,
selected through SRH_LIB_PATH); the kernel prints its lap counters once per launch (workgroup 0), this script runs each case in a
child process, parses the last launch's lines and writes one JSON file:
    python tools/probes/lean_phase_clocks.py [out.json]
Clocks are s_memtime ticks of the profile build (100 MHz-independent shader clock, ~2.4 GHz under this load); the profile build
is slower than the product build (extra barriers at the gusto-level laps), so the ms figures next to them come from the PRODUCT
library timed in the same run (SRH_LIB_PATH unset in that child)."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = (('closed_loop_N5', 5, 0.05, 1, 5), ('hardware_closed_loop_N3', 3, 0.1, 0, 5), ('c2_N50', 50, 0.05, 1, 5))


def child(N, dt, with_X, cap):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
    import numpy as np
    import torch
    torch.cuda.init()
    import bench
    import workloads as wl
    from scipy.interpolate import interp1d
    from sofacontrol_amd import _lib
    from sofacontrol_amd.mor.pod import POD
    from sofacontrol_amd.scp.gusto import GuSTO
    from sofacontrol_amd.utils import Polyhedron
    w = wl.diamond_c2(N=N, dt=dt, with_X=bool(with_X))
    m, r = w['m'], w['r']
    rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
    tp, gm = bench.build_model(w, 1354)
    xc, fc = gm.get_characteristic_vals()
    reps = 8
    X = wl.snapshots(w['q_ref'], reps, seed=2)
    x0 = np.concatenate((np.zeros((reps, r)), rom.compute_RO_state(qf=X)), axis=1)
    u0 = np.zeros((N, m))
    x_init, _ = tp.rollout(x0, np.zeros((reps, N, m)), dt)
    zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
    z = np.stack([zi(b * 10.0 / reps + dt * np.arange(N + 1)) for b in range(reps)])
    g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[0], u0, x_init[0], z=z[0], U=Polyhedron(w['UA'], w['Ub']),
              X=Polyhedron(w['XA'], w['Xb']) if with_X else None, x_char=xc, f_char=fc, convg_thresh=1e-3, max_trace=0, max_gusto_iters=cap)
    _lib.sync()
    print('=== timed solves', flush=True)
    ts, its = [], []
    for b in range(reps):
        t0 = time.perf_counter()
        g.solve(x0[b], u0, x_init[b], z=z[b])
        ts.append(time.perf_counter() - t0)
        its.append(int(g.iters[0]))
    _lib.sync()
    print('=== result', json.dumps({'kernel': g.kernel_info['kernel'], 'scp_iterations': its, 'ms_per_solve': [t * 1e3 for t in ts],
                                    'ms_per_scp_iteration_median': sorted(t / max(1, i) for t, i in zip(ts, its))[reps // 2] * 1e3}), flush=True)


LINE = re.compile(r'^lean (gusto clocks|qp laps|newton laps|step laps|qp tail|split)')


def parse(text):
    """the LAST launch's lines (the last solve of the child: rollout 7)"""
    blocks, cur = [], None
    for ln in text.splitlines():
        if ln.startswith('lean gusto clocks'):
            cur = [ln]
            blocks.append(cur)
        elif cur is not None and LINE.match(ln):
            cur.append(ln)
    if not blocks:
        return None
    out = {}
    for ln in blocks[-1]:
        head, _, rest = ln.partition(':')
        if ln.startswith('lean gusto clocks'):
            out['scp_iterations'] = int(re.search(r'\((\d+) iterations\)', ln).group(1))
        if ln.startswith('lean split'):
            rest = ln.split('):', 1)[1]
        key = LINE.match(ln).group(1).replace(' ', '_')
        out[key] = {k.strip(): int(v) for k, v in re.findall(r'([A-Za-z_+()0-9 \-]+?) (\d+)(?= |$|;)', rest)}
    return out


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'lean_phase_clocks.json')
    prof_lib = os.environ.get('SRH_PROF_LIB', os.path.join(ROOT, 'gpurun_variants', 'libsofacontrol_hip_prof.so'))
    res = {'what': __doc__.split('\n')[0], 'profile_library': os.path.basename(prof_lib), 'cases': {}}
    for key, N, dt, with_X, cap in CASES:
        entry = {'N': N, 'dt': dt, 'X_rows': 4 if with_X else 0, 'max_gusto_iters': cap}
        for flavour, env in (('product', {k: v for k, v in os.environ.items() if k != 'SRH_LIB_PATH'}), ('profile', dict(os.environ, SRH_LIB_PATH=prof_lib))):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(N), str(dt), str(with_X), str(cap)], env=env,
                               capture_output=True, text=True, timeout=600)
            txt = p.stdout
            m = re.search(r'^=== result (.*)$', txt, re.M)
            entry[flavour] = json.loads(m.group(1)) if m else {'error': (p.stderr or txt)[-400:]}
            if flavour == 'profile':
                entry['clocks_last_solve'] = parse(txt.split('=== timed solves')[-1])
                with open(out_path.replace('.json', '_%s.log' % key), 'w') as f:
                    f.write(txt[-20000:])
        res['cases'][key] = entry
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
