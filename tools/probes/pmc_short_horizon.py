"""Workload for `rocprofv3 --pmc ...` passes over the short-horizon lean kernels: 32 consecutive N = 5 (X box) and N = 3 solves, one rollout at
a time (the series of bench.py: scp_reference_horizons), nothing else on the GPU.  tools/probes/pmc_short_horizon.sh runs the passes and
summarises instructions / wave cycles per SCP iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
torch.cuda.init()
import bench
import workloads as wl
from scipy.interpolate import interp1d
from sofacontrol_amd.mor.pod import POD
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron
tot = {}
for N, dt, with_X in ((5, 0.05, True), (3, 0.1, False)):
    w = wl.diamond_c2(N=N, dt=dt, with_X=with_X)
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
    g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[0], u0, x_init[0], z=z[0], U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']) if with_X else None,
              x_char=xc, f_char=fc, convg_thresh=1e-3, max_trace=0, max_gusto_iters=5, first_solve_cap=5)
    its = int(g.iters[0])
    for i in range(32):
        b = i % reps
        g.solve(x0[b], u0, x_init[b], z=z[b])
        its += int(g.iters[0])
    tot['N%d' % N] = {'solves': 33, 'scp_iterations': its, 'kernel': g.kernel_info['kernel']}
import json
print('PMC_WORKLOAD ' + json.dumps(tot))
