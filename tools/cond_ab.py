"""A/B of the condensed QP path inside the fused GuSTO kernel: SRH_QP_NO_COND=1 python tools/cond_ab.py  vs  python tools/cond_ab.py"""
import sys, time, io, contextlib, os
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import workloads as wl
from test_gusto_bench_shapes_gpu import problem
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron
for name, w, tip in (('C2', wl.diamond_c2(), 1354), ('C5', wl.trunk_c5(), 51)):
    B = int(os.environ.get('B', '1'))
    gm, xc, fc, x0, u_init, x_init, z = problem(w, B, 2, tip)
    X = Polyhedron(w['XA'], w['Xb']) if w['XA'] is not None else None
    g = GuSTO(gm, w['N'], w['dt'], w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=X, x_char=xc, f_char=fc,
              convg_thresh=1e-3, batch=B, max_trace=16, max_gusto_iters=5)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); g.solve_batch(x0, u_init, x_init, z=z); ts.append(time.perf_counter() - t0)
    print('%s NO_COND=%s batch %d: %.2f ms per solve, %d SCP iterations -> %.2f ms per SCP iteration; J trace %s' % (
        name, os.environ.get('SRH_QP_NO_COND'), B, min(ts) * 1e3, g.iters.sum(), min(ts) * 1e3 / g.iters.max(), g.trace[0, :int(g.iters[0]), 0]))
