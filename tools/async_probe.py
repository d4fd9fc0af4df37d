"""Which host-pointer calls block behind an in-flight asynchronous GuSTO request?  (times in ms)"""
import sys, time, io, contextlib
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, workloads as wl
from sofacontrol_amd.scp.standalone import GuSTOSolverNode
from sofacontrol_amd.utils import Polyhedron
w = wl.diamond_c2()
tp, gm = bench.build_model(w)
with contextlib.redirect_stdout(io.StringIO()):
    node = GuSTOSolverNode(gm, w['N'], w['dt'], w['Qz'], w['R'], np.zeros(60), t=w['t'], z=w['z'], U=Polyhedron(w['UA'], w['Ub']),
                           X=Polyhedron(w['XA'], w['Xb']), convg_thresh=1e-3, max_gusto_iters=500)
x0 = node.xopt[2] + 1e-2
xfull = np.concatenate((w['v_ref'], w['q_ref']))
def T(fn):
    t0 = time.perf_counter(); r = fn(); return (time.perf_counter() - t0) * 1e3, r
for what in ('nearest', 'project', 'nothing'):
    t_begin, _ = T(lambda: node.gusto_callback_begin(2 * w['dt'], x0))
    d0 = node.gusto_callback_done()
    if what == 'nearest':
        t_call, _ = T(lambda: tp.calc_nearest_point(x0))
    elif what == 'project':
        t_call, _ = T(lambda: tp.rom.compute_RO_state(xf=xfull))
    else:
        t_call = 0.0
    d1 = node.gusto_callback_done()
    t_end, _ = T(node.gusto_callback_end)
    print('%-8s begin %.2f ms, done-after-begin %s, call %.2f ms, done-after-call %s, end (wait) %.2f ms' % (what, t_begin, d0, t_call, d1, t_end))
