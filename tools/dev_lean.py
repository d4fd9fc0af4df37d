"""Development helper: lean vs fused kernels on one C2 / C5 QP and a few rollouts, with timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import workloads as wl
import test_lean_gpu as T

which = sys.argv[1] if len(sys.argv) > 1 else 'c2'
w = wl.diamond_c2() if which == 'c2' else wl.trunk_c5()
qp = T.first_qp(w, b=0)
for lean in (True, False):
    r = T.locp_solve(w, qp, 1e4, lean)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = T.locp_solve(w, qp, 1e4, lean); ts.append(time.perf_counter() - t0)
    print('lean' if lean else 'fused', 'J %.9e ok %s iters %d  %.2f ms' % (r[0], r[1], r[2], min(ts) * 1e3))
    if lean: rl = r
print('rel x %.2e u %.2e' % (T.rel(rl[3], r[3]), T.rel(rl[4], r[4])))
tip = 1354 if which == 'c2' else w['tip_node']
for lean in (True, False):
    t0 = time.perf_counter()
    first, second = T.gusto_case(w, tip, 8, 2 if which == 'c2' else 9, lean, 5)
    print('gusto', 'lean' if lean else 'fused', 'iters', first[0], second[0], 'status', first[1], second[1], '%.1f ms total' % ((time.perf_counter() - t0) * 1e3))
    if lean: gl = (first, second)
for a, b in zip(gl, (first, second)):
    print('rel x %.2e u %.2e' % (T.rel(a[3], b[3]), T.rel(a[4], b[4])))
