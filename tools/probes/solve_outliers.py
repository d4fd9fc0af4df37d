"""Where do the ~84 ms outliers of consecutive short-horizon solves come from (bench.py: scp_reference_horizons, ms_per_solve_max)?
64 consecutive host-pointer solves at N = 5 after (a) 2 s of host-only work (GPU idle), (b) right after other GPU work, with
SRH_TRACE_SOLVE=5 (the library reports the host-side segments of any call above 5 ms)."""
import os, sys, time
os.environ.setdefault('SRH_TRACE_SOLVE', '5')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
N, dt = 5, 0.05
w = wl.diamond_c2(N=N, dt=dt)
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
g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0[0], u0, x_init[0], z=z[0], U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']), x_char=xc,
          f_char=fc, convg_thresh=1e-3, max_trace=0, max_gusto_iters=5)


def burst(tag, n=64, gap=0.0):
    ts = []
    for i in range(n):
        b = i % reps
        t0 = time.perf_counter()
        g.solve(x0[b], u0, x_init[b], z=z[b])
        ts.append((time.perf_counter() - t0) * 1e3)
        if gap:
            time.sleep(gap)
    s = sorted(ts)
    print('%s: median %.3f ms, max %.3f ms at solve %d, solves above 5 ms: %s' % (tag, s[len(s) // 2], max(ts), int(np.argmax(ts)),
                                                                                  [(i, round(t, 1)) for i, t in enumerate(ts) if t > 5.0]), flush=True)


burst('right after the constructor')
time.sleep(2.0)
burst('after 2 s of idle')
time.sleep(2.0)
burst('after 2 s of idle, 10 ms between solves (a 100 Hz controller)', gap=0.01)
a = torch.randn(4096, 4096, device='cuda')
for _ in range(50):
    a = a @ a * 1e-4
torch.cuda.synchronize()
burst('right after 50 torch matmuls')
burst('again')
