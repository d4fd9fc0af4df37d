import sys, os, io, contextlib
ROOT='/root/repo' if os.path.isdir('/root/repo/tools') else os.getcwd()
sys.path[:0]=[ROOT, os.path.join(ROOT,'soft-robot-control_amd'), os.path.join(ROOT,'tools')]
import torch; torch.cuda.init()
import numpy as np, bench, workloads as wl
from scipy.interpolate import interp1d
from sofacontrol_amd import _lib
from sofacontrol_amd.mor.pod import POD
from sofacontrol_amd.scp.gusto import GuSTO
from sofacontrol_amd.utils import Polyhedron
_lib.set_device(0)
w = wl.diamond_c2(); N, m, r, dt = w['N'], w['m'], w['r'], w['dt']
rom = POD(dict(U=w['U'], q_ref=w['q_ref'], v_ref=w['v_ref']))
tp, gm = bench.build_model(w)
xc, fc = gm.get_characteristic_vals()
X = wl.snapshots(w['q_ref'], 1, seed=2)
x0 = np.concatenate((np.zeros((1, r)), rom.compute_RO_state(qf=X)), axis=1)
u_init = np.zeros((1, N, m)); x_init, _ = tp.rollout(x0, u_init, dt)
zi = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
z = np.stack([zi(dt * np.arange(N + 1))])
g = GuSTO(gm, N, dt, w['Qz'], w['R'], x0, u_init, x_init, z=z, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']) if w.get('XA') is not None else None,
          x_char=xc, f_char=fc, convg_thresh=1e-3, batch=1, max_trace=0, max_gusto_iters=5)
g.solve_batch(x0, u_init, x_init, z=z)
_lib.sync()
