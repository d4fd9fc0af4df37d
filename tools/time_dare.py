"""Start-up cost of the scp controller: DARE gains for every TPWL point (tpwl/controllers.py:238-246), Diamond shape."""
import io, contextlib, sys, time
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.')
import workloads as wl
from sofacontrol_amd.lqr.lqr import dare_batch
w = wl.diamond_c2()
r, m = w['r'], w['m']
Ad, Bd, dd = wl.zoh_tables(w['tab'], 0.01)
H = w['H']
Q = H.T @ w['Qz'] @ H + 1e-3 * np.eye(2 * r); R = 1e-4 * np.eye(m)
dare_batch(Ad[:2], Bd[:2], Q, R)
t0 = time.perf_counter(); L, P = dare_batch(Ad, Bd, Q, R); t = time.perf_counter() - t0
print('DARE gains for %d points (n_x = %d): %.1f ms' % (Ad.shape[0], 2 * r, t * 1e3))
from scipy.linalg import solve_discrete_are
t0 = time.perf_counter()
for i in range(4):
    Ps = solve_discrete_are(Ad[i], Bd[i], Q, R)
print('scipy solve_discrete_are: %.1f ms per point; max |P - P_scipy| / |P| = %.1e' % ((time.perf_counter() - t0) / 4 * 1e3, np.abs(P[3] - Ps).max() / np.abs(Ps).max()))
