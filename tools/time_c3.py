"""BASELINE C3 (SSM n_x = 10, n_u = 8, N = 100, dt = 0.05, backward Euler; workloads.ssm_c3): the iLQR kernel with one wave per
problem (default for small models) next to the 512-thread form (SRH_ILQR_THREADS=512), 256 problems and one problem; the results
of the two forms compared bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch; torch.cuda.init()
import workloads as wl
from sofacontrol_amd.SSM.ssm import SSMDynamics
from sofacontrol_amd.lqr.ilqr import iLQR
from sofacontrol_amd.utils import QuadraticCost
c3 = wl.ssm_c3(256)
n, m, N, dt = c3['n'], c3['m'], c3['N'], c3['dt']
model = c3['model']
def mat(v):
    a = np.empty((1, 1), dtype=object); a[0, 0] = np.asarray(v); return a
sc = lambda v: mat(np.array([[v]]))
s = SSMDynamics(model['z_ref'].copy(), discrete=False, discr_method=c3['discr'],
                model=dict(Ts=sc(dt), w_coeff=mat(model['W']), v_coeff=mat(model['V']), r_coeff=mat(model['R']), B=mat(model['B']), rd_coeff=mat(model['Rd']), Bd=mat(model['Bd'])),
                params=dict(state_dim=sc(n), input_dim=sc(m), output_dim=sc(n), SSM_order=sc(2), ROM_order=sc(3)))
s.H = model['W'][:, :n].copy()
res = {}
for threads in ('512', '64'):
    os.environ['SRH_ILQR_THREADS'] = threads
    for Bn in (256, 1):
        il = iLQR(dt, s, QuadraticCost(Q=c3['Qz'], R=c3['R'], Qf=c3['Qf']), N)
        il.set_target(c3['zt'] if Bn > 1 else c3['zt'][0])
        x0 = c3['x0'] if Bn > 1 else c3['x0'][0]
        x, u, K = il.ilqr_computation(x0)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); x, u, K = il.ilqr_computation(x0); ts.append(time.perf_counter() - t0)
        its = np.atleast_1d(il.iters)
        res[(threads, Bn)] = (x.copy(), u.copy(), K.copy(), its.copy())
        print('threads %s, %d problems: %.2f ms (min of 5, host-pointer API), iterations max %d sum %d -> %.1f k iterations/s' %
              (threads, Bn, min(ts) * 1e3, its.max(), its.sum(), its.sum() / min(ts) / 1e3))
for Bn in (256, 1):
    a, b = res[('512', Bn)], res[('64', Bn)]
    print('%d problems, 64 vs 512 threads: max |dx| %.3e |du| %.3e |dK| %.3e, iteration counts differ in %d' %
          (Bn, np.abs(a[0] - b[0]).max(), np.abs(a[1] - b[1]).max(), np.abs(a[2] - b[2]).max(), int((a[3] != b[3]).sum())))
