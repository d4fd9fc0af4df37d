"""Repeated solves of the C3 SSM iLQR batch (in-kernel wave-level Gauss-Jordan, two inverses side by side) must be bit-identical."""
import os, sys
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
for discr in ('be', 'bil'):
    s = SSMDynamics(model['z_ref'].copy(), discrete=False, discr_method=discr,
                    model=dict(Ts=sc(dt), w_coeff=mat(model['W']), v_coeff=mat(model['V']), r_coeff=mat(model['R']), B=mat(model['B']), rd_coeff=mat(model['Rd']), Bd=mat(model['Bd'])),
                    params=dict(state_dim=sc(n), input_dim=sc(m), output_dim=sc(n), SSM_order=sc(2), ROM_order=sc(3)))
    s.H = model['W'][:, :n].copy()
    il = iLQR(dt, s, QuadraticCost(Q=c3['Qz'], R=c3['R'], Qf=c3['Qf']), N)
    il.set_target(c3['zt'])
    x, u, K = il.ilqr_computation(c3['x0'])
    first = (x.copy(), u.copy(), K.copy(), il.iters.copy())
    bad = []
    for rep in range(10):
        x, u, K = il.ilqr_computation(c3['x0'])
        bad.append((float(np.abs(x - first[0]).max()), float(np.abs(u - first[1]).max()), float(np.abs(K - first[2]).max()), int((il.iters != first[3]).sum())))
    print(discr, 'max differences over 10 repeated solves of 256 problems (x, u, K, iteration counts):', max(b[0] for b in bad), max(b[1] for b in bad), max(b[2] for b in bad), max(b[3] for b in bad))
    # --dump file.npz / --compare file.npz: the results of another build of the library, bit for bit
    for flag in ('--dump', '--compare'):
        if flag in sys.argv:
            f = sys.argv[sys.argv.index(flag) + 1].replace('.npz', '_%s.npz' % discr)
            if flag == '--dump':
                np.savez(f, x=first[0], u=first[1], K=first[2], iters=first[3])
            else:
                g = np.load(f)
                print(discr, 'against', f, ': max |dx| %.3e |du| %.3e |dK| %.3e, iteration counts differ in %d problems' %
                      (np.abs(g['x'] - first[0]).max(), np.abs(g['u'] - first[1]).max(), np.abs(g['K'] - first[2]).max(), int((g['iters'] != first[3]).sum())))
