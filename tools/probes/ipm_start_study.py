"""Starting points of the interior point compared on the first QPs of C2 / C5 with the numpy statement (oracle/condensed_ipm.py): the
shipped one (unit-weight Newton step, uniform shifts 1 + max / 1 - min), Mehrotra's full heuristic (1.5 x shifts + the two balancing
terms) and a central-path start (slacks from the rows with a floor, multipliers mu0 / t).  Prints iteration counts and the distance of the
minimisers.  CPU only:  python tools/probes/ipm_start_study.py"""
import os, sys, re, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'soft-robot-control_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import workloads as wl
import test_lean_gpu as T
from oracle import condensed_ipm as cipm, riccati_ipm as ripm
src = open(os.path.join(ROOT, 'oracle', 'condensed_ipm.py')).read()
old = '''        sh_t = (1.0 + allg.max()) if allg.max() >= 0 else 0.0
        sh_l = (1.0 - allg.min()) if allg.min() <= 0 else 0.0
        tx = [None] + [-g + sh_t for g in gx[1:]]; tu = [-g + sh_t for g in gu]
        lx = [None] + [g + sh_l for g in gx[1:]]; lu = [g + sh_l for g in gu]'''
assert old in src
variants = {
 'mehrotra': '''        tt = -allg; ll = allg.copy()
        dt_ = max(-1.5 * tt.min(), 0.0); dl_ = max(-1.5 * ll.min(), 0.0)
        tt = tt + dt_; ll = ll + dl_
        dot = float(tt @ ll)
        dp = 0.5 * dot / max(ll.sum(), 1e-300); dd = 0.5 * dot / max(tt.sum(), 1e-300)
        sh_t = dt_ + dp; sh_l = dl_ + dd
        tx = [None] + [-g + sh_t for g in gx[1:]]; tu = [-g + sh_t for g in gu]
        lx = [None] + [g + sh_l for g in gx[1:]]; lu = [g + sh_l for g in gu]''',
 'central1': '''        sp_ = max(1.0, abs(p.delta), np.abs(Ub).max() if nU else 1.0); sd_ = max(1.0, p.omega, np.abs(p.grad_x(1, np.zeros(n))).max())
        flo = 1e-2 * sp_
        mu0 = MU0 * sd_ * sp_ * 1e-2
        tx = [None] + [np.maximum(-g, flo) for g in gx[1:]]; tu = [np.maximum(-g, flo) for g in gu]
        lx = [None] + [mu0 / t for t in tx[1:]]; lu = [mu0 / t for t in tu]''',
}
qps = []
for which in ('c2','c5'):
    w = wl.diamond_c2() if which=='c2' else wl.trunk_c5()
    for b in (0,2,5):
        qp = T.first_qp(w, b=b, B=8, seed=9)
        p = ripm.Problem(w['N'], w['H'], w['Qz'], w['R'], qp['A'], qp['B'], qp['d'], qp['x0'], qp['xk'], 1e4, 1.0, z=qp['z'],
                         U=(w['UA'], w['Ub']), X=(w['XA'], w['Xb']) if w['XA'] is not None else None, x_scale=1.0 / np.abs(qp['xc']))
        qps.append((which, b, p))
base = {}
for which, b, p in qps:
    xe, ue, Je, inf = cipm.solve(p)
    base[(which,b)] = (xe, ue, Je, inf['iters'])
print('baseline iters', {k: v[3] for k, v in base.items()})
for name, body in variants.items():
    for MU0 in ((1.0,) if name == 'mehrotra' else (1.0, 0.1, 0.01, 1e-3)):
        mod = types.ModuleType('cipm_' + name)
        mod.__dict__['__file__'] = cipm.__file__
        code = src.replace(old, body.replace('MU0', repr(MU0)))
        code = code.replace('from . import', 'from oracle import').replace('from .', 'from oracle.')
        exec(compile(code, 'cipm_' + name, 'exec'), mod.__dict__)
        res = {}
        for which, b, p in qps:
            try:
                x, u, J, inf = mod.solve(p)
                xe, ue, Je, it0 = base[(which, b)]
                res[(which, b)] = (inf['iters'], inf['status'][:3], '%.1e' % (np.abs(x - xe).max() / np.abs(xe).max()))
            except Exception as e:
                res[(which, b)] = ('err', str(e)[:40])
        print(name, MU0, res)
