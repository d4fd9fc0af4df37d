import sys, io, contextlib, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.')
from helpers import golden_problem, product_tpwl, Poly
from sofacontrol_amd.scp.models.tpwl import TPWLGuSTO
from sofacontrol_amd.scp.gusto import GuSTO
from scipy.interpolate import interp1d
g = np.load('tests/golden/g6_gusto.npz')
model, U, q_ref, v_ref, Hf = golden_problem(4, 3, 7, 20, 30, q_scale=0.05)
tp = product_tpwl(model, U, q_ref, v_ref, Hf)
gm = TPWLGuSTO(tp)
with contextlib.redirect_stdout(io.StringIO()):
    gm.pre_discretize(0.05)
N, dt, B = 12, 0.05, 300
rng = np.random.default_rng(5)
x0 = 1e-3 * rng.standard_normal((B, 8)) * rng.uniform(0.1, 30.0, (B, 1))
u_init = np.zeros((B, N, 3)); x_init, _ = gm.rollout(x0, u_init, dt)
zi = interp1d(g['t'], g['zt'], axis=0)
z = np.stack([zi(0.003 * b + dt * np.arange(N + 1)) for b in range(B)])
kw = dict(x_char=g['x_char'], f_char=g['f_char'], convg_thresh=1e-3, U=Poly(g['U_A'], g['U_b']))
gb = GuSTO(gm, N, dt, g['Qz'], g['R'], x0, u_init, x_init, z=z, batch=B, **kw)
first = (gb.xopt.copy(), gb.iters.copy())
bad = []
for rep in range(12):
    gb.solve_batch(x0, u_init, x_init, z=z)
    d = np.abs(gb.xopt - first[0]).reshape(B, -1).max(1)
    bad.append((int((d > 0).sum()), np.nonzero(d > 0)[0][:4].tolist(), float(d.max()), int((gb.iters != first[1]).sum())))
print(bad)

if len(sys.argv) > 1 and sys.argv[1] == 'c2':
    # the same check at the benchmark shape (specialised kernel, two rounds of workgroups)
    sys.path.insert(0, '.')
    import workloads as wl, bench
    w = wl.diamond_c2(); N2, m2, r2, dt2 = w['N'], w['m'], w['r'], w['dt']
    tp2, gm2 = bench.build_model(w); xc, fc = gm2.get_characteristic_vals()
    zi2 = interp1d(w['t'], w['z'], axis=0, bounds_error=False, fill_value=(w['z'][0], w['z'][-1]))
    R = 600
    x02 = np.concatenate((np.zeros((R, r2)), np.random.default_rng(3).standard_normal((R, r2)) * 2.0), axis=1)
    u02 = np.zeros((R, N2, m2)); xi2, _ = tp2.rollout(x02, u02, dt2)
    z2 = np.stack([zi2(b * 10.0 / R + dt2 * np.arange(N2 + 1)) for b in range(R)])
    from sofacontrol_amd.utils import Polyhedron
    g2 = GuSTO(gm2, N2, dt2, w['Qz'], w['R'], x02, u02, xi2, z=z2, U=Polyhedron(w['UA'], w['Ub']), X=Polyhedron(w['XA'], w['Xb']),
               x_char=xc, f_char=fc, convg_thresh=1e-3, batch=R, max_trace=0, max_gusto_iters=5)
    g2.max_gusto_iters = 5
    g2.solve_batch(x02, u02, xi2, z=z2)
    ref = (g2.xopt.copy(), g2.iters.copy())
    out = []
    for rep in range(4):
        g2.solve_batch(x02, u02, xi2, z=z2)
        d = np.abs(g2.xopt - ref[0]).reshape(R, -1).max(1)
        out.append((int((d > 0).sum()), float(d.max()), int((g2.iters != ref[1]).sum())))
    print('c2 shape:', out)
