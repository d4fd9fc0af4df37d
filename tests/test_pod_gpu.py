"""GPU parity: POD projection / lift / U^T M U through the C ABI against the oracle and the
golden vectors of the imported reference.  float64; tolerance 1e-12 relative to the magnitude of the
data (the kernels reorder the dot products; the reference is BLAS-ordered as well)."""
import numpy as np
import pytest

from oracle import pod as opod

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-12):
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * scale * 50)


def make_rom(n_f, r, seed=0):
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((n_f, r)))
    q_ref = np.random.default_rng(seed + 1).uniform(-108, 107, n_f)
    v_ref = 0.01 * np.random.default_rng(seed + 2).standard_normal(n_f)
    return U, q_ref, v_ref


def test_golden_g1(golden):
    from sofacontrol_amd.mor.pod import POD
    g = golden('g1_pod')
    rom = POD(dict(U=g['U'], q_ref=g['q_ref'], v_ref=g['v_ref']))
    close(rom.V, g['V'])
    close(rom.x_ref, g['x_ref'])
    for i in range(5):   # single-vector calls, as tpwl/controllers.py:96
        close(rom.compute_RO_state(qf=g['Xq'][i]), g['proj_q'][i])
        close(rom.compute_RO_state(vf=g['Xv'][i]), g['proj_v'][i])
    Xx = opod.qv2x(g['Xq'], g['Xv'])
    close(rom.compute_RO_state(xf=Xx), g['proj_x'])
    close(rom.compute_RO_state(xf=Xx[2]), g['proj_x'][2])
    pr = g['proj_x']
    close(rom.compute_FO_state(q=pr[:, 8:]), g['lift_q'])
    close(rom.compute_FO_state(v=pr[1, :8]), g['lift_v'][1])
    close(rom.compute_FO_state(x=pr), g['lift_x'])
    close(rom.compute_RO_matrix(g['M']), g['UMU'])
    close(rom.compute_RO_matrix(g['M'], left=True, right=True), g['UMU'])
    close(rom.compute_RO_matrix(g['M'], left=True), g['UM'])
    close(rom.compute_RO_matrix(g['M'], right=True), g['MU'])
    close(rom.compute_RO_matrix(g['Hm'], left=True), g['UH'])
    from scipy.sparse import coo_matrix
    close(rom.compute_RO_matrix(coo_matrix(g['Mc_dense'])), g['UMcU'])
    with pytest.raises(RuntimeError):
        rom.compute_RO_state()
    with pytest.raises(RuntimeError):
        rom.compute_FO_state()
    with pytest.raises(RuntimeError):
        rom.compute_RO_matrix([[1.0]])


@pytest.mark.parametrize('n_f,r', [(4884, 30), (4884, 36), (2127, 10), (2127, 30), (50, 3), (1000, 64),
                                   (300, 7), (300, 22), (500, 41), (500, 55)])   # every (full, 4-column) tile mix
@pytest.mark.parametrize('B', [1, 7, 128, 1000])
def test_project_lift_vs_oracle(n_f, r, B):
    from sofacontrol_amd.mor.pod import POD
    U, q_ref, v_ref = make_rom(n_f, r)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    rng = np.random.default_rng(B)
    Xq = q_ref + 5 * rng.standard_normal((B, n_f))
    Xv = rng.standard_normal((B, n_f))
    close(rom.compute_RO_state(qf=Xq), opod.project(U, q_ref, Xq))
    close(rom.compute_RO_state(vf=Xv), opod.project(U, v_ref, Xv))
    Xx = opod.qv2x(Xq, Xv)
    pr = opod.project_x(U, q_ref, v_ref, Xx)
    close(rom.compute_RO_state(xf=Xx), pr)
    close(rom.compute_FO_state(q=pr[:, r:]), opod.lift(U, q_ref, pr[:, r:]))
    close(rom.compute_FO_state(x=pr), opod.lift_x(U, q_ref, v_ref, pr))


@pytest.mark.parametrize('n_f,r,B,pitch_extra,base_off', [
    (4884, 30, 300, 0, 0),      # pitch = 4 (mod 16): four alignment classes, partial last super-block
    (4884, 30, 257, 0, 3),      # output base not line aligned
    (2127, 9, 1100, 0, 0),      # odd pitch: sixteen classes
    (2127, 30, 70, 5, 1),       # pitch > n_f: the gap between rows stays untouched
    (4896, 36, 130, 0, 0),      # aligned pitch: one class
    (333, 64, 65, 0, 7),
    (50, 3, 1, 0, 0),
])
def test_lift_alignment_classes(n_f, r, B, pitch_extra, base_off):
    """The lift picks the rows of an MFMA tile by the 128-byte alignment class of their start and shifts the column
    window per class: every (pitch mod 16, base offset, ragged B) must still write exactly out[b, :n_f] and nothing
    else (sentinel check around and between the rows).  Reference: sofacontrol/mor/pod.py:54-66."""
    import ctypes as C
    from sofacontrol_amd import _lib
    from sofacontrol_amd.mor.pod import POD
    U, q_ref, v_ref = make_rom(n_f, r, seed=3)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    rng = np.random.default_rng(B)
    Xr = rng.standard_normal((B, r))
    ldo = n_f + pitch_extra
    guard = 64
    total = guard + base_off + B * ldo + guard
    sentinel = -777.25
    dO = _lib.DeviceBuffer.from_array(np.full(total, sentinel))
    dX = _lib.DeviceBuffer.from_array(Xr)
    optr = C.c_void_p(dO.ptr.value + 8 * (guard + base_off))
    L = _lib.lib()
    _lib.check(L.srom_lift_dev(rom.handle, 0, dX.ptr, C.c_int64(B), C.c_int64(r), optr, C.c_int64(ldo), None), 'lift')
    _lib.sync()
    got = dO.to_array((total,))
    body = got[guard + base_off:guard + base_off + B * ldo].reshape(B, ldo)
    close(body[:, :n_f], opod.lift(U, q_ref, Xr))
    assert np.all(body[:, n_f:] == sentinel)
    assert np.all(got[:guard + base_off] == sentinel) and np.all(got[guard + base_off + B * ldo:] == sentinel)


def test_project_transpose_detecting():
    """Asymmetric basis with identity-like snapshots: catches swapped MFMA operands / C layout."""
    from sofacontrol_amd.mor.pod import POD
    n_f, r = 200, 20
    U = np.arange(n_f * r, dtype=np.float64).reshape(n_f, r) / 7.0
    rom = POD(dict(U=U, q_ref=np.zeros(n_f), v_ref=np.zeros(n_f)))
    X = np.eye(n_f)[:150]
    np.testing.assert_array_equal(rom.compute_RO_state(qf=X), U[:150])
    np.testing.assert_array_equal(rom.compute_FO_state(q=np.eye(r)), U.T)


def test_reduce_matrix_diamond_size():
    from sofacontrol_amd.mor.pod import POD
    n_f, r = 4884, 30
    U, q_ref, v_ref = make_rom(n_f, r)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    rng = np.random.default_rng(5)
    M = rng.standard_normal((n_f, n_f))
    close(rom.compute_RO_matrix(M), opod.reduce_matrix(U, M), rtol=1e-11)
    H = rng.standard_normal((n_f, 4))
    close(rom.compute_RO_matrix(H, left=True), U.T @ H)
    b = rng.standard_normal(n_f)
    close(rom.compute_RO_matrix(b, left=True), U.T @ b)


@pytest.mark.parametrize('n_f,r', [(131, 5), (300, 16), (777, 22), (1000, 30), (1539, 36), (640, 48), (515, 64)])
def test_reduce_matrix_one_pass_shapes(n_f, r, monkeypatch):
    """U^T M U in one pass over M (the r x r partial of every workgroup, then one reduction): every accumulator
    layout of the projection kernel (full 16-column tiles, 4-column tiles, both), ragged last row tile and last
    column chunk; against the oracle and against the older two-pass path (T = M U through HBM)."""
    from sofacontrol_amd.mor.pod import POD
    U, q_ref, v_ref = make_rom(n_f, r)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    M = np.random.default_rng(n_f + r).standard_normal((n_f, n_f))
    got = rom.compute_RO_matrix(M)
    close(got, opod.reduce_matrix(U, M), rtol=1e-12)
    monkeypatch.setenv('SRH_UTMU_TWO_PASS', '1')
    close(got, rom.compute_RO_matrix(M), rtol=1e-12)


@pytest.mark.parametrize('n_f,r,count', [(300, 8, 4), (1000, 30, 3), (777, 36, 6), (4884, 30, 4)])
def test_reduce_matrices_batch_equals_one_by_one(n_f, r, count):
    """K, D, M, S of one TPWL point in ONE call (srom_reduce_matrices: groups of four share a launch pair, blockIdx.z = matrix):
    the oracle's U^T M U to rounding, the single-matrix form's to rounding (a group uses fewer K-slices per row tile so that all
    its workgroups are resident at once: another summation order), and the same bits when the same batch is reduced twice.
    count = 6: a full group and a group of two; count = 3: a ragged group."""
    from sofacontrol_amd.mor.pod import POD
    U, q_ref, v_ref = make_rom(n_f, r)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    rng = np.random.default_rng(n_f + r + count)
    Ms = [rng.standard_normal((n_f, n_f)) * (1.0 + 10.0 * i) for i in range(count)]
    got = rom.compute_RO_matrices(Ms)
    assert len(got) == count
    for M, g in zip(Ms, got):
        assert g.shape == (r, r)
        close(g, rom.compute_RO_matrix(M), rtol=1e-12)
        close(g, opod.reduce_matrix(U, M), rtol=1e-12)
    again = rom.compute_RO_matrices(Ms)
    assert all(np.array_equal(a, b) for a, b in zip(got, again))
    with pytest.raises(RuntimeError):
        rom.compute_RO_matrices([Ms[0][:, :-1]])


def test_projection_round_trip_full_size():
    """BASELINE size property test: lifting then projecting is the identity on the reduced space
    (U orthonormal), and projection is linear -- independent of any CPU result."""
    from sofacontrol_amd.mor.pod import POD
    n_f, r, B = 4884, 30, 4096
    U, q_ref, v_ref = make_rom(n_f, r)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    rng = np.random.default_rng(9)
    Xr = rng.standard_normal((B, r))
    Xf = rom.compute_FO_state(q=Xr)
    back = rom.compute_RO_state(qf=Xf)
    np.testing.assert_allclose(back, Xr, rtol=0, atol=1e-11)
    a, b = rom.compute_RO_state(qf=Xf[:64]), rom.compute_RO_state(qf=Xf[64:128])
    mid = rom.compute_RO_state(qf=0.5 * (Xf[:64] + Xf[64:128]))
    np.testing.assert_allclose(mid, 0.5 * (a + b), rtol=0, atol=1e-11)


@pytest.mark.parametrize('n_s,n_f', [(40, 300), (257, 1001), (300, 4884), (129, 64)])
def test_gramian_vs_numpy(n_s, n_f):
    from sofacontrol_amd.mor.pod import gramian
    rng = np.random.default_rng(n_s)
    S = rng.standard_normal((n_s, n_f)) * rng.uniform(0.1, 10.0, (n_s, 1))
    G = gramian(S)
    ref = S @ S.T
    np.testing.assert_allclose(G, ref, rtol=0, atol=1e-12 * np.abs(ref).max() * 10)
    np.testing.assert_array_equal(G, G.T)


def test_compute_pod_gramian_route_vs_reference_svd(golden):
    """compute_POD (pod.py:181-200): same k, singular values and modes (up to sign) as the reference's SVD."""
    from sofacontrol_amd.mor.pod import compute_POD
    g = golden('g1_pod')
    for tol in (1e-2, 1e-4, 1e-7):
        _, U, k, Sig = compute_POD(g['pod_S'], tol)
        assert k == int(g['pod_k_%g' % tol])
        np.testing.assert_allclose(Sig[:6], g['pod_Sigma'][:6], rtol=1e-9)
        np.testing.assert_allclose(np.abs(U), g['pod_Ufull_abs'][:, :k], rtol=0, atol=1e-8)
        np.testing.assert_allclose(U.T @ U, np.eye(k), rtol=0, atol=1e-9)


def test_tpwl_assembly_golden(golden):
    """TPWLSnapshotData.add_point / add_continuous_TPWL / add_discrete_TPWL / evaluate_point_dist
    (tpwl/tpwl_utils.py:84-117, 170-196, 263-290): full-order points reduced on the device, model assembled."""
    import io, contextlib
    from types import SimpleNamespace
    from helpers import small_rom, assembly_points
    from sofacontrol_amd.mor.pod import POD
    from sofacontrol_amd.tpwl.tpwl_utils import TPWLSnapshotData
    from sofacontrol_amd.utils import Point
    g = golden('g12_assembly')
    n_nodes, r, m = 30, 5, 3
    U, q_ref, v_ref = small_rom(n_nodes, r, 120)
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    cfg = SimpleNamespace(eval_type='distance', save_continuous_TPWL=True, save_discrete_TPWL=True,
                          TPWL_weighting_factors={'q': 1.0, 'v': 0.2}, TPWL_separate_calculation=False,
                          TPWL_threshold=1.0, TPWL_type='ATV', discr_type='zoh')
    data = TPWLSnapshotData(rom, cfg)
    pts = []
    for d in assembly_points(3 * n_nodes, m, q_ref, 121):
        p = Point()
        for k, v in d.items():
            setattr(p, k, v)
        pts.append(p)
    with contextlib.redirect_stdout(io.StringIO()):
        assert data.evaluate_point(pts[0], None) == bool(g['eval0'])
        data.add_point(pts[0])
        assert data.evaluate_point(pts[1], pts[0]) == bool(g['eval1'])
        data.add_point(pts[1])
        near = Point(); near.q = pts[1].q + 1e-3; near.v = pts[1].v
        assert data.evaluate_point(near, pts[1]) == bool(g['eval_near'])
        data.add_point(pts[2])
    for k in ('q', 'v', 'K', 'D', 'M', 'S', 'H', 'b', 'f', 'q+', 'v+'):
        ref = g['out_' + k]
        np.testing.assert_allclose(np.asarray(data.dict[k]), ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))
    for k in ('A_c', 'B_c', 'd_c', 'A_d', 'B_d', 'd_d'):
        ref = g['out_' + k]
        np.testing.assert_allclose(np.asarray(data.dict[k]), ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


def test_file_round_trips(tmp_path):
    """run_POD / load_POD (mor/pod.py:93-141) and TPWLSnapshotData.simulation_end -> TPWLATV(data=<file>)
    (tpwl_utils.py:130-153, tpwl.py:26-29): the pickle formats of the reference, written and read back."""
    import io, contextlib
    from types import SimpleNamespace
    from helpers import small_rom, assembly_points
    from sofacontrol_amd import utils as scutils
    from sofacontrol_amd.mor.pod import run_POD, load_POD, pod_config
    from sofacontrol_amd.tpwl.tpwl_utils import TPWLSnapshotData
    from sofacontrol_amd.tpwl.tpwl import TPWLATV
    from sofacontrol_amd.utils import Point
    rng = np.random.default_rng(3)
    n_f, n_s = 90, 40
    L = rng.standard_normal((n_s, 4)) * np.array([30, 10, 3, 1.0])
    q = L @ rng.standard_normal((4, n_f)) + 1e-4 * rng.standard_normal((n_s, n_f))
    snap, podf, tpwlf = str(tmp_path / 'snap.pkl'), str(tmp_path / 'pod.pkl'), str(tmp_path / 'tpwl.pkl')
    scutils.save_data(snap, {'q': list(q + 5.0), 'v': list(q)})
    cfg = pod_config(); cfg.pod_type = 'v'; cfg.pod_tolerance = 1e-6
    with contextlib.redirect_stdout(io.StringIO()):
        res = run_POD(snap, podf, cfg)
    rom = load_POD(podf)
    assert rom.rom_dim == res['POD_info']['U'].shape[1] == 4
    s = np.linalg.svd(q.T, compute_uv=False)
    np.testing.assert_allclose(res['Sigma'][:4], s[:4], rtol=1e-9)
    np.testing.assert_allclose(rom.q_ref, q[0] + 5.0)
    # TPWL data collected with this ROM, saved, loaded back as a model
    tcfg = SimpleNamespace(eval_type='distance', save_continuous_TPWL=True, save_discrete_TPWL=True,
                           TPWL_weighting_factors={'q': 1.0, 'v': 0.0}, TPWL_separate_calculation=False,
                           TPWL_threshold=1.0, TPWL_type='ATV', discr_type='zoh')
    data = TPWLSnapshotData(rom, tcfg)
    with contextlib.redirect_stdout(io.StringIO()):
        for d in assembly_points(n_f, 3, rom.q_ref, 7, count=2):
            p = Point()
            for k, v in d.items():
                setattr(p, k, v)
            data.add_point(p)
        data.simulation_end(tpwlf)
    tp = TPWLATV(data=tpwlf, params=dict(tpwl_method='nn', dist_weights={'q': 1.0, 'v': 0.0}), discr_method='zoh')
    assert tp.num_points == 2 and tp.get_state_dim() == 8 and tp.get_input_dim() == 3
    saved = scutils.load_data(tpwlf)
    A, B, d = tp.get_jacobians(np.concatenate((saved['v'][1], saved['q'][1])))
    np.testing.assert_array_equal(A, saved['A_c'][1])
    assert saved['info']['nbr_lin'] == '2' and saved['rom_info']['type'] == 'POD'


def test_sharded_pod_build_device_resident_single_rank():
    """distributed.pod_from_column_shards on the device (world = 1: no collective): Gramian in a torch-owned HBM tensor ->
    eigh in place -> mode selection -> local rows of U; against the reference SVD route (oracle.pod.compute_pod)."""
    import torch
    from oracle import pod as opod
    from sofacontrol_amd.distributed import pod_from_column_shards
    rng = np.random.default_rng(4)
    n_s, n_f = 60, 501
    Lm = rng.standard_normal((n_s, 6)) * np.array([50, 20, 8, 3, 1, 0.3])
    S = Lm @ rng.standard_normal((6, n_f)) + 1e-3 * rng.standard_normal((n_s, n_f))
    tm = {}
    U, k, Sig = pod_from_column_shards(S, 1e-4, timings=tm)
    _, U_ref, k_ref, S_ref = opod.compute_pod(S.T, 1e-4)
    assert k == k_ref and tm['collective'] == 'none' and set(tm) >= {'gramian_s', 'eigh_s', 'modes_s'}
    np.testing.assert_allclose(Sig[:6], S_ref[:6], rtol=1e-9)
    np.testing.assert_allclose(np.abs(U), np.abs(U_ref), rtol=0, atol=1e-8)
    # the same with torch-owned buffers (what a process group uses so that RCCL reduces the Gramian in place), from a shard
    # that is already resident, result kept on the device
    Ud, k2, _ = pod_from_column_shards(torch.from_numpy(S).cuda(), 1e-4, keep_on_device=True, force_torch=True)
    assert Ud.is_cuda and k2 == k
    np.testing.assert_allclose(np.abs(Ud.cpu().numpy()), np.abs(U_ref), rtol=0, atol=1e-8)


def test_gramian_full_size_c4_shard_properties():
    """BASELINE config C4 at its per-GPU size (10 000 snapshots x 6250 DoF columns): size-independent properties of
    G = S S^T -- symmetry, trace(G) = |S|_F^2, G v = S (S^T v) for random v."""
    import ctypes as C
    import torch
    from sofacontrol_amd import _lib
    n_s, n_f = 10000, 6250
    gen = torch.Generator(device='cuda'); gen.manual_seed(11)
    S = torch.randn((n_s, n_f), dtype=torch.float64, device='cuda', generator=gen)
    G = torch.empty((n_s, n_s), dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    _lib.check(_lib.lib().srom_gramian_dev(C.c_void_p(S.data_ptr()), C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f),
                                           C.c_void_p(G.data_ptr()), None), 'srom_gramian_dev')
    _lib.sync()
    assert torch.equal(G, G.T)                                            # mirrored, not recomputed: exactly symmetric
    fro2 = float((S * S).sum())
    assert abs(float(torch.diagonal(G).sum()) - fro2) <= 1e-11 * fro2
    v = torch.randn((n_s, 3), dtype=torch.float64, device='cuda', generator=gen)
    ref = S @ (S.T @ v)
    got = G @ v
    assert float((got - ref).abs().max()) <= 1e-10 * float(ref.abs().max())


def test_leading_eigenpairs_vs_numpy():
    """srom_eigh_topk_dev (blocked subspace iteration + Rayleigh-Ritz) against numpy.linalg.eigh on Gramians with a decaying
    and with a clustered spectrum: eigenvalues to 1e-10 relative, residuals |G v - lambda v|, orthonormality, trace."""
    import ctypes as C
    from sofacontrol_amd import _lib
    rng = np.random.default_rng(5)
    for n, k, spec in ((600, 20, 'decay'), (900, 64, 'cluster'), (300, 8, 'lowrank')):
        Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
        if spec == 'decay':
            lam = 10.0 ** (-0.08 * np.arange(n))
        elif spec == 'cluster':
            lam = np.concatenate((1.0 + 0.01 * rng.standard_normal(70) ** 2, 1e-6 * rng.uniform(0.5, 1.0, n - 70)))
            lam = np.sort(lam)[::-1]
        else:
            lam = np.concatenate((np.linspace(5.0, 1.0, 12), np.zeros(n - 12)))
        G = (Qm * lam) @ Qm.T
        G = 0.5 * (G + G.T)
        dG = _lib.DeviceBuffer.from_array(G)
        dw, dV, dW = _lib.DeviceBuffer(k * 8), _lib.DeviceBuffer(k * n * 8), _lib.DeviceBuffer(k * n * 8)
        tr, it = C.c_double(), C.c_int()
        _lib.check(_lib.lib().srom_eigh_topk_dev(dG.ptr, C.c_int64(n), C.c_int(k), C.c_int(-1), dw.ptr, dW.ptr, dV.ptr, C.byref(tr),
                                                 C.byref(it), None), 'srom_eigh_topk_dev')
        w, Vt, Wk = dw.to_array((k,)), dV.to_array((k, n)), dW.to_array((n, k))
        wref = np.linalg.eigvalsh(G)[::-1][:k]
        np.testing.assert_allclose(w, wref, rtol=1e-10, atol=1e-12 * wref[0], err_msg=spec)
        assert abs(tr.value - np.trace(G)) <= 1e-12 * np.trace(G)
        assert np.abs(Vt @ Vt.T - np.eye(k)).max() <= 1e-10, spec
        assert np.abs(G @ Vt.T - Vt.T * w).max() <= 1e-9 * wref[0], (spec, np.abs(G @ Vt.T - Vt.T * w).max())
        np.testing.assert_allclose(Wk, Vt.T / np.sqrt(w), rtol=1e-12, atol=0)
        assert 1 <= it.value < 30


def test_sharded_pod_build_leading_modes_vs_svd():
    """More than 2048 snapshots with rom_dim given: pod_from_column_shards takes the leading-eigenpair route
    (srom_eigh_topk_dev) -- modes and singular values against numpy's SVD (pod.py:190) of the same snapshot matrix; with a
    tolerance instead of rom_dim the block grows until the tail energy fits and the same k as the reference rule comes out."""
    from sofacontrol_amd.distributed import pod_from_column_shards
    from sofacontrol_amd.mor.pod import energy_truncation
    rng = np.random.default_rng(3)
    n_s, n_f, rank = 2600, 700, 24
    S = (rng.standard_normal((n_s, rank)) * np.linspace(30.0, 2.0, rank)) @ rng.standard_normal((rank, n_f)) + 1e-3 * rng.standard_normal((n_s, n_f))
    Uref, Sref, _ = np.linalg.svd(S.T, full_matrices=False)
    tm = {}
    U, k, Sig = pod_from_column_shards(S, 1e-4, rom_dim=16, timings=tm)
    assert tm['spectrum'] == 'leading' and k == 16 and U.shape == (n_f, 16)
    np.testing.assert_allclose(Sig[:16], Sref[:16], rtol=1e-9)
    np.testing.assert_allclose(np.abs(U), np.abs(Uref[:, :16]), rtol=0, atol=1e-7)
    tm = {}
    U2, k2, Sig2 = pod_from_column_shards(S, 1e-6, timings=tm)
    assert tm['spectrum'] == 'leading' and k2 == energy_truncation(Sref, 1e-6), (k2, energy_truncation(Sref, 1e-6))
    np.testing.assert_allclose(np.abs(U2), np.abs(Uref[:, :k2]), rtol=0, atol=1e-6)
    # the full spectrum on request (what the reference stores as Sigma)
    tm = {}
    U3, k3, Sig3 = pod_from_column_shards(S, 1e-6, timings=tm, spectrum='full')
    assert tm['spectrum'] == 'full' and k3 == k2 and len(Sig3) == n_s
    np.testing.assert_allclose(Sig3[:rank], Sref[:rank], rtol=1e-8)


def test_pod_build_c4_shard_end_to_end_properties():
    """BASELINE config C4 at its per-GPU size, END TO END (Gramian -> leading eigenpairs -> modes) on a resident shard of
    10 000 snapshots x 6250 DoF columns (low rank 64 + noise, as bench.py): size-independent properties checked in float64
    on the host -- U^T U = I, G W = W Lambda through S (S^T W) on the kept modes, tail energy = trace - sum, sigma_i = |S u_i|."""
    import torch
    from sofacontrol_amd.distributed import pod_from_column_shards
    n_s, n_f, k = 10000, 6250, 64
    gen = torch.Generator(device='cuda'); gen.manual_seed(21)
    Lr = torch.randn((n_s, k), dtype=torch.float64, device='cuda', generator=gen) * torch.linspace(40.0, 4.0, k, dtype=torch.float64, device='cuda')
    S_t = Lr @ torch.randn((k, n_f), dtype=torch.float64, device='cuda', generator=gen) + 1e-3 * torch.randn((n_s, n_f), dtype=torch.float64, device='cuda', generator=gen)
    del Lr
    tm = {}
    U_t, kk, Sig = pod_from_column_shards(S_t, 1e-4, rom_dim=k, timings=tm, keep_on_device=True, force_torch=True)
    assert kk == k and tm['spectrum'] == 'leading' and tm['subspace_iterations'] <= 10, tm
    U = U_t.cpu().numpy()
    S = S_t.cpu().numpy()
    assert np.abs(U.T @ U - np.eye(k)).max() <= 1e-9
    SU = S @ U                                         # = W Sigma
    np.testing.assert_allclose(np.linalg.norm(SU, axis=0), Sig[:k], rtol=1e-10)
    W = SU / Sig[:k]
    R = S @ (S.T @ W) - W * Sig[:k] ** 2               # G W - W Lambda
    assert np.abs(R).max() <= 1e-9 * Sig[0] ** 2, np.abs(R).max()
    fro2 = float((S * S).sum())
    tail = (fro2 - float((Sig[:k] ** 2).sum())) / fro2
    assert 0.0 <= tail <= 1e-6                          # the 1e-3 noise floor


def test_shipped_pod_model(golden):
    """The reference's shipped Diamond POD model (examples/diamond/pod_model.pkl: U 4884 x 36 -- two 16-column MFMA tiles
    + one 4-column tile -- q_ref, v_ref) and rest state through the device kernels, against what the imported reference
    POD class computes on the same files (golden g18; seeded inputs are regenerated here)."""
    from sofacontrol_amd.mor.pod import POD
    from sofacontrol_amd.utils import qv2x
    g = golden('g18_pod_shipped')
    U, q_ref, v_ref = g['U'], g['q_ref'], g['v_ref']
    rom = POD(dict(U=U, q_ref=q_ref, v_ref=v_ref))
    n_f, r = U.shape
    rng = np.random.default_rng(int(g['seed']))
    Xq = q_ref + 5 * rng.standard_normal((5, n_f))
    Xv = rng.standard_normal((5, n_f))
    close(rom.compute_RO_state(qf=g['rest']), g['proj_rest'], 1e-12)
    close(rom.compute_RO_state(qf=Xq), g['proj_q'], 1e-12)
    close(rom.compute_RO_state(xf=qv2x(Xq, Xv)), g['proj_x'], 1e-12)
    close(rom.compute_FO_state(x=g['proj_x']), g['lift_x'], 1e-12)
    M = np.random.default_rng(int(g['seed']) + 1).standard_normal((n_f, n_f))
    close(rom.compute_RO_matrix(M), g['UMU'], 1e-11)
    close(rom.compute_RO_matrix(M, right=True)[:64], g['MU_rows'], 1e-12)
    Hm = np.random.default_rng(int(g['seed']) + 2).standard_normal((n_f, 4))
    close(rom.compute_RO_matrix(Hm, left=True), g['UH'], 1e-12)
