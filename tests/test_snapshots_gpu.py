"""Snapshot preprocessing of the POD build on the device (csrc/snapshots.hip) against the imported reference's outputs
(golden g19: sofacontrol/mor/pod.py:157-178, 207-216) and against numpy."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g19_preprocess.npz')


@pytest.fixture(scope='module')
def g19():
    return np.load(GOLD)


@pytest.mark.parametrize('shape', [(5, 3), (64, 257), (300, 1000), (1031, 777)])
def test_column_statistics_normalize_center_vs_numpy(shape):
    from sofacontrol_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(shape[0])
    S = rng.standard_normal(shape) * (1.0 + 10.0 * rng.random(shape[1])) + rng.standard_normal(shape[1])
    n_s, n_f = shape
    dS = _lib.DeviceBuffer.from_array(S)
    dmin, dmax, dmean = (_lib.DeviceBuffer(n_f * 8) for _ in range(3))
    n64 = (C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f))
    _lib.check(L.srom_snapshot_stats_dev(dS.ptr, *n64, dmin.ptr, dmax.ptr, dmean.ptr, None), 'stats')
    assert np.array_equal(dmin.to_array((n_f,)), S.min(axis=0)) and np.array_equal(dmax.to_array((n_f,)), S.max(axis=0))
    np.testing.assert_allclose(dmean.to_array((n_f,)), S.mean(axis=0), rtol=0, atol=1e-13 * np.abs(S).max())
    _lib.check(L.srom_snapshot_normalize_dev(dS.ptr, *n64, dmin.ptr, dmax.ptr, None), 'normalize')
    _lib.sync()
    ref = (S - S.min(axis=0)) / (S.max(axis=0) + 1e-15 - S.min(axis=0))
    np.testing.assert_allclose(dS.to_array(shape), ref, rtol=1e-15, atol=1e-16)
    dS2 = _lib.DeviceBuffer.from_array(S)
    _lib.check(L.srom_snapshot_center_dev(dS2.ptr, *n64, dmean.ptr, None), 'center')
    _lib.sync()
    np.testing.assert_allclose(dS2.to_array(shape), S - S.mean(axis=0, keepdims=True), rtol=0, atol=1e-12 * np.abs(S).max())


def test_pitched_rows_and_single_outputs():
    from sofacontrol_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    full = rng.standard_normal((40, 90))
    n_s, n_f, lds = 40, 70, 90                                   # the first 70 columns of a 90-column buffer
    dS = _lib.DeviceBuffer.from_array(full)
    dmean = _lib.DeviceBuffer(n_f * 8)
    _lib.check(L.srom_snapshot_stats_dev(dS.ptr, C.c_int64(n_s), C.c_int64(n_f), C.c_int64(lds), None, None, dmean.ptr, None), 'stats')
    np.testing.assert_allclose(dmean.to_array((n_f,)), full[:, :70].mean(axis=0), atol=1e-14)
    _lib.check(L.srom_snapshot_center_dev(dS.ptr, C.c_int64(n_s), C.c_int64(n_f), C.c_int64(lds), dmean.ptr, None), 'center')
    _lib.sync()
    got = dS.to_array((40, 90))
    np.testing.assert_allclose(got[:, :70], full[:, :70] - full[:, :70].mean(axis=0), atol=1e-14)
    assert np.array_equal(got[:, 70:], full[:, 70:])             # the columns behind n_f are not touched


@pytest.mark.parametrize('name', ['blobs', 'cloud'])
def test_process_snapshots_matches_the_imported_reference(g19, name):
    from sofacontrol_amd.mor import pod
    S = g19[name]
    k = int(g19[name + '_k'])
    assert np.abs(pod.process_snapshots(S.copy(), ['normalize'], {}) - g19[name + '_normalize']).max() <= 1e-15
    assert np.abs(pod.process_snapshots(S.copy(), ['substract_mean'], {}) - g19[name + '_mean']).max() <= 1e-13 * np.abs(S).max()
    assert np.abs(pod.process_snapshots(S.copy(), ['normalize', 'substract_mean'], {}) - g19[name + '_both']).max() <= 1e-14
    # k-means: the same seeding draws, the same Lloyd iterates -> the same centroids, in the same order
    got = pod.process_snapshots(S.copy(), ['clustering'], dict(nbr_clusters=k))
    assert got.shape == (k, S.shape[1])
    assert np.abs(got - g19[name + '_centroids']).max() <= 1e-11 * np.abs(S).max()
    got = pod.process_snapshots(S.copy(), ['normalize', 'substract_mean', 'clustering'], dict(nbr_clusters=k))
    assert np.abs(got - g19[name + '_all']).max() <= 1e-11
    # nbr_clusters missing / zero: the reference prints a note and keeps the snapshots
    assert np.array_equal(pod.process_snapshots(S.copy(), ['clustering'], dict(nbr_clusters=0)), S)


def test_lloyd_run_properties_and_empty_cluster_relocation():
    """One Lloyd run from chosen centres: fixed point of the update, labels = nearest centre, inertia; a centre far away
    from every snapshot starts with an empty cluster and is relocated to the farthest snapshot (sklearn's rule)."""
    from sofacontrol_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    X = np.concatenate([c + 0.2 * rng.standard_normal((30, 50)) for c in (np.zeros(50), 3.0 * np.ones(50))])
    X[7] += 40.0                                                  # an outlier: the farthest snapshot
    n_s, n_f, k = X.shape[0], X.shape[1], 3
    C0 = np.stack([X[0], X[40], 1e3 * np.ones(n_f)])             # third centre: nobody's nearest
    dX, dC, dl = _lib.DeviceBuffer.from_array(X), _lib.DeviceBuffer.from_array(C0), _lib.DeviceBuffer(n_s * 4)
    inertia, iters = C.c_double(), C.c_int()
    _lib.check(L.srom_kmeans_lloyd_dev(dX.ptr, C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), C.c_int(k), dC.ptr, C.c_int(100),
                                       C.c_double(0.0), dl.ptr, C.byref(inertia), C.byref(iters), None), 'lloyd')
    Cf, lab = dC.to_array((k, n_f)), dl.to_array((n_s,), dtype=np.int32)
    assert lab[7] == 2 and (lab == 2).sum() == 1 and np.abs(Cf[2] - X[7]).max() <= 1e-12      # the outlier founds cluster 2
    d = ((X[:, None, :] - Cf[None, :, :]) ** 2).sum(axis=2)
    assert np.array_equal(lab, d.argmin(axis=1))
    for j in range(k):
        np.testing.assert_allclose(Cf[j], X[lab == j].mean(axis=0), atol=1e-12)
    assert abs(inertia.value - d[np.arange(n_s), lab].sum()) <= 1e-10 * inertia.value and 2 <= iters.value <= 10


def test_kmeans_against_sklearn_on_a_larger_set():
    sk = pytest.importorskip('sklearn.cluster')
    from sofacontrol_amd.mor import pod
    rng = np.random.default_rng(11)
    X = np.concatenate([c + rng.standard_normal((60, 130)) for c in 6.0 * rng.standard_normal((5, 130))])
    ref = sk.KMeans(5, n_init=3, max_iter=1000, random_state=0).fit(X.copy()).cluster_centers_
    got = pod.compute_kmeans_centroids(X, 5, n_init=3)
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(X).max()


def test_run_pod_with_preprocessing_matches_the_reference_pipeline(tmp_path):
    """run_POD (pod.py:110-141) with config.preprocess = ['substract_mean', 'clustering']: get_snapshots -> process_snapshots
    -> compute_POD on one resident copy; the basis spans what the reference pipeline (numpy + the KMeans restatement + SVD)
    spans and the singular values agree."""
    import contextlib, io
    from oracle import pod as opod
    from sofacontrol_amd import utils as scutils
    from sofacontrol_amd.mor.pod import run_POD, pod_config
    rng = np.random.default_rng(21)
    centres = 5.0 * rng.standard_normal((6, 4)) @ rng.standard_normal((4, 70))          # rank 4 structure
    q = np.concatenate([c + 0.05 * rng.standard_normal((12, 70)) for c in centres])
    snap, podf = str(tmp_path / 's.pkl'), str(tmp_path / 'p.pkl')
    scutils.save_data(snap, {'q': list(q), 'v': list(q)})
    cfg = pod_config(); cfg.pod_type = 'v'; cfg.pod_tolerance = 1e-8
    cfg.preprocess = ['substract_mean', 'clustering']; cfg.preprocess_args = dict(nbr_clusters=6)
    with contextlib.redirect_stdout(io.StringIO()):
        res = run_POD(snap, podf, cfg)
    ref_snap = opod.process_snapshots(opod.get_snapshots({'q': list(q), 'v': list(q)}, 'v'), cfg.preprocess, cfg.preprocess_args)
    _, Uref, kref, Sref = opod.compute_pod(ref_snap.T, cfg.pod_tolerance)
    U = res['POD_info']['U']
    assert U.shape == Uref.shape == (70, kref)
    np.testing.assert_allclose(res['Sigma'][:kref], Sref[:kref], rtol=1e-8)
    assert np.abs(U @ (U.T @ Uref) - Uref).max() <= 1e-7            # same subspace
