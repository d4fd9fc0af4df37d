"""CPU: the C-ABI library loads and exports every symbol declared in include/sofacontrol_hip.h; the host
mirror keeps the reference's class / method surface; calling compute without a GPU fails loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'sofacontrol_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'^\s*(?:const\s+char\s*\*|int|void)\s+\**\s*([a-z]+_[a-z0-9_]+)\s*\(', src, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from sofacontrol_amd import _lib
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.srh_version() >= 100


def test_no_cpu_fallback_compute_fails_loudly_without_gpu():
    from sofacontrol_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('GPU present')
    from sofacontrol_amd.mor.pod import POD
    U = np.linalg.qr(np.random.default_rng(0).standard_normal((30, 3)))[0]
    with pytest.raises((RuntimeError, _lib.HipError)):
        POD(dict(U=U, q_ref=np.zeros(30), v_ref=np.zeros(30)))


def test_host_mirror_surface_matches_reference_names():
    import sofacontrol_amd.mor.pod as pod
    import sofacontrol_amd.tpwl.tpwl as tpwl
    import sofacontrol_amd.scp.gusto as gusto
    import sofacontrol_amd.scp.locp as locp
    import sofacontrol_amd.scp.standalone as sa
    import sofacontrol_amd.scp.models.tpwl as mt
    import sofacontrol_amd.lqr.ilqr as ilqr
    import sofacontrol_amd.lqr.lqr as lqr
    import sofacontrol_amd.lqr.traj_tracking_lqr as tt
    import sofacontrol_amd.utils as utils
    import sofacontrol_amd.SSM.controllers as sctl
    import sofacontrol_amd.scp.models.dubins_car as dub
    assert hasattr(dub, 'DubinsCar')
    import sofacontrol_amd.tpwl.controllers as tctl
    for mod, names in [(sctl, ['TemplateController', 'scp', 'SSMObserver']),
                       (tctl, ['TemplateController', 'scp', 'ilqr', 'TrajTracking', 'StateDLQR', 'GuSTOClient']),
                       (pod, ['POD', 'pod_config', 'load_POD', 'run_POD', 'get_snapshots', 'process_snapshots', 'compute_POD']),
                       (tpwl, ['TPWL', 'TPWLATV']), (gusto, ['GuSTO']), (locp, ['LOCP']),
                       (sa, ['runGuSTOSolverStandAlone', 'GuSTOSolverNode']), (mt, ['TPWLGuSTO']),
                       (ilqr, ['iLQR']), (lqr, ['solve_riccati', 'dare', 'DLQR']), (tt, ['TrajTrackingLQR']),
                       (utils, ['QuadraticCost', 'qv2x', 'x2qv', 'Polyhedron', 'HyperRectangle', 'arr2np', 'np2arr'])]:
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)
    for m in ['compute_RO_state', 'compute_FO_state', 'compute_RO_matrix', 'get_info']:
        assert hasattr(pod.POD, m)
    for m in ['get_jacobians', 'update_state', 'update_dynamics', 'rollout', 'pre_discretize', 'discretize_dynamics',
              'calc_nearest_point', 'x_to_zfyf', 'zfyf_to_zy', 'get_characteristic_dx']:
        assert hasattr(tpwl.TPWLATV, m)
    for m in ['solve', 'get_solution', 'is_converged', 'is_in_trust_region', 'compute_accuracy', 'state_constraints_violated']:
        assert hasattr(gusto.GuSTO, m)
    for m in ['update', 'solve', 'get_solution']:
        assert hasattr(locp.LOCP, m)
    hr = utils.HyperRectangle([2., 3.], [-1., 0.])
    np.testing.assert_array_equal(hr.A, [[1, 0], [-1, 0], [0, 1], [0, -1]])
    np.testing.assert_array_equal(hr.b, [2., 1., 3., 0.])
    assert hr.contains(np.array([0., 1.])) and not hr.contains(np.array([3., 1.]))
    np.testing.assert_allclose(hr.get_constraint_violation(np.array([3., -1.])), np.sqrt(2.0))


def test_measurement_models_match_reference_selector(golden):
    """linearModel (measurement_models.py:7-44) builds the same selector as the reference (golden g3 'Hf' =
    linearModel(nodes=[10], num_nodes=20).C); pure host data format, no GPU needed."""
    import numpy as np
    from sofacontrol_amd.measurement_models import linearModel, MeasurementModel, buildCq, buildCv
    g = golden('g3_tpwl')
    m = linearModel(nodes=[10], num_nodes=20)
    np.testing.assert_array_equal(m.C.toarray(), g['Hf'])
    x = np.arange(120.0)
    np.testing.assert_array_equal(m.evaluate(x), g['Hf'] @ x)
    assert buildCq([1, 3], 5).shape == (6, 30) and buildCv([2], 5).toarray()[1, 7] == 1.0
    mm = MeasurementModel([0, 4], 5, pos=True, vel=False)
    assert mm.C.shape == (6, 30) and mm.mean.shape == (6,) and not mm.covariance.any()
    np.testing.assert_array_equal(mm.evaluate(np.arange(30.0)), mm.C @ np.arange(30.0))
