"""Device eigensolver (srom_eigh_dev): time and residuals over n (LDS Jacobi <= 128, HBM Jacobi <= 2048, rocSOLVER above)."""
import sys, time
import numpy as np
sys.path.insert(0, 'soft-robot-control_amd'); sys.path.insert(0, '.')
from sofacontrol_amd.mor.pod import _device_eigh
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 129, 300, 601, 1000, 2048]:
    rng = np.random.default_rng(n)
    S = rng.standard_normal((n, n + 3)) * np.logspace(0, -3, n + 3)
    G = S @ S.T
    _device_eigh(G[:4, :4])
    t0 = time.perf_counter(); w, W = _device_eigh(G); t = time.perf_counter() - t0
    t0 = time.perf_counter(); we = np.linalg.eigvalsh(G); tn = time.perf_counter() - t0
    sc = max(1.0, np.abs(we).max())
    print('n=%5d: %8.1f ms (numpy eigvalsh %7.1f ms)  |w-we| %.1e  |W^T W - I| %.1e  |G W - W w| %.1e' % (
        n, t * 1e3, tn * 1e3, np.abs(w - we).max() / sc, np.abs(W.T @ W - np.eye(n)).max(), np.abs(G @ W - W * w).max() / sc))
