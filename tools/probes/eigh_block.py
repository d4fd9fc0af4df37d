"""Block Jacobi (csrc/eigh.hip: jacobi_block) against numpy / rocSOLVER: accuracy and time.  usage: eigh_block.py n [n ...] [--roc]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'soft-robot-control_amd'))
import numpy as np
from sofacontrol_amd.mor.pod import _device_eigh

roc = '--roc' in sys.argv
for n in [int(a) for a in sys.argv[1:] if not a.startswith('--')]:
    rng = np.random.default_rng(n)
    k = min(n + 3, 4000)
    S = rng.standard_normal((n, k)) * np.logspace(0, -3, k)
    G = S @ S.T
    t0 = time.perf_counter(); w, W = _device_eigh(G); t1 = time.perf_counter()
    t2 = time.perf_counter(); w2, W2 = _device_eigh(G); t3 = time.perf_counter()
    we = np.linalg.eigvalsh(G) if n <= 6000 else None
    res = np.abs(G @ W[:, -16:] - W[:, -16:] * w[-16:]).max() / np.abs(w).max()
    orth = np.abs(W[:, -64:].T @ W[:, -64:] - np.eye(64)).max() if n >= 64 else 0.0
    print('n %5d: block Jacobi %.3f s (first call %.3f s); |w - numpy| / w_max %s; residual of the 16 leading pairs %.2e; orthogonality %.2e' %
          (n, t3 - t2, t1 - t0, 'n/a' if we is None else '%.2e' % (np.abs(w - we).max() / np.abs(we).max()), res, orth), flush=True)
    if roc:
        os.environ['SRH_EIGH_ROCSOLVER'] = '1'
        t0 = time.perf_counter(); wr, Wr = _device_eigh(G); t1 = time.perf_counter()
        t2 = time.perf_counter(); wr, Wr = _device_eigh(G); t3 = time.perf_counter()
        del os.environ['SRH_EIGH_ROCSOLVER']
        print('         rocSOLVER dsyevd %.3f s (first call %.3f s); |w - w_roc| / w_max %.2e' % (t3 - t2, t1 - t0, np.abs(w - wr).max() / np.abs(wr).max()), flush=True)
