"""Batched device discretisation (csrc/discretize.hip) against the host formulas it replaced (numpy / scipy.linalg.expm per model):
51 models of the Diamond size (n_x = 60, n_u = 4) -- one horizon of a weighting-mode TPWL linearisation -- and the 100 stored points
of a pre_discretize call."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'soft-robot-control_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import numpy as np
from sofacontrol_amd import _lib
from oracle import tpwl as otpwl

rng = np.random.default_rng(0)
n, m = 60, 4
h = n // 2
for batch in (51, 100):
    A = np.zeros((batch, n, n)); B = rng.standard_normal((batch, n, m)); d = rng.standard_normal((batch, n))
    for b in range(batch):
        Q, _ = np.linalg.qr(rng.standard_normal((h, h)))
        K = Q @ np.diag(np.logspace(0, 4.5, h)) @ Q.T
        A[b, :h, :h], A[b, :h, h:], A[b, h:, :h] = -(0.02 * K + 0.5 * np.eye(h)), -K, np.eye(h)
    Ad = np.empty_like(A); Bd = np.empty_like(B); dd = np.empty_like(d)
    for name, code in (('fe', 0), ('be', 1), ('bil', 2), ('zoh', 3)):
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            _lib.check(_lib.lib().stpwl_discretize(C.c_int(code), C.c_int(n), C.c_int(m), C.c_int64(batch), _lib.dptr(A), _lib.dptr(B), _lib.dptr(d),
                                                   C.c_double(0.05), _lib.dptr(Ad), _lib.dptr(Bd), _lib.dptr(dd)), 'stpwl_discretize')
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for b in range(batch):
            otpwl.discretize(A[b], B[b], d[b], 0.05, name)
        th = time.perf_counter() - t0
        print('%3d models, %-3s: device %.3f ms (host buffers, min of 5 warm calls), host formulas %.1f ms' % (batch, name, min(ts[1:]) * 1e3, th * 1e3), flush=True)
