"""Oracle: discrete EKF over a prediscretised nn-TPWL model (test infrastructure only).

Restates sofacontrol/tpwl/observer.py:97-126 (DiscreteEKFObserver.predict_state / update_state).
Pinned by tests/golden/g9_ekf.npz (imported reference)."""
import numpy as np

from . import tpwl as otpwl


def predict(model, Ad, Bd, dd, x, Sigma, u, W):
    """observer.py:97-106."""
    i = otpwl.nearest_point(model, x)
    A = Ad[i]
    return A @ x + Bd[i] @ u + dd[i], A @ Sigma @ A.T + W


def update(C, y_ref, x, Sigma, y, V):
    """observer.py:108-126."""
    y = y - y_ref
    S = C @ Sigma @ C.T + V
    K = Sigma @ C.T @ np.linalg.inv(S)
    x = x + K @ (y - C @ x)
    Sigma = (np.eye(x.shape[0]) - K @ C) @ Sigma
    return x, Sigma
