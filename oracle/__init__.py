"""CPU oracle for the sofacontrol hot path -- TEST INFRASTRUCTURE ONLY.

A numpy/scipy (float64) restatement of the reference's arithmetic for the path
named in BASELINE.json (POD projection -> TPWL linearisation/rollout -> GuSTO/LOCP
SCP solve and iLQR/LQR Riccati solves).  Every function cites the reference
file:line it follows (paths relative to the reference repository root).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package -- as the checker, never as the product.  The product
path (`soft-robot-control_amd/`) never imports it and fails loudly when the HIP
extension is missing.

Pinning status
--------------
* pod / tpwl / lqr (iLQR, TV-LQR, Riccati): pinned against golden vectors produced
  by importing the reference itself in the build container
  (`tests/golden/make_golden.py`, fixtures under `tests/golden/*.npz`).
* gusto outer loop: pinned against the imported reference `GuSTO` class driven
  with this package's QP solver injected for `LOCP` (same fixtures script).
* locp (the QP): **parity unpinned** against the reference's solver -- the reference
  delegates the QP to cvxpy -> OSQP/GUROBI (third-party, not vendored, versions not
  pinned in requirements.txt, not installed here, no network).  The QP *data* follow
  sofacontrol/scp/locp.py:218-342 line by line; the solution is pinned by exact KKT
  solves / KKT-residual certificates and by a restatement of the published OSQP
  algorithm (Stellato et al., "OSQP: an operator splitting solver for quadratic
  programs", Math. Prog. Comp. 2020), see oracle/locp.py.
"""
