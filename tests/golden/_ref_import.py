"""Import harness for the *reference* python package (only usable in the build container).

Registers empty stand-in modules for the third-party packages that are absent here
(recipe from SURVEY.md section 8c) so that `sofacontrol.*` from /root/reference imports.
Never used on the GPU box: /root/reference does not exist there.
"""
import sys
import types

import numpy as np

REF = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if not hasattr(np, 'infty'):
        np.infty = np.inf

    def _jit(*a, **k):
        if a and callable(a[0]):
            return a[0]
        return lambda f: f

    def _jacobian(f, argnums=0):
        """Stand-in for jax.jacobian on the reference's polynomial maps: complex-step differentiation
        (exact to rounding for polynomials; jax itself is absent from the build container)."""
        def run(*args):
            nums = argnums if isinstance(argnums, (tuple, list)) else (argnums,)
            outs = []
            for a in nums:
                base = np.asarray(args[a], dtype=float)
                cols = []
                for j in range(base.shape[0]):
                    pert = base.astype(complex)
                    pert[j] += 1e-30j
                    aa = list(args)
                    aa[a] = pert
                    cols.append(np.imag(np.asarray(f(*aa))) / 1e-30)
                outs.append(np.stack(cols, axis=-1))
            return tuple(outs) if isinstance(argnums, (tuple, list)) else outs[0]
        return run

    _mod('osqp', OSQP=object)
    _mod('control')
    sofa = _mod('Sofa')
    sofa.Core = _mod('Sofa.Core', Controller=object, Node=object)
    sofa.Simulation = _mod('Sofa.Simulation')
    rclpy = _mod('rclpy')
    rclpy.node = _mod('rclpy.node', Node=object)
    srr = _mod('soft_robot_control_ros')
    srr.srv = _mod('soft_robot_control_ros.srv', GuSTOsrv=object)
    import scipy.special
    jax = _mod('jax', jit=_jit, jacobian=_jacobian)
    # the reference's SSM maps only need dot / asarray / eye / linalg from jax.numpy: alias numpy
    jax.numpy = _mod('jax.numpy', dot=np.dot, asarray=np.asarray, eye=np.eye, linalg=np.linalg, ndarray=np.ndarray,
                     array=np.array, zeros=np.zeros)
    jax.scipy = _mod('jax.scipy', special=scipy.special)
    # cvxpy: an evaluating stand-in (expressions compute their numpy value), enough to EXECUTE scp/locp.py's objective
    # and constraint code at given points -- see _cvxpy_eval.py
    import _cvxpy_eval
    _cvxpy_eval.install(sys.modules)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mpl = _mod('matplotlib')
        mpl.pyplot = _mod('matplotlib.pyplot')
    return True
