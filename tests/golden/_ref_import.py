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
        if a and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _mod('osqp', OSQP=object)
    _mod('control')
    sofa = _mod('Sofa')
    sofa.Core = _mod('Sofa.Core', Controller=object, Node=object)
    sofa.Simulation = _mod('Sofa.Simulation')
    rclpy = _mod('rclpy')
    rclpy.node = _mod('rclpy.node', Node=object)
    srr = _mod('soft_robot_control_ros')
    srr.srv = _mod('soft_robot_control_ros.srv', GuSTOsrv=object)
    jax = _mod('jax', jit=_jit)
    jax.numpy = _mod('jax.numpy')
    jax.scipy = _mod('jax.scipy')
    cp = _mod('cvxpy')
    _mod('cvxpy.atoms')
    _mod('cvxpy.atoms.affine')
    _mod('cvxpy.atoms.affine.wraps', psd_wrap=lambda x: x)
    _mod('cvxpy.atoms.affine.reshape', reshape=lambda x, s: x)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mpl = _mod('matplotlib')
        mpl.pyplot = _mod('matplotlib.pyplot')
    return True
