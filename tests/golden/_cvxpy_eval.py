"""An EVALUATING stand-in for the handful of cvxpy symbols sofacontrol/scp/locp.py uses (build container only).

cvxpy / OSQP are absent and cannot be installed, so the reference's `LOCP` cannot *solve* here -- but its objective
(locp.py:218-263) and constraint (locp.py:265-342) code can still be *executed*: this module provides `Variable`,
`Parameter`, `quad_form`, `sum`, `norm`, `norm2`, `reshape`, `multiply`, `bmat`, `Minimize`, `Problem` and the
arithmetic / comparison operators as lazy expression nodes whose `.value` is computed with numpy from the current
`.value` of the variables and parameters.  With it the golden generator instantiates the reference class, calls its
`update(...)`, assigns a point to (x, u, st) and records J(w) and every constraint's residual -- outputs of the
reference's own statement of the QP, which tests/test_oracle_golden.py holds `oracle.locp.build_qp` to.

Semantics follow cvxpy's documented behaviour for the constructs used: `reshape` is column-major (order='F'),
`norm(X, 'inf', axis=0)` is the column-wise max-abs, `quad_form(x, P)` = x^T P x, `*` with a scalar parameter scales,
`@` is the matrix product, comparisons build constraint objects (`lhs <= rhs`, `lhs == rhs`, `lhs >= rhs`).
`Problem.solve` is deliberately absent.
"""
import numpy as np
import scipy.sparse as sp


def _val(a):
    if isinstance(a, Expr):
        return a.value
    if sp.issparse(a):
        return a.toarray()
    return np.asarray(a, dtype=float)


class Expr:
    """Operator surface shared by computed nodes (`Node`) and leaves (`Variable`, `Parameter`).  The two are SIBLING
    classes on purpose: if one were a subclass of the other, python would try the reflected comparison of the
    right-hand operand first and `a == b` would come out as Constraint(b, a)."""
    __array_ufunc__ = None          # numpy defers `ndarray (op) Expr` to the reflected operators below

    @property
    def T(self):
        return Node(lambda a: a.T, self)

    def __getitem__(self, key):
        return Node(lambda a: a[key], self)

    def __neg__(self):
        return Node(lambda a: -a, self)

    def __add__(self, o):
        return Node(lambda a, b: a + b, self, o)

    __radd__ = __add__

    def __sub__(self, o):
        return Node(lambda a, b: a - b, self, o)

    def __rsub__(self, o):
        return Node(lambda a, b: b - a, self, o)

    def __mul__(self, o):
        return Node(lambda a, b: a * b, self, o)

    __rmul__ = __mul__

    def __matmul__(self, o):
        return Node(lambda a, b: a @ b, self, o)

    def __rmatmul__(self, o):
        return Node(lambda a, b: b @ a, self, o)

    def __le__(self, o):
        return Constraint('<=', self, o)

    def __ge__(self, o):
        return Constraint('>=', self, o)

    def __eq__(self, o):
        return Constraint('==', self, o)

    __hash__ = object.__hash__


class Node(Expr):
    def __init__(self, fn, *kids):
        self._fn, self._kids = fn, kids

    @property
    def value(self):
        return self._fn(*[_val(k) for k in self._kids])


class Leaf(Expr):
    def __init__(self, shape=(), **attrs):
        self.shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        self.attrs = attrs
        self._value = None

    @property
    def value(self):
        if self._value is None:
            raise ValueError('no value assigned to %s of shape %r' % (type(self).__name__, self.shape))
        return self._value

    @value.setter
    def value(self, v):
        if v is None:               # cvxpy lets a value be cleared (locp.py:135 with xk=None in the MPC baselines)
            self._value = None
            return
        v = np.asarray(v, dtype=float)
        if v.shape != self.shape:
            raise ValueError('value of shape %r assigned to %s of shape %r' % (v.shape, type(self).__name__, self.shape))
        self._value = v


class Variable(Leaf):
    pass


class Parameter(Leaf):
    pass


class Constraint:
    """kind in '<=', '>=', '=='; `residual()` = lhs - rhs ('>=': rhs - lhs), i.e. feasible iff <= 0 (== 0)."""

    def __init__(self, kind, lhs, rhs):
        self.kind, self.lhs, self.rhs = kind, lhs, rhs

    def residual(self):
        l, r = _val(self.lhs), _val(self.rhs)
        out = (r - l) if self.kind == '>=' else (l - r)
        return np.atleast_1d(np.asarray(out, dtype=float)).ravel()


def quad_form(x, P):
    Pd = _val(P)
    return Node(lambda a: float(a @ Pd @ a), x)


def sum(x):            # noqa: A001  (cvxpy's name)
    return Node(lambda a: float(np.sum(a)), x)


def norm(x, p=2, axis=None):
    if p == 'inf':
        return Node(lambda a: np.max(np.abs(a), axis=axis), x)
    if p == 2 and axis is None:
        return norm2(x)
    raise NotImplementedError('norm(%r, axis=%r)' % (p, axis))


def norm2(x):
    return Node(lambda a: float(np.linalg.norm(a)), x)


def reshape(x, shape, order='F'):
    return Node(lambda a: np.reshape(a, shape, order=order), x)


def multiply(a, b):
    return Node(lambda u, v: u * v, a, b)


def bmat(blocks):
    flat = [b for row in blocks for b in row]
    ncol = len(blocks[0])

    def build(*vals):
        rows = [list(vals[i * ncol:(i + 1) * ncol]) for i in range(len(blocks))]
        return np.block(rows)
    return Node(build, *flat)


class Minimize:
    def __init__(self, expr):
        self.expr = expr

    @property
    def value(self):
        return float(_val(self.expr))


class Problem:
    def __init__(self, objective, constraints=()):
        self.objective, self.constraints = objective, list(constraints)
        self.status = None
        self.solver_stats = None


def install(sys_modules):
    """Register this module as `cvxpy` (+ the two atom sub-modules locp.py imports from)."""
    import sys
    import types
    me = sys.modules[__name__]
    sys_modules['cvxpy'] = me
    for name in ('cvxpy.atoms', 'cvxpy.atoms.affine'):
        sys_modules[name] = types.ModuleType(name)
    w = types.ModuleType('cvxpy.atoms.affine.wraps')
    w.psd_wrap = lambda x: x
    sys_modules['cvxpy.atoms.affine.wraps'] = w
    r = types.ModuleType('cvxpy.atoms.affine.reshape')
    r.reshape = reshape
    sys_modules['cvxpy.atoms.affine.reshape'] = r
