"""Multi-GPU paths of the hot path (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

* POD build (BASELINE config C4): the snapshot matrix S (n_s x n_f) is sharded by DoF COLUMNS; each rank
  forms the partial Gramian S_g S_g^T with the f64 MFMA kernel, ONE all-reduce (sum) of the n_s x n_s
  Gramian is the only collective; the eigen-decomposition is replicated and every rank recovers its own
  rows of U (U stays row-sharded for later projections).
* Batched SCP rollouts (C5) and POD projection batches shard by independent rows: no data-path collective
  (`shard_range`); a sharded rollout batch ends with ONE all_gather of the per-rollout optimal costs
  (`gather_rollout_costs`: every rank learns the global best rollout).
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items independent units for `rank` of `world`."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rollout_costs(J_local, n_total=None, group=None):
    """The reduction step of a batch of SCP rollouts sharded over the ranks ("final all_gather of the
    costs / pick the best"; the solves themselves need no collective).  J_local: this rank's per-rollout optimal costs
    (`GuSTO.costs`; +inf / nan for a rollout without an accepted step) in the order of its `shard_range(n_total, rank,
    world)` slice.  Returns (J_all (n_total,), best): every rank gets the costs of ALL rollouts in global order and the
    global index of the cheapest one (first index on ties, nan never wins; -1 when no rollout has a finite cost).
    One all_gather of at most ceil(n_total / world) doubles per rank -- RCCL on device tensors with the nccl backend
    (pass a CUDA tensor), gloo on host memory.  Without a process group (or world 1) it is the local argmin."""
    try:
        import torch
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    except ImportError:
        multi = False
    if not multi:
        J_all = np.asarray(J_local.detach().cpu() if hasattr(J_local, 'detach') else J_local, dtype=np.float64).ravel()
        if n_total is not None and J_all.size != int(n_total):
            raise ValueError('gather_rollout_costs: %d local costs for %d rollouts on a single rank' % (J_all.size, int(n_total)))
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        on_device = torch.is_tensor(J_local) and J_local.is_cuda
        Jt = J_local.to(torch.float64).reshape(-1) if torch.is_tensor(J_local) else torch.from_numpy(np.ascontiguousarray(J_local, dtype=np.float64).ravel())
        if n_total is None:                          # equal shards assumed unless told otherwise: agree on the total
            cnt = torch.tensor([Jt.numel()], dtype=torch.int64, device=Jt.device)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
            n_total = int(cnt.item())
        lo, hi = shard_range(n_total, rank, world)
        # a wrong shard size on ONE rank must fail on EVERY rank: agree on validity before the data collective (a rank that
        # raised alone would leave the others waiting in all_gather until the backend's timeout)
        ok = torch.tensor([1 if Jt.numel() == hi - lo else 0], dtype=torch.int64, device=Jt.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise ValueError('gather_rollout_costs: rank %d holds %d costs, its shard of %d rollouts has %d (a shard size is wrong on at least '
                             'one rank)' % (rank, Jt.numel(), n_total, hi - lo))
        width = -(-int(n_total) // world)            # shards differ by at most one: pad to the widest
        mine = torch.full((width,), float('inf'), dtype=torch.float64, device=Jt.device)
        mine[:hi - lo] = Jt
        allv = torch.empty((world * width,), dtype=torch.float64, device=Jt.device)
        dist.all_gather_into_tensor(allv, mine, group=group)
        allv = allv.cpu().numpy() if on_device else allv.numpy()
        J_all = np.concatenate([allv[r * width:r * width + (shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0])]
                                for r in range(world)])
    key = np.where(np.isfinite(J_all), J_all, np.inf)
    best = int(np.argmin(key)) if key.size and np.isfinite(key).any() else -1
    return J_all, best


class _HostSteps:
    """numpy stand-ins for the four device steps -- the CPU (gloo) tests inject these to run the sharding and exchange
    logic of `pod_from_column_shards` without a GPU (no product path uses it)."""

    def __init__(self, gramian, eigh, modes):
        self._gramian, self._eigh, self._modes = gramian, eigh, modes

    def upload(self, S):
        return np.ascontiguousarray(S, dtype=np.float64)

    def gramian(self, S):
        import torch
        return torch.from_numpy(np.ascontiguousarray(self._gramian(S)))          # reduced in place by torch.distributed

    def eigenvalues_descending(self, G):
        w, W = self._eigh(G.numpy())
        self._W = W[:, ::-1]
        return np.maximum(w[::-1], 0.0)

    def leading(self, G, k):
        """(k largest eigenvalues, trace) -- stand-in for srom_eigh_topk_dev."""
        Gh = G.numpy()
        w, W = self._eigh(Gh)
        self._W = W[:, ::-1]
        return np.maximum(w[::-1][:k], 0.0), float(np.trace(Gh))

    def modes(self, S, G, k, sigma):
        return self._modes(S, np.ascontiguousarray(self._W[:, :k] / sigma[:k]))

    def sync(self):
        pass


class _DeviceSteps:
    """The same four steps on the GPU with everything resident in HBM: the snapshot shard is uploaded once, the
    eigen-decomposition overwrites the Gramian, and only the n_s eigenvalues (for the energy truncation) and the local
    mode rows come back to the host.  With a process group the buffers are torch CUDA tensors, so that RCCL reduces the
    Gramian in place; a single rank uses the library's own allocations and needs no torch at all.  (Torch brings its own
    HIP runtime: a process that uses both must initialise torch.cuda BEFORE the first call into libsofacontrol_hip --
    bench.py and tests/conftest.py do.)"""

    def __init__(self, use_torch):
        from . import _lib
        self._lib, self.L, self.use_torch = _lib, _lib.lib(), use_torch
        if use_torch:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError('pod_from_column_shards: torch sees no HIP GPU -- in a process that uses both, torch.cuda must be '
                                   'initialised (torch.cuda.init()) BEFORE the first call into libsofacontrol_hip')
            self.torch = torch

    def _alloc(self, shape):
        if self.use_torch:
            return self.torch.empty(shape, dtype=self.torch.float64, device='cuda')
        return self._lib.DeviceBuffer(int(np.prod(shape)) * 8)

    def _ptr(self, a):
        import ctypes as C
        return C.c_void_p(a.data_ptr()) if self.use_torch else a.ptr

    def upload(self, S):
        self.shape = tuple(S.shape)
        if self.use_torch:
            if self.torch.is_tensor(S):
                assert S.is_cuda and S.dtype == self.torch.float64 and S.is_contiguous()
                return S
            return self.torch.from_numpy(np.ascontiguousarray(S, dtype=np.float64)).cuda()
        if not isinstance(S, np.ndarray):           # a resident torch tensor on a single rank: use its memory as it is
            self.use_torch, self.torch = True, __import__('torch')
            return self.upload(S)
        return self._lib.DeviceBuffer.from_array(np.ascontiguousarray(S, dtype=np.float64))

    def gramian(self, S):
        import ctypes as C
        n_s, n_f = self.shape
        G = self._alloc((n_s, n_s))
        self.sync()
        self._lib.check(self.L.srom_gramian_dev(self._ptr(S), C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), self._ptr(G), None),
                        'srom_gramian_dev')
        self._lib.sync()
        return G

    def eigenvalues_descending(self, G):
        import ctypes as C
        n_s = self.shape[0]
        self._w = self._alloc((n_s,))
        self.sync()
        self._lib.check(self.L.srom_eigh_dev(self._ptr(G), C.c_int64(n_s), self._ptr(self._w), None), 'srom_eigh_dev')
        self._lib.sync()                                                 # rows of G are now the eigenvectors (ascending)
        w = self._w.cpu().numpy() if self.use_torch else self._w.to_array((n_s,))
        return np.maximum(w[::-1], 0.0)

    def leading(self, G, k):
        """The k largest eigenpairs of the (reduced) Gramian by blocked subspace iteration (srom_eigh_topk_dev): returns
        (eigenvalues descending (k,), trace(G)); the scaled eigenvector columns W_k stay on the device for `modes`."""
        import ctypes as C
        n_s = self.shape[0]
        self._wk, self._Wk = self._alloc((k,)), self._alloc((n_s, k))
        tr, it = C.c_double(0.0), C.c_int(0)
        self.sync()
        self._lib.check(self.L.srom_eigh_topk_dev(self._ptr(G), C.c_int64(n_s), C.c_int(k), C.c_int(-1), self._ptr(self._wk), self._ptr(self._Wk),
                                                  None, C.byref(tr), C.byref(it), None), 'srom_eigh_topk_dev')
        self._lib.sync()
        self.subspace_iterations = it.value
        w = self._wk.cpu().numpy() if self.use_torch else self._wk.to_array((k,))
        return np.maximum(w, 0.0), tr.value

    def modes(self, S, G, k, sigma):
        import ctypes as C
        n_s, n_f = self.shape
        if getattr(self, '_Wk', None) is not None:          # leading eigenpairs: W_k is there already
            U = self._alloc((n_f, k))
            self.sync()
            self._lib.check(self.L.srom_modes_dev(self._ptr(S), C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), self._ptr(self._Wk), C.c_int(k),
                                                  self._ptr(U), None), 'srom_modes_dev')
            self._lib.sync()
            self._k = k
            return U
        Wk, U = self._alloc((n_s, k)), self._alloc((n_f, k))
        self.sync()
        self._lib.check(self.L.srom_select_modes_dev(self._ptr(G), self._ptr(self._w), C.c_int64(n_s), C.c_int(k), self._ptr(Wk), None),
                        'srom_select_modes_dev')
        self._lib.check(self.L.srom_modes_dev(self._ptr(S), C.c_int64(n_s), C.c_int64(n_f), C.c_int64(n_f), self._ptr(Wk), C.c_int(k),
                                              self._ptr(U), None), 'srom_modes_dev')
        self._lib.sync()
        self._k = k
        return U

    def to_host(self, U):
        return U.cpu().numpy() if self.use_torch else U.to_array((self.shape[1], self._k))

    def sync(self):
        if self.use_torch:
            self.torch.cuda.synchronize()
        else:
            self._lib.sync()


RANK_FLOOR = 1e-7       # relative singular value below which a Gramian-derived mode is noise (see _usable_modes)


def _usable_modes(Sigma, k):
    """rom_dim larger than the numerical rank of the METHOD OF SNAPSHOTS: the spectrum comes from the Gramian, whose eigenvalues
    carry an absolute error of ~1e-16 lambda_max, so a singular value below ~1e-8 sigma_0 is rounding noise and
    U = S^T W Sigma^-1 for it is neither accurate nor orthonormal (the reference's SVD, pod.py:190-200, still returns orthonormal
    vectors there: a documented deviation).  Refuse with a clear message instead of returning such modes."""
    Sigma = np.asarray(Sigma)
    if k > len(Sigma) or Sigma[k - 1] <= RANK_FLOOR * max(Sigma[0], 1e-300):
        rank = int((Sigma > RANK_FLOOR * max(Sigma[0], 1e-300)).sum())
        raise RuntimeError('rom_dim = %d exceeds the numerical rank %d of the snapshot matrix (singular value %d is %.3e of the largest)'
                           % (k, rank, k, (Sigma[k - 1] / Sigma[0]) if k <= len(Sigma) and Sigma[0] > 0 else 0.0))
    return k


def reduce_gramian(G, group=None, collective='auto'):
    """The one exchange step of the path: sum the partial Gramians over the ranks, in place on the tensor G (HBM with
    the nccl = RCCL backend, host memory with gloo).  'rs_ag' = reduce-scatter of row blocks + all-gather (keeps all
    xGMI links of a node busy; needs n_s divisible by the world size), 'all_reduce' = the plain collective; 'auto' picks
    rs_ag when it applies."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 'none'
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 'none'
    world = dist.get_world_size(group)
    can_rs = G.shape[0] % world == 0
    if collective == 'rs_ag' and not can_rs:
        raise RuntimeError('rs_ag needs n_s divisible by the world size')
    if collective == 'rs_ag' or (collective == 'auto' and can_rs):
        rows = G.shape[0] // world
        mine = G.new_empty((rows, G.shape[1]))
        dist.reduce_scatter_tensor(mine, G, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(G, mine, group=group)
        return 'rs_ag'
    dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    return 'all_reduce'


LEADING_MIN_SNAPSHOTS = 2048      # above this many snapshots the full spectrum is a library call (rocSOLVER dsyevd, ~1 s at 10 000)
LEADING_MAX_MODES = 112           # block of the subspace iteration <= 128 including its oversampling
LEADING_MAX_ITERATIONS = 30       # srom_eigh_topk_dev reports more than this many when the block did not settle: full spectrum then
LEADING_MIN_TOL = 1e-12           # below this the tail energy trace - sum(w) is cancellation noise: full spectrum


def pod_from_column_shards(S_shard, tol, group=None, rom_dim=None, local_gramian=None, local_modes=None,
                           local_eigh=None, collective='auto', timings=None, keep_on_device=False, force_torch=False,
                           spectrum='auto'):
    """Distributed method-of-snapshots POD (sofacontrol/mor/pod.py:181-200 for a snapshot matrix sharded by DoF columns).

    S_shard: this rank's (n_s x n_f_local) block of snapshot columns -- a numpy array (uploaded once) or a CUDA float64
    torch tensor (used in place).  Steps: local Gramian (f64 MFMA kernel) -> `reduce_gramian` (the only collective) ->
    replicated eigen-decomposition on the device -> this rank's rows of U = S^T W Sigma^-1.  The Gramian never visits the
    host.  Returns (U_local (n_f_local x k), k, Sigma); U_local is a numpy array unless keep_on_device.
    spectrum: 'full' = every eigenvalue of the Gramian (what the reference's SVD returns as `Sigma`); 'leading' = only the
    kept modes by blocked subspace iteration, the truncation rule evaluated as (trace(G) - sum of the leading eigenvalues) /
    trace(G) (pod.py:192-197 needs nothing else), `Sigma` then holds the computed leading singular values only; 'auto' =
    'leading' for more than 2048 snapshots when at most 112 modes are asked for (rom_dim) or suffice (tol), else 'full'.
    `local_gramian` / `local_eigh` / `local_modes`: numpy stand-ins for the device steps (CPU tests of the exchange
    logic).  `timings`: optional dict that receives the seconds of each phase.  keep_on_device: return the device
    array (a torch tensor with a process group or force_torch, else a _lib.DeviceBuffer)."""
    import time
    from .mor import pod as _pod
    host = local_gramian is not None or local_modes is not None or local_eigh is not None
    if host:
        steps = _HostSteps(local_gramian or (lambda A: A @ A.T), local_eigh or np.linalg.eigh, local_modes or (lambda A, W: A.T @ W))
    else:
        try:
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        except ImportError:                  # a single rank needs no torch at all
            multi = False
        steps = _DeviceSteps(use_torch=multi or force_torch)

    def lap(name, t0):
        steps.sync()
        if timings is not None:
            timings[name] = time.perf_counter() - t0
        return time.perf_counter()
    t0 = time.perf_counter()
    S = steps.upload(S_shard)
    t0 = lap('upload_s', t0)
    G = steps.gramian(S)
    t0 = lap('gramian_s', t0)
    how = reduce_gramian(G, group, collective)
    t0 = lap('collective_s', t0)
    if timings is not None:
        timings['collective'] = how
    n_s = int(np.shape(S_shard)[0])
    lead = spectrum == 'leading' or (spectrum == 'auto' and n_s > LEADING_MIN_SNAPSHOTS and (rom_dim is None or int(rom_dim) <= LEADING_MAX_MODES))
    if lead and rom_dim is None and tol < LEADING_MIN_TOL:
        if spectrum == 'leading':
            raise ValueError('pod_from_column_shards: spectrum="leading" evaluates the truncation rule as trace(G) - sum of the leading '
                             'eigenvalues, which is rounding noise for tol < %g; use spectrum="full"' % LEADING_MIN_TOL)
        lead = False
    k = None
    settled = lambda: getattr(steps, 'subspace_iterations', 0) <= LEADING_MAX_ITERATIONS
    if lead:
        # leading eigenpairs only: with rom_dim that many; with a tolerance blocks of growing size until the tail energy fits
        for kk in ([int(rom_dim)] if rom_dim is not None else [32, 64, LEADING_MAX_MODES]):
            kk = min(kk, n_s)
            w, trace = steps.leading(G, kk)
            Sigma = np.sqrt(w)
            if not settled():
                break                                                   # a flat spectrum: the iteration has not settled -> full spectrum
            if rom_dim is not None:
                k = kk
                break
            tail = np.maximum(trace - np.cumsum(w), 0.0) / trace        # tail[i]: energy beyond the first i + 1 modes
            ok = np.nonzero(tail <= tol)[0]
            if ok.size and ok[0] + 1 < kk:                              # (the last value of a block is not trusted)
                k = int(ok[0]) + 1
                # W_k in the layout of exactly k modes: a second run with a smaller block (k + oversampling), i.e. a smaller
                # gap behind it -- if THAT run does not settle its Ritz basis must not be used: full spectrum instead
                w, trace = steps.leading(G, k)
                Sigma = np.sqrt(w)
                if not settled():
                    k = None
                break
        if timings is not None:
            timings['spectrum'] = 'leading' if k is not None else 'full (leading blocks did not reach the tolerance)'
            timings['subspace_iterations'] = getattr(steps, 'subspace_iterations', None)
    if k is None:
        if hasattr(steps, '_Wk'):
            steps._Wk = None
        w = steps.eigenvalues_descending(G)
        Sigma = np.sqrt(w)
        k = _pod.energy_truncation(Sigma, tol) if rom_dim is None else int(rom_dim)
        if timings is not None and 'spectrum' not in timings:
            timings['spectrum'] = 'full'
    t0 = lap('eigh_s', t0)
    k = _usable_modes(Sigma, k)
    U_local = steps.modes(S, G, k, Sigma)
    lap('modes_s', t0)
    if not host and not keep_on_device:
        U_local = steps.to_host(U_local)
    return U_local, k, Sigma
