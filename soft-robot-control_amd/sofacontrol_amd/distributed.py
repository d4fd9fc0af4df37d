"""Multi-GPU paths of the hot path (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

* POD build (BASELINE config C4): the snapshot matrix S (n_s x n_f) is sharded by DoF COLUMNS; each rank
  forms the partial Gramian S_g S_g^T with the f64 MFMA kernel, ONE all-reduce (sum) of the n_s x n_s
  Gramian is the only collective; the eigen-decomposition is replicated and every rank recovers its own
  rows of U (U stays row-sharded for later projections).
* Batched SCP rollouts (C5) and POD projection batches shard by independent rows: no data-path collective
  (`shard_range`); only the bench gathers counts.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items independent units for `rank` of `world`."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _hip_local_gramian(S_shard):
    from .mor.pod import gramian
    return gramian(S_shard)


def pod_from_column_shards(S_shard, tol, group=None, rom_dim=None, local_gramian=None, local_modes=None,
                           local_eigh=None):
    """Distributed method-of-snapshots POD.

    S_shard: this rank's (n_s x n_f_local) block of snapshot columns (numpy).  Returns
    (U_local (n_f_local x k), k, Sigma).  `local_gramian` / `local_eigh` / `local_modes` default to the device; the
    CPU (gloo) tests inject numpy stand-ins to exercise the sharding / reduction logic without a GPU."""
    import torch
    import torch.distributed as dist
    from .mor import pod as _pod
    lg = local_gramian or _hip_local_gramian
    G = np.ascontiguousarray(lg(np.ascontiguousarray(S_shard, dtype=np.float64)))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        use_cuda = dist.get_backend(group) == 'nccl'
        t = torch.from_numpy(G)
        if use_cuda:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)      # the one exchange step of the path
        G = t.cpu().numpy()
    Wk, k, Sigma = _pod.modes_from_gramian(None, G, tol, rom_dim, eigh=local_eigh)
    if local_modes is not None:
        U_local = local_modes(S_shard, Wk)
    else:
        import ctypes as C
        from . import _lib
        n_s, n_fl = S_shard.shape
        dS, dW = _lib.DeviceBuffer.from_array(np.ascontiguousarray(S_shard)), _lib.DeviceBuffer.from_array(Wk)
        dU = _lib.DeviceBuffer(n_fl * k * 8)
        _lib.check(_lib.lib().srom_modes_dev(dS.ptr, C.c_int64(n_s), C.c_int64(n_fl), C.c_int64(n_fl), dW.ptr, C.c_int(k),
                                             dU.ptr, None), 'srom_modes_dev')
        _lib.sync()
        U_local = dU.to_array((n_fl, k))
    return U_local, k, Sigma
