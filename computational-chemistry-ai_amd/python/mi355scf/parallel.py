"""Multi-GPU plumbing: one process per GPU, `torch.distributed` over RCCL/xGMI.

The resident-ERI tile runs are dealt to ranks longest-processing-time first by bytes inside
`mi_eri_prepare(rank, nranks)` -- no ERI ever crosses xGMI.  Each Fock build ends in ONE all-reduce of ONE fused FP64
buffer ([J|K] for HF, [J|K|Vxc|N_elec|E_xc] for Kohn-Sham with the grid sharded too); D is replicated, the
purification / diagonalisation / DIIS algebra is replicated and -- being free of atomics (fixed-order partial sums) --
bit-identical on every rank, so no control-scalar broadcast is needed inside the SCF loop.
With the `gloo` backend (CPU tests; or two ranks sharing one GPU) tensors are staged through host memory.
"""
import os

import torch
import torch.distributed as dist


# collectives issued by this process (world > 1 only): lets the tests assert "one all-reduce, zero broadcasts per SCF cycle"
STATS = {"all_reduce": 0, "broadcast": 0, "all_reduce_bytes": 0}


def reset_stats():
    for k in STATS:
        STATS[k] = 0


def init(backend=None, device=None):
    """Initialise the default process group from torchrun-style env vars; returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def all_reduce_sum(t, group=None):
    """In-place sum over ranks of a (device or host) tensor; returns it."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    backend = dist.get_backend(group)
    STATS["all_reduce"] += 1
    STATS["all_reduce_bytes"] += t.numel() * t.element_size()
    if backend == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def all_reduce_fused(tensors, group=None):
    """One collective for several tensors (fused buffer), results written back in place."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    flat = torch.cat([x.reshape(-1) for x in tensors])
    all_reduce_sum(flat, group)
    off = 0
    for x in tensors:
        n = x.numel()
        x.copy_(flat[off:off + n].view_as(x))
        off += n
    return tensors


def agree_max(values, group=None):
    """Element-wise MAX over ranks of a short list of host floats (one small collective; set-up time only).  Used to make
    rank-local decisions identical on every rank: pass `-x` to agree on a minimum."""
    vals = [float(v) for v in values]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return vals
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(vals, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    STATS["agree"] = STATS.get("agree", 0) + 1
    return [float(v) for v in t.cpu()]


_ATOMICS_OFF = False


def blas_atomics_off():
    """Sharded runs keep their replicated algebra (Cholesky transforms, commutators, purification GEMMs) bit-identical across
    ranks; rocBLAS may pick kernels that accumulate with atomics (run-to-run last-bit differences) unless told not to.
    Sets `rocblas_atomics_not_allowed` on torch's rocBLAS handle of the current device.  Returns True when it took effect."""
    global _ATOMICS_OFF
    if _ATOMICS_OFF:
        return True
    try:
        import ctypes
        h = torch.cuda.current_blas_handle()
        # the handle belongs to the rocBLAS torch has loaded (its bundled copy), not to whatever "librocblas.so" resolves to
        import os
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so")
        rb = ctypes.CDLL(path if os.path.exists(path) else "librocblas.so")
        rb.rocblas_set_atomics_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _ATOMICS_OFF = rb.rocblas_set_atomics_mode(ctypes.c_void_p(h), 0) == 0   # 0 = rocblas_atomics_not_allowed
    except Exception:
        _ATOMICS_OFF = False
    return _ATOMICS_OFF


def split_range(n, rank, nranks):
    """Contiguous, exhaustive, non-overlapping split of range(n)."""
    per = (n + nranks - 1) // nranks
    return min(rank * per, n), min((rank + 1) * per, n)


def broadcast0(t, group=None):
    """Make rank 0's copy of a (device or host) tensor authoritative on every rank.  Used for the handful of
    scalars that steer control flow (energy, |g|, DIIS Gram row, SP2 traces, nuclear gradient): replicated
    FP64 work can differ in the last bit between ranks (atomic summation order), and a borderline convergence
    test must not let one rank leave a loop whose body contains a collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    STATS["broadcast"] += 1
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=0, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=0, group=group)
    return t
