"""Data-parallel plumbing: one process per GPU, frames sharded, two small all-reduces per step.

The loss of ``EigenFunctionTask`` is a rational function of batch sums,
so a global batch of ``B_g`` frames is split into ``world`` contiguous local slices; every rank
reduces its slice to the statistics vector, the vectors are summed across ranks (collective #1,
fp64, <= 70 doubles), every rank evaluates the identical scalar tail and back-propagates its
slice with the *global* coefficients, and the flat parameter gradients are summed (collective
#2, fp32, P floats) before the identical fused Adam update.  With ``backend='nccl'`` these are
RCCL all-reduces over xGMI; the same code runs over ``gloo`` on CPU tensors in the tests.
"""

import os

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def collectives():
    """True when the step must run its two all-reduces: more than one rank, or CVF_FORCE_COLLECTIVES=1 in an
    initialised (possibly one-rank) group - the latter exercises the data-parallel code path, including its hipGraph
    capture with the RCCL kernels inside, on a single GPU."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CVF_FORCE_COLLECTIVES", "0") == "1"


def backend():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run)."""
    forced = os.environ.get("CVF_FORCE_COLLECTIVES", "0") == "1" and "MASTER_PORT" in os.environ
    if world() > 1 or (int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not forced) or (dist.is_available() and dist.is_initialized()):
        return
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend)


def local_slice(n_global, r=None, w=None):
    """Contiguous block of ``n_global`` items owned by rank ``r`` (sizes differ by at most one)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, rem = divmod(n_global, w)
    start = r * base + min(r, rem)
    return start, start + base + (1 if r < rem else 0)


def shard_batches(n, bs, r=None, w=None):
    """The static batches of a permuted set of ``n`` items - ``DataLoader(batch_size=bs, drop_last=True, shuffle=False)``,
    core.py:470-481 - split over the ranks: global batch ``j`` is items ``[j bs, (j+1) bs)`` and rank ``r`` owns the
    contiguous slice ``local_slice(bs, r, w)`` of every batch.

    Returns ``(pos, nb)``: ``pos`` = the positions (into the permuted set) this rank keeps RESIDENT, batch-major, and
    ``nb`` = frames per local batch, so that local batch ``j`` is resident rows ``[j nb, (j+1) nb)``.  The union over the
    ranks of batch ``j``'s rows is exactly global batch ``j``; items of the dropped tail are resident nowhere.
    """
    import numpy as np
    r = rank() if r is None else r
    w = world() if w is None else w
    n_batches = n // bs if bs > 0 else 0
    a, b = local_slice(bs, r, w)
    if n_batches == 0:
        return np.zeros(0, dtype=np.int64), b - a
    pos = (np.arange(n_batches, dtype=np.int64)[:, None] * bs + np.arange(a, b, dtype=np.int64)[None, :]).reshape(-1)
    return pos, b - a


_abi_comm = None   # CVF_COMM=abi: the communicator of include/cvf.h's cvf_comm_* (RCCL behind the C ABI)


def _abi():
    """With ``CVF_COMM=abi`` the two sums go through ``cvf_comm_allreduce_*`` of the C ABI (the same RCCL all-reduce, issued on
    the current stream by the library instead of by torch.distributed) - the path a host without PyTorch would take; the
    unique id travels from rank 0 over the torch process group.  Default: torch.distributed."""
    global _abi_comm
    if os.environ.get("CVF_COMM", "") != "abi" or not torch.cuda.is_available():
        return None
    if _abi_comm is None:
        import ctypes
        from . import _hip
        lib = _hip.lib()
        nbytes = lib.cvf_comm_unique_id_bytes()
        uid = torch.zeros(nbytes, dtype=torch.uint8)
        if rank() == 0:
            _hip.check(lib.cvf_comm_unique_id(uid.data_ptr()), "cvf_comm_unique_id")
        if world() > 1:
            dev_uid = uid.cuda() if backend() == "nccl" else uid
            dist.broadcast(dev_uid, src=0)
            uid = dev_uid.cpu()
        handle = ctypes.c_void_p()
        _hip.check(lib.cvf_comm_init(ctypes.byref(handle), rank(), world(), uid.data_ptr()), "cvf_comm_init")
        _abi_comm = handle
    return _abi_comm


_p2p_comm = None   # CVF_COMM=p2p: the one-shot peer-to-peer reduce of include/cvf.h's cvf_p2p_* (csrc/p2p.hip)
_P2P_MAX_BYTES = int(os.environ.get("CVF_P2P_MAX_BYTES", str(1 << 20)))


def _p2p():
    """With ``CVF_COMM=p2p`` the two sums go through ``cvf_p2p_allreduce_*``: every rank writes its vector into a slot of every
    peer's IPC-shared window and adds the slots in rank order (one xGMI hop instead of a ring's 2 (N - 1),
    bitwise the same sum on every rank).  The IPC handles travel over the torch process group, once."""
    global _p2p_comm
    if os.environ.get("CVF_COMM", "") != "p2p" or not torch.cuda.is_available():
        return None
    if _p2p_comm is None:
        import ctypes
        from . import _hip
        lib = _hip.lib()
        nb = lib.cvf_p2p_handle_bytes()
        mine = torch.zeros(nb, dtype=torch.uint8)
        handle = ctypes.c_void_p()
        _hip.check(lib.cvf_p2p_create(ctypes.byref(handle), rank(), world(), _P2P_MAX_BYTES, mine.data_ptr()), "cvf_p2p_create")
        if world() > 1:
            on_dev = backend() == "nccl"
            src = mine.cuda() if on_dev else mine
            parts = [torch.zeros_like(src) for _ in range(world())]
            dist.all_gather(parts, src)
            allh = torch.cat([p_.cpu() for p_ in parts]).contiguous()
        else:
            allh = mine
        _hip.check(lib.cvf_p2p_connect(handle, allh.data_ptr()), "cvf_p2p_connect")
        if world() > 1:
            dist.barrier()      # every rank has opened every window before the first kernel writes into one
        _p2p_comm = handle
    return _p2p_comm


def p2p_error():
    """0, or the number of the all-reduce in which a peer did not arrive (``CVF_COMM=p2p`` only; synchronises the device)."""
    if _p2p_comm is None:
        return 0
    from . import _hip
    return int(_hip.lib().cvf_p2p_error(_p2p_comm))


def allreduce_sum_(t):
    """In-place sum over ranks; a no-op in a single-process run."""
    if collectives():
        plain = t.is_cuda and t.dtype in (torch.float32, torch.float64) and t.is_contiguous()
        p2p = _p2p() if plain and t.numel() * t.element_size() <= _P2P_MAX_BYTES else None
        if p2p is not None:
            from . import _hip
            fn = _hip.lib().cvf_p2p_allreduce_f64 if t.dtype == torch.float64 else _hip.lib().cvf_p2p_allreduce_f32
            _hip.check(fn(p2p, _hip.ptr(t), t.numel(), _hip.stream()), "cvf_p2p_allreduce")
            return t
        comm = _abi() if plain else None
        if comm is not None:
            from . import _hip
            fn = _hip.lib().cvf_comm_allreduce_f64 if t.dtype == torch.float64 else _hip.lib().cvf_comm_allreduce_f32
            _hip.check(fn(comm, _hip.ptr(t), t.numel(), _hip.stream()), "cvf_comm_allreduce")
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t, src=0):
    if world() > 1:
        dist.broadcast(t, src=src)
    return t
