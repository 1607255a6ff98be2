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
    if comm_mode() != "abi" or not torch.cuda.is_available():
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


_p2p_comm = None    # the one-shot peer-to-peer communicator of include/cvf.h's cvf_p2p_* (csrc/p2p.hip), once it is up
_p2p_state = None   # None: not decided; "on" / "off"
_P2P_MAX_BYTES = int(os.environ.get("CVF_P2P_MAX_BYTES", str(1 << 20)))


def comm_mode():
    """``CVF_COMM``: ``auto`` (default) - the peer-to-peer windows when the job runs over RCCL (``nccl`` backend: one process per
    GPU of one node) and a start-up self-test of the windows passes on every rank, else torch.distributed; ``p2p`` - the windows,
    or an error; ``rccl`` - torch.distributed's all-reduce; ``abi`` - RCCL through the C ABI's cvf_comm_*."""
    return os.environ.get("CVF_COMM", "auto")


def _agree(ok):
    """True when `ok` is true on every rank (decisions that change the launch sequence must be taken by all ranks alike)."""
    if world() <= 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


def _p2p_bring_up():
    """Create the windows, hand the IPC handles around over the torch process group, map the peers', run one all-reduce through
    them and compare it with the known answer.  Returns the communicator handle or None; every step is agreed across the ranks."""
    import ctypes
    from . import _hip
    lib = _hip.lib()
    nb = lib.cvf_p2p_handle_bytes()
    mine = torch.zeros(nb, dtype=torch.uint8)
    handle = ctypes.c_void_p()
    rc = lib.cvf_p2p_create(ctypes.byref(handle), rank(), world(), _P2P_MAX_BYTES, mine.data_ptr())
    why = _hip.lib().cvf_last_error().decode() if rc != 0 else ""
    if not _agree(rc == 0):
        return None, f"cvf_p2p_create failed on a rank ({why})"
    if world() > 1:
        on_dev = backend() == "nccl"
        src = mine.cuda() if on_dev else mine
        parts = [torch.zeros_like(src) for _ in range(world())]
        dist.all_gather(parts, src)
        allh = torch.cat([p_.cpu() for p_ in parts]).contiguous()
    else:
        allh = mine
    rc = lib.cvf_p2p_connect(handle, allh.data_ptr())
    why = _hip.lib().cvf_last_error().decode() if rc != 0 else ""
    if not _agree(rc == 0):       # (every rank has opened every window before the first kernel writes into one)
        return None, f"cvf_p2p_connect failed on a rank ({why})"
    # self-test: sum over ranks of (rank + 1) * [1, 2, .., 512] through the windows, against the closed form
    n = 512
    base = torch.arange(1, n + 1, device="cuda", dtype=torch.float32)
    v = base * float(rank() + 1)
    rc = lib.cvf_p2p_allreduce_f32(handle, _hip.ptr(v), n, _hip.stream())
    torch.cuda.synchronize()
    want = base * float(world() * (world() + 1) // 2)
    good = rc == 0 and int(lib.cvf_p2p_error(handle)) == 0 and bool(torch.equal(v, want))
    if good:   # ... and the low-latency form the fused launches use: 70 doubles, then a one-row slab summed by cvf_slab_reduce_dp
        d = (base[:70] * float(rank() + 1)).double()
        rc = lib.cvf_p2p_exchange_f64(handle, _hip.ptr(d), 70, _hip.stream())
        slab, grad = base * float(rank() + 1), torch.empty_like(base)
        rc2 = lib.cvf_slab_reduce_dp(_hip.ptr(slab), 1, n, _hip.ptr(grad), None, handle, _hip.stream())
        torch.cuda.synchronize()
        good = (rc == 0 and rc2 == 0 and int(lib.cvf_p2p_error(handle)) == 0 and bool(torch.equal(grad, want)) and
                bool(torch.equal(d, want[:70].double())))
    if not _agree(good):
        return None, "the self-test all-reduce through the windows did not give the known sum on every rank"
    return handle, ""


def _p2p():
    """The peer-to-peer communicator, or None when this job sums through torch.distributed / RCCL (see comm_mode)."""
    global _p2p_comm, _p2p_state
    mode = comm_mode()
    if mode not in ("auto", "p2p"):
        return None
    if _p2p_state is None:
        want = torch.cuda.is_available() and (mode == "p2p" or (mode == "auto" and backend() == "nccl" and world() > 1))
        _p2p_state = "off"
        if want:
            handle, why = _p2p_bring_up()
            if handle is not None:
                _p2p_comm, _p2p_state = handle, "on"
            elif mode == "p2p":
                raise RuntimeError(f"CVF_COMM=p2p: {why}")
            elif rank() == 0:
                print(f"[colvarsfinder] peer-to-peer windows not usable ({why}); the cross-rank sums go through RCCL", flush=True)
    return _p2p_comm


def fused_comm():
    """The communicator for the launches that carry a cross-rank sum inside them (cvf_ef16_finish_dp, cvf_ef_stats_dp,
    cvf_ef_loss_dp, cvf_slab_reduce_dp: the data-parallel step in as many launches as the single-process step), or None: the
    sums are separate all-reduce launches.  ``CVF_FUSED_COMM=0`` keeps them separate over the same windows."""
    if not collectives() or os.environ.get("CVF_FUSED_COMM", "1") == "0":
        return None
    return _p2p()


def p2p_error():
    """0, or the number of the exchange in which a peer's data did not arrive in time (a host-visible word: no device call)."""
    if _p2p_comm is None:
        return 0
    from . import _hip
    return int(_hip.lib().cvf_p2p_error(_p2p_comm))


def check_comm():
    """Raise if an exchange over the peer-to-peer windows has timed out: the kernels poison their results with NaN and leave the
    communicator's error word set; the tasks call this wherever they read results back (after a device synchronisation).  A
    communicator that has seen a time-out is dead - the job must stop."""
    err = p2p_error()
    if err:
        raise RuntimeError(f"colvarsfinder: a peer did not deliver its share of cross-rank exchange {err} within the time-out "
                           f"(CVF_P2P_TIMEOUT_MS, default 20 s) - rank {rank()} of {world()}; the batch sums / gradients of that "
                           "step are NaN and the ranks are no longer in step. Aborting.")


def allreduce_sum_(t):
    """In-place sum over ranks; a no-op in a single-process run."""
    if collectives():
        plain = t.is_cuda and t.dtype in (torch.float32, torch.float64) and t.is_contiguous()
        p2p = _p2p() if plain and t.numel() * t.element_size() <= _P2P_MAX_BYTES else None   # (None also when the job sums over RCCL)
        if p2p is not None:
            from . import _hip
            if t.dtype == torch.float64 and t.numel() <= 80:   # batch sums / loss terms: the low-latency words (one hop, no fence)
                _hip.check(_hip.lib().cvf_p2p_exchange_f64(p2p, _hip.ptr(t), t.numel(), _hip.stream()), "cvf_p2p_exchange_f64")
                return t
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
