"""ctypes binding of ``libcvf_hip.so`` (C ABI declared in ``include/cvf.h``).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes
through the C ABI with raw device pointers.  There is no CPU or eager-PyTorch fallback: if
the shared library is missing, or a call fails, this module raises.
"""

import ctypes as C
import os

import torch

TILE = 64
MAX_NETS = 8
MAX_LAYERS = 12
AUX_ROWS = 18
EF_HIDDEN_WIDTHS = (8, 12, 16, 20, 24, 32, 48, 64)   # hidden widths the eigenfunction kernels are instantiated for (csrc/ef_mfma.hip)


def ef_widths(n_hidden):
    """Hidden widths the eigenfunction kernels are instantiated for at this depth (csrc/ef_mfma.hip: ef_dispatch)."""
    if n_hidden >= 4:
        return (20, 32) if n_hidden <= 5 else ()
    return tuple(w for w in EF_HIDDEN_WIDTHS if w < 24 or n_hidden >= 2)   # (48 and 64: the plain 64-frame kernels, two or three hidden layers)


FEAT_ANGLE, FEAT_BOND, FEAT_DIHEDRAL, FEAT_POSITION = 0, 1, 2, 3
PP_IDENTITY, PP_ALIGN = 0, 1
PP_ALIGN_CONTIG, PP_PURE_POSITION, PP_SLOT_BATCHED = 1, 2, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcvf_hip.so")


class PPDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_coord", C.c_int32), ("n_align", C.c_int32), ("n_rec", C.c_int32),
                ("d_r", C.c_int32), ("use_angle_value", C.c_int32), ("has_position", C.c_int32), ("flags", C.c_int32),
                ("align_idx", C.c_void_p), ("ref_c", C.c_void_p), ("rec", C.c_void_p),
                ("atom_align", C.c_void_p), ("atom_slot", C.c_void_p), ("rec_slot", C.c_void_p), ("slot_atom", C.c_void_p), ("n_slot", C.c_int32),
                ("n_rec_slot", C.c_int32), ("align_w", C.c_void_p), ("mrec", C.c_void_p), ("slot_row", C.c_void_p),
                ("n_mrec", C.c_int32), ("n_ref", C.c_int32)]


class MLPDesc(C.Structure):
    _fields_ = [("n_nets", C.c_int32), ("n_layers", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1)),
                ("act", C.c_int32 * MAX_LAYERS), ("w_off", (C.c_int32 * MAX_LAYERS) * MAX_NETS),
                ("b_off", (C.c_int32 * MAX_LAYERS) * MAX_NETS), ("n_params", C.c_int32)]


class EFCfg(C.Structure):
    _fields_ = [("k", C.c_int32), ("lag_idx", C.c_int32), ("sort_eigvals", C.c_int32), ("pad_", C.c_int32),
                ("alpha", C.c_double), ("beta", C.c_double), ("dt", C.c_double), ("eig_w", C.c_double * MAX_NETS)]


class AdamArgs(C.Structure):
    _fields_ = [("theta", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("step_count", C.c_void_p), ("mlp", C.POINTER(MLPDesc)),
                ("packed", C.c_void_p), ("lr_dev", C.c_void_p)]


_SIGNATURES = {
    "cvf_version": (C.c_int, []),
    "cvf_last_error": (C.c_char_p, []),
    "cvf_ef_nstats": (C.c_int, [C.c_int, C.c_int]),
    "cvf_align_feature_scratch_bytes": (C.c_int64, [C.POINTER(PPDesc), C.c_int64]),
    "cvf_align_feature_fwd": (C.c_int, [C.POINTER(PPDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "cvf_metric_apply": (C.c_int, [C.POINTER(PPDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_metric_stats_scratch_doubles": (C.c_int64, [C.c_int64, C.c_int]),
    "cvf_metric_apply_stats": (C.c_int, [C.POINTER(PPDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(EFCfg), C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_metric_dense_doubles": (C.c_int64, [C.POINTER(PPDesc)]),
    "cvf_metric_dense_tensors": (C.c_int, [C.POINTER(PPDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_pack_floats": (C.c_int64, [C.POINTER(MLPDesc)]),
    "cvf_ef_pack": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_mlp_fwd": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "cvf_ef_saved_floats": (C.c_int64, [C.POINTER(MLPDesc), C.c_int64]),
    "cvf_ef_fwd_metric_supported": (C.c_int, [C.POINTER(MLPDesc), C.POINTER(PPDesc)]),
    "cvf_ef_fused_stats_rows": (C.c_int64, [C.POINTER(MLPDesc), C.POINTER(PPDesc), C.c_int64, C.c_int]),
    "cvf_ef_stats_finish_rows": (C.c_int, [C.POINTER(EFCfg), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_align_fwd_metric_supported": (C.c_int, [C.POINTER(MLPDesc), C.POINTER(PPDesc)]),
    "cvf_ef_align_fwd_metric_stats": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PPDesc), C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(EFCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p]),
    "cvf_ef_fwd_metric_stats": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PPDesc), C.c_void_p,
                                          C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(EFCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef16_supported": (C.c_int, [C.POINTER(MLPDesc), C.POINTER(PPDesc)]),
    "cvf_ef16_scratch_doubles": (C.c_int64, [C.c_int64, C.c_int]),
    "cvf_ef16_saved_floats": (C.c_int64, [C.POINTER(MLPDesc), C.c_int64]),
    "cvf_ef16_rows": (C.c_int64, [C.c_int64]),
    "cvf_ef16_finish": (C.c_int, [C.POINTER(EFCfg), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef16_front": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PPDesc), C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(EFCfg), C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef16_backward": (C.c_int, [C.POINTER(EFCfg), C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "cvf_ef16_front_transfer": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PPDesc), C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef16_transfer_rows": (C.c_int64, [C.c_int64, C.c_int]),
    "cvf_ef16_front_transfer_rows": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PPDesc), C.c_void_p,
                                               C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef16_backward_transfer": (C.c_int, [C.POINTER(EFCfg), C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "cvf_comm_unique_id_bytes": (C.c_int, []),
    "cvf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cvf_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "cvf_comm_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvf_comm_allreduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvf_comm_destroy": (C.c_int, [C.c_void_p]),
    "cvf_p2p_handle_bytes": (C.c_int, []),
    "cvf_p2p_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "cvf_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cvf_p2p_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvf_p2p_allreduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvf_p2p_error": (C.c_int, [C.c_void_p]),
    "cvf_p2p_destroy": (C.c_int, [C.c_void_p]),
    "cvf_ef16_finish_dp": (C.c_int, [C.POINTER(EFCfg), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_stats_dp": (C.c_int, [C.POINTER(EFCfg), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_loss_dp": (C.c_int, [C.POINTER(EFCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_probe_stream_out_floats": (C.c_int64, [C.c_int, C.c_int64]),
    "cvf_probe_stream": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvf_p2p_exchange_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "cvf_slab_reduce_dp": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(AdamArgs), C.c_void_p, C.c_void_p]),
    "cvf_ef_stats_scratch_doubles": (C.c_int64, [C.c_int, C.c_int]),
    "cvf_ef_stats": (C.c_int, [C.POINTER(EFCfg), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_loss": (C.c_int, [C.POINTER(EFCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_ef_backward_slab_rows": (C.c_int64, [C.c_int64]),
    "cvf_ef_backward": (C.c_int, [C.POINTER(EFCfg), C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "cvf_ef_align_fwd": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PPDesc), C.c_void_p, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_slab_reduce": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(AdamArgs), C.c_void_p]),
    "cvf_ae_scratch_floats": (C.c_int64, [C.POINTER(MLPDesc), C.c_int64]),
    "cvf_ae_step": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamArgs), C.c_void_p]),
    "cvf_regae_scratch_floats": (C.c_int64, [C.POINTER(MLPDesc), C.c_int64]),
    "cvf_regae_forward": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_regae_backward": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamArgs),
                                     C.c_void_p]),
    "cvf_regae_forward_keep": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_regae_backward_reuse": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AdamArgs),
                                     C.c_void_p]),
    "cvf_regae_enc_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cvf_regae_loss_row": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_double,
                                     C.c_double, C.c_void_p, C.c_void_p]),
    "cvf_mlp_eval_rows": (C.c_int, [C.POINTER(MLPDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "cvf_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_double,
                                C.c_double, C.c_double, C.c_void_p, C.POINTER(MLPDesc), C.c_void_p, C.c_void_p]),
    "cvf_sgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.POINTER(MLPDesc), C.c_void_p,
                               C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """The loaded shared library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"colvarsfinder: HIP extension not built ({LIB_PATH} missing). Run `python __graft_entry__.py` "
                "or `make -C colvars-finder_amd/csrc`. There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().cvf_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP path needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def align_scratch(desc, B, device):
    """Scratch buffer cvf_align_feature_fwd needs for B frames (None for small molecules)."""
    n = lib().cvf_align_feature_scratch_bytes(desc, B)
    return torch.empty((n + 7) // 8, device=device, dtype=torch.float64) if n > 0 else None


def ntiles(B):
    return (B + TILE - 1) // TILE


def require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(
            "colvarsfinder (MI355X build) computes on the GPU only: pass device=torch.device('cuda'). "
            "There is no CPU path in this package.")
    if not torch.cuda.is_available():
        raise RuntimeError("colvarsfinder: no HIP device visible")
    lib()
    return device


def upload_f32(a, device, chunk_bytes=64 << 20):
    """Host array -> contiguous fp32 device tensor (trajectory ingestion into the HBM-resident shard).

    A pageable ``tensor.to(device)`` stages through the driver's own bounce buffer.  Here an fp32 C-contiguous source is
    page-locked in place (``hipHostRegister``) and copied by one DMA; anything else (fp64 text trajectories, strided
    arrays) is converted chunk by chunk into two pinned staging buffers whose copies run on a side stream while the
    next chunk is converted.  Small arrays take the plain path."""
    t = torch.as_tensor(a)
    if t.is_cuda:
        return t.to(device=device, dtype=torch.float32).contiguous()
    n = t.numel()
    if n * 4 <= (8 << 20) or not torch.cuda.is_available():
        return t.to(dtype=torch.float32).contiguous().to(device)
    out = torch.empty(tuple(t.shape), device=device, dtype=torch.float32)
    if t.dtype == torch.float32 and t.is_contiguous():
        rt = torch.cuda.cudart()
        if int(rt.cudaHostRegister(t.data_ptr(), n * 4, 0)) == 0:
            try:
                out.copy_(t, non_blocking=True)
                torch.cuda.current_stream(device).synchronize()
            finally:
                rt.cudaHostUnregister(t.data_ptr())
            return out
    src, dst = t.reshape(-1), out.reshape(-1)
    ce = max(1, chunk_bytes // 4)
    bufs = [torch.empty(ce, dtype=torch.float32).pin_memory() for _ in range(2)]
    done = [None, None]
    side = torch.cuda.Stream(device)
    # `out` came from the caching allocator on the current stream: work queued there may still use the block it recycled
    side.wait_stream(torch.cuda.current_stream(device))
    for i, s0 in enumerate(range(0, n, ce)):
        e0, b = min(s0 + ce, n), bufs[i % 2]
        if done[i % 2] is not None:
            done[i % 2].synchronize()            # the copy that last used this staging buffer has finished
        b[:e0 - s0].copy_(src[s0:e0])            # host side: gather / convert into page-locked memory
        with torch.cuda.stream(side):
            dst[s0:e0].copy_(b[:e0 - s0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
            done[i % 2] = ev
    torch.cuda.current_stream(device).wait_stream(side)
    side.synchronize()
    return out
