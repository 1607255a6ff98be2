"""Training tasks with the API of the reference's ``colvarsfinder.core`` on MI355X.

API twin (own code) of reference ``colvarsfinder/core.py``: ``TrainingTask`` (core.py:60-249),
``EigenFunctionTask`` (core.py:251-566) and ``AutoEncoderTask`` (core.py:569-744) keep their
constructor arguments, defaults, public methods and attributes, so the reference's example
scripts run unchanged (``device`` defaults to ``torch.device('cuda')``, PyTorch-ROCm's name for the HIP device; the
modules returned by ``colvar_model()`` / ``reg_model()`` take CPU tensors and answer on the CPU, as the notebooks'
evaluation cells expect).  What differs is *how* a
step is computed: the model's parameters live in one flat HBM buffer, the trajectory shard is
resident in HBM and pre-permuted once (batches are static: ``shuffle=False``, core.py:472-481),
and every step is a short sequence of hand-written gfx950 kernels called through the C ABI of
``libcvf_hip.so`` (``include/cvf.h``) - no autograd graph, no per-step host synchronisation.

Per-step pipeline of ``EigenFunctionTask`` (generator mode, core.py:418-426,438):
    K1  align + features           cvf_align_feature_fwd     replaces pp_layer(X)           core.py:403
    K4a nets forward + dY/dfeat    cvf_ef_mlp_fwd            replaces model(...)            core.py:403
    K2/3 q = J A J^T g, E          cvf_metric_apply          replaces autograd.grad x k     core.py:424-426
    K5  batch sums (fp64)          cvf_ef_stats              core.py:406-410,446-452
    C1  all-reduce of the sums     torch.distributed (RCCL)  (absent in the reference)
        scalar tail + d loss/d sum cvf_ef_loss               core.py:426-457
    K4b parameter gradient         cvf_ef_backward           replaces loss.backward()       core.py:517
    C2  all-reduce of the gradient torch.distributed (RCCL)
    K6  Adam                       cvf_adam_step             replaces optimizer.step()      core.py:522

There is no CPU path: ``device`` must be a HIP device (``torch.device('cuda')`` is PyTorch's name
for it) and the HIP extension must be built, otherwise construction raises.
"""

import copy
import math
import os
import types
from abc import ABC, abstractmethod

import numpy as np
import pandas as pd
import torch

import ctypes

from . import _dist, _hip
from .nn import AutoEncoder, EigenFunctions, RegAutoEncoder, RegModel, mlp_layout  # noqa: F401
from .pp import AlignFeatureLayer, identity_desc

try:  # logging sink of the reference (core.py:50,143); optional here
    from tensorboardX import SummaryWriter as _SummaryWriter
except Exception:  # pragma: no cover - tensorboardX is not part of the image
    _SummaryWriter = None

try:
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    def _tqdm(it, **_):
        return it


def C_pointer(struct):
    return ctypes.pointer(struct)


class _ScalarLog:
    """Stand-in for ``tensorboardX.SummaryWriter`` when that package is absent: keeps the scalars."""

    def __init__(self, logdir=None):
        self.logdir, self.scalars = logdir, []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), int(step)))


def _split(n, test_ratio):
    """Same draw as ``sklearn.model_selection.train_test_split`` at core.py:465,468,672: one permutation
    from NumPy's global RNG; test = first ceil(ratio n) entries, train = the rest."""
    n_test = int(math.ceil(test_ratio * n))
    perm = np.random.permutation(n)
    return perm[n_test:], perm[:n_test]


class _AsyncEpochLog:
    """The per-epoch loss rows leave the device through a small ring of pinned buffers, asynchronously: the host side of an
    epoch (``loss_list``, ``_cvec``, the writer's scalars - ``on_epoch(epoch, train_rows, test_rows)``) runs up to
    ``depth - 1`` epochs later, in order, instead of draining the stream after every epoch (with a few steps per epoch that
    drain was most of the epoch: 1.38 -> 0.54 ms per epoch at config 3).  Epochs that save or plot ``flush_all()`` first."""

    def __init__(self, log_tr, log_te, n_tr, n_te, on_epoch, depth=4):
        self.log_tr, self.log_te, self.n_tr, self.n_te, self.on_epoch = log_tr, log_te, n_tr, n_te, on_epoch
        self.ring = [dict(tr=torch.zeros(log_tr.shape, dtype=log_tr.dtype).pin_memory(),
                          te=torch.zeros(log_te.shape, dtype=log_te.dtype).pin_memory(), ev=torch.cuda.Event(), pending=None)
                     for _ in range(depth)]

    def _flush(self, slot):
        slot["ev"].synchronize()
        _dist.check_comm()    # a cross-rank exchange that timed out has poisoned these rows with NaN: stop here, loudly
        ep, slot["pending"] = slot["pending"], None
        self.on_epoch(ep, slot["tr"][:self.n_tr].clone(), slot["te"][:self.n_te].clone())

    def reserve(self, epoch):
        """Finish the epoch that still occupies this epoch's ring slot (callers with side buffers of their own do this first)."""
        slot = self.ring[epoch % len(self.ring)]
        if slot["pending"] is not None:
            self._flush(slot)
        return slot

    def push(self, epoch):
        slot = self.reserve(epoch)
        slot["tr"][:self.n_tr].copy_(self.log_tr[:self.n_tr], non_blocking=True)
        slot["te"][:self.n_te].copy_(self.log_te[:self.n_te], non_blocking=True)
        slot["ev"].record()
        slot["pending"] = epoch

    def flush_all(self):
        for slot in sorted((s_ for s_ in self.ring if s_["pending"] is not None), key=lambda s_: s_["pending"]):
            self._flush(slot)


class _CVModel(torch.nn.Sequential):
    """What ``colvar_model()`` / ``reg_model()`` return: ``Sequential(preprocessing_layer, nets)`` as in the reference
    (core.py:372-382, 640-647, 855-877), evaluated on the task's GPU whatever device the input lives on.  The reference's
    callers pass CPU tensors made from the trajectory and call ``.detach().numpy()`` on the result (2d.ipynb:437-446,
    main.ipynb:561-562, and ``plot_class.plot(self.colvar_model(), ...)`` at core.py:530-532): the input is moved to the
    device, alignment kernel and nets run there, the result comes back on the input's device and floating-point type."""

    def __init__(self, *modules, device=None):
        super().__init__(*modules)
        self._cvf_device = None if device is None else torch.device(device)

    def _compute_device(self):
        if self._cvf_device is not None:
            return self._cvf_device
        for t in list(self.parameters()) + list(self.buffers()):   # (a slice of this Sequential: where its tensors live)
            return t.device
        return torch.device("cuda")

    def _autograd_twin(self):
        """The same map written with torch operators only (the alignment layer replaced by export.ScriptableAlignFeature, the
        nets as they are), on the compute device - what the reference's own ``Sequential`` is (core.py:372-382): used when the
        caller wants d xi / d x through autograd.  The HIP kernels carry no autograd graph."""
        from .export import ScriptableAlignFeature
        mods = []
        for m in self.children():
            mods.append(ScriptableAlignFeature(m).to(self._compute_device()) if hasattr(m, "pp_desc") and hasattr(m, "rec") else m)
        return torch.nn.Sequential(*mods)

    def forward(self, x):
        x = torch.as_tensor(x)
        dev = _hip.require_gpu(self._compute_device())
        src_dev, src_dt = x.device, (x.dtype if x.dtype.is_floating_point else torch.float32)
        if x.requires_grad and torch.is_grad_enabled():
            # differentiable call (the reference returns a plain differentiable Sequential: d xi / d x by autograd, e.g. for the
            # Dirichlet energy of a learned CV): torch operators on the GPU in fp32, graph attached, answer in the caller's type
            with torch.cuda.device(dev):
                out = self._autograd_twin()(x.to(device=dev, dtype=torch.float32))   # (the nets' parameters are fp32)
            return out.to(device=src_dev, dtype=src_dt)
        with torch.cuda.device(dev):
            out = super().forward(x.detach().to(device=dev, dtype=torch.float32))
        return out.to(device=src_dev, dtype=src_dt)


class _FlatParams:
    """The model's parameters as views of one fp32 device buffer (+ gradient and Adam moments)."""

    def __init__(self, model, device):
        model.to(device=device, dtype=torch.float32)
        self.model = model
        self._views = None
        if isinstance(model, EigenFunctions) and self._init_padded(model, device):
            return
        lay = mlp_layout(model)
        self.n = lay["n_params"]
        self.theta = torch.empty(self.n, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.n, device=device, dtype=torch.float32)
        pos = 0
        for p in model.parameters():
            n = p.numel()
            self.theta[pos:pos + n].copy_(p.data.reshape(-1))
            p.data = self.theta[pos:pos + n].view(p.shape)  # the module now aliases the flat buffer
            pos += n
        nets = lay["nets"]
        assert 1 <= len(nets) <= _hip.MAX_NETS, f"between 1 and {_hip.MAX_NETS} nets are supported, got {len(nets)}"
        L = len(nets[0])
        assert all(len(c) == L for c in nets) and L <= _hip.MAX_LAYERS, "nets must have the same depth (<= 12 layers)"
        d = _hip.MLPDesc()
        d.n_nets, d.n_layers, d.n_params = len(nets), L, self.n
        for l, (_, _, fin, fout, act) in enumerate(nets[0]):
            d.dims[l], d.dims[l + 1], d.act[l] = fin, fout, act
        for i, chain in enumerate(nets):
            for l, (wo, bo, fin, fout, act) in enumerate(chain):
                assert (fin, fout, act) == (d.dims[l], d.dims[l + 1], d.act[l]), "nets must share one architecture"
                d.w_off[i][l], d.b_off[i][l] = wo, bo
        self.desc = d
        # second copy of the weights as MFMA fragments (EigenFunctions nets only; see csrc/cvf_pack.hpp)
        n_pack = _hip.lib().cvf_ef_pack_floats(d) if isinstance(model, EigenFunctions) else 0
        self.packed = torch.zeros(n_pack, device=device, dtype=torch.float32) if n_pack > 0 else None
        self.repack()

    def _init_padded(self, model, device):
        """EigenFunctions whose hidden widths are NOT one of the kernels' widths (csrc/ef_mfma.hip: d0 -> H -> .. -> H -> 1 with
        H in _hip.EF_HIDDEN_WIDTHS): lay the nets out with every hidden layer padded to the next kernel width H.  The padding
        rows / columns hold zeros, and stay zero: a padded unit's pre-activation is 0, tanh(0) = 0, its outgoing weights are 0,
        so every gradient entry that touches the padding is exactly 0 and Adam / SGD leave a 0 with zero moments where it is.
        The module's parameters alias the unpadded sub-blocks (strided views), so state_dict(), load_state_dict(), forward and
        the per-CV text export see the nets the user built.  Returns False when no padding is needed or possible."""
        from .nn import _chain_layers
        chains = [_chain_layers(net) for net in model.eigen_funcs]
        L = len(chains[0])
        dims = [chains[0][0][0].in_features] + [lin.out_features for lin, _ in chains[0]]
        hidden = dims[1:-1]
        if not (2 <= L <= 6) or dims[-1] != 1 or not hidden or max(hidden) > max(_hip.EF_HIDDEN_WIDTHS):
            return False
        from .nn import ACT_TANH, ACT_RELU, ACT_ELU, ACT_LEAKY_RELU
        act0 = chains[0][0][1]
        if act0 not in (ACT_TANH, ACT_RELU, ACT_ELU, ACT_LEAKY_RELU):   # (a padded unit must emit act(0) = 0: not Sigmoid / Softplus)
            return False
        for c in chains:
            if [c[0][0].in_features] + [lin.out_features for lin, _ in c] != dims or [a for _, a in c] != [act0] * (L - 1) + [0]:
                return False
        cand = [w for w in _hip.ef_widths(L - 1) if w >= max(hidden)]
        if not cand:                          # e.g. ONE hidden layer of 21..32 units: no kernel width to pad to - the task then
            return False                      # raises its NotImplementedError (plain layout, no fragment copy)
        H = min(cand)
        if all(h == H for h in hidden):
            return False                      # already a kernel shape: plain layout
        pdims = [dims[0]] + [H] * (L - 1) + [1]
        per_net = sum(pdims[l + 1] * pdims[l] + pdims[l + 1] for l in range(L))
        assert len(chains) <= _hip.MAX_NETS, f"between 1 and {_hip.MAX_NETS} nets are supported, got {len(chains)}"
        self.n = per_net * len(chains)
        self.theta = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.n, device=device, dtype=torch.float32)
        d = _hip.MLPDesc()
        d.n_nets, d.n_layers, d.n_params = len(chains), L, self.n
        for l in range(L):
            d.dims[l], d.dims[l + 1], d.act[l] = pdims[l], pdims[l + 1], (act0 if l < L - 1 else 0)
        self._views, pos = [], 0
        for i, chain in enumerate(chains):
            for l, (lin, _) in enumerate(chain):
                fo, fi = pdims[l + 1], pdims[l]
                d.w_off[i][l], d.b_off[i][l] = pos, pos + fo * fi
                for prm, off, shape in ((lin.weight, pos, (fo, fi)), (lin.bias, pos + fo * fi, (fo,))):
                    sl = tuple(slice(0, n_) for n_ in prm.shape)
                    tv = self.theta[off:off + int(np.prod(shape))].view(shape)[sl]
                    gv = self.grad[off:off + int(np.prod(shape))].view(shape)[sl]
                    tv.copy_(prm.data)
                    prm.data = tv                       # the module aliases its sub-block of the padded layer
                    self._views.append((prm, gv))
                pos += fo * fi + fo
        # (grad_views() must follow model.parameters() order for the optimizer's param_groups; both walk nets, then layers)
        assert [id(p_) for p_, _ in self._views] == [id(p_) for p_ in model.parameters()]
        self.desc = d
        n_pack = _hip.lib().cvf_ef_pack_floats(d)
        self.packed = torch.zeros(n_pack, device=device, dtype=torch.float32) if n_pack > 0 else None
        self.repack()
        return True

    def grad_views(self):
        """[(module parameter, its slice of the flat gradient)] in flat order."""
        if self._views is not None:
            return list(self._views)
        out, pos = [], 0
        for p in self.model.parameters():
            out.append((p, self.grad[pos:pos + p.numel()].view(p.shape)))
            pos += p.numel()
        return out

    def repack(self):
        """Rebuild the fragment copy from theta (needed after the parameters were written from outside)."""
        if self.packed is not None:
            _hip.check(_hip.lib().cvf_ef_pack(self.desc, _hip.ptr(self.theta), _hip.ptr(self.packed), _hip.stream()),
                       "cvf_ef_pack")


class _FusedOptimizer:
    """``optimizer`` attribute of the tasks: Adam / SGD as ONE kernel over the flat buffer (same update rule and
    defaults as the ``torch.optim`` objects built at core.py:163-166), with the parts of the ``torch.optim.Optimizer``
    surface user code touches: ``param_groups`` (``param_groups[0]['lr']`` may be edited between steps - a manual decay, or a
    schedule driven by user code; ``torch.optim.lr_scheduler`` classes insist on a ``torch.optim.Optimizer`` and do not take this
    object - and is honoured by captured hipGraphs too: the kernels read it from a device scalar), ``zero_grad``, ``step``,
    ``state_dict`` / ``load_state_dict`` (the reference never checkpoints the optimizer; SURVEY 8f row 2 asks for it)."""

    def __init__(self, flat, name, lr):
        self.flat, self.name, self.lr = flat, name.lower(), float(lr)
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self.exp_avg = torch.zeros_like(flat.theta)
        self.exp_avg_sq = torch.zeros_like(flat.theta)
        self.step_count = torch.zeros(1, device=flat.theta.device, dtype=torch.int32)
        self.param_groups = [dict(params=[p for p, _ in flat.grad_views()], lr=self.lr, betas=self.betas, eps=self.eps,
                                  weight_decay=0, amsgrad=False)]
        self.lr_dev = torch.full((1,), self.lr, device=flat.theta.device, dtype=torch.float32)
        self._lr_written = self.lr

    def sync_lr(self):
        """Write ``param_groups[0]['lr']`` to the device scalar the kernels read, if it changed since the last write
        (called before every eager step and before every graph replay)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_written:
            self.lr_dev.fill_(lr)
            self._lr_written = lr
        return lr

    def zero_grad(self, set_to_none=True):
        # the backward kernels overwrite the whole flat gradient; module-level .grad tensors (copies made by the
        # tasks' backward()) are dropped or zeroed as torch.optim does
        for p, _ in self.flat.grad_views():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def fused_args(self):
        """``cvf_adam_args`` for the kernels that reduce the gradient and apply Adam in one launch
        (single-process runs only: with several ranks the gradient all-reduce sits in between)."""
        if self.name != "adam":
            return None
        f, a = self.flat, _hip.AdamArgs()
        a.theta, a.m, a.v = f.theta.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.lr, a.beta1, a.beta2, a.eps = self.sync_lr(), self.betas[0], self.betas[1], self.eps
        a.lr_dev = self.lr_dev.data_ptr()
        a.step_count = self.step_count.data_ptr()
        if f.packed is not None:
            a.mlp, a.packed = C_pointer(f.desc), f.packed.data_ptr()
        return a

    def step(self, advance=True):
        """``advance=False`` when the gradient kernel of this step has already advanced the step counter
        (the fused training loops); the public ``loss -> backward() -> optimizer.step()`` path advances here and
        takes the gradient from the modules' ``.grad`` tensors where they are set (so that edits made after
        ``backward()`` - clipping, masking - count, as with ``torch.optim``)."""
        f, lib = self.flat, _hip.lib()
        lr = self.sync_lr()
        if advance:
            self.step_count += 1
            for p, gv in f.grad_views():
                if p.grad is not None:
                    gv.copy_(p.grad.to(device=gv.device, dtype=gv.dtype))
        if self.name == "adam":
            _hip.check(lib.cvf_adam_step(_hip.ptr(f.theta), _hip.ptr(f.grad), _hip.ptr(self.exp_avg), _hip.ptr(self.exp_avg_sq),
                                         f.n, lr, _hip.ptr(self.lr_dev), self.betas[0], self.betas[1], self.eps,
                                         _hip.ptr(self.step_count), f.desc, _hip.ptr(f.packed), _hip.stream()), "cvf_adam_step")
        else:
            _hip.check(lib.cvf_sgd_step(_hip.ptr(f.theta), _hip.ptr(f.grad), f.n, lr, _hip.ptr(self.lr_dev), f.desc,
                                        _hip.ptr(f.packed), _hip.stream()), "cvf_sgd_step")

    def state_dict(self):
        """Adam moments, step number and hyper-parameters as CPU tensors / numbers (flat order = ``parameters()`` order for
        EigenFunctions / AutoEncoder models, the chain order of ``_RegFlatParams`` for RegAutoEncoder)."""
        return dict(name=self.name, n=int(self.flat.n), step=int(self.step_count.item()),
                    exp_avg=self.exp_avg.detach().cpu().clone(), exp_avg_sq=self.exp_avg_sq.detach().cpu().clone(),
                    param_groups=[dict(lr=float(self.param_groups[0]["lr"]), betas=tuple(self.betas), eps=float(self.eps))])

    def load_state_dict(self, sd):
        assert sd["name"] == self.name and int(sd["n"]) == int(self.flat.n), \
            f"optimizer state is for {sd['name']} over {sd['n']} parameters, this one is {self.name} over {self.flat.n}"
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count.fill_(int(sd["step"]))
        self.param_groups[0]["lr"] = float(sd["param_groups"][0]["lr"])
        self.sync_lr()


class TrainingTask(ABC):
    """Base class of the training tasks (constructor arguments and attributes as core.py:102-143)."""

    def __init__(self, traj_obj, pp_layer, model, model_path, learning_rate, load_model_filename, save_model_every_step,
                 k, batch_size, num_epochs, test_ratio, optimizer_name, device, plot_class, plot_frequency, verbose,
                 debug_mode):
        self.device = _hip.require_gpu(device)
        self.traj_obj = traj_obj
        self.preprocessing_layer = pp_layer.to(self.device)
        self.learning_rate = learning_rate
        self.batch_size = batch_size
        self.num_epochs = num_epochs
        self.test_ratio = test_ratio
        self.k = k
        self.model = model
        self.load_model_filename = load_model_filename
        self.save_model_every_step = save_model_every_step
        self.model_path = model_path
        self.optimizer_name = optimizer_name
        self.plot_class = plot_class
        self.plot_frequency = plot_frequency
        self.verbose = verbose
        self.debug_mode = debug_mode
        self.model_name = type(self).__name__
        if self.verbose:
            print('\n[Info] Log directory: {}\n'.format(self.model_path), flush=True)
        self.writer = _SummaryWriter(self.model_path) if _SummaryWriter is not None else _ScalarLog(self.model_path)

    # -- every kernel launch goes through here; bench.py sets ``_events`` to time launches with HIP events
    _events = None
    _last_call = None      # with ``_events``: name -> (fn, args) of the most recent launch (bench.py re-times the dominant one)

    def _call(self, name, fn, *args):
        ev = self._events
        if ev is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args)
            b.record()
            ev.setdefault(name, []).append((a, b))
            if self._last_call is not None:
                self._last_call[name] = (fn, args)
        else:
            rc = fn(*args)
        _hip.check(rc, name)

    def _allreduce(self, name, t):
        """One of the step's two cross-rank sums, bracketed by HIP events like a C-ABI call when the
        per-call timing pass is on (bench.py reports them per collective: where a multi-GPU step's time goes)."""
        ev = self._events
        if ev is not None and _dist.collectives() and t.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _dist.allreduce_sum_(t)
            b.record()
            ev.setdefault(name, []).append((a, b))
        else:
            _dist.allreduce_sum_(t)
        return t

    # -- description of r(x) for the kernels
    def _pp_desc(self, n_coord):
        pp = self.preprocessing_layer
        if isinstance(pp, torch.nn.Identity):
            return identity_desc(n_coord)
        if isinstance(pp, AlignFeatureLayer):
            assert n_coord == 3 * pp.n_atoms, f"trajectory frames have {n_coord} coordinates, layer expects {3 * pp.n_atoms}"
            return pp.pp_desc()
        raise TypeError("pp_layer must be torch.nn.Identity or a colvarsfinder.pp.AlignFeatureLayer "
                        f"(PreprocessingANN); the GPU kernels cannot run an arbitrary module ({type(pp).__name__})")

    def init_model_and_optimizer(self):
        """core.py:145-166: optional restart from ``load_model_filename``; Adam (case-insensitive) else SGD."""
        if self.load_model_filename:
            if os.path.isfile(self.load_model_filename):
                self.model.load_state_dict(torch.load(self.load_model_filename, map_location="cpu"), strict=False)
                if self.verbose:
                    print(f'model parameters loaded from: {self.load_model_filename}')
            elif self.verbose:
                print(f'model file not found: {self.load_model_filename}')
        self._flat = _FlatParams(self.model, self.device)
        self.optimizer = _FusedOptimizer(self._flat, self.optimizer_name, self.learning_rate)

    def save_model(self, epoch, description="latest"):
        """core.py:168-227: ``model.pt`` + per-CV text files (+ TorchScript CV when the layer is scriptable)."""
        _dist.check_comm()
        if _dist.rank() != 0:
            return
        if self.verbose:
            print(f"\n\nEpoch={epoch}:")
        sd = {k_: v.detach().cpu() for k_, v in self.model.state_dict().items()}
        if self.debug_mode is True:
            os.makedirs(f'{self.model_path}/models', exist_ok=True)
            torch.save(sd, f'{self.model_path}/models/model_{epoch}.pt')
        out_dir = f'{self.model_path}/{description}'
        os.makedirs(out_dir, exist_ok=True)
        torch.save(sd, f'{out_dir}/model.pt')
        for idx in range(self.k):
            for name, param in self.model.get_params_of_cv(idx):
                np.savetxt('%s/%d_' % (out_dir, idx) + name.replace('.', '_') + '.txt', param.detach().cpu().numpy())
        if self.verbose:
            print(f'  trained model saved at:\n\t{out_dir}/model.pt')
        # TorchScript export (core.py:205-226): the kernel-backed alignment layer is replaced by its pure-torch export
        # twin (export.py), so the saved files are self-contained; the GPU variant is the same module moved to the device
        from .export import scriptable_cv
        cv = scriptable_cv(self.colvar_model())
        torch.jit.script(cv).save(f'{out_dir}/scripted_cv_cpu.pt')
        torch.jit.script(copy.deepcopy(cv).to(self.device)).save(f'{out_dir}/scripted_cv_gpu.pt')
        if self.verbose:
            print(f'  script models for CVs saved at:\n\t{out_dir}/scripted_cv_cpu.pt\n\t{out_dir}/scripted_cv_gpu.pt\n', flush=True)

    @abstractmethod
    def train(self):
        pass

    @abstractmethod
    def colvar_model(self):
        pass

    @abstractmethod
    def reg_model(self):
        pass


class _EFWorkspace:
    """Device buffers of one batch size (all sizes follow include/cvf.h)."""

    def __init__(self, B, k, d_r, n_params, lag, mlp_desc, device, ef16=False):
        lib = _hip.lib()
        T = _hip.ntiles(B)
        Tt = 2 * T if lag > 0 else T
        f32 = dict(device=device, dtype=torch.float32)
        f64 = dict(device=device, dtype=torch.float64)
        self.B, self.T, self.Tt = B, T, Tt
        # features / alignment data are double-buffered: the alignment kernel of the NEXT batch does not depend on
        # the parameters, so the training loop runs it beside this batch's step (EigenFunctionTask.train_step)
        self.slot = 0
        self._feat = [torch.empty(Tt * d_r * _hip.TILE, **f32) for _ in range(2)]
        self._aux = [torch.empty(T * _hip.AUX_ROWS * _hip.TILE, **f32) for _ in range(2)]
        self.y = torch.empty(Tt * k * _hip.TILE, **f32)
        if lag == 0:
            self.g = None if ef16 else torch.empty(T * k * d_r * _hip.TILE, **f32)   # (the 16-frame step keeps g on the chip)
            self.q = torch.empty(T * k * d_r * _hip.TILE, **f32)
            self.e = torch.empty(T * k * _hip.TILE, **f32)
        else:
            self.g = self.q = self.e = None
        self.scratch = torch.zeros(lib.cvf_ef16_scratch_doubles(B, k) if ef16 else
                                   lib.cvf_metric_stats_scratch_doubles(B, k) if lag == 0 else
                                   lib.cvf_ef_stats_scratch_doubles(k, lag), **f64)
        self.stats = torch.empty(lib.cvf_ef_nstats(k, lag), **f64)
        self.loss_vec = torch.empty(3 + 2 * k, **f64)
        self.coef = torch.empty(4 * k + k * k, **f64)
        self._k1_scratch, self.k1_scratch_checked = [None, None], False   # large-molecule alignment scratch, sized on first use
        # hidden activations handed from the forward kernel to the backward kernel (0 floats: shape without hand-off)
        n_saved = lib.cvf_ef16_saved_floats(mlp_desc, Tt) if ef16 else lib.cvf_ef_saved_floats(mlp_desc, Tt)
        self.saved = torch.empty(n_saved, **f32) if n_saved > 0 else None
        self.slab_rows = lib.cvf_ef_backward_slab_rows(Tt)
        self.slab = torch.empty(self.slab_rows * n_params, **f32)

    feat = property(lambda self: self._feat[self.slot])
    aux = property(lambda self: self._aux[self.slot])
    k1_scratch = property(lambda self: self._k1_scratch[self.slot])


class _HostFrames:
    """The trajectory on the host as the tasks touch it: ``shape``, ``rows(idx)`` (the frames a process keeps resident) and
    ``all()`` (single process: everything, once).  ``traj_obj.trajectory`` may be an array (the reference's
    ``WeightedTrajectory``: held whole, core.py:343), an ``np.memmap`` (indexing it reads the pages of the rows asked for), or
    any object with ``shape`` and ``take_rows(rows)`` - ``utils.RowReader`` - so that a rank of a data-parallel job never
    holds more than its own 1/world of the frames on the host either."""

    def __init__(self, src):
        self.reader = src if hasattr(src, "take_rows") else None
        self.array = None if self.reader is not None else np.asarray(src)
        self.shape = tuple((self.reader if self.reader is not None else self.array).shape)

    def rows(self, idx):
        return self.reader.take_rows(idx) if self.reader is not None else self.array[idx]

    def all(self):
        return self.reader.take_rows(np.arange(self.shape[0])) if self.reader is not None else self.array


class EigenFunctionTask(TrainingTask):
    """Eigenfunctions of the generator (``lag_tau == 0``) or of the transfer operator (``lag_tau > 0``).

    Arguments, defaults and attributes as reference core.py:293-354 (see the module docstring for
    what replaces what).  ``loss_list`` / ``train_loss_df`` / ``test_loss_df`` hold the same
    per-step rows ``[loss, eigen_non_penalty, eigen_penalty, eig_1..k]`` as the reference.
    """

    def __init__(self, traj_obj, pp_layer, model, model_path, alpha, eig_weights, diag_coeff=None, beta=1.0, lag_tau=0,
                 learning_rate=0.01, load_model_filename=None, save_model_every_step=10, sort_eigvals_in_training=True,
                 k=1, batch_size=1000, num_epochs=10, test_ratio=0.2, optimizer_name='Adam',
                 device=torch.device('cuda'), plot_class=None, plot_frequency=0, verbose=True, debug_mode=True):
        super().__init__(traj_obj, pp_layer, model, model_path, learning_rate, load_model_filename, save_model_every_step,
                         k, batch_size, num_epochs, test_ratio, optimizer_name, device, plot_class, plot_frequency,
                         verbose, debug_mode)
        assert isinstance(model, EigenFunctions), 'model must be an object of the class EigenFunctions'
        assert k == len(model.eigen_funcs), \
            f'number of cv ({k}) must equal the number of eigenfunctions ({len(model.eigen_funcs)})'
        assert len(eig_weights) >= k, f'{k} eigenvalue weights are needed, got {len(eig_weights)}'
        self._alpha = alpha
        self._sort_eigvals_in_training = sort_eigvals_in_training
        self._eig_w = eig_weights
        self._cvec = None
        self.traj_dt = traj_obj.dt
        lag_idx = lag_tau / self.traj_dt
        assert abs(lag_idx - int(lag_idx)) < 1e-6, \
            f'lag-time ({lag_tau}) not divisable by the timestep {self.traj_dt} of the trajectory'
        self.lag_idx = int(lag_idx)
        if self.verbose:
            print('\nEigenfunctions:\n', self.model, flush=True)
        self.init_model_and_optimizer()
        if self._flat.packed is None:   # no matrix-core kernel instance for this architecture: say so here, not at the first step
            d = self._flat.desc
            raise NotImplementedError(
                "EigenFunctionTask on MI355X: no kernel instance for nets with layer widths "
                f"{[d.dims[i] for i in range(d.n_layers + 1)]}. Supported: 2 to 5 hidden layers of at most "
                f"{max(_hip.EF_HIDDEN_WIDTHS)} units each, ONE hidden layer of at most {max(_hip.ef_widths(1))} units "
                f"(kernel widths {_hip.EF_HIDDEN_WIDTHS}, for 4 or 5 hidden layers 20 and 32; "
                f"other widths are zero-padded to the next one - not with Sigmoid / Softplus, whose padding would not stay zero), "
                f"scalar output, ONE activation of include/cvf.h after every hidden layer, k <= {_hip.MAX_NETS} "
                "(csrc/ef_mfma.hip: ef_shape / ef_dispatch).")

        # The frames stay resident in HBM (core.py:343-344 keeps CPU copies and moves every batch, core.py:500).  One process:
        # the whole trajectory.  Data-parallel job (one process per GPU): NOT here - train() uploads only the rows of this
        # rank's slices of the static batches, 1/world of the trajectory per GPU.
        self._traj_host = _HostFrames(traj_obj.trajectory)
        self._n_frames = int(self._traj_host.shape[0])
        self._sharded = _dist.world() > 1
        self._traj = None if self._sharded else _hip.upload_f32(self._traj_host.all(), self.device)
        self._weights = torch.as_tensor(np.asarray(traj_obj.weights)).to(device=self.device, dtype=torch.float32).contiguous()
        self.resident_bytes = 0 if self._sharded else self._traj.numel() * 4    # frames held in HBM (train() adds its gathers)
        self.tot_dim = int(np.prod(self._traj_host.shape[1:]))
        self._beta = beta
        if self.lag_idx == 0:
            if diag_coeff is not None:
                assert diag_coeff.dim() == 1 and diag_coeff.size(dim=0) == self.tot_dim, \
                    f'diag_coeff should be a 1d tensor of length {self.tot_dim}, current shape: {diag_coeff}'
                self._diag_coeff = diag_coeff.detach().to(device=self.device, dtype=torch.float32).contiguous()
            else:
                self._diag_coeff = torch.ones(self.tot_dim, device=self.device, dtype=torch.float32)

        self._pp = self._pp_desc(self.tot_dim)
        assert self._pp.d_r == self._flat.desc.dims[0], \
            f'preprocessing layer emits {self._pp.d_r} features but the networks take {self._flat.desc.dims[0]}'
        cfg = _hip.EFCfg()
        cfg.k, cfg.lag_idx, cfg.sort_eigvals = k, self.lag_idx, int(bool(sort_eigvals_in_training))
        cfg.alpha, cfg.beta, cfg.dt = float(alpha), float(beta), float(self.traj_dt)
        for i in range(k):
            cfg.eig_w[i] = float(eig_weights[i])
        self._cfg = cfg
        # large molecules (streaming alignment path): moments of (diag_coeff, reference) used by the derivative kernel
        self._dense = None
        if self.lag_idx == 0 and _hip.lib().cvf_align_feature_scratch_bytes(self._pp, 64) > 0:
            why = self.preprocessing_layer.derivative_table_limits()
            if why is not None:   # (here, not as a generic message at the first step)
                raise NotImplementedError(f"EigenFunctionTask (generator mode) on MI355X: the feature list has {why} "
                                          "(csrc/metric_large.hip). Use lag_tau > 0 (no derivative through the features) or fewer features per atom.")
            self._dense = torch.zeros(_hip.lib().cvf_metric_dense_doubles(self._pp), device=self.device, dtype=torch.float64)
            _hip.check(_hip.lib().cvf_metric_dense_tensors(self._pp, _hip.ptr(self._diag_coeff), _hip.ptr(self._dense),
                                                           _hip.stream()), "cvf_metric_dense_tensors")
        self._ws = {}
        self._graphs = {}
        # whole-step hipGraph replay (CVF_GRAPH=0 turns it off).  In a data-parallel job the two RCCL all-reduces are
        # captured inside the graph (backend nccl only; a failed capture falls back to eager launches for good)
        self._use_graphs = os.environ.get("CVF_GRAPH", "1") != "0" and (not _dist.collectives() or _dist.backend() == "nccl" or
                                                                        _dist.fused_comm() is not None)   # (the peer-to-peer kernels capture on any backend)
        # CVF_PIPELINE=1: the next batch's alignment (independent of the parameters) runs on this stream beside the
        # current step's backward kernel.  Off by default: at 20 000 frames per step it measured 133 us/step against
        # 126 serial - the two-branch graph costs more at the fork/join than the 14 us kernel it hides.
        self._ef16 = None
        # (RegAutoEncoderTask drives an inner task of this class: `_local_only` - evaluate the batch handed in on this rank alone, no
        #  cross-rank sums; `_grad_local` - leave the parameter gradient un-reduced, the caller sums its own flat gradient later)
        self._local_only = self._grad_local = False
        self._fused_fm = self._fused_k1 = self._fused_tr = None   # decided on first use: cvf_ef_[align_]fwd_metric_supported(nets, layer)
        self._side = torch.cuda.Stream(device=self.device)
        self._pipeline = os.environ.get("CVF_PIPELINE", "0") == "1"

    # ---------------------------------------------------------------- model views
    def get_reordered_eigenfunctions(self, model, cvec):
        """core.py:356-370: deep copy whose module list is re-ordered by ``cvec``."""
        new = copy.deepcopy(model)
        new.eigen_funcs = torch.nn.ModuleList([copy.deepcopy(model.eigen_funcs[int(i)]) for i in cvec])
        return new

    def colvar_model(self):
        """core.py:372-382: ``Sequential(preprocessing_layer, nets re-ordered by the last cvec)``."""
        if self._cvec is None:
            self._cvec = torch.arange(self.k)
        return _CVModel(self.preprocessing_layer, self.get_reordered_eigenfunctions(self.model, self._cvec), device=self.device)

    def reg_model(self):
        return None

    # ---------------------------------------------------------------- GPU step
    def _workspace(self, B):
        ws = self._ws.get(B)
        if ws is None:
            ws = self._ws[B] = _EFWorkspace(B, self.k, self._pp.d_r, self._flat.n, self.lag_idx, self._flat.desc, self.device,
                                            ef16=self._use_ef16())
        return ws

    def _use_ef16(self):
        """The fast layout: the 16-frames-per-wave step (csrc/ef16_front.hip, ef16_back.hip; generator and transfer-operator mode), decided once."""
        if self._ef16 is None:
            self._ef16 = (not self._pipeline and os.environ.get("CVF_NO_EF16_TRANSFER" if self.lag_idx > 0 else "CVF_NO_EF16") is None
                          and bool(_hip.lib().cvf_ef16_supported(self._flat.desc, self._pp)))
        return self._ef16

    def _sum_stats_and_tail(self, ws):
        """Data-parallel step, collective #1 + the loss tail on this rank's batch sums in ``ws.stats``:
        one launch over the peer-to-peer windows (cvf_ef_loss_dp), else an all-reduce followed by cvf_ef_loss."""
        lib, P = _hip.lib(), _hip.ptr
        comm = _dist.fused_comm()
        if comm is not None:
            self._call("cvf_ef_loss_dp", lib.cvf_ef_loss_dp, self._cfg, P(ws.stats), P(ws.loss_out), P(ws.coef), comm, _hip.stream())
        else:
            self._allreduce("allreduce_batch_sums", ws.stats)
            self._call("cvf_ef_loss", lib.cvf_ef_loss, self._cfg, P(ws.stats), P(ws.loss_out), P(ws.coef), _hip.stream())

    def _align(self, ws, slot, X, X_lag):
        """K1 of one batch (and of its lagged partner) into feature buffer ``slot``."""
        lib, P = _hip.lib(), _hip.ptr
        B, d_r = ws.B, self._pp.d_r
        self._call("cvf_align_feature_fwd", lib.cvf_align_feature_fwd, self._pp, P(X), B, P(ws._feat[slot]), None, P(ws._aux[slot]),
                   P(ws._k1_scratch[slot]), _hip.stream())
        if self.lag_idx > 0:
            feat_lag = ws._feat[slot][ws.T * d_r * _hip.TILE:]
            self._call("cvf_align_feature_fwd", lib.cvf_align_feature_fwd, self._pp, P(X_lag), B, P(feat_lag), None, None,
                       P(ws._k1_scratch[slot]), _hip.stream())

    def _forward(self, X, w, X_lag=None, w_lag=None, slot=0, aligned=False, out=None):
        """Everything up to the loss for one (local) batch; leaves loss_vec / coef on the device.
        ``aligned``: buffer ``slot`` already holds this batch's features (a previous step prefetched them).
        ``out``: fp64 device row of length 3 + 2k that receives the loss vector instead of ``ws.loss_vec`` (the training
        loops pass the step's slot of the epoch log: no copy kernel per step)."""
        lib, s, P = _hip.lib(), _hip.stream(), _hip.ptr
        B = X.shape[0]
        ws = self._workspace(B)
        ws.slot = slot
        fl, k, d_r = self._flat, self.k, self._pp.d_r
        lag = self.lag_idx
        single = not _dist.collectives() or self._local_only   # no cross-rank reduction: the loss tail runs inside the stats launch
        ws.loss_out = ws.loss_vec if out is None else out
        assert ws.loss_out.is_contiguous() and ws.loss_out.dtype == torch.float64 and ws.loss_out.numel() == 3 + 2 * k
        lv, cf = (P(ws.loss_out), P(ws.coef)) if single else (None, None)
        if self._use_ef16() and not aligned and lag > 0:
            # transfer-operator mode: coordinates of the frames and of their lagged partners -> features, y, hidden activations
            # in one launch (16 frames per wave), then the time-lagged batch sums and the loss tail
            comm = None if single else _dist.fused_comm()
            if lib.cvf_ef16_transfer_rows(B, k) > 0 and os.environ.get("CVF_NO_TRANSFER_ROWS") is None:
                # a unit and its lagged partner in one block: the units' rows of the time-lagged sums leave the front launch,
                # the finishing launch adds them (+ collective #1 in a data-parallel job) and evaluates the loss tail
                self._call("cvf_ef16_front_transfer", lib.cvf_ef16_front_transfer_rows, fl.desc, P(fl.theta), P(fl.packed), P(ws.feat),
                           self._pp, P(X), P(X_lag), B, P(ws.y), P(ws.saved), P(w), P(w_lag), P(ws.scratch), s)
                if comm is not None:
                    self._call("cvf_ef16_finish_dp", lib.cvf_ef16_finish_dp, self._cfg, B, P(ws.scratch), P(ws.stats), P(ws.loss_out),
                               P(ws.coef), comm, s)
                    return ws
                self._call("cvf_ef16_finish", lib.cvf_ef16_finish, self._cfg, B, P(ws.scratch), P(ws.stats), lv, cf, s)
                if not single:
                    self._sum_stats_and_tail(ws)                                                            # collective #1
                return ws
            self._call("cvf_ef16_front_transfer", lib.cvf_ef16_front_transfer, fl.desc, P(fl.theta), P(fl.packed), P(ws.feat),
                       self._pp, P(X), P(X_lag), B, P(ws.y), P(ws.saved), s)
            y_lag = ws.y[ws.T * k * _hip.TILE:]
            if comm is not None:   # collective #1 and the loss tail inside the finishing launch of the sums
                self._call("cvf_ef_stats_dp", lib.cvf_ef_stats_dp, self._cfg, B, P(w), P(ws.y), None, P(w_lag), P(y_lag),
                           P(ws.scratch), P(ws.stats), P(ws.loss_out), P(ws.coef), comm, s)
                return ws
            self._call("cvf_ef_stats", lib.cvf_ef_stats, self._cfg, B, P(w), P(ws.y), None, P(w_lag), P(y_lag),
                       P(ws.scratch), P(ws.stats), lv, cf, s)
            if not single:
                self._sum_stats_and_tail(ws)                                                                # collective #1
            return ws
        if self._use_ef16() and not aligned:
            # coordinates -> features, y, hidden activations, q = J A J^T g, E and the batch sums in one launch, 16 frames per
            # wave (+ the short launch that adds the units' rows and evaluates the loss tail)
            rows = lib.cvf_ef16_rows(B) > 0   # the units' rows of batch sums are added by a second, short launch
            self._call("cvf_ef16_front", lib.cvf_ef16_front, fl.desc, P(fl.theta), P(fl.packed), P(ws.feat), self._pp, P(X), B,
                       P(self._diag_coeff), P(ws.y), P(ws.saved), P(ws.q), P(ws.e), self._cfg, P(w), P(ws.scratch),
                       None if rows else P(ws.stats), lv, cf, s)
            comm = None if single else _dist.fused_comm()
            if rows and comm is not None:   # the units' rows -> sums -> collective #1 -> loss tail: one launch
                self._call("cvf_ef16_finish_dp", lib.cvf_ef16_finish_dp, self._cfg, B, P(ws.scratch), P(ws.stats), P(ws.loss_out),
                           P(ws.coef), comm, s)
                return ws
            if rows:
                self._call("cvf_ef16_finish", lib.cvf_ef16_finish, self._cfg, B, P(ws.scratch), P(ws.stats), lv, cf, s)
            if not single:
                self._sum_stats_and_tail(ws)                                                                # collective #1
            return ws
        if not ws.k1_scratch_checked:
            ws._k1_scratch = [_hip.align_scratch(self._pp, B, self.device) for _ in range(2)]
            ws.k1_scratch_checked = True
        if self._fused_fm is None:
            self._fused_fm = lag == 0 and bool(lib.cvf_ef_fwd_metric_supported(fl.desc, self._pp))
            self._fused_k1 = self._fused_fm and bool(lib.cvf_ef_align_fwd_metric_supported(fl.desc, self._pp))
        with_k1 = self._fused_k1 and not aligned   # the alignment runs inside the fused launch
        if self._fused_tr is None:                 # transfer mode: alignment + forward of both frame sets in one launch
            self._fused_tr = (lag > 0 and os.environ.get("CVF_NO_ALIGN_FWD") is None and
                              bool(lib.cvf_ef_align_fwd_metric_supported(fl.desc, self._pp)))
        with_tr = self._fused_tr and not aligned
        if not aligned and not with_k1 and not with_tr:
            self._align(ws, slot, X, X_lag)
        if self._fused_fm:   # [alignment,] nets forward, q = J A J^T g, E and the batch sums in one launch: g never leaves the chip
            name, fn = (("cvf_ef_align_fwd_metric_stats", lib.cvf_ef_align_fwd_metric_stats) if with_k1 else
                        ("cvf_ef_fwd_metric_stats", lib.cvf_ef_fwd_metric_stats))
            rows = lib.cvf_ef_fused_stats_rows(fl.desc, self._pp, B, int(with_k1))   # > 0: per-tile sums left for a short second launch
            self._call(name, fn, fl.desc, P(fl.theta), P(fl.packed), P(ws.feat),
                       self._pp, P(X), B, P(ws.aux), P(self._diag_coeff), P(ws.y), P(ws.saved), P(ws.q), P(ws.e), self._cfg,
                       P(w), P(ws.scratch), None if rows > 0 else P(ws.stats), lv, cf, s)
            if rows > 0:
                self._call("cvf_ef_stats_finish_rows", lib.cvf_ef_stats_finish_rows, self._cfg, rows, P(ws.scratch), P(ws.stats),
                           lv, cf, s)
            if not single:
                self._sum_stats_and_tail(ws)                                                                # collective #1
            return ws
        if with_tr:
            self._call("cvf_ef_align_fwd", lib.cvf_ef_align_fwd, fl.desc, P(fl.theta), P(fl.packed), P(ws.feat), self._pp, P(X),
                       P(X_lag), B, P(ws.y), P(ws.saved), s)
        else:
            self._call("cvf_ef_mlp_fwd", lib.cvf_ef_mlp_fwd, fl.desc, P(fl.theta), P(fl.packed), P(ws.feat), ws.Tt, P(ws.y),
                       P(ws.g) if lag == 0 else None, P(ws.saved), s)
        if lag == 0:   # q = J A J^T g, E, and the batch sums (K2/K3 + K5) in one launch
            self._call("cvf_metric_apply", lib.cvf_metric_apply_stats, self._pp, P(X), B, P(ws.aux), P(self._diag_coeff), k,
                       P(ws.g), P(ws.q), P(ws.e), P(ws.k1_scratch), P(self._dense), self._cfg, P(w), P(ws.y),
                       P(ws.scratch), P(ws.stats), lv, cf, s)
        else:
            y_lag = ws.y[ws.T * k * _hip.TILE:]
            self._call("cvf_ef_stats", lib.cvf_ef_stats, self._cfg, B, P(w), P(ws.y), None, P(w_lag), P(y_lag),
                       P(ws.scratch), P(ws.stats), lv, cf, s)
        if not single:
            self._sum_stats_and_tail(ws)                                                                    # collective #1
        return ws

    def _backward(self, ws, w, w_lag=None, advance=False, fuse_adam=False):
        """Parameter gradient into the flat buffer.  ``advance``: this gradient belongs to an optimiser step
        (the kernel advances the device step counter).  ``fuse_adam``: single-process training - the kernel
        that sums the per-block partial gradients applies the Adam update in the same launch."""
        lib, fl, P = _hip.lib(), self._flat, _hip.ptr
        if self._use_ef16() and self.lag_idx > 0:
            self._call("cvf_ef16_backward_transfer", lib.cvf_ef16_backward_transfer, self._cfg, fl.desc, P(fl.theta), P(fl.packed), ws.B,
                       P(w), P(w_lag), P(ws.feat), P(ws.y), P(ws.coef), P(ws.slab),
                       P(self.optimizer.step_count) if advance else None, P(ws.saved), _hip.stream())
        elif self._use_ef16():
            self._call("cvf_ef16_backward", lib.cvf_ef16_backward, self._cfg, fl.desc, P(fl.theta), P(fl.packed), ws.B, P(w),
                       P(ws.feat), P(ws.y), P(ws.q), P(ws.coef), P(ws.slab),
                       P(self.optimizer.step_count) if advance else None, P(ws.saved), _hip.stream())
        else:
            self._call("cvf_ef_backward", lib.cvf_ef_backward, self._cfg, fl.desc, P(fl.theta), P(fl.packed), ws.B, P(w), P(w_lag),
                       P(ws.feat), P(ws.y), P(ws.q) if self.lag_idx == 0 else None, P(ws.coef), P(ws.slab),
                       P(self.optimizer.step_count) if advance else None, P(ws.saved), _hip.stream())
        local = self._local_only or self._grad_local
        comm = None if local else _dist.fused_comm()
        if comm is not None:   # sum of the slab rows -> collective #2 -> (train_step) the identical Adam update: one launch
            adam = self.optimizer.fused_args() if advance else None
            self._call("cvf_slab_reduce_dp", lib.cvf_slab_reduce_dp, P(ws.slab), ws.slab_rows, fl.n, P(fl.grad), adam, comm, _hip.stream())
            return adam is not None
        adam = self.optimizer.fused_args() if fuse_adam else None
        self._call("cvf_slab_reduce", lib.cvf_slab_reduce, P(ws.slab), ws.slab_rows, fl.n, P(fl.grad), adam, _hip.stream())
        if not fuse_adam and not local:
            self._allreduce("allreduce_gradient", fl.grad)                                                # collective #2
        return adam is not None

    def train_step(self, X, w, X_lag=None, w_lag=None, slot=0, aligned=False, prefetch=None, out=None):
        """One optimisation step on device tensors; returns the device vector
        ``[loss, npl, pen, eig_1..k, cvec_1..k]`` (fp64) without synchronising the host.
        ``prefetch = (X_next, X_lag_next)``: align that batch into the other feature buffer on a side stream while this
        step's backward kernel runs; the next call then passes ``slot ^ 1, aligned=True``."""
        ws = self._forward(X, w, X_lag, w_lag, slot, aligned, out=out)
        if prefetch is not None:   # beside the backward kernel, which leaves SIMD slots and LDS free at these sizes
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._align(ws, slot ^ 1, prefetch[0], prefetch[1])
        fused = self._backward(ws, w, w_lag, advance=True, fuse_adam=not _dist.collectives())
        if not fused:
            self.optimizer.step(advance=False)
        if prefetch is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        return ws.loss_out

    # -- hipGraph replay of a whole step: batches are static (shuffle=False, core.py:472-481), so every
    #    (batch, kind) pair is captured once and replayed in all later epochs: one host call per step
    def _graph_step(self, key, fn, out_slot, takes_out=False):
        """Run ``fn()`` (a step that returns the device loss vector) and copy its result into ``out_slot``;
        captured into a hipGraph on first use when graphs are enabled."""
        def run():
            # (steps that take the slot write their loss vector there themselves; others return a tensor to copy)
            got = fn(out_slot) if takes_out else fn()
            if got is not out_slot and got.data_ptr() != out_slot.data_ptr():
                out_slot.copy_(got)

        if not self._use_graphs:
            run()
            return
        g = self._graphs.get(key)
        if g is None:
            run()                                     # eager warm-up: allocates the workspace of this batch size
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    run()
            except Exception as exc:                   # e.g. a collective that cannot be captured on this stack
                if not _dist.collectives():
                    raise
                self._use_graphs = False
                self._graphs.clear()
                torch.cuda.synchronize()
                print(f"[colvarsfinder] hipGraph capture of the data-parallel step failed ({type(exc).__name__}: {exc}); "
                      "continuing with eager launches", flush=True)
                return
            self._graphs[key] = g
            return                                     # the warm-up call already did this step's work once... see note
        self.optimizer.sync_lr()                       # the captured kernels read the learning rate from a device scalar
        g.replay()

    def _graph_call(self, key, body):
        """Run ``body()`` - any sequence of steps on static batches that writes its results into fixed device buffers, e.g.
        a whole epoch - through ONE hipGraph: captured on first use (after an eager run that allocates the workspaces),
        replayed afterwards.  One replay per epoch instead of one per step: the ~9 us the GPU idles between two graph
        launches (rocprofv3 kernel trace of bench.py) is paid once per epoch."""
        if not self._use_graphs:
            body()
            return
        g = self._graphs.get(key)
        if g is None:
            body()
            torch.cuda.current_stream().synchronize()
            self.optimizer.sync_lr()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    body()
            except Exception as exc:                   # e.g. a collective that cannot be captured on this stack
                if not _dist.collectives():
                    raise
                self._use_graphs = False
                self._graphs.clear()
                torch.cuda.synchronize()
                print(f"[colvarsfinder] hipGraph capture of the data-parallel epoch failed ({type(exc).__name__}: {exc}); "
                      "continuing with eager launches", flush=True)
                return
            self._graphs[key] = g
            return                                     # (the eager run above already did this call's work)
        self.optimizer.sync_lr()
        g.replay()

    def _dev(self, t, dtype=torch.float32):
        return None if t is None else torch.as_tensor(t).detach().to(device=self.device, dtype=dtype).contiguous()

    def loss_func(self, X, weight, X_lagged, weight_lagged):
        """core.py:387-457.  Returns ``(loss, eig_vals, non_penalty_loss, penalty, cvec)``; the parameter
        gradient of ``loss`` is obtained with :meth:`backward` (there is no autograd graph)."""
        X, weight = self._dev(X), self._dev(weight)
        X_lagged, weight_lagged = self._dev(X_lagged), self._dev(weight_lagged)
        self._flat.repack()  # the caller may have modified the parameters through the nn.Module
        ws = self._forward(X, weight, X_lagged, weight_lagged)
        self._last = (ws, weight, weight_lagged)
        v = ws.loss_vec.cpu()
        _dist.check_comm()
        k = self.k
        dt = torch.get_default_dtype()
        cvec = v[3 + k:3 + 2 * k].round().to(torch.long).numpy()
        return v[0].to(dt), v[3:3 + k].to(dt), v[1].to(dt), v[2].to(dt), cvec

    def backward(self):
        """Parameter gradient of the last :meth:`loss_func` call -> ``p.grad`` of the model's parameters."""
        ws, w, w_lag = self._last
        self._backward(ws, w, w_lag)
        for p, gv in self._flat.grad_views():
            p.grad = gv.clone()

    # ---------------------------------------------------------------- training loop
    def train(self):
        """core.py:459-566 with the trajectory resident in HBM and one host copy of the losses per epoch."""
        k, lag = self.k, self.lag_idx
        self._flat.repack()
        self._graphs = {}   # captured steps hold pointers into the previous call's batches
        ll = self._n_frames - lag
        _split(ll, self.test_ratio)                                  # core.py:465 (drawn, discarded)
        idx_train, idx_test = _split(ll, self.test_ratio)            # core.py:468
        world, rank = _dist.world(), _dist.rank()
        if world > 1:  # every rank must use rank 0's permutation
            both = torch.as_tensor(np.concatenate([idx_train, idx_test]), device=self.device)
            _dist.broadcast_(both)
            both = both.cpu().numpy()
            idx_train, idx_test = both[:len(idx_train)], both[len(idx_train):]

        def resident(idx, bs):
            """Frames (weights, lagged partners) of the permuted set `idx` that this process works on, gathered once in
            batch order, and its batches as row ranges of that copy: DataLoader(batch_size=bs, drop_last=True,
            shuffle=False), core.py:470-481.  One process: the whole set, gathered on the device from the resident
            trajectory.  Data-parallel: only this rank's slice of every global batch (_dist.shard_batches), gathered on
            the host and uploaded - the lagged partners are pre-gathered too, so a frame may live on two ranks."""
            if not self._sharded:
                it = torch.as_tensor(idx, device=self.device, dtype=torch.long)
                X, w = self._traj[it].contiguous(), self._weights[it].contiguous()
                Xl, wl = (self._traj[it + lag].contiguous(), self._weights[it + lag].contiguous()) if lag > 0 else (None, None)
                bl = [(s, s + bs) for s in range(0, len(idx) - bs + 1, bs)] if bs > 0 else []
            else:
                assert bs == 0 or bs >= world, f"batch size {bs} is smaller than the number of ranks {world}"
                pos, nb = _dist.shard_batches(len(idx), bs, rank, world)
                rows = np.asarray(idx)[pos]
                it = torch.as_tensor(rows, device=self.device, dtype=torch.long)
                X, w = _hip.upload_f32(self._traj_host.rows(rows), self.device), self._weights[it].contiguous()
                Xl, wl = ((_hip.upload_f32(self._traj_host.rows(rows + lag), self.device), self._weights[it + lag].contiguous())
                          if lag > 0 else (None, None))
                bl = [(j * nb, (j + 1) * nb) for j in range(len(pos) // nb)] if nb > 0 else []
            self.resident_bytes += sum(t.numel() * t.element_size() for t in (X, w, Xl, wl) if t is not None)
            return (X, w, Xl, wl), bl

        bs_train = min(self.batch_size, len(idx_train))
        bs_test = min(self.batch_size, len(idx_test))
        self.resident_bytes = 0 if self._sharded else self._traj.numel() * 4
        Xtr, tr_batches = resident(idx_train, bs_train)
        Xte, te_batches = resident(idx_test, bs_test)

        self.loss_list = []
        min_loss = float("inf")
        if rank == 0:
            print("\nTraining starts.\n%d epochs in total, batch sizes (train/test): %d/%d" % (self.num_epochs, bs_train, bs_test))
            print("\nTrain set:\n\t%d data, %d iterations per epoch, %d iterations in total." %
                  (len(idx_train), len(tr_batches), len(tr_batches) * self.num_epochs), flush=True)
            print("Test set:\n\t%d data, %d iterations per epoch, %d iterations in total." %
                  (len(idx_test), len(te_batches), len(te_batches) * self.num_epochs), flush=True)
        loss_names = ['loss', 'eigen_non_penalty', 'eigen_penalty'] + ['eig_%d' % (i + 1) for i in range(k)]
        nrow = 3 + 2 * k
        log_tr = torch.zeros(max(len(tr_batches), 1), nrow, device=self.device, dtype=torch.float64)
        log_te = torch.zeros(max(len(te_batches), 1), nrow, device=self.device, dtype=torch.float64)

        def sl(data, a, b):
            return tuple(None if t is None else t[a:b] for t in data)

        def on_epoch(ep, tr, te):
            """Host side of one finished epoch: loss_list (core.py:553), _cvec (core.py:515), the writer (core.py:559-561)."""
            dt = torch.get_default_dtype()
            if len(tr_batches) > 0:
                self._cvec = tr[-1, 3 + k:].round().to(torch.long).numpy()
            self.loss_list.append([tr[:, :3 + k].to(dt), te[:, :3 + k].to(dt)])
            mean_tr = self.loss_list[-1][0].mean(0) if len(tr_batches) else torch.full((3 + k,), float("nan"))
            mean_te = self.loss_list[-1][1].mean(0) if len(te_batches) else torch.full((3 + k,), float("nan"))
            for i, name in enumerate(loss_names):
                self.writer.add_scalar('%s/train' % name, mean_tr[i], ep)
                self.writer.add_scalar('%s/test' % name, mean_te[i], ep)

        elog = _AsyncEpochLog(log_tr, log_te, len(tr_batches), len(te_batches), on_epoch)

        def train_one(it):
            X, w, Xl, wl = sl(Xtr, *tr_batches[it])
            nxt = None
            if self._pipeline and it + 1 < len(tr_batches):
                Xn, _, Xln, _ = sl(Xtr, *tr_batches[it + 1])
                nxt = (Xn, Xln)
            self.train_step(X, w, Xl, wl, slot=it % 2, aligned=self._pipeline and it > 0, prefetch=nxt, out=log_tr[it])

        def test_one(it):                                          # core.py:535-551 (same loss, no update)
            X, w, Xl, wl = sl(Xte, *te_batches[it])
            self._forward(X, w, Xl, wl, out=log_te[it])

        # every step of the epoch (static batches) replays from hipGraphs: ONE graph per epoch, or - small batches on a long
        # trajectory - one per kEpochChunk steps (a graph of many thousands of kernel nodes takes seconds to instantiate)
        kEpochChunk = 256
        steps = [(train_one, it) for it in range(len(tr_batches))] + [(test_one, it) for it in range(len(te_batches))]
        chunks = [steps[c:c + kEpochChunk] for c in range(0, len(steps), kEpochChunk)]

        for epoch in _tqdm(range(self.num_epochs), disable=(rank != 0)):
            self.model.train()
            for ci, chunk in enumerate(chunks):
                self._graph_call(("epoch", ci), lambda chunk=chunk: [fn(it) for fn, it in chunk])
            elog.push(epoch)        # (the host reads the epoch's numbers later; epochs that save or plot flush first)
            saving = self.save_model_every_step > 0 and epoch % self.save_model_every_step == self.save_model_every_step - 1
            plotting = self.plot_frequency > 0 and epoch % self.plot_frequency == self.plot_frequency - 1
            if saving or plotting:
                elog.flush_all()
            if saving:
                self.save_model(epoch)
                last = float(self.loss_list[-1][0][-1, 0]) if len(tr_batches) > 0 else float("inf")
                if last < min_loss:                                             # core.py:526-528
                    min_loss = last
                    self.save_model(epoch, 'best')
            if plotting:
                if self.plot_class is not None and rank == 0:
                    self.plot_class.plot(self.colvar_model(), epoch=epoch)
        elog.flush_all()

        self.train_loss_df = pd.DataFrame(torch.cat([e[0].mean(dim=0, keepdim=True) for e in self.loss_list]).numpy(),
                                          columns=loss_names)
        self.test_loss_df = pd.DataFrame(torch.cat([e[1].mean(dim=0, keepdim=True) for e in self.loss_list]).numpy(),
                                         columns=loss_names)


class AutoEncoderTask(TrainingTask):
    """Autoencoder trained with the weighted reconstruction loss (arguments as core.py:610-625).

    The feature trajectory ``r(x)`` is computed once for all frames (core.py:635) by kernel K1 and
    stays in HBM; each step is one fused kernel (forward, weighted MSE, parameter gradient) + Adam.
    """

    def __init__(self, traj_obj, pp_layer, model, model_path, learning_rate=0.01, load_model_filename=None,
                 save_model_every_step=10, batch_size=1000, num_epochs=10, test_ratio=0.2, optimizer_name='Adam',
                 device=torch.device('cuda'), plot_class=None, plot_frequency=0, verbose=True, debug_mode=True):
        super().__init__(traj_obj, pp_layer, model, model_path, learning_rate, load_model_filename, save_model_every_step,
                         model.encoded_dim, batch_size, num_epochs, test_ratio, optimizer_name, device, plot_class,
                         plot_frequency, verbose, debug_mode)
        assert isinstance(model, AutoEncoder), 'model must be an object of the class AutoEncoder'
        self.init_model_and_optimizer()
        self._traj_host = _HostFrames(traj_obj.trajectory)
        self._n_frames = int(self._traj_host.shape[0])
        self._weights = torch.as_tensor(np.asarray(traj_obj.weights)).to(device=self.device, dtype=torch.float32).contiguous()
        # pp_layer: the alignment + feature kernel (AlignFeatureLayer / Identity) or ANY torch module (core.py:65,122,635): this
        # task applies it exactly once, to build the feature trajectory - a foreign module is run there with torch, on the
        # device, and the training step (cvf_ae_step on the resident feature rows) is the same either way
        self._foreign_pp = not isinstance(self.preprocessing_layer, (torch.nn.Identity, AlignFeatureLayer))
        if self._foreign_pp:
            with torch.no_grad():
                probe = self.preprocessing_layer(torch.as_tensor(self._traj_host.rows(np.arange(1))).to(device=self.device, dtype=torch.float32))
            pp = types.SimpleNamespace(d_r=int(probe.reshape(1, -1).shape[1]))
            self._pp = pp
        else:
            self._pp = pp = self._pp_desc(int(np.prod(self._traj_host.shape[1:])))
        # core.py:635: the feature trajectory r(x) of ALL frames, once.  In a data-parallel job (one process per GPU) each
        # rank computes the features of its own rows only, in train(), once the split is known.
        self._sharded = _dist.world() > 1
        self._feature_traj = None if self._sharded else self._features(self._traj_host.all())
        self.resident_bytes = 0 if self._sharded else self._feature_traj.numel() * 4
        assert pp.d_r == self._flat.desc.dims[0] == self._flat.desc.dims[self._flat.desc.n_layers], \
            'autoencoder input/output width must equal the feature dimension'
        if self.verbose:
            print('\nShape of trajectory data array:\n {}'.format((self._n_frames, pp.d_r)), flush=True)
        self._out2 = torch.zeros(3, device=self.device, dtype=torch.float64)
        self._scratch = {}

    def _features(self, rows):
        """K1 over host frames -> row-major feature rows resident in HBM."""
        X = _hip.upload_f32(rows, self.device)
        n = X.shape[0]
        if self._foreign_pp:     # a module this package has no kernel for: torch, on the device, in bounded chunks, once
            out = torch.empty(n, self._pp.d_r, device=self.device, dtype=torch.float32)
            with torch.no_grad():
                for s0 in range(0, n, 65536):
                    out[s0:s0 + 65536] = self.preprocessing_layer(X[s0:s0 + 65536]).reshape(-1, self._pp.d_r)
            return out
        out = torch.empty(n, self._pp.d_r, device=self.device, dtype=torch.float32)
        if n > 0:
            _hip.check(_hip.lib().cvf_align_feature_fwd(self._pp, _hip.ptr(X), n, None, _hip.ptr(out), None,
                                                        _hip.ptr(_hip.align_scratch(self._pp, n, self.device)), _hip.stream()),
                       "cvf_align_feature_fwd")
        return out

    def colvar_model(self):
        """core.py:640-647."""
        return _CVModel(self.preprocessing_layer, self.model.encoder, device=self.device)

    def reg_model(self):
        return None

    def _step(self, feat, idx, w, with_grad, inv_wsum=None, advance=False, fuse_adam=False):
        """One fused kernel: forward, weighted MSE, parameter gradient (+ Adam when ``fuse_adam``).
        ``inv_wsum`` = 1 / (global sum of the batch weights); it does not depend on the model, so ``train``
        computes it once per (static) batch."""
        lib, fl = _hip.lib(), self._flat
        B = w.shape[0]
        sc = self._scratch.get(B)
        if sc is None:
            sc = self._scratch[B] = torch.empty(lib.cvf_ae_scratch_floats(fl.desc, B), device=self.device, dtype=torch.float32)
        if inv_wsum is None:
            wsum = w.sum(dtype=torch.float64)
            _dist.allreduce_sum_(wsum)
            inv_wsum = 1.0 / float(wsum)
        adam = self.optimizer.fused_args() if (fuse_adam and with_grad) else None
        _hip.check(lib.cvf_ae_step(fl.desc, _hip.ptr(fl.theta), _hip.ptr(feat), _hip.ptr(idx), B, _hip.ptr(w),
                                   inv_wsum, _hip.ptr(sc), _hip.ptr(self._out2),
                                   _hip.ptr(fl.grad) if with_grad else None,
                                   _hip.ptr(self.optimizer.step_count) if advance else None, adam, _hip.stream()), "cvf_ae_step")
        if _dist.world() == 1:
            # (the kernel left the ratio beside the two sums: no arithmetic launches on the host side of a step; the caller
            #  copies the value out before the next step overwrites it)
            if with_grad and adam is None and advance:
                self.optimizer.step(advance=False)
            return self._out2[2]
        out = self._out2[:2].clone()
        _dist.allreduce_sum_(out)
        if with_grad and adam is None:
            self._allreduce("allreduce_gradient", fl.grad)
            if advance:
                self.optimizer.step(advance=False)
        return out[0] / out[1]

    def weighted_MSE_loss(self, X, weight):
        """core.py:652-666 on a feature batch ``X [B, d_r]``; ``backward()`` afterwards fills ``p.grad``."""
        X = torch.as_tensor(X).detach().to(device=self.device, dtype=torch.float32).contiguous()
        weight = torch.as_tensor(weight).detach().to(device=self.device, dtype=torch.float32).contiguous()
        loss = self._step(X, None, weight, with_grad=True)
        return loss.to(torch.get_default_dtype(), copy=True)

    def backward(self):
        pos = 0
        for p in self.model.parameters():
            p.grad = self._flat.grad[pos:pos + p.numel()].view(p.shape).clone()
            pos += p.numel()

    def train(self):
        """core.py:668-744."""
        n = self._n_frames
        idx_train, idx_test = _split(n, self.test_ratio)             # core.py:672 (one draw)
        world, rank = _dist.world(), _dist.rank()
        if world > 1:
            both = torch.as_tensor(np.concatenate([idx_train, idx_test]), device=self.device)
            _dist.broadcast_(both)
            both = both.cpu().numpy()
            idx_train, idx_test = both[:len(idx_train)], both[len(idx_train):]
        bs_train, bs_test = min(self.batch_size, len(idx_train)), min(self.batch_size, len(idx_test))

        def resident(idx, bs):
            """(feature rows, per-batch row indices or None, weights, batches) of the permuted set `idx` for this process.
            One process: batches index the resident feature trajectory.  Data-parallel: only this rank's slice of every
            global batch is uploaded and run through K1 (_dist.shard_batches); its batches are contiguous row ranges."""
            if not self._sharded:
                it = torch.as_tensor(idx, device=self.device, dtype=torch.long)
                bl = [(s, s + bs) for s in range(0, len(idx) - bs + 1, bs)] if bs > 0 else []
                return self._feature_traj, it, self._weights[it].contiguous(), bl
            assert bs == 0 or bs >= world, f"batch size {bs} is smaller than the number of ranks {world}"
            pos, nb = _dist.shard_batches(len(idx), bs, rank, world)
            rows = np.asarray(idx)[pos]
            feat = self._features(self._traj_host.rows(rows))
            self.resident_bytes += feat.numel() * 4
            wv = self._weights[torch.as_tensor(rows, device=self.device, dtype=torch.long)].contiguous()
            return feat, None, wv, ([(j * nb, (j + 1) * nb) for j in range(len(pos) // nb)] if nb > 0 else [])

        if self._sharded:
            self.resident_bytes = 0
        ftr, itr, wtr, tr_batches = resident(idx_train, bs_train)
        fte, ite, wte, te_batches = resident(idx_test, bs_test)
        if self._sharded:
            self._feature_traj = ftr      # (this rank's training rows; the reference's attribute holds all frames)

        def rows_of(feat, it, a, b):
            return (feat, it[a:b]) if it is not None else (feat[a:b], None)

        def inv_wsums(wv, bl):
            # batches are static (shuffle=False): their weight sums are known before the first step
            if not bl:
                return []
            sums = torch.stack([wv[a:b].sum(dtype=torch.float64) for a, b in bl])
            _dist.allreduce_sum_(sums)
            return [1.0 / float(v) for v in sums.cpu()]

        iw_tr, iw_te = inv_wsums(wtr, tr_batches), inv_wsums(wte, te_batches)
        self.loss_list = []
        min_loss = float("inf")
        if rank == 0:
            print("\nTraining starts.\n%d epochs in total, batch sizes (train/test): %d/%d" % (self.num_epochs, bs_train, bs_test))
            print("\nTrain set:\n\t%d data, %d iterations per epoch, %d iterations in total." %
                  (len(idx_train), len(tr_batches), len(tr_batches) * self.num_epochs), flush=True)
            print("Test set:\n\t%d data, %d iterations per epoch, %d iterations in total." %
                  (len(idx_test), len(te_batches), len(te_batches) * self.num_epochs), flush=True)
        log_tr = torch.zeros(max(len(tr_batches), 1), device=self.device, dtype=torch.float64)
        log_te = torch.zeros(max(len(te_batches), 1), device=self.device, dtype=torch.float64)

        def on_epoch(ep, tr, te):
            dt = torch.get_default_dtype()
            tr, te = tr.to(dt), te.to(dt)
            self.loss_list.append([tr, te])                                        # core.py:736
            self.writer.add_scalar('Loss/train', tr.mean() if len(tr) else float("nan"), ep)   # core.py:738-739
            self.writer.add_scalar('Loss/test', te.mean() if len(te) else float("nan"), ep)

        elog = _AsyncEpochLog(log_tr, log_te, len(tr_batches), len(te_batches), on_epoch)
        for epoch in _tqdm(range(self.num_epochs), disable=(rank != 0)):
            self.model.train()
            for it, (a, b) in enumerate(tr_batches):
                log_tr[it] = self._step(*rows_of(ftr, itr, a, b), wtr[a:b], True, iw_tr[it], advance=True,
                                        fuse_adam=(world == 1))
            self.model.eval()
            for it, (a, b) in enumerate(te_batches):
                log_te[it] = self._step(*rows_of(fte, ite, a, b), wte[a:b], False, iw_te[it])
            elog.push(epoch)
            saving = self.save_model_every_step > 0 and epoch % self.save_model_every_step == self.save_model_every_step - 1
            plotting = self.plot_frequency > 0 and epoch % self.plot_frequency == self.plot_frequency - 1
            if saving or plotting:
                elog.flush_all()
            if saving:
                self.save_model(epoch)
                last = float(self.loss_list[-1][0][-1]) if len(tr_batches) else float("inf")
                if last < min_loss:
                    min_loss = last
                    self.save_model(epoch, 'best')
            if plotting:
                if self.plot_class is not None and rank == 0:
                    self.plot_class.plot(self.colvar_model(), epoch=epoch)
        elog.flush_all()
        self.train_loss_df = pd.DataFrame(torch.cat([e[0].mean(dim=0, keepdim=True) for e in self.loss_list]).numpy(),
                                          columns=['loss'])
        self.test_loss_df = pd.DataFrame(torch.cat([e[1].mean(dim=0, keepdim=True) for e in self.loss_list]).numpy(),
                                         columns=['loss'])


# ----------------------------------------------------------------------------------------------------------------------
# RegAutoEncoderTask (core.py:746-1217)
# ----------------------------------------------------------------------------------------------------------------------
class _RegFlatParams:
    """A :class:`RegAutoEncoder` as ONE chain over a flat fp32 buffer: the encoder's layers, then the decoder and
    the K regulariser nets side by side.  Merged layer m maps ``[dec_m | reg_1,m | .. | reg_K,m]`` to the same of
    m + 1 with a block-structured matrix (the first one: every block reads the whole latent vector), so the last
    layer emits ``[reconstruction | y_1..y_K]`` and one pass of the chain kernel serves ``forward_ae`` and
    ``forward_reg`` (nn.py:174-198).  Every module parameter aliases its block of the buffer (a strided view for the
    block-diagonal layers); ``mask`` is 1 on real parameters, 0 on structural zeros and frozen (encoder) entries.
    """

    def __init__(self, model, device, freeze_encoder=False):
        from .nn import _chain_layers
        model.to(device=device, dtype=torch.float32)
        enc, dec = _chain_layers(model.encoder), _chain_layers(model.decoder)
        regs = [_chain_layers(r) for r in model.reg] if model.num_reg > 0 else []
        K, Ld = len(regs), len(dec)
        for r in regs:
            if len(r) != Ld or any(ra != da for (_, ra), (_, da) in zip(r, dec)):
                raise NotImplementedError("the MI355X path runs decoder and regulariser nets side by side: they must have the "
                                          "same number of layers (as in the reference's notebooks)")
            if r[-1][0].out_features != 1:
                raise NotImplementedError("regulariser nets must be scalar-valued")
        assert len(enc) + Ld <= _hip.MAX_LAYERS, f"at most {_hip.MAX_LAYERS} layers (encoder + decoder)"
        # sizes
        shapes = [(lin.out_features, lin.in_features, act) for lin, act in enc]
        for m in range(Ld):
            fin = dec[m][0].in_features if m == 0 else dec[m][0].in_features + sum(r[m][0].in_features for r in regs)
            fout = dec[m][0].out_features + sum(r[m][0].out_features for r in regs)
            shapes.append((fout, fin, dec[m][1]))
        self.n = sum(fo * fi + fo for fo, fi, _ in shapes)
        self.theta = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.mask = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.packed = None
        self.K = K
        d = _hip.MLPDesc()
        d.n_nets, d.n_layers, d.n_params = 1, len(shapes), self.n
        self.views = []   # (parameter, view of theta, view of grad)

        def alias(p, region, r0, r1, c0=None, c1=None, frozen=False):
            tv = region(self.theta)[r0:r1] if c0 is None else region(self.theta)[r0:r1, c0:c1]
            gv = region(self.grad)[r0:r1] if c0 is None else region(self.grad)[r0:r1, c0:c1]
            mv = region(self.mask)[r0:r1] if c0 is None else region(self.mask)[r0:r1, c0:c1]
            tv.copy_(p.data)
            mv.fill_(0.0 if frozen else 1.0)
            p.data = tv
            self.views.append((p, tv, gv))

        pos = 0
        for l, (fo, fi, act) in enumerate(shapes):
            d.dims[l], d.dims[l + 1], d.act[l] = fi, fo, int(act)
            d.w_off[0][l], d.b_off[0][l] = pos, pos + fo * fi
            wreg = (lambda t, a=pos, fo=fo, fi=fi: t[a:a + fo * fi].view(fo, fi))
            breg = (lambda t, a=pos + fo * fi, fo=fo: t[a:a + fo])
            if l < len(enc):
                lin = enc[l][0]
                alias(lin.weight, wreg, 0, fo, 0, fi, frozen=freeze_encoder)
                alias(lin.bias, breg, 0, fo, frozen=freeze_encoder)
            else:
                m = l - len(enc)
                r0 = c0 = 0
                for chain in [dec] + regs:
                    lin = chain[m][0]
                    cw = lin.in_features
                    if m == 0:
                        alias(lin.weight, wreg, r0, r0 + lin.out_features, 0, cw)       # every block reads the latent vector
                    else:
                        alias(lin.weight, wreg, r0, r0 + lin.out_features, c0, c0 + cw)
                        c0 += cw
                    alias(lin.bias, breg, r0, r0 + lin.out_features)
                    r0 += lin.out_features
            pos += fo * fi + fo
        self.desc = d

    def repack(self):
        pass

    def grad_views(self):
        return [(p, gv) for p, _, gv in self.views]


class _EncGradPenalty:
    """``eta[0]``: ``reg_enc_grad_loss`` (core.py:896-910) = sum_i (1/W) sum_b w_b |d enc_i / d r (r_b)|^2, the derivative
    taken with respect to the FEATURES r = pp(x) (``Y.requires_grad_()`` there), and its gradient with respect to the
    encoder's parameters (second order through the encoder).

    That is the energy term of the eigenfunction task's generator mode with an identity preprocessing layer and
    ``diag_coeff = 1``, for k "virtual" scalar nets that share the encoder's hidden layers and end in row i of its last layer.
    So it runs on the eigenfunction kernels: the encoder's parameters are gathered into the k-net layout, ``cvf_ef_mlp_fwd``
    (y, g = dy/dr) + ``cvf_metric_apply_stats`` (E_i = |g_i|^2 and the batch sums) give the term, ``cvf_ef_backward`` with the
    coefficient vector [0 .. eta_0 / W on the E entries .. 0] + ``cvf_slab_reduce`` give d(eta_0 term)/d(virtual parameters), and
    the k copies of the shared layers are added back (fixed order) into the encoder's block of the flat gradient."""

    def __init__(self, task):
        from .nn import _chain_layers
        enc = _chain_layers(task.model.encoder)
        self.task, self.k, self.L = task, task.k, len(enc)
        fl, dev, k, L = task._flat, task.device, task.k, len(enc)
        dims = [enc[0][0].in_features] + [lin.out_features for lin, _ in enc]
        assert dims[0] == task.tot_dim, \
            (f"eta[0] > 0: the gradient-norm penalty reshapes the encoder's input gradient to [-1, {task.tot_dim}] (core.py:907), "
             f"so the preprocessing layer must emit {task.tot_dim} features, not {dims[0]}")
        act0 = enc[0][1] if L >= 2 else 0
        if L < 2 or act0 == 0 or [a for _, a in enc] != [act0] * (L - 1) + [0]:
            raise NotImplementedError("eta[0] on MI355X: the encoder must be Linear + activation layers (one activation) ending in a Linear layer")
        d = _hip.MLPDesc()
        vd = dims[:-1] + [1]
        d.n_nets, d.n_layers = k, L
        for l in range(L):
            d.dims[l], d.dims[l + 1], d.act[l] = vd[l], vd[l + 1], (act0 if l < L - 1 else 0)
        # virtual net i = [shared layers 0..L-2 | row i of the last weight | entry i of the last bias]
        ed = fl.desc
        n_shared = ed.w_off[0][L - 1] - ed.w_off[0][0]
        assert ed.w_off[0][0] == 0 and ed.b_off[0][L - 2] + dims[L - 1] == n_shared   # the encoder opens the flat buffer
        h = dims[L - 1]
        per = n_shared + h + 1
        src, pos = [], 0
        for i in range(k):
            for l in range(L - 1):
                d.w_off[i][l], d.b_off[i][l] = pos + ed.w_off[0][l], pos + ed.b_off[0][l]
            d.w_off[i][L - 1], d.b_off[i][L - 1] = pos + n_shared, pos + n_shared + h
            src += list(range(n_shared)) + list(range(ed.w_off[0][L - 1] + i * h, ed.w_off[0][L - 1] + (i + 1) * h)) + [ed.b_off[0][L - 1] + i]
            pos += per
        d.n_params = pos
        n_pack = _hip.lib().cvf_ef_pack_floats(d)
        if k > _hip.MAX_NETS or n_pack <= 0:
            raise NotImplementedError(
                f"eta[0] on MI355X runs on the eigenfunction kernels: encoder widths {dims} need 1 to 3 equal hidden layers of one of "
                f"{_hip.EF_HIDDEN_WIDTHS} units and at most {_hip.MAX_NETS} latent components")
        self.desc, self.n, self.per, self.n_shared, self.h = d, pos, per, n_shared, h
        self.last_w, self.last_b = int(ed.w_off[0][L - 1]), int(ed.b_off[0][L - 1])
        self.src = torch.tensor(src, device=dev, dtype=torch.long)
        self.theta = torch.zeros(pos, device=dev, dtype=torch.float32)
        self.packed = torch.zeros(n_pack, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(pos, device=dev, dtype=torch.float32)
        self.ones = torch.ones(dims[0], device=dev, dtype=torch.float32)
        self.pp = identity_desc(dims[0])
        cfg = _hip.EFCfg()
        cfg.k, cfg.lag_idx, cfg.sort_eigvals, cfg.alpha, cfg.beta, cfg.dt = k, 0, 0, 0.0, 1.0, 1.0
        for i in range(k):
            cfg.eig_w[i] = 1.0
        self.cfg, self.d_r = cfg, dims[0]
        self._ws = {}

    def _workspace(self, B):
        ws = self._ws.get(B)
        if ws is None:
            lib, k, d_r, dev = _hip.lib(), self.k, self.d_r, self.task.device
            T = _hip.ntiles(B)
            f32, f64 = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.float64)
            n_saved = lib.cvf_ef_saved_floats(self.desc, T)
            rows = lib.cvf_ef_backward_slab_rows(T)
            ws = dict(T=T, feat=torch.empty(T * d_r * _hip.TILE, **f32), y=torch.empty(T * k * _hip.TILE, **f32),
                      g=torch.empty(T * k * d_r * _hip.TILE, **f32), q=torch.empty(T * k * d_r * _hip.TILE, **f32),
                      e=torch.empty(T * k * _hip.TILE, **f32),
                      scratch=torch.zeros(lib.cvf_metric_stats_scratch_doubles(B, k), **f64),
                      stats=torch.zeros(lib.cvf_ef_nstats(k, 0), **f64), coef=torch.zeros(4 * k + k * k, **f64),
                      saved=torch.empty(n_saved, **f32) if n_saved > 0 else None, rows=rows,
                      slab=torch.empty(rows * self.n, **f32))
            self._ws[B] = ws
        return ws

    def run(self, rows, w, eta0, with_grad, dp=False):
        """``rows``: [B, d_r] feature rows of the batch (contiguous), ``w``: [B].  Returns the term (0-dim fp64 device tensor);
        with ``with_grad`` adds d(eta0 * term)/d(encoder parameters) to the encoder's block of the task's flat gradient.
        ``dp``: the batch is this rank's slice of a global batch - the batch sums are added across the ranks, the gradient
        contribution stays this rank's share (the caller sums the flat gradient)."""
        task, lib, P, s = self.task, _hip.lib(), _hip.ptr, _hip.stream()
        fl, k = task._flat, self.k
        B = int(rows.shape[0])
        ws = self._workspace(B)
        torch.index_select(fl.theta, 0, self.src, out=self.theta)
        task._call("cvf_ef_pack", lib.cvf_ef_pack, self.desc, P(self.theta), P(self.packed), s)
        task._call("cvf_align_feature_fwd", lib.cvf_align_feature_fwd, self.pp, P(rows), B, P(ws["feat"]), None, None, None, s)
        task._call("cvf_ef_mlp_fwd", lib.cvf_ef_mlp_fwd, self.desc, P(self.theta), P(self.packed), P(ws["feat"]), ws["T"], P(ws["y"]),
                   P(ws["g"]), P(ws["saved"]), s)
        task._call("cvf_metric_apply", lib.cvf_metric_apply_stats, self.pp, P(rows), B, None, P(self.ones), k, P(ws["g"]), P(ws["q"]),
                   P(ws["e"]), None, None, self.cfg, P(w), P(ws["y"]), P(ws["scratch"]), P(ws["stats"]), None, None, s)
        st = ws["stats"]                                   # [W, S1(k), S2(i<=j), E(k)]
        if dp:
            task._allreduce("allreduce_batch_sums", st)
        e0 = 1 + k + k * (k + 1) // 2
        term = st[e0:e0 + k].sum() / st[0]
        if with_grad:
            ge = ws["coef"][k + k * k:2 * k + k * k]       # [gS1(k), gS2(k*k), gE(k), ...]: d(eta0 term)/dE_i = eta0 / W
            ge.copy_((eta0 / st[0]).expand(k))
            task._call("cvf_ef_backward", lib.cvf_ef_backward, self.cfg, self.desc, P(self.theta), P(self.packed), B, P(w), None,
                       P(ws["feat"]), P(ws["y"]), P(ws["q"]), P(ws["coef"]), P(ws["slab"]), None, P(ws["saved"]), s)
            task._call("cvf_slab_reduce", lib.cvf_slab_reduce, P(ws["slab"]), ws["rows"], self.n, P(self.grad), None, s)
            gv = self.grad.view(k, self.per)
            fl.grad[:self.n_shared] += gv[:, :self.n_shared].sum(0)
            fl.grad[self.last_w:self.last_w + k * self.h].view(k, self.h).add_(gv[:, self.n_shared:self.n_shared + self.h])
            fl.grad[self.last_b:self.last_b + k] += gv[:, self.n_shared + self.h]
        return term


class _RegGenerator:
    """The eigenfunction regulariser in GENERATOR mode (``gamma`` with ``lag_tau_reg = 0``, the constructor's default;
    core.py:990,1008-1022): y_i = reg_i(encoder(r(x))) differentiated with respect to the coordinates x, the same loss as
    :class:`EigenFunctionTask`'s generator mode with ``diag_coeff = 1`` (core.py:851).

    The K chains reg_i o encoder ARE scalar nets d_r -> .. -> 1: the encoder's hidden layers (shared), one merged layer
    (the encoder's last Linear has no activation, so it folds into the regulariser's first: W = W_r1 W_e, b = W_r1 b_e + b_r1),
    the regulariser's remaining layers.  They are laid out as an :class:`EigenFunctions` model of K nets, every hidden layer
    zero-padded to one kernel width, and handed to an inner ``EigenFunctionTask`` - so the step runs on that task's kernels
    (alignment derivative, forward, q = J J^T g, batch sums, loss tail, backward), whichever it picks for the shape.  The
    gradient with respect to the virtual parameters goes back to the module's parameters through the (tiny) map that built
    them, by autograd on the host-side torch graph: shared layers summed over i, the merged layer by the product rule."""

    def __init__(self, task, beta):
        from .nn import _chain_layers
        m = task.model
        self.task, self.K = task, m.num_reg
        self.enc = _chain_layers(m.encoder)
        self.regs = [_chain_layers(r) for r in m.reg]
        Le, Lr = len(self.enc), len(self.regs[0])
        act0 = self.regs[0][0][1] if Lr >= 2 else 0
        ok = act0 != 0 and [a for _, a in self.enc] == [act0] * (Le - 1) + [0] and all(
            len(r) == Lr and [a for _, a in r] == [act0] * (Lr - 1) + [0] and r[-1][0].out_features == 1 for r in self.regs)
        if not ok or Lr < 2:
            raise NotImplementedError("generator-mode regulariser on MI355X: encoder and regulariser nets must be Linear + activation "
                                      "chains (one activation) ending in a Linear layer, the regulariser nets with at least one hidden layer")
        d_r = self.enc[0][0].in_features
        hidden = [lin.out_features for lin, _ in self.enc[:-1]] + [lin.out_features for lin, _ in self.regs[0][:-1]]
        for r in self.regs:
            assert [lin.out_features for lin, _ in r] == [lin.out_features for lin, _ in self.regs[0]], "regulariser nets differ in shape"
        widths = [w for w in _hip.ef_widths(len(hidden)) if w >= max(hidden)] if 1 <= len(hidden) <= 5 else []
        if not widths or self.K > _hip.MAX_NETS:
            raise NotImplementedError(
                f"generator-mode regulariser on MI355X: the chain regulariser o encoder has hidden widths {hidden}; the eigenfunction "
                f"kernels take 1 to 5 hidden layers of at most {max(_hip.EF_HIDDEN_WIDTHS)} units and at most {_hip.MAX_NETS} nets "
                "(use lag_tau_reg > 0 - the transfer operator - for other shapes)")
        self.H = H = widths[0]
        from .nn import ACT_SIGMOID, ACT_SOFTPLUS
        if act0 in (ACT_SIGMOID, ACT_SOFTPLUS) and any(h != H for h in hidden):   # (zero padding needs act(0) = 0)
            raise NotImplementedError(f"generator-mode regulariser on MI355X with Sigmoid / Softplus: every hidden width of the chain "
                                      f"{hidden} must be one kernel width (no zero padding)")
        self.pdims = [d_r] + [H] * len(hidden) + [1]
        act_module = next(mod for mod in m.reg[0]._modules.values() if not isinstance(mod, torch.nn.Linear))
        with torch.random.fork_rng(devices=[]):   # (the virtual nets' initial values are overwritten: leave the caller's RNG stream alone)
            model = EigenFunctions(self.pdims, self.K, copy.deepcopy(act_module))
        tok = np.zeros((4,) + tuple(task._traj_host.shape[1:]), dtype=np.float32)

        class _Tok:
            trajectory, weights, dt, n_frames = tok, np.ones(4), task.traj_dt, 4

        if isinstance(task.preprocessing_layer, AlignFeatureLayer):   # (a frame the alignment accepts: the reference itself)
            _Tok.trajectory = np.zeros_like(tok) + np.random.RandomState(0).normal(size=tok.shape[1:]).astype(np.float32)
        g0, g1 = float(task.gamma[0]), float(task.gamma[1])
        import tempfile
        # (its own scratch log directory: a second SummaryWriter must not open event files in the user's model_path)
        self._logdir = tempfile.mkdtemp(prefix="cvf_reg_generator_")
        self.inner = EigenFunctionTask(_Tok, task.preprocessing_layer, model, self._logdir, g1 / g0, [float(v) for v in task._eig_w],
                                       diag_coeff=None, beta=beta, lag_tau=0, learning_rate=task.learning_rate, k=self.K,
                                       batch_size=task.batch_size, device=task.device, verbose=False, save_model_every_step=0)
        try:   # nothing is ever logged or stepped through the inner task: close its writer, drop the directory
            close = getattr(self.inner.writer, "close", None)
            if close is not None:
                close()
            import shutil
            shutil.rmtree(self._logdir, ignore_errors=True)
        except Exception:
            pass
        self.n = self.inner._flat.n

    def _params(self):
        ps = [p for lin, _ in self.enc for p in (lin.weight, lin.bias)]
        for r in self.regs:
            ps += [p for lin, _ in r for p in (lin.weight, lin.bias)]
        return ps

    def _theta(self):
        """The virtual parameters as a (differentiable) function of the module's: flat, in the inner model's order."""
        pd, out = self.pdims, []
        We, be = self.enc[-1][0].weight, self.enc[-1][0].bias
        for i in range(self.K):
            r = self.regs[i]
            layers = [(lin.weight, lin.bias) for lin, _ in self.enc[:-1]]
            layers.append((r[0][0].weight @ We, r[0][0].weight @ be + r[0][0].bias))
            layers += [(lin.weight, lin.bias) for lin, _ in r[1:]]
            for l, (W, b) in enumerate(layers):
                fo, fi = pd[l + 1], pd[l]
                out.append(torch.nn.functional.pad(W, (0, fi - W.shape[1], 0, fo - W.shape[0])).reshape(-1))
                out.append(torch.nn.functional.pad(b, (0, fo - b.shape[0])))
        return torch.cat(out)

    def forward(self, X, w, with_grad, dp=False):
        """Loss terms of the batch ``X`` (raw coordinates on the device): returns the inner task's device loss vector
        ``[npl + (gamma_1/gamma_0) pen, npl, pen, eig_1..K sorted, cvec]`` (fp64).  ``dp``: ``X`` is this rank's slice of a global
        batch - the inner task adds its batch sums across the ranks (collective #1) and leaves its gradient local; else the
        batch is evaluated on the calling rank alone."""
        inner = self.inner
        inner._local_only, inner._grad_local = not dp, True
        with torch.enable_grad() if with_grad else torch.no_grad():
            self._tv = self._theta()
        inner._flat.theta.copy_(self._tv.detach())
        inner._flat.repack()
        self._ws = inner._forward(X, w)
        return self._ws.loss_out

    def backward(self, w, scale):
        """Adds ``scale`` * d(npl + (gamma_1/gamma_0) pen)/d(parameter) to the module parameters' slices of the flat gradient."""
        inner, task = self.inner, self.task
        inner._backward(self._ws, w, advance=False, fuse_adam=False)
        views = {id(p): gv for p, _, gv in task._flat.views}
        ps = [p for p in self._params() if not (task.freeze_encoder and any(p is q for lin, _ in self.enc for q in (lin.weight, lin.bias)))]
        grads = torch.autograd.grad(self._tv, ps, grad_outputs=inner._flat.grad * scale, allow_unused=True)
        for p, g in zip(ps, grads):
            if g is not None:
                views[id(p)].add_(g)


class RegAutoEncoderTask(TrainingTask):
    """Regularised autoencoder (arguments, defaults and attributes as core.py:792-816).

    Built on the MI355X path: the time-lagged reconstruction loss (``alpha``, ``lag_tau_ae``, core.py:883-885) and the
    transfer-operator eigenfunction regulariser (``gamma``, ``lag_tau_reg > 0``, core.py:973-1036) - the configurations
    of the reference's notebooks (2d.ipynb:743-760, main.ipynb:452-458) - with ``freeze_encoder``, the variance /
    covariance penalties on the latent vector (``eta[1]``, ``eta[2]``, core.py:912-971) and the gradient-norm penalty of the
    encoder (``eta[0]``, core.py:887-910: :class:`_EncGradPenalty`, on the eigenfunction task's kernels) and the
    regulariser in generator mode (``lag_tau_reg = 0``, core.py:990,1008-1022: :class:`_RegGenerator`, an inner
    :class:`EigenFunctionTask` over the chains regulariser o encoder).

    A step is three launches + the reduction: ``cvf_regae_forward`` (one chain: encoder, then decoder and regulariser
    nets side by side; y on the batch's frames and on their lagged partners, reconstruction error), ``cvf_ef_stats``
    (batch sums, loss tail, d loss / d sum - the eigenfunction task's own kernel), ``cvf_regae_backward`` (forward
    again, output gradients, parameter gradient on the matrix cores, fixed-order reduction, Adam).
    """

    def __init__(self, traj_obj, pp_layer, model, model_path, eig_weights=[], learning_rate=0.01, load_model_filename=None,
                 save_model_every_step=10, batch_size=1000, num_epochs=10, test_ratio=0.2, optimizer_name='Adam', alpha=1.0,
                 gamma=[0.0, 0.0], eta=[0.0, 0.0, 0.0], lag_tau_ae=0, lag_tau_reg=0, beta=1.0, device=torch.device('cuda'),
                 plot_class=None, plot_frequency=0, freeze_encoder=False, verbose=True, debug_mode=True):
        super().__init__(traj_obj, pp_layer, model, model_path, learning_rate, load_model_filename, save_model_every_step,
                         model.encoded_dim, batch_size, num_epochs, test_ratio, optimizer_name, device, plot_class,
                         plot_frequency, verbose, debug_mode)
        assert isinstance(model, RegAutoEncoder), 'model must be an object of the class RegAutoEncoder'
        assert model.num_reg == len(eig_weights), 'number of weights does not match the number of eigenfunctions!'
        self.alpha, self.gamma, self.eta = alpha, gamma, eta
        self.num_reg = model.num_reg
        self._eps = 1e-5
        self._eig_w = eig_weights
        self._cvec = None
        self.freeze_encoder = freeze_encoder
        self.traj_dt = traj_obj.dt
        lag_ae_idx, lag_idx = lag_tau_ae / self.traj_dt, lag_tau_reg / self.traj_dt
        assert abs(lag_ae_idx - int(lag_ae_idx)) < 1e-6 and abs(lag_idx - int(lag_idx)) < 1e-6, \
            f'lag-times ({lag_tau_ae}, {lag_tau_reg}) not divisable by the timestep {self.traj_dt} of the trajectory'
        self.lag_ae_idx, self.lag_idx = int(lag_ae_idx), int(lag_idx)
        self._use_reg = self.gamma[0] + self.gamma[1] > self._eps
        if self._use_reg:
            assert self.num_reg > 0, 'number of eigenfunctions must be positive!'
            if self.gamma[0] <= self._eps:
                raise NotImplementedError("RegAutoEncoderTask on MI355X: gamma[0] must be positive when gamma[1] is")
            self._beta = beta
        self._use_enc = max(self.eta[1], self.eta[2]) > self._eps
        # Data-parallel job (one process per GPU): the batch's frames are split over the ranks, the three kinds of batch sums
        # (reconstruction error, latent statistics, regulariser heads) and the gradient are summed across them;
        # the feature trajectory itself is small (d_r floats per frame) and stays whole on every rank.  The two parts that run on
        # an inner EigenFunctionTask (generator-mode regulariser, gradient-norm penalty eta[0]) add their batch sums across the
        # ranks too and leave their gradient shares in the flat gradient, which is summed once, at the end of the step.
        self.init_model_and_optimizer()
        # --- data: the feature trajectory r(x) of every frame, once (the layer has no parameters), resident in HBM
        traj = _HostFrames(traj_obj.trajectory).all()
        self.tot_dim = int(np.prod(traj.shape[1:]))
        self._weights = torch.as_tensor(np.asarray(traj_obj.weights)).to(device=self.device, dtype=torch.float32).contiguous()
        self._pp = self._pp_desc(self.tot_dim)
        self._feature_traj = self._features(torch.as_tensor(traj))
        d = self._flat.desc
        assert self._pp.d_r == d.dims[0] and d.dims[d.n_layers] == d.dims[0] + self.num_reg, \
            'encoder input / decoder output width must equal the feature dimension'
        if self.verbose:
            print('\nShape of trajectory data array:\n {}'.format(tuple(traj.shape)), flush=True)
        cfg = _hip.EFCfg()
        K = max(self.num_reg, 1)
        cfg.k, cfg.lag_idx, cfg.sort_eigvals = K, max(self.lag_idx, 1), 1          # cvec = argsort always (core.py:1016)
        cfg.alpha = float(self.gamma[1] / self.gamma[0]) if self._use_reg else 0.0  # gamma_0 (npl + gamma_1/gamma_0 pen)
        cfg.beta, cfg.dt = float(beta), float(self.traj_dt)
        for i in range(self.num_reg):
            cfg.eig_w[i] = float(eig_weights[i])
        self._cfg = cfg
        ecfg = _hip.EFCfg()                      # batch sums of the latent vector (the generator-mode layout, E unused)
        ecfg.k, ecfg.lag_idx, ecfg.sort_eigvals, ecfg.alpha, ecfg.beta, ecfg.dt = self.k, 0, 0, 0.0, 1.0, 1.0
        self._ecfg = ecfg
        self._n_enc_layers = len([m for m in self.model.encoder if isinstance(m, torch.nn.Linear)])
        self._ws = {}
        self._enc_grad = _EncGradPenalty(self) if self.eta[0] > self._eps else None   # (raises here when the encoder does not fit)
        # generator-mode regulariser (lag_tau_reg = 0): needs the coordinates themselves (its derivative runs through r(x))
        self._traj_host = traj
        self._gen = _RegGenerator(self, beta) if self._use_reg and self.lag_idx == 0 else None
        self._traj = _hip.upload_f32(traj, self.device) if self._gen is not None else None

    # -- the base class builds the flat buffer from mlp_layout(); this model needs the side-by-side chain
    def init_model_and_optimizer(self):
        if self.load_model_filename:
            if os.path.isfile(self.load_model_filename):
                self.model.load_state_dict(torch.load(self.load_model_filename, map_location="cpu"), strict=False)
                if self.verbose:
                    print(f'model parameters loaded from: {self.load_model_filename}')
            elif self.verbose:
                print(f'model file not found: {self.load_model_filename}')
        self._flat = _RegFlatParams(self.model, self.device, self.freeze_encoder)
        self.optimizer = _FusedOptimizer(self._flat, self.optimizer_name, self.learning_rate)

    def _features(self, X):
        X = _hip.upload_f32(torch.as_tensor(X).detach(), self.device)
        n = X.shape[0]
        out = torch.empty(n, self._pp.d_r, device=self.device, dtype=torch.float32)
        _hip.check(_hip.lib().cvf_align_feature_fwd(self._pp, _hip.ptr(X), n, None, _hip.ptr(out), None,
                                                    _hip.ptr(_hip.align_scratch(self._pp, n, self.device)), _hip.stream()),
                   "cvf_align_feature_fwd")
        return out

    def colvar_model(self):
        """core.py:855-863."""
        return _CVModel(self.preprocessing_layer, self.model.encoder, device=self.device)

    def reg_model(self):
        """core.py:865-877."""
        if self._cvec is None:
            self._cvec = torch.arange(self.model.num_reg)
        return _CVModel(self.preprocessing_layer, RegModel(self.model, self._cvec), device=self.device)

    def _workspace(self, B):
        ws = self._ws.get(B)
        if ws is None:
            lib, K, dev = _hip.lib(), max(self.num_reg, 1), self.device
            T = _hip.ntiles(B)
            ws = dict(
                scratch=torch.empty(lib.cvf_regae_scratch_floats(self._flat.desc, B), device=dev, dtype=torch.float32),
                y=torch.zeros(2 * T * K * 64, device=dev, dtype=torch.float32),
                out2=torch.zeros(3, device=dev, dtype=torch.float64),
                stats=torch.zeros(lib.cvf_ef_nstats(K, 1), device=dev, dtype=torch.float64),
                sscratch=torch.zeros(lib.cvf_ef_stats_scratch_doubles(K, 1), device=dev, dtype=torch.float64),
                loss_vec=torch.zeros(3 + 2 * K, device=dev, dtype=torch.float64),
                coef=torch.zeros(4 * K + K * K, device=dev, dtype=torch.float64), T=T)
            if True:   # (small) latent-vector buffers: the public reg_enc_* functions need them whatever eta is
                k = self.k
                ws.update(enc=torch.zeros(T * k * 64, device=dev, dtype=torch.float32),
                          ezero=torch.zeros(T * k * 64, device=dev, dtype=torch.float32),
                          estats=torch.zeros(lib.cvf_ef_nstats(k, 0), device=dev, dtype=torch.float64),
                          esscratch=torch.zeros(lib.cvf_ef_stats_scratch_doubles(k, 0), device=dev, dtype=torch.float64),
                          eterms=torch.zeros(2, device=dev, dtype=torch.float64),
                          ecoef=torch.zeros(k + k * k, device=dev, dtype=torch.float64))
            self._ws[B] = ws
        return ws

    def _step(self, feat, idx, w, w_lag, lag_ae, lag_reg, with_grad, advance=False, wsum=None, out=None, X=None, dp=False):
        """Loss terms (and, with ``with_grad``, gradient + optimizer step) of one batch: rows ``idx`` of ``feat``
        (``None``: rows 0..B-1), targets at ``+lag_ae``, lagged partners at ``+lag_reg``.  ``wsum``: the batch's weight
        sum when the caller knows it (static batches), else one host read.  Returns (or fills ``out`` with) the device
        vector [loss, ae, npl, pen, eig_1..K, enc_grad, enc_norm, enc_orth] (fp64).  ``X``: the batch's raw coordinates for the
        generator-mode regulariser (``None``: rows ``idx`` of the resident trajectory).  ``dp``: the batch is this rank's slice of a
        global batch (``train()`` of a data-parallel job) - batch sums and gradient are summed across the ranks; the public loss
        functions below evaluate the batch they are handed, on the calling rank alone."""
        lib, fl, P = _hip.lib(), self._flat, _hip.ptr
        B, K = int(w.shape[0]), self.num_reg
        ws = self._workspace(B)
        use_reg = self._use_reg and K > 0
        gen = self._gen if use_reg and lag_reg == 0 else None    # generator mode: the regulariser runs on the eigenfunction task's kernels
        assert not (use_reg and lag_reg == 0 and gen is None), 'generator-mode regulariser needs gamma and lag_tau_reg = 0 at construction'
        use_reg = use_reg and gen is None     # (from here on: the transfer-operator regulariser inside the chain kernels)
        alpha = float(self.alpha) if self.alpha > self._eps else 0.0
        use_enc = self._use_enc
        eta1 = float(self.eta[1]) if self.eta[1] > self._eps else 0.0
        eta2 = float(self.eta[2]) if self.eta[2] > self._eps else 0.0
        # (with a gradient to follow, the statistics pass leaves its activations in the scratch buffer for the gradient pass)
        self._call("cvf_regae_forward", lib.cvf_regae_forward_keep if with_grad else lib.cvf_regae_forward, fl.desc, P(fl.theta),
                   P(feat), P(idx), B, lag_ae, lag_reg if use_reg else 0, K, P(w), P(ws["scratch"]), P(ws["y"]), self._n_enc_layers,
                   P(ws["enc"]) if use_enc else None, P(ws["out2"]), _hip.stream())
        if dp:
            self._allreduce("allreduce_batch_sums", ws["out2"])     # [sum w err, sum w, (ratio: recomputed by cvf_regae_loss_row)]
        if use_enc:   # core.py:912-971: weighted means / variances / covariances of the latent vector, then the two penalties
            self._call("cvf_ef_stats", lib.cvf_ef_stats, self._ecfg, B, P(w), P(ws["enc"]), P(ws["ezero"]), None, None,
                       P(ws["esscratch"]), P(ws["estats"]), None, None, _hip.stream())
            if dp:
                self._allreduce("allreduce_batch_sums", ws["estats"])
            self._call("cvf_regae_enc_loss", lib.cvf_regae_enc_loss, P(ws["estats"]), self.k, eta1, eta2, P(ws["eterms"]),
                       P(ws["ecoef"]), _hip.stream())
        if use_reg:
            y_lag = ws["y"][ws["T"] * K * 64:]
            self._call("cvf_ef_stats", lib.cvf_ef_stats, self._cfg, B, P(w), P(ws["y"]), None, P(w_lag), P(y_lag),
                       P(ws["sscratch"]), P(ws["stats"]), None if dp else P(ws["loss_vec"]), None if dp else P(ws["coef"]), _hip.stream())
            if dp:
                self._allreduce("allreduce_batch_sums", ws["stats"])
                self._call("cvf_ef_loss", lib.cvf_ef_loss, self._cfg, P(ws["stats"]), P(ws["loss_vec"]), P(ws["coef"]), _hip.stream())
            self._cvec_dev = ws["loss_vec"][3 + K:3 + 2 * K]
        if out is None:
            out = torch.zeros(7 + K, device=self.device, dtype=torch.float64)
        self._call("cvf_regae_loss_row", lib.cvf_regae_loss_row, P(ws["out2"]), P(ws["loss_vec"]) if use_reg else None, alpha,
                   float(self.gamma[0]) if use_reg else 0.0, float(self.gamma[1]) if use_reg else 0.0, K,
                   P(ws["eterms"]) if use_enc else None, eta1, eta2, P(out), _hip.stream())
        eg = self._enc_grad if self.eta[0] > self._eps else None
        rows = None
        if eg is not None:   # core.py:1092-1095: the gradient-norm penalty of the encoder, on the eigenfunction kernels
            rows = feat[:B] if idx is None else feat.index_select(0, idx)
        if with_grad:
            if wsum is None:
                wsum_t = w.sum(dtype=torch.float64)
                if dp:
                    _dist.allreduce_sum_(wsum_t)
                wsum = float(wsum_t)
            adam = self.optimizer.fused_args() if advance and eg is None and gen is None and not dp else None   # (their gradients are added before the update)
            self._call("cvf_regae_backward", lib.cvf_regae_backward_reuse, fl.desc, P(fl.theta), P(feat), P(idx), B, lag_ae,
                       lag_reg if use_reg else 0, K, P(w), P(w_lag) if use_reg else None, alpha / wsum,
                       float(self.gamma[0]) if use_reg else 0.0, P(ws["y"]) if use_reg else None,
                       P(ws["coef"]) if use_reg else None, self._n_enc_layers, P(ws["ecoef"]) if use_enc else None,
                       P(ws["scratch"]), P(fl.grad), P(fl.mask),
                       P(self.optimizer.step_count) if advance else None, adam, _hip.stream())
        if gen is not None:
            lv = gen.forward(X if X is not None else (self._traj[:B] if idx is None else self._traj.index_select(0, idx)), w, with_grad, dp)
            out[2:4] = lv[1:3]
            out[4:4 + K] = lv[3:3 + K]
            out[0] += float(self.gamma[0]) * lv[1] + float(self.gamma[1]) * lv[2]
            self._cvec_dev = lv[3 + K:3 + 2 * K]
            if with_grad:
                gen.backward(w, float(self.gamma[0]))
        if eg is not None:
            term = eg.run(rows, w, float(self.eta[0]), with_grad and not self.freeze_encoder, dp)
            out[4 + K] = term
            out[0] += float(self.eta[0]) * term
        if with_grad and dp:
            self._allreduce("allreduce_gradient", fl.grad)      # collective #2: every part's share, once
        if with_grad and advance and adam is None:
            self.optimizer.step(advance=False)
        return out

    # -- the reference's public loss functions, evaluated on raw coordinate batches (forward values; the training
    #    loop below uses the resident feature trajectory instead)
    def _dev(self, t):
        return torch.as_tensor(t).detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _terms(self, feat, w, w_lag, lag_ae, lag_reg, alpha, use_reg, use_enc):
        """Forward values with only the named parts of the loss switched on."""
        keep = (self.alpha, self._use_reg, self._use_enc, self.eta)
        self.alpha, self._use_reg, self._use_enc, self.eta = alpha, use_reg, use_enc, [0.0, 1.0, 1.0] if use_enc else [0.0, 0.0, 0.0]
        try:
            return self._step(feat, None, w, w_lag, lag_ae, lag_reg, with_grad=False)
        finally:
            self.alpha, self._use_reg, self._use_enc, self.eta = keep

    def weighted_MSE_loss(self, X, X_lagged, weight):
        """core.py:879-885."""
        B = int(torch.as_tensor(X).shape[0])
        feat = torch.cat([self._features(X), self._features(X_lagged)])
        return self._terms(feat, self._dev(weight), None, B, 0, 1.0, False, False)[1].to(torch.get_default_dtype())

    def reg_eigen_loss(self, X, weight, X_lagged, weight_lagged):
        """core.py:973-1036 (transfer operator): ``(eig_vals, non_penalty_loss, penalty, cvec)``."""
        assert self.num_reg > 0, 'needs regularisers'
        if self.lag_idx == 0:   # generator mode (core.py:990,1008-1022): X_lagged / weight_lagged are not used
            assert self._gen is not None, 'generator-mode regulariser needs gamma at construction'
            Xd = _hip.upload_f32(torch.as_tensor(X).detach(), self.device)
            lv = self._gen.forward(Xd, self._dev(weight), False)
            dt, K = torch.get_default_dtype(), self.num_reg
            return lv[3:3 + K].to(dt).cpu(), lv[1].to(dt), lv[2].to(dt), lv[3 + K:3 + 2 * K].cpu().to(torch.long).numpy()
        B = int(torch.as_tensor(X).shape[0])
        feat = torch.cat([self._features(X), self._features(X_lagged)])
        gam, self.gamma = self.gamma, (self.gamma if self._use_reg else [1.0, 1.0])
        try:
            out = self._terms(feat, self._dev(weight), self._dev(weight_lagged), 0, B, 0.0, True, False)
        finally:
            self.gamma = gam
        dt = torch.get_default_dtype()
        cvec = self._cvec_dev.cpu().to(torch.long).numpy()
        return out[4:4 + self.num_reg].to(dt).cpu(), out[2].to(dt), out[3].to(dt), cvec

    def backward(self):
        """Fill ``p.grad`` of the module's parameters from the flat gradient of the last ``_step(..., with_grad=True)``."""
        for p, _, gv in self._flat.views:
            p.grad = gv.clone()

    def reg_enc_grad_loss(self, X, weight):
        """core.py:887-910 (forward value)."""
        if self._enc_grad is None:
            self._enc_grad = _EncGradPenalty(self)
        return self._enc_grad.run(self._features(X), self._dev(weight), 1.0, False).to(torch.get_default_dtype())

    def _enc_terms(self, X, weight):
        out = self._terms(self._features(X), self._dev(weight), None, 0, 0, 0.0, False, True)
        return out[5 + self.num_reg:].to(torch.get_default_dtype())

    def reg_enc_norm_loss(self, X, weight):
        """core.py:912-934."""
        return self._enc_terms(X, weight)[0]

    def reg_enc_orthognal_loss(self, X, weight):
        """core.py:936-971."""
        return self._enc_terms(X, weight)[1]

    def train(self):
        """core.py:1038-1217."""
        n = self._feature_traj.shape[0]
        ll = n - max(self.lag_idx, self.lag_ae_idx)                      # core.py:1042
        idx_train, idx_test = _split(ll, self.test_ratio)                # core.py:1044 (one draw)
        bs_train, bs_test = min(self.batch_size, len(idx_train)), min(self.batch_size, len(idx_test))
        world, rank = _dist.world(), _dist.rank()
        if world > 1:   # rank 0's permutation for everyone, then this rank's contiguous slice of every static batch (_dist.shard_batches)
            both = torch.as_tensor(np.concatenate([idx_train, idx_test]), device=self.device)
            _dist.broadcast_(both)
            both = both.cpu().numpy()
            idx_train, idx_test = both[:len(idx_train)], both[len(idx_train):]
            assert (bs_train == 0 or bs_train >= world) and (bs_test == 0 or bs_test >= world), \
                f"batch sizes {bs_train} / {bs_test} smaller than the number of ranks {world}"   # (each set by itself: a rank without steps would leave the others waiting)
            (ptr, nb_tr), (pte, nb_te) = _dist.shard_batches(len(idx_train), bs_train, rank, world), _dist.shard_batches(len(idx_test), bs_test, rank, world)
            idx_train, idx_test = np.asarray(idx_train)[ptr], np.asarray(idx_test)[pte]
        else:
            nb_tr, nb_te = bs_train, bs_test
        itr = torch.as_tensor(idx_train, device=self.device, dtype=torch.long)
        ite = torch.as_tensor(idx_test, device=self.device, dtype=torch.long)
        wtr, wte = self._weights[itr].contiguous(), self._weights[ite].contiguous()
        wtr_lag, wte_lag = self._weights[itr + self.lag_idx].contiguous(), self._weights[ite + self.lag_idx].contiguous()
        tr_batches = [(s, s + nb_tr) for s in range(0, len(idx_train) - nb_tr + 1, nb_tr)] if nb_tr > 0 else []
        te_batches = [(s, s + nb_te) for s in range(0, len(idx_test) - nb_te + 1, nb_te)] if nb_te > 0 else []
        # batches are static (shuffle=False): their (global) weight sums are known before the first step
        wsum_t = torch.stack([wtr[a:b].sum(dtype=torch.float64) for a, b in tr_batches]) if tr_batches else torch.zeros(0, device=self.device, dtype=torch.float64)
        if world > 1 and len(tr_batches):
            _dist.allreduce_sum_(wsum_t)
        wsum_tr = [float(v) for v in wsum_t.cpu()]
        self.loss_list = []
        min_loss = float("inf")
        if rank == 0:
            print("\nTraining starts.\n%d epochs in total, batch sizes (train/test): %d/%d" % (self.num_epochs, bs_train, bs_test))
            print("\nTrain set:\n\t%d data, %d iterations per epoch, %d iterations in total." %
                  (len(idx_train) * world, len(tr_batches), len(tr_batches) * self.num_epochs), flush=True)
            print("Test set:\n\t%d data, %d iterations per epoch, %d iterations in total." %
                  (len(idx_test) * world, len(te_batches), len(te_batches) * self.num_epochs), flush=True)
        K = self.num_reg
        loss_names = ['loss', 'ae_loss', 'eigen_non_penalty', 'eigen_penalty'] + ['eig_%d' % i for i in range(K)] + \
                     ['encoder_gradient', 'encoder_norm', 'encoder_orthogonality']
        ncol = len(loss_names)
        log_tr = torch.zeros(max(len(tr_batches), 1), ncol, device=self.device, dtype=torch.float64)
        log_te = torch.zeros(max(len(te_batches), 1), ncol, device=self.device, dtype=torch.float64)
        cvec_log = [torch.zeros(max(K, 1), dtype=torch.float64).pin_memory() for _ in range(4)]   # the ordering, with the ring below

        def on_epoch(ep, tr, te):
            dt = torch.get_default_dtype()
            tr, te = tr.to(dt), te.to(dt)
            if self._use_reg and (tr_batches or te_batches):
                self._cvec = cvec_log[ep % len(cvec_log)][:K].clone().to(torch.long)          # core.py:1110 (last evaluation)
            self.loss_list.append([tr, te])                                                       # core.py:1202
            mean_tr = tr.mean(0) if len(tr) else torch.full((ncol,), float("nan"))
            mean_te = te.mean(0) if len(te) else torch.full((ncol,), float("nan"))
            for i, name in enumerate(loss_names):                                                 # core.py:1208-1212
                self.writer.add_scalar('%s/train' % name, mean_tr[i], ep)
                self.writer.add_scalar('%s/test' % name, mean_te[i], ep)

        elog = _AsyncEpochLog(log_tr, log_te, len(tr_batches), len(te_batches), on_epoch)
        dp = _dist.collectives()
        for epoch in _tqdm(range(self.num_epochs), disable=(rank != 0)):
            self.model.train()
            for it, (a, b) in enumerate(tr_batches):
                self._step(self._feature_traj, itr[a:b], wtr[a:b], wtr_lag[a:b], self.lag_ae_idx, self.lag_idx, with_grad=True,
                           advance=True, wsum=wsum_tr[it], out=log_tr[it], dp=dp)
            saving = self.save_model_every_step > 0 and epoch % self.save_model_every_step == self.save_model_every_step - 1
            plotting = self.plot_frequency > 0 and epoch % self.plot_frequency == self.plot_frequency - 1
            if saving or plotting:                       # (order of the reference: save / plot before the test pass, core.py:1130-1140)
                if self._use_reg and tr_batches:
                    self._cvec = self._cvec_dev.cpu().to(torch.long)
            if saving:
                self.save_model(epoch)
                last = float(log_tr[len(tr_batches) - 1, 0]) if tr_batches else float("inf")
                if last < min_loss:
                    min_loss = last
                    self.save_model(epoch, 'best')
            if plotting:
                if self.plot_class is not None and rank == 0:
                    self.plot_class.plot(self.colvar_model(), self.reg_model(), epoch=epoch)
            for it, (a, b) in enumerate(te_batches):
                self._step(self._feature_traj, ite[a:b], wte[a:b], wte_lag[a:b], self.lag_ae_idx, self.lag_idx, with_grad=False,
                           out=log_te[it], dp=dp)
            elog.reserve(epoch)
            if self._use_reg and (tr_batches or te_batches):
                cvec_log[epoch % len(cvec_log)].copy_(self._cvec_dev, non_blocking=True)
            elog.push(epoch)
        elog.flush_all()
        self.train_loss_df = pd.DataFrame(torch.cat([e[0].mean(dim=0, keepdim=True) for e in self.loss_list]).numpy(),
                                          columns=loss_names)
        self.test_loss_df = pd.DataFrame(torch.cat([e[1].mean(dim=0, keepdim=True) for e in self.loss_list]).numpy(),
                                         columns=loss_names)
