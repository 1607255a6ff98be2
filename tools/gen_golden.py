#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (it needs /root/reference).  The reference's Python
files are imported read-only from where they lie; two cosmetic third-party imports the
image lacks are satisfied by empty in-memory modules (``tensorboardX.SummaryWriter``
used only for logging at core.py:143,559-561,738-739; ``openmm.unit`` imported at
core.py:58 and never used).  Nothing of the reference is copied: the fixtures are
inputs (seeded synthetic data + explicit initial weights) and the numbers the
reference computes from them.

Cases:
  G1  loss_func known-answer tests  (EigenFunctionTask.loss_func, core.py:387-457)
  G2  EigenFunctionTask.train traces (core.py:459-566)
  G3  AutoEncoderTask.train traces   (core.py:668-744)
  G4  colvarsfinder.nn structure     (nn.py:29-114,242-293)
  G7  RegAutoEncoderTask.train traces (core.py:746-1217: time-lagged reconstruction + transfer-operator regulariser)
  G8  the bench-sized cases: BASELINE configs 2 / 3 and config 3 in transfer mode at 100 000 frames, batches of 20 000
      (seed + outputs only; tests/goldens.py regenerates the frames from the seed)
Every case is emitted for torch default dtype float32 and float64 with identical
initial weights (drawn in fp32, then cast).

The alignment/feature layer is NOT part of the reference (third-party molann, absent):
cases with ``pp='align'`` pass the oracle's ``oracle.pp.AlignFeature`` to the reference
as its opaque ``pp_layer`` - they pin the reference's loss/training code composed with
that layer, not the layer itself.

Usage:  python tools/gen_golden.py            (writes tests/golden/*.npz)
"""

import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

REF = "/root/reference"


def import_reference():
    tbx = types.ModuleType("tensorboardX")

    class SummaryWriter:  # logging sink only
        def __init__(self, *a, **k):
            self.scalars = []

        def add_scalar(self, tag, value, step):
            self.scalars.append((tag, float(value), int(step)))

    tbx.SummaryWriter = SummaryWriter
    sys.modules.setdefault("tensorboardX", tbx)
    omm = types.ModuleType("openmm")
    omm.unit = types.ModuleType("openmm.unit")
    omm.app = types.ModuleType("openmm.app")
    sys.modules.setdefault("openmm", omm)
    sys.modules.setdefault("openmm.unit", omm.unit)
    sys.modules.setdefault("openmm.app", omm.app)
    sys.path.insert(0, REF)
    import colvarsfinder.core as core
    import colvarsfinder.nn as rnn
    assert core.__file__.startswith(REF), core.__file__
    return core, rnn


from oracle import nnref  # noqa: E402
from oracle.pp import AlignFeature  # noqa: E402
from tests.synth import make_molecule_traj, make_2d_traj, Traj  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sd_np(sd):
    return {f"sd/{k}": v.detach().cpu().numpy() for k, v in sd.items()}


def build_pp(case):
    if case["pp"] == "identity":
        return torch.nn.Identity()
    return AlignFeature(case["align_idx"], case["ref_pos"], case["features"], case.get("use_angle_value", False))


def pp_meta(case):
    if case["pp"] == "identity":
        return {"pp": "identity"}
    feats = case["features"]
    return {
        "pp": "align",
        "align_idx": np.asarray(case["align_idx"], dtype=np.int64),
        "ref_pos": np.asarray(case["ref_pos"], dtype=np.float64),
        "feat_types": np.asarray([t for t, _ in feats]),
        "feat_atoms": np.asarray([list(a) + [-1] * (4 - len(a)) if t != "position" else [-1] * 4 for t, a in feats],
                                 dtype=np.int64),
        "feat_pos_atoms": np.asarray(
            [i for t, a in feats if t == "position" for i in a], dtype=np.int64),
        "use_angle_value": np.asarray(case.get("use_angle_value", False)),
    }


def molecule_case(n_atoms, n_frames, seed, features="position", lag=False):
    traj, w, ref = make_molecule_traj(n_atoms, n_frames, seed)
    if features == "position":
        feats = [("position", tuple(range(n_atoms)))]
    else:  # mixed: positions of a subset + a few internal coordinates
        feats = [("position", (0, 2, 3, 5)), ("bond", (0, 1)), ("bond", (2, 7)), ("angle", (1, 2, 3)),
                 ("dihedral", (0, 1, 2, 3)), ("dihedral", (4, 5, 6, 7)), ("angle", (6, 8, 9))]
    return dict(pp="align", traj=traj, w=w, align_idx=list(range(n_atoms)) if features == "position" else [0, 1, 2, 4, 5, 8],
                ref_pos=ref if features == "position" else ref[[0, 1, 2, 4, 5, 8]], features=feats)


def run_loss_kat(core, rnn, name, case, k, layer_dims, mode, dtype, alpha, eig_w, beta, sort, seed):
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd0 = nnref.init_eigenfunctions(layer_dims, k, g, dtype)
    model = rnn.EigenFunctions(layer_dims, k)
    model.load_state_dict(sd0)
    pp = build_pp(case)
    traj, w = case["traj"], case["w"]
    dt = 0.5
    lag_idx = 0 if mode == "generator" else 3
    tot_dim = traj[0].size
    a = None
    if mode == "generator":
        rs = np.random.RandomState(seed + 7)
        a = torch.tensor(1.0 / rs.choice([1.0, 12.0, 14.0, 16.0], size=tot_dim // (3 if traj.ndim == 3 else 1)).repeat(
            3 if traj.ndim == 3 else 1), dtype=dtype)
    with tempfile.TemporaryDirectory() as tmp:
        task = core.EigenFunctionTask(Traj(traj, w, dt), pp, model, tmp, alpha, eig_w, diag_coeff=a, beta=beta,
                                      lag_tau=lag_idx * dt, k=k, sort_eigvals_in_training=sort, verbose=False,
                                      save_model_every_step=0)
    B = traj.shape[0] - lag_idx
    X = torch.tensor(traj[:B]).to(dtype)
    wt = torch.tensor(w[:B]).to(dtype)
    Xl = wl = None
    if lag_idx == 0:
        X.requires_grad_()
    else:
        Xl = torch.tensor(traj[lag_idx:lag_idx + B]).to(dtype)
        wl = torch.tensor(w[lag_idx:lag_idx + B]).to(dtype)
    loss, eig, npl, pen, cvec = task.loss_func(X, wt, Xl, wl)
    loss.backward()
    out = dict(kind="ef_loss", mode=mode, k=k, layer_dims=np.asarray(layer_dims), alpha=alpha, eig_w=np.asarray(eig_w, dtype=np.float64),
               beta=beta, dt=dt, lag_idx=lag_idx, sort=sort, traj=traj, w=w,
               loss=float(loss), eig=eig.detach().numpy().astype(np.float64), npl=float(npl), pen=float(pen),
               cvec=np.asarray(cvec, dtype=np.int64), y=model(pp(X)).detach().numpy())
    if a is not None:
        out["diag_coeff"] = a.numpy().astype(np.float64)
    out.update(sd_np(sd0))
    out.update({f"grad/{n}": p.grad.numpy() for n, p in model.named_parameters()})
    out.update(pp_meta(case))
    tag = "f32" if dtype == torch.float32 else "f64"
    np.savez_compressed(os.path.join(OUT, f"{name}_{tag}.npz"), **out)
    print(f"  {name}_{tag}: loss={float(loss):.9g} eig={eig.tolist()} cvec={list(cvec)}")


def run_ef_train(core, rnn, name, case, k, layer_dims, mode, dtype, alpha, eig_w, beta, lr, bs, epochs, seed, lag_idx=0, dt=0.5):
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd0 = nnref.init_eigenfunctions(layer_dims, k, g, dtype)
    model = rnn.EigenFunctions(layer_dims, k)
    model.load_state_dict(sd0)
    pp = build_pp(case)
    traj, w = case["traj"], case["w"]
    tot_dim = traj[0].size
    a = None
    if mode == "generator":
        lag_idx = 0
        rs = np.random.RandomState(seed + 7)
        per = 3 if traj.ndim == 3 else 1
        a = torch.tensor(1.0 / rs.choice([1.0, 12.0, 14.0, 16.0], size=tot_dim // per).repeat(per), dtype=dtype)
    with tempfile.TemporaryDirectory() as tmp:
        task = core.EigenFunctionTask(Traj(traj, w, dt), pp, model, tmp, alpha, eig_w, diag_coeff=a, beta=beta,
                                      lag_tau=lag_idx * dt, learning_rate=lr, k=k, batch_size=bs, num_epochs=epochs,
                                      test_ratio=0.2, verbose=False, save_model_every_step=0)
        np.random.seed(seed)
        task.train()
    # the split the reference drew (second of two permutations, core.py:465-468)
    np.random.seed(seed)
    n = traj.shape[0] - lag_idx
    n_test = int(np.ceil(0.2 * n))
    np.random.permutation(n)
    perm = np.random.permutation(n)
    probe = torch.tensor(traj[:64]).to(dtype)
    cv = task.colvar_model()(probe).detach().numpy()
    out = dict(kind="ef_train", mode=mode, k=k, layer_dims=np.asarray(layer_dims), alpha=alpha, eig_w=np.asarray(eig_w, dtype=np.float64),
               beta=beta, dt=dt, lag_idx=lag_idx, lr=lr, batch_size=bs, num_epochs=epochs, seed=seed, traj=traj, w=w,
               train_idx=perm[n_test:], test_idx=perm[:n_test],
               train_loss=np.stack([e[0].numpy() for e in task.loss_list]),
               test_loss=np.stack([e[1].numpy() for e in task.loss_list]),
               cvec=np.asarray(task._cvec, dtype=np.int64), colvar_probe=cv,
               train_loss_df=task.train_loss_df.to_numpy(), test_loss_df=task.test_loss_df.to_numpy(),
               loss_names=np.asarray(list(task.train_loss_df.columns)))
    if a is not None:
        out["diag_coeff"] = a.numpy().astype(np.float64)
    out.update(sd_np(sd0))
    out.update({f"final/{n}": p.detach().numpy() for n, p in model.state_dict().items()})
    out.update(pp_meta(case))
    tag = "f32" if dtype == torch.float32 else "f64"
    np.savez_compressed(os.path.join(OUT, f"{name}_{tag}.npz"), **out)
    print(f"  {name}_{tag}: last train row {out['train_loss'][-1, -1]}")


def run_ae_train(core, rnn, name, case, e_dims, d_dims, dtype, lr, bs, epochs, seed):
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd0 = nnref.init_autoencoder(e_dims, d_dims, g, dtype)
    model = rnn.AutoEncoder(e_dims, d_dims)
    model.load_state_dict(sd0)
    pp = build_pp(case)
    traj, w = case["traj"], case["w"]
    with tempfile.TemporaryDirectory() as tmp:
        task = core.AutoEncoderTask(Traj(traj, w, 0.5), pp, model, tmp, learning_rate=lr, batch_size=bs, num_epochs=epochs,
                                    test_ratio=0.2, verbose=False, save_model_every_step=0)
        feat = task._feature_traj.detach().numpy()
        # known-answer for weighted_MSE_loss + grads at the initial weights (core.py:652-666)
        Fb = task._feature_traj[:min(256, len(w))]
        wb = task._weights[:min(256, len(w))]
        l0 = task.weighted_MSE_loss(Fb, wb)
        l0.backward()
        grads0 = {f"grad/{n}": p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        np.random.seed(seed)
        task.train()
    np.random.seed(seed)
    n = traj.shape[0]
    n_test = int(np.ceil(0.2 * n))
    perm = np.random.permutation(n)
    probe = torch.tensor(traj[:64]).to(dtype)
    cv = task.colvar_model()(probe).detach().numpy()
    out = dict(kind="ae_train", e_dims=np.asarray(e_dims), d_dims=np.asarray(d_dims), lr=lr, batch_size=bs, num_epochs=epochs,
               seed=seed, traj=traj, w=w, train_idx=perm[n_test:], test_idx=perm[:n_test], features=feat,
               loss0=float(l0),
               train_loss=np.stack([e[0].numpy() for e in task.loss_list]),
               test_loss=np.stack([e[1].numpy() for e in task.loss_list]),
               colvar_probe=cv, train_loss_df=task.train_loss_df.to_numpy(), test_loss_df=task.test_loss_df.to_numpy())
    out.update(sd_np(sd0))
    out.update(grads0)
    out.update({f"final/{n}": p.detach().numpy() for n, p in model.state_dict().items()})
    out.update(pp_meta(case))
    tag = "f32" if dtype == torch.float32 else "f64"
    np.savez_compressed(os.path.join(OUT, f"{name}_{tag}.npz"), **out)
    print(f"  {name}_{tag}: loss0={float(l0):.9g} last train {out['train_loss'][-1, -1]:.9g}")


def run_regae_train(core, rnn, name, case, e_dims, d_dims, r_dims, K, dtype, alpha, gamma, eig_w, lag_ae, lag_reg, dt, lr, bs,
                    epochs, seed, freeze=False, eta=(0.0, 0.0, 0.0), beta=1.0):
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd0 = nnref.init_regautoencoder(e_dims, d_dims, r_dims, K, g, dtype)
    model = rnn.RegAutoEncoder(e_dims, d_dims, r_dims, K)
    model.load_state_dict(sd0)
    pp = build_pp(case)
    traj, w = case["traj"], case["w"]
    with tempfile.TemporaryDirectory() as tmp:
        task = core.RegAutoEncoderTask(Traj(traj, w, dt), pp, model, tmp, eig_weights=eig_w, learning_rate=lr, batch_size=bs,
                                       num_epochs=epochs, test_ratio=0.2, alpha=alpha, gamma=gamma, eta=list(eta),
                                       lag_tau_ae=lag_ae * dt, lag_tau_reg=lag_reg * dt, beta=beta, freeze_encoder=freeze, verbose=False,
                                       save_model_every_step=0)
        # known answer at the initial weights: the loss terms of the first 200 frames + all parameter gradients
        nb = min(200, traj.shape[0] - max(lag_ae, lag_reg))
        X, wb = task._traj[:nb], task._weights[:nb]
        ae0 = task.weighted_MSE_loss(X, task._traj[lag_ae:lag_ae + nb], wb)
        Xg = X.clone() if lag_reg == 0 else X     # (generator mode sets requires_grad on its input, core.py:990)
        eig0, npl0, pen0, cvec0 = task.reg_eigen_loss(Xg, wb, task._traj[lag_reg:lag_reg + nb], task._weights[lag_reg:lag_reg + nb])
        en0 = task.reg_enc_norm_loss(X, wb) if eta[1] > 0 else torch.zeros(())
        eo0 = task.reg_enc_orthognal_loss(X, wb) if eta[2] > 0 else torch.zeros(())
        eg0 = task.reg_enc_grad_loss(X.clone(), wb) if eta[0] > 0 else torch.zeros(())   # (the call sets requires_grad on its input)
        l0 = alpha * ae0 + gamma[0] * npl0 + gamma[1] * pen0 + eta[0] * eg0 + eta[1] * en0 + eta[2] * eo0
        l0.backward()
        grads0 = {f"grad/{n}": p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        np.random.seed(seed)
        task.train()
    np.random.seed(seed)
    ll = traj.shape[0] - max(lag_ae, lag_reg)
    n_test = int(np.ceil(0.2 * ll))
    perm = np.random.permutation(ll)
    probe = torch.tensor(traj[:64]).to(dtype)
    out = dict(kind="regae_train", e_dims=np.asarray(e_dims), d_dims=np.asarray(d_dims), r_dims=np.asarray(r_dims), K=K, lr=lr,
               batch_size=bs, num_epochs=epochs, seed=seed, alpha=alpha, gamma=np.asarray(gamma, dtype=np.float64),
               eig_w=np.asarray(eig_w, dtype=np.float64), lag_ae=lag_ae, lag_reg=lag_reg, dt=dt, freeze=freeze, traj=traj, w=w,
               eta=np.asarray(eta, dtype=np.float64), kat_enc=np.asarray([float(en0), float(eo0)]), kat_enc_grad=float(eg0), beta=beta,
               train_idx=perm[n_test:], test_idx=perm[:n_test], kat_n=nb,
               kat=np.asarray([float(l0), float(ae0), float(npl0), float(pen0)] + [float(e) for e in eig0]),
               kat_cvec=np.asarray(cvec0),
               train_loss=np.stack([e[0].numpy() for e in task.loss_list]),
               test_loss=np.stack([e[1].numpy() for e in task.loss_list]),
               cvec=np.asarray(task._cvec), colvar_probe=task.colvar_model()(probe).detach().numpy(),
               reg_probe=task.reg_model()(probe).detach().numpy(),
               train_loss_df=task.train_loss_df.to_numpy(), test_loss_df=task.test_loss_df.to_numpy())
    out.update(sd_np(sd0))
    out.update(grads0)
    out.update({f"final/{n}": p.detach().numpy() for n, p in model.state_dict().items()})
    out.update(pp_meta(case))
    tag = "f32" if dtype == torch.float32 else "f64"
    np.savez_compressed(os.path.join(OUT, f"{name}_{tag}.npz"), **out)
    print(f"  {name}_{tag}: kat={out['kat'][:4]} last train {out['train_loss'][-1, -1, :4]}")


def run_regae_eta_cases(core, rnn, id2, mol10):
    for dtype in (torch.float32, torch.float64):
        # variance / covariance penalties on the latent vector (eta_1, eta_2) on top of the main.ipynb shape
        run_regae_train(core, rnn, "train_regae_mol10_k2_eta", mol10, [30, 20, 20, 20, 2], [2, 10, 10, 30], [2, 10, 10, 1], 2, dtype, 1.0,
                        [1.0, 10.0], [1.0, 0.5], 0, 1, 0.5, 2e-3, 64, 3, 704, eta=(0.0, 0.5, 2.0))
        # ... and alone (no eigenfunction regulariser), three latent components, identity preprocessing
        run_regae_train(core, rnn, "train_regae_id2_k1_eta", id2, [2, 16, 3], [3, 16, 2], [3, 8, 1], 1, dtype, 1.0,
                        [1.0, 5.0], [1.0], 1, 2, 0.5, 5e-3, 100, 2, 705, eta=(0.0, 1.5, 0.7))
    torch.set_default_dtype(torch.float32)


def run_regae_grad_cases(core, rnn, id2, mol10):
    for dtype in (torch.float32, torch.float64):
        # the gradient-norm penalty on the encoder (eta_0, core.py:896-910) with the transfer-operator regulariser, main.ipynb shape
        run_regae_train(core, rnn, "train_regae_mol10_k2_eg", mol10, [30, 20, 20, 20, 2], [2, 10, 10, 30], [2, 10, 10, 1], 2, dtype, 1.0,
                        [1.0, 10.0], [1.0, 0.5], 0, 1, 0.5, 2e-3, 64, 3, 706, eta=(0.4, 0.0, 0.0))
        # ... with the two other penalties, three latent components, one hidden layer, identity layer
        run_regae_train(core, rnn, "train_regae_id2_k1_eg", id2, [2, 16, 3], [3, 16, 2], [3, 8, 1], 1, dtype, 1.0,
                        [1.0, 5.0], [1.0], 1, 2, 0.5, 5e-3, 100, 2, 707, eta=(0.8, 1.5, 0.7))
    torch.set_default_dtype(torch.float32)


def run_regae_generator_cases(core, rnn, id2, mol10):
    for dtype in (torch.float32, torch.float64):
        # the eigenfunction regulariser in GENERATOR mode (lag_tau_reg = 0, the constructor's default; core.py:1008-1022):
        # regulariser o encoder o r(x) differentiated with respect to the coordinates.  2d.ipynb shape, K = 1
        run_regae_train(core, rnn, "train_regae_id2_k1_gen", id2, [2, 20, 20, 20, 1], [1, 20, 20, 2], [1, 20, 20, 1], 1, dtype, 1.0,
                        [1.0, 20.0], [1.0], 2, 0, 0.5, 5e-3, 100, 3, 711, beta=1.5)
        # main.ipynb shape through the alignment layer, K = 2, with the encoder's gradient-norm penalty on top
        run_regae_train(core, rnn, "train_regae_mol10_k2_gen", mol10, [30, 20, 20, 20, 2], [2, 10, 10, 30], [2, 10, 10, 1], 2, dtype, 1.0,
                        [1.0, 10.0], [1.0, 0.5], 0, 0, 0.5, 2e-3, 64, 3, 712, eta=(0.2, 0.0, 0.0), beta=1.0)
        # a shallow encoder (three hidden layers in the chain regulariser o encoder), frozen
        run_regae_train(core, rnn, "train_regae_id2_k2_gen_frozen", id2, [2, 12, 2], [2, 12, 12, 2], [2, 12, 12, 1], 2, dtype, 0.5,
                        [2.0, 8.0], [1.0, 0.3], 1, 0, 0.5, 5e-3, 100, 2, 713, freeze=True)
    torch.set_default_dtype(torch.float32)


def run_regae_cases(core, rnn, id2, mol10):
    for dtype in (torch.float32, torch.float64):
        # 2d.ipynb:743-760 shape: K = 1, both lags equal
        run_regae_train(core, rnn, "train_regae_id2_k1", id2, [2, 20, 20, 20, 1], [1, 20, 20, 2], [1, 20, 20, 1], 1, dtype, 1.0,
                        [1.0, 20.0], [1.0], 2, 2, 0.5, 5e-3, 100, 3, 701)
        # main.ipynb:452-458 shape: K = 2, lag_tau_ae = 0
        run_regae_train(core, rnn, "train_regae_mol10_k2", mol10, [30, 20, 20, 20, 2], [2, 10, 10, 30], [2, 10, 10, 1], 2, dtype, 1.0,
                        [1.0, 10.0], [1.0, 0.5], 0, 1, 0.5, 2e-3, 64, 3, 702)
        # frozen encoder, different lags, alpha != 1
        run_regae_train(core, rnn, "train_regae_id2_k2_frozen", id2, [2, 12, 12, 2], [2, 12, 2], [2, 12, 1], 2, dtype, 0.5,
                        [2.0, 8.0], [1.0, 0.3], 1, 3, 0.5, 5e-3, 100, 2, 703, freeze=True)
    torch.set_default_dtype(torch.float32)


# ---------------------------------------------------------------------------------------------------------------------------
# Fixtures at the sizes bench.py times: BASELINE configs 2 and 3 - 22 atoms x 100 000 frames, batches of
# 20 000 (examples/dipeptide/main.ipynb:266) - run by the imported reference.  The trajectory is NOT stored: the fixture keeps
# the arguments of tests.synth.make_molecule_traj (``traj_gen`` = atoms, frames, seed) and the reference's outputs only.
BIG_ATOMS, BIG_FRAMES, BIG_BATCH = 22, 100_000, 20_000


def big_case(seed):
    traj, w, ref = make_molecule_traj(BIG_ATOMS, BIG_FRAMES, seed)
    return dict(pp="align", traj=traj, w=w, align_idx=list(range(BIG_ATOMS)), ref_pos=ref,
                features=[("position", tuple(range(BIG_ATOMS)))], traj_gen=np.asarray([BIG_ATOMS, BIG_FRAMES, seed], dtype=np.int64))


def run_ef_train_big(core, rnn, name, case, k, layer_dims, mode, dtype, alpha, eig_w, beta, lr, epochs, seed, lag_idx=0, dt=0.5):
    """config 3 (generator, diag_coeff) / the same in transfer mode: one loss_func call on the first 20 000 frames with every
    parameter gradient (core.py:387-457), then EigenFunctionTask.train (core.py:459-566) at batch_size 20 000."""
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd0 = nnref.init_eigenfunctions(layer_dims, k, g, dtype)
    model = rnn.EigenFunctions(layer_dims, k)
    model.load_state_dict(sd0)
    pp = build_pp(case)
    traj, w = case["traj"], case["w"]
    a = None
    if mode == "generator":
        lag_idx = 0
        rs = np.random.RandomState(seed + 7)
        a = torch.tensor(1.0 / rs.choice([1.0, 12.0, 14.0, 16.0], size=traj.shape[1]).repeat(3), dtype=dtype)
    with tempfile.TemporaryDirectory() as tmp:
        task = core.EigenFunctionTask(Traj(traj, w, dt), pp, model, tmp, alpha, eig_w, diag_coeff=a, beta=beta,
                                      lag_tau=lag_idx * dt, learning_rate=lr, k=k, batch_size=BIG_BATCH, num_epochs=epochs,
                                      test_ratio=0.2, verbose=False, save_model_every_step=0)
        B = BIG_BATCH
        X = torch.tensor(traj[:B]).to(dtype)
        wt = torch.tensor(w[:B]).to(dtype)
        Xl = wl = None
        if lag_idx == 0:
            X.requires_grad_()
        else:
            Xl = torch.tensor(traj[lag_idx:lag_idx + B]).to(dtype)
            wl = torch.tensor(w[lag_idx:lag_idx + B]).to(dtype)
        loss, eig, npl, pen, cvec = task.loss_func(X, wt, Xl, wl)
        loss.backward()
        kat = dict(kat_loss=float(loss), kat_eig=eig.detach().numpy().astype(np.float64), kat_npl=float(npl), kat_pen=float(pen),
                   kat_cvec=np.asarray(cvec, dtype=np.int64), kat_n=B)
        kat.update({f"grad/{n}": p.grad.numpy().copy() for n, p in model.named_parameters()})
        model.zero_grad(set_to_none=True)
        np.random.seed(seed)
        task.train()
    probe = torch.tensor(traj[:64]).to(dtype)
    cv = task.colvar_model()(probe).detach().numpy()
    out = dict(kind="ef_train_big", mode=mode, k=k, layer_dims=np.asarray(layer_dims), alpha=alpha, eig_w=np.asarray(eig_w, dtype=np.float64),
               beta=beta, dt=dt, lag_idx=lag_idx, lr=lr, batch_size=BIG_BATCH, num_epochs=epochs, seed=seed, traj_gen=case["traj_gen"],
               train_loss=np.stack([e[0].numpy() for e in task.loss_list]),
               test_loss=np.stack([e[1].numpy() for e in task.loss_list]),
               cvec=np.asarray(task._cvec, dtype=np.int64), colvar_probe=cv,
               train_loss_df=task.train_loss_df.to_numpy(), test_loss_df=task.test_loss_df.to_numpy(),
               loss_names=np.asarray(list(task.train_loss_df.columns)))
    out.update(kat)
    if a is not None:
        out["diag_coeff"] = a.numpy().astype(np.float64)
    out.update(sd_np(sd0))
    out.update({f"final/{n}": p.detach().numpy() for n, p in model.state_dict().items()})
    out.update(pp_meta(case))
    tag = "f32" if dtype == torch.float32 else "f64"
    np.savez_compressed(os.path.join(OUT, f"{name}_{tag}.npz"), **out)
    print(f"  {name}_{tag}: kat loss {float(loss):.9g}; {out['train_loss'].shape[0]} train steps, last row {out['train_loss'][-1, -1]}")


def run_ae_train_big(core, rnn, name, case, e_dims, d_dims, dtype, lr, epochs, seed):
    """config 2: AutoEncoderTask (core.py:610-744) on 100 000 x 22 atoms at batch_size 20 000; weighted_MSE_loss and its gradient on
    the first 20 000 feature rows; 512 rows of _feature_traj (first and last 256) instead of the whole array."""
    torch.set_default_dtype(dtype)
    g = torch.Generator().manual_seed(seed)
    sd0 = nnref.init_autoencoder(e_dims, d_dims, g, dtype)
    model = rnn.AutoEncoder(e_dims, d_dims)
    model.load_state_dict(sd0)
    pp = build_pp(case)
    traj, w = case["traj"], case["w"]
    with tempfile.TemporaryDirectory() as tmp:
        task = core.AutoEncoderTask(Traj(traj, w, 0.5), pp, model, tmp, learning_rate=lr, batch_size=BIG_BATCH, num_epochs=epochs,
                                    test_ratio=0.2, verbose=False, save_model_every_step=0)
        feat = task._feature_traj.detach().numpy()
        l0 = task.weighted_MSE_loss(task._feature_traj[:BIG_BATCH], task._weights[:BIG_BATCH])
        l0.backward()
        grads0 = {f"grad/{n}": p.grad.detach().numpy().copy() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
        np.random.seed(seed)
        task.train()
    probe = torch.tensor(traj[:64]).to(dtype)
    cv = task.colvar_model()(probe).detach().numpy()
    out = dict(kind="ae_train_big", e_dims=np.asarray(e_dims), d_dims=np.asarray(d_dims), lr=lr, batch_size=BIG_BATCH, num_epochs=epochs,
               seed=seed, traj_gen=case["traj_gen"], feature_rows=np.concatenate([feat[:256], feat[-256:]]), loss0=float(l0), kat_n=BIG_BATCH,
               train_loss=np.stack([e[0].numpy() for e in task.loss_list]),
               test_loss=np.stack([e[1].numpy() for e in task.loss_list]),
               colvar_probe=cv, train_loss_df=task.train_loss_df.to_numpy(), test_loss_df=task.test_loss_df.to_numpy())
    out.update(sd_np(sd0))
    out.update(grads0)
    out.update({f"final/{n}": p.detach().numpy() for n, p in model.state_dict().items()})
    out.update(pp_meta(case))
    tag = "f32" if dtype == torch.float32 else "f64"
    np.savez_compressed(os.path.join(OUT, f"{name}_{tag}.npz"), **out)
    print(f"  {name}_{tag}: loss0={float(l0):.9g} last train {out['train_loss'][-1, -1]:.9g}")


def run_big_cases(core, rnn):
    import time
    for dtype in (torch.float32, torch.float64):
        t0 = time.time()
        run_ef_train_big(core, rnn, "big_gen_c3", big_case(2603), 3, [66, 20, 20, 20, 1], "generator", dtype, 20.0, [1.0, 0.75, 0.5], 1.0,
                         1e-3, 2, 2603)
        run_ef_train_big(core, rnn, "big_tr_c3", big_case(2604), 3, [66, 20, 20, 20, 1], "transfer", dtype, 20.0, [1.0, 0.75, 0.5], 1.0,
                         1e-3, 2, 2604, lag_idx=3)
        run_ae_train_big(core, rnn, "big_ae_c2", big_case(2602), [66, 20, 20, 20, 2], [2, 10, 10, 66], dtype, 1e-3, 2, 2602)
        print(f"  big cases {dtype}: {time.time() - t0:.1f} s")
    torch.set_default_dtype(torch.float32)


def run_nn_structure(rnn):
    torch.set_default_dtype(torch.float32)
    ef = rnn.EigenFunctions([30, 20, 20, 20, 1], 3)
    ae = rnn.AutoEncoder([66, 20, 20, 20, 2], [2, 10, 10, 66])
    g = torch.Generator().manual_seed(5)
    sd_ef = nnref.init_eigenfunctions([30, 20, 20, 20, 1], 3, g)
    sd_ae = nnref.init_autoencoder([66, 20, 20, 20, 2], [2, 10, 10, 66], g)
    ef.load_state_dict(sd_ef)
    ae.load_state_dict(sd_ae)
    x30 = torch.randn(17, 30, generator=g)
    x66 = torch.randn(17, 66, generator=g)
    out = dict(kind="nn",
               ef_keys=np.asarray(list(ef.state_dict().keys())), ae_keys=np.asarray(list(ae.state_dict().keys())),
               ef_nparams=sum(p.numel() for p in ef.parameters()), ae_nparams=sum(p.numel() for p in ae.parameters()),
               ef_modules=np.asarray([n for n, _ in ef.named_modules()]), ae_modules=np.asarray([n for n, _ in ae.named_modules()]),
               ef_cv1_names=np.asarray([n for n, _ in ef.get_params_of_cv(1)]),
               ae_cv1_names=np.asarray([n for n, _ in ae.get_params_of_cv(1)]),
               ae_cv1_shapes=np.asarray([list(p.shape) + [0] * (2 - p.dim()) for _, p in ae.get_params_of_cv(1)]),
               ae_cv1_last_w=ae.get_params_of_cv(1)[-2][1].detach().numpy(),
               x30=x30.numpy(), x66=x66.numpy(), ef_out=ef(x30).detach().numpy(), ae_out=ae(x66).detach().numpy(),
               enc_out=ae.encoder(x66).detach().numpy(), ae_encoded_dim=ae.encoded_dim)
    out.update({f"ef/{k}": v.numpy() for k, v in sd_ef.items()})
    out.update({f"ae/{k}": v.numpy() for k, v in sd_ae.items()})
    np.savez_compressed(os.path.join(OUT, "nn_structure.npz"), **out)
    print("  nn_structure: ef params", out["ef_nparams"], "ae params", out["ae_nparams"])


def run_utils():
    """G6: integrate_sde_overdamped / calc_weights / WeightedTrajectory(text) of the reference (utils.py:62-169,257-417)."""
    import contextlib
    import io
    import colvarsfinder.utils as rutils
    assert rutils.__file__.startswith(REF)

    class Pot:
        dim, beta = 2, 1.5

        def V(self, x):
            return (x[0] ** 2 - 1) ** 2 + 2.0 * x[1] ** 2

        def gradV(self, x):
            return np.array([4 * x[0] * (x[0] ** 2 - 1), 4.0 * x[1]])

    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(77)
        rutils.integrate_sde_overdamped(Pot(), 3000, tmp, pre_steps=50, step_size=0.01, report_interval=10,
                                        report_interval_stdout=1000)
        after = np.random.rand()
        rutils.calc_weights(os.path.join(tmp, "output.csv"), 1.5, 1.0, traj_weight_filename=os.path.join(tmp, "weights.txt"))
        traj_txt = open(os.path.join(tmp, "traj.txt")).read()
        csv_txt = open(os.path.join(tmp, "output.csv")).read()
        w_txt = open(os.path.join(tmp, "weights.txt")).read()
        t = rutils.WeightedTrajectory(traj_filename=os.path.join(tmp, "traj.txt"),
                                      weight_filename=os.path.join(tmp, "weights.txt"), min_w=0.2, max_w=3.0, verbose=False)
    np.savez_compressed(os.path.join(OUT, "utils_2d.npz"), traj_txt=np.asarray(traj_txt), csv_txt=np.asarray(csv_txt),
                        w_txt=np.asarray(w_txt), rng_after=after, wt_traj=t.trajectory, wt_weights=t.weights, wt_dt=t.dt,
                        wt_n_frames=t.n_frames)
    print("  utils_2d:", t.trajectory.shape, "kept of", t.n_frames)


def main():
    os.makedirs(OUT, exist_ok=True)
    core, rnn = import_reference()
    print("reference imported from", core.__file__)
    if "--big-only" in sys.argv:     # add the bench-sized cases without rewriting the other fixtures
        run_big_cases(core, rnn)
        return
    if "--regae-generator-only" in sys.argv:
        t2, w2 = make_2d_traj(600, seed=11)
        run_regae_generator_cases(core, rnn, dict(pp="identity", traj=t2, w=w2), molecule_case(10, 300, seed=21))
        return
    if "--regae-grad-only" in sys.argv:
        t2, w2 = make_2d_traj(600, seed=11)
        run_regae_grad_cases(core, rnn, dict(pp="identity", traj=t2, w=w2), molecule_case(10, 300, seed=21))
        return
    if "--regae-eta-only" in sys.argv:
        t2, w2 = make_2d_traj(600, seed=11)
        run_regae_eta_cases(core, rnn, dict(pp="identity", traj=t2, w=w2), molecule_case(10, 300, seed=21))
        return
    if "--regae-only" in sys.argv:   # add the G7 cases without rewriting the other fixtures
        t2, w2 = make_2d_traj(600, seed=11)
        run_regae_cases(core, rnn, dict(pp="identity", traj=t2, w=w2), molecule_case(10, 300, seed=21))
        return
    run_nn_structure(rnn)
    run_utils()
    if "--utils-only" in sys.argv:
        return

    t2, w2 = make_2d_traj(600, seed=11)
    id2 = dict(pp="identity", traj=t2, w=w2)
    mol10 = molecule_case(10, 300, seed=21)
    mol22 = molecule_case(22, 256, seed=22)
    mix10 = molecule_case(10, 300, seed=23, features="mixed")

    for dtype in (torch.float32, torch.float64):
        # G1: loss_func KATs
        run_loss_kat(core, rnn, "kat_gen_id2_k1", id2, 1, [2, 20, 20, 20, 1], "generator", dtype, 10.0, [1.0], 1.0, True, 101)
        run_loss_kat(core, rnn, "kat_gen_id2_k3", id2, 3, [2, 12, 12, 1], "generator", dtype, 20.0, [1.0, 0.7, 0.4], 2.0, True, 102)
        run_loss_kat(core, rnn, "kat_tr_id2_k2", id2, 2, [2, 20, 20, 20, 1], "transfer", dtype, 20.0, [1.0, 0.2], 1.0, True, 103)
        run_loss_kat(core, rnn, "kat_gen_mol10_k2", mol10, 2, [30, 20, 20, 20, 1], "generator", dtype, 20.0, [1.0, 0.2], 1.0, True, 104)
        run_loss_kat(core, rnn, "kat_gen_mol22_k3", mol22, 3, [66, 20, 20, 20, 1], "generator", dtype, 20.0, [1.0, 0.75, 0.5], 1.0, True, 105)
        run_loss_kat(core, rnn, "kat_tr_mol10_k2", mol10, 2, [30, 20, 20, 20, 1], "transfer", dtype, 20.0, [1.0, 0.2], 1.0, True, 106)
        run_loss_kat(core, rnn, "kat_gen_mix10_k2", mix10, 2, [20, 16, 16, 1], "generator", dtype, 15.0, [1.0, 0.5], 1.5, False, 107)
        # G2: train traces
        run_ef_train(core, rnn, "train_gen_id2_k1", id2, 1, [2, 20, 20, 20, 1], "generator", dtype, 10.0, [1.0], 1.0, 5e-3, 100, 3, 201)
        run_ef_train(core, rnn, "train_tr_id2_k2", id2, 2, [2, 20, 20, 20, 1], "transfer", dtype, 20.0, [1.0, 0.2], 1.0, 5e-3, 100, 3, 202, lag_idx=2)
        run_ef_train(core, rnn, "train_gen_mol22_k3", mol22, 3, [66, 20, 20, 20, 1], "generator", dtype, 20.0, [1.0, 0.75, 0.5], 1.0, 1e-3, 64, 3, 203)
        run_ef_train(core, rnn, "train_tr_mol10_k2", mol10, 2, [30, 20, 20, 20, 1], "transfer", dtype, 20.0, [1.0, 0.2], 1.0, 2e-3, 64, 3, 204, lag_idx=1)
        # G3: autoencoder traces
        run_ae_train(core, rnn, "train_ae_id2", id2, [2, 20, 20, 20, 1], [1, 20, 20, 2], dtype, 5e-3, 100, 4, 301)
        run_ae_train(core, rnn, "train_ae_mol22", mol22, [66, 20, 20, 20, 2], [2, 10, 10, 66], dtype, 1e-3, 64, 3, 302)
    torch.set_default_dtype(torch.float32)
    run_regae_cases(core, rnn, id2, mol10)
    run_regae_eta_cases(core, rnn, id2, mol10)
    run_regae_grad_cases(core, rnn, id2, mol10)
    run_regae_generator_cases(core, rnn, id2, mol10)
    run_big_cases(core, rnn)


if __name__ == "__main__":
    main()
