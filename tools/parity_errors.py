#!/usr/bin/env python3
"""Achieved parity errors of the HIP path against every reference-generated fixture -> profiles/<tag>_parity_errors.json.

north_star: "per-step loss and learned-CV outputs within 1e-5 relative" of the reference's CPU path.  The tests state
tolerances; this script records the DISTANCES: for each fixture (the reference's own fp32 run and its fp64 run, written by
tools/gen_golden.py) the maximum relative error of
  * the loss and the eigenvalues of one loss_func call (known-answer fixtures),
  * every per-step loss of a training trace (train and test rows, column 0),
  * the learned-CV outputs on the probe frames after training (relative to the largest |CV|; eigenfunction CVs up to the
    additive constant the loss does not determine, as in tests/test_gpu_parity.py).
It also records the reference's own fp32-vs-fp64 distance on the same quantity, which is the noise floor a comparison with
its fp32 run cannot go below.  Runs on the GPU box:  python tools/parity_errors.py [tag]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from tests import goldens  # noqa: E402
from tests.synth import Traj  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def rel_scale(a, b):
    """max |a - b| / max |b| (for vectors with entries near zero)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def main():
    tag_out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r4"
    from tests.test_gpu_parity import build_task, make_layer
    from colvarsfinder import core, nn
    dev = torch.device("cuda:0")
    out = {"_what": "max relative error of the MI355X path vs the reference's own runs (fixtures of tools/gen_golden.py); "
                    "ref_f32_vs_f64 = the same distance between the reference's fp32 and fp64 runs (its own noise floor)",
           "_north_star_tolerance": 1e-5, "kat": {}, "ef_train": {}, "ae_train": {}}

    for name in goldens.KAT_CASES:
        row = {}
        g32, g64 = goldens.load(name, "f32"), goldens.load(name, "f64")
        for tag, g in (("f32", g32), ("f64", g64)):
            task, _ = build_task(g, dev)
            lag = int(g["lag_idx"])
            traj, w = np.array(g["traj"]), np.array(g["w"])
            B = traj.shape[0] - lag
            Xl = torch.tensor(traj[lag:lag + B]) if lag else None
            wl = torch.tensor(w[lag:lag + B]) if lag else None
            loss, eig, npl, pen, cvec = task.loss_func(torch.tensor(traj[:B]), torch.tensor(w[:B]), Xl, wl)
            row[f"loss_vs_{tag}"] = rel(float(loss), float(g["loss"]))
            row[f"eig_vs_{tag}"] = rel(eig.numpy(), g["eig"])
            row[f"npl_vs_{tag}"] = rel(float(npl), float(g["npl"]))
            row[f"pen_vs_{tag}"] = abs(float(pen) - float(g["pen"])) / max(abs(float(g["pen"])), abs(float(g["loss"])) / float(g["alpha"]))
            task.backward()
            names = [n for n, _ in task.model.named_parameters()]
            gmax = max(float(np.abs(g["grad/" + n]).max()) for n in names)
            row[f"grad_vs_{tag}"] = max(float(np.abs(p_.grad.cpu().numpy() - g["grad/" + n]).max()) for n, p_ in task.model.named_parameters()) / gmax
        row["ref_f32_vs_f64_loss"] = rel(float(g32["loss"]), float(g64["loss"]))
        row["ref_f32_vs_f64_eig"] = rel(g32["eig"], g64["eig"])
        out["kat"][name] = row

    for name in goldens.EF_TRAIN_CASES:
        row = {}
        g32, g64 = goldens.load(name, "f32"), goldens.load(name, "f64")
        for tag, g in (("f32", g32), ("f64", g64)):
            task, _ = build_task(g, dev)
            np.random.seed(int(g["seed"]))
            task.train()
            tr = np.stack([e[0].numpy() for e in task.loss_list])
            te = np.stack([e[1].numpy() for e in task.loss_list])
            row[f"step_loss_vs_{tag}"] = max(rel(tr[..., 0], np.array(g["train_loss"])[..., 0]), rel(te[..., 0], np.array(g["test_loss"])[..., 0]))
            # every column of the rows [loss, npl, pen, eig_1..k], as the test compares them: |got - want| / (|want| + 1)
            row[f"step_rows_vs_{tag}"] = max(float(np.max(np.abs(a_ - np.array(b_)) / (np.abs(np.array(b_)) + 1.0)))
                                             for a_, b_ in ((tr, g["train_loss"]), (te, g["test_loss"])))
            last_bias = f".{len(g['layer_dims']) - 1}.bias"
            row[f"params_vs_{tag}"] = max(float(np.max(np.abs(p_.cpu().numpy() - g["final/" + n]) / (np.abs(g["final/" + n]) + 1.0)))
                                          for n, p_ in task.model.state_dict().items() if not n.endswith(last_bias))
            cv = task.colvar_model()(torch.tensor(np.array(g["traj"])[:64], dtype=torch.float32)).detach().numpy()
            rc = np.array(g["colvar_probe"])
            row[f"cv_vs_{tag}"] = rel_scale(cv - cv.mean(0), rc - rc.mean(0))
            row["steps"] = int(tr.shape[0] * tr.shape[1])
        row["ref_f32_vs_f64_step_loss"] = rel(np.array(g32["train_loss"])[..., 0], np.array(g64["train_loss"])[..., 0])
        a, b = np.array(g32["colvar_probe"]), np.array(g64["colvar_probe"])
        row["ref_f32_vs_f64_cv"] = rel_scale(a - a.mean(0), b - b.mean(0))
        out["ef_train"][name] = row

    for name in goldens.AE_TRAIN_CASES:
        row = {}
        g32, g64 = goldens.load(name, "f32"), goldens.load(name, "f64")
        for tag, g in (("f32", g32), ("f64", g64)):
            e_dims, d_dims = [int(d) for d in g["e_dims"]], [int(d) for d in g["d_dims"]]
            model = nn.AutoEncoder(e_dims, d_dims)
            model.load_state_dict(goldens.state_dict(g, dtype=torch.float32))
            traj = np.array(g["traj"])
            layer = make_layer(goldens.pp_spec(g), traj.shape[1] if traj.ndim == 3 else 0, dev)
            task = core.AutoEncoderTask(Traj(traj, np.array(g["w"]), 0.5), layer, model, "/tmp/cvf_parity", learning_rate=float(g["lr"]),
                                        batch_size=int(g["batch_size"]), num_epochs=int(g["num_epochs"]), device=dev, verbose=False,
                                        save_model_every_step=0)
            np.random.seed(int(g["seed"]))
            task.train()
            tr = np.stack([e[0].numpy() for e in task.loss_list])
            te = np.stack([e[1].numpy() for e in task.loss_list])
            row[f"step_loss_vs_{tag}"] = max(rel(tr, g["train_loss"]), rel(te, g["test_loss"]))
            row[f"params_vs_{tag}"] = max(float(np.max(np.abs(p_.cpu().numpy() - g["final/" + n]) / (np.abs(g["final/" + n]) + 1.0)))
                                          for n, p_ in task.model.state_dict().items())
            cv = task.colvar_model()(torch.tensor(traj[:64], dtype=torch.float32)).detach().numpy()
            row[f"cv_vs_{tag}"] = rel_scale(cv, g["colvar_probe"])
            row["steps"] = int(tr.size)
        row["ref_f32_vs_f64_step_loss"] = rel(g32["train_loss"], g64["train_loss"])
        row["ref_f32_vs_f64_cv"] = rel_scale(g32["colvar_probe"], g64["colvar_probe"])
        out["ae_train"][name] = row

    out["regae_train"] = {}
    for name in goldens.REGAE_TRAIN_CASES:   # RegAutoEncoderTask: every option incl. eta[0] and the generator-mode regulariser
        row = {}
        g32, g64 = goldens.load(name, "f32"), goldens.load(name, "f64")
        for tag, g in (("f32", g32), ("f64", g64)):
            e_dims, d_dims, r_dims = ([int(d) for d in g[k_]] for k_ in ("e_dims", "d_dims", "r_dims"))
            K, lag_ae, lag_reg, dt = int(g["K"]), int(g["lag_ae"]), int(g["lag_reg"]), float(g["dt"])
            model = nn.RegAutoEncoder(e_dims, d_dims, r_dims, K)
            model.load_state_dict(goldens.state_dict(g, dtype=torch.float32))
            traj = np.array(g["traj"])
            layer = make_layer(goldens.pp_spec(g), traj.shape[1] if traj.ndim == 3 else 0, dev)
            task = core.RegAutoEncoderTask(Traj(traj, np.array(g["w"]), dt), layer, model, "/tmp/cvf_parity", eig_weights=[float(v) for v in g["eig_w"]],
                                           learning_rate=float(g["lr"]), batch_size=int(g["batch_size"]), num_epochs=int(g["num_epochs"]),
                                           alpha=float(g["alpha"]), gamma=[float(v) for v in g["gamma"]],
                                           eta=[float(v) for v in g["eta"]] if "eta" in g.files else [0.0, 0.0, 0.0],
                                           lag_tau_ae=lag_ae * dt, lag_tau_reg=lag_reg * dt, beta=float(g["beta"]) if "beta" in g.files else 1.0,
                                           freeze_encoder=bool(g["freeze"]), device=dev, verbose=False, save_model_every_step=0)
            np.random.seed(int(g["seed"]))
            task.train()
            tr = np.stack([e[0].numpy() for e in task.loss_list])
            row[f"first_step_loss_vs_{tag}"] = rel(tr[0, 0, 0], np.array(g["train_loss"])[0, 0, 0])
            row[f"step_loss_vs_{tag}"] = rel(tr[..., 0], np.array(g["train_loss"])[..., 0])
            cv = task.colvar_model()(torch.tensor(traj[:64], dtype=torch.float32)).detach().numpy()
            row[f"cv_vs_{tag}"] = rel_scale(cv, g["colvar_probe"])
            row["steps"] = int(tr.shape[0] * tr.shape[1])
            ad = lambda n, p_: float(np.max(np.abs(p_.cpu().numpy() - g["final/" + n]) / (np.abs(g["final/" + n]) + 1.0)))  # noqa: E731
            rlb = f".{len(r_dims) - 1}.bias"
            sdict = task.model.state_dict()
            row[f"encdec_params_vs_{tag}"] = max(ad(n, p_) for n, p_ in sdict.items() if not n.startswith("reg."))
            row[f"reg_params_vs_{tag}"] = max([ad(n, p_) for n, p_ in sdict.items() if n.startswith("reg.") and not n.endswith(rlb)] + [0.0])
            rp, rr = task.reg_model()(torch.tensor(traj[:64], dtype=torch.float32)).detach().numpy(), np.array(g["reg_probe"])
            row[f"reg_probe_vs_{tag}"] = float(np.max(np.abs((rp - rp.mean(0)) - (rr - rr.mean(0)))) / max(float(np.abs(rr - rr.mean(0)).max()), 1e-300))
        row["ref_f32_vs_f64_step_loss"] = rel(np.array(g32["train_loss"])[..., 0], np.array(g64["train_loss"])[..., 0])
        row["ref_f32_vs_f64_cv"] = rel_scale(g32["colvar_probe"], g64["colvar_probe"])
        out["regae_train"][name] = row

    # the bench-sized fixtures (BASELINE configs 2 / 3 at 100 000 frames, batches of 20 000; tools/gen_golden.py run_big_cases)
    out["bench_size"] = {}
    for name in goldens.BIG_EF_CASES:
        row = {}
        g32, g64 = goldens.load_big(name, "f32"), goldens.load_big(name, "f64")
        for tag, g in (("f32", g32), ("f64", g64)):
            task, _ = build_task(g, dev)
            lag, B = int(g["lag_idx"]), int(g["kat_n"])
            traj, w = g["traj"], g["w"]
            Xl = torch.tensor(traj[lag:lag + B]) if lag else None
            wl = torch.tensor(w[lag:lag + B]) if lag else None
            loss, eig, npl, pen, cvec = task.loss_func(torch.tensor(traj[:B]), torch.tensor(w[:B]), Xl, wl)
            row[f"loss_vs_{tag}"] = rel(float(loss), float(g["kat_loss"]))
            row[f"eig_vs_{tag}"] = rel(eig.numpy(), g["kat_eig"])
            row[f"npl_vs_{tag}"] = rel(float(npl), float(g["kat_npl"]))
            task.backward()
            names = [n for n, _ in task.model.named_parameters()]
            gmax = max(float(np.abs(g["grad/" + n]).max()) for n in names)
            row[f"grad_vs_{tag}"] = max(float(np.abs(p_.grad.cpu().numpy() - g["grad/" + n]).max()) for n, p_ in task.model.named_parameters()) / gmax
            np.random.seed(int(g["seed"]))
            task.train()
            tr = np.stack([e[0].numpy() for e in task.loss_list])
            te = np.stack([e[1].numpy() for e in task.loss_list])
            row[f"step_loss_vs_{tag}"] = max(rel(tr[..., 0], np.array(g["train_loss"])[..., 0]), rel(te[..., 0], np.array(g["test_loss"])[..., 0]))
            row[f"step_rows_vs_{tag}"] = max(float(np.max(np.abs(a_ - np.array(b_)) / (np.abs(np.array(b_)) + 1.0)))
                                             for a_, b_ in ((tr, g["train_loss"]), (te, g["test_loss"])))
            last_bias = f".{len(g['layer_dims']) - 1}.bias"
            row[f"params_vs_{tag}"] = max(float(np.max(np.abs(p_.cpu().numpy() - g["final/" + n]) / (np.abs(g["final/" + n]) + 1.0)))
                                          for n, p_ in task.model.state_dict().items() if not n.endswith(last_bias))
            cv = task.colvar_model()(torch.tensor(traj[:64], dtype=torch.float32)).detach().numpy()
            rc = np.array(g["colvar_probe"])
            row[f"cv_vs_{tag}"] = rel_scale(cv - cv.mean(0), rc - rc.mean(0))
            row["steps"] = int(tr.shape[0] * tr.shape[1])
        row["ref_f32_vs_f64_step_loss"] = rel(np.array(g32["train_loss"])[..., 0], np.array(g64["train_loss"])[..., 0])
        row["ref_f32_vs_f64_step_rows"] = float(np.max(np.abs(np.array(g32["train_loss"]) - g64["train_loss"]) / (np.abs(g64["train_loss"]) + 1.0)))
        a, b = np.array(g32["colvar_probe"]), np.array(g64["colvar_probe"])
        row["ref_f32_vs_f64_cv"] = rel_scale(a - a.mean(0), b - b.mean(0))
        out["bench_size"][name] = row
    row = {}
    g32, g64 = goldens.load_big("big_ae_c2", "f32"), goldens.load_big("big_ae_c2", "f64")
    for tag, g in (("f32", g32), ("f64", g64)):
        model = nn.AutoEncoder([int(d) for d in g["e_dims"]], [int(d) for d in g["d_dims"]])
        model.load_state_dict(goldens.state_dict(g, dtype=torch.float32))
        traj = g["traj"]
        task = core.AutoEncoderTask(Traj(traj, g["w"], 0.5), make_layer(goldens.pp_spec(g), traj.shape[1], dev), model, "/tmp/cvf_parity",
                                    learning_rate=float(g["lr"]), batch_size=int(g["batch_size"]), num_epochs=int(g["num_epochs"]), device=dev,
                                    verbose=False, save_model_every_step=0)
        np.random.seed(int(g["seed"]))
        task.train()
        tr = np.stack([e[0].numpy() for e in task.loss_list])
        te = np.stack([e[1].numpy() for e in task.loss_list])
        row[f"step_loss_vs_{tag}"] = max(rel(tr, g["train_loss"]), rel(te, g["test_loss"]))
        row[f"params_vs_{tag}"] = max(float(np.max(np.abs(p_.cpu().numpy() - g["final/" + n]) / (np.abs(g["final/" + n]) + 1.0)))
                                      for n, p_ in task.model.state_dict().items())
        cv = task.colvar_model()(torch.tensor(traj[:64], dtype=torch.float32)).detach().numpy()
        row[f"cv_vs_{tag}"] = rel_scale(cv, g["colvar_probe"])
    row["ref_f32_vs_f64_step_loss"] = rel(g32["train_loss"], g64["train_loss"])
    out["bench_size"]["big_ae_c2"] = row

    # config-5 shape at the batch sizes bench.py --workload c5 times, vs the chunked fp64 oracle (no reference fixture exists at
    # 5000 atoms: the oracle's alignment layer is the unpinned part)
    if "--no-c5" not in sys.argv:
        from tests.test_gpu_parity import config5_bench_batch_errors
        out["config5_bench_batches"] = {str(B): config5_bench_batch_errors(dev, B) for B in (2000, 16000)}

    def worst(key):
        return max(r[key] for sec in ("kat", "ef_train", "ae_train") for r in out[sec].values() if key in r)

    out["_summary_more"] = {key: worst(key) for key in ("loss_vs_f32", "eig_vs_f64", "eig_vs_f32", "npl_vs_f64", "npl_vs_f32", "pen_vs_f64", "pen_vs_f32",
                                                        "grad_vs_f64", "grad_vs_f32", "step_rows_vs_f64", "step_rows_vs_f32", "params_vs_f64",
                                                        "params_vs_f32")}
    out["_summary"] = {"worst_loss_vs_f64_single_call": worst("loss_vs_f64"), "worst_step_loss_vs_f64_traces": worst("step_loss_vs_f64"),
                       "worst_step_loss_vs_f32_traces": worst("step_loss_vs_f32"), "worst_cv_vs_f64": worst("cv_vs_f64"),
                       "worst_cv_vs_f32": worst("cv_vs_f32"),
                       "reference_own_f32_vs_f64_worst_step_loss": worst("ref_f32_vs_f64_step_loss"),
                       "reference_own_f32_vs_f64_worst_cv": worst("ref_f32_vs_f64_cv"),
                       "regae_worst_first_step_loss_vs_f64": max(r["first_step_loss_vs_f64"] for r in out["regae_train"].values()),
                       "regae_worst_step_loss_vs_f64_traces": max(r["step_loss_vs_f64"] for r in out["regae_train"].values()),
                       "regae_reference_own_f32_vs_f64_worst_step_loss": max(r["ref_f32_vs_f64_step_loss"] for r in out["regae_train"].values()),
                       "regae_worst_cv_vs_f64": max(r["cv_vs_f64"] for r in out["regae_train"].values()),
                       "regae_worst_encdec_params_vs_f64": max(r["encdec_params_vs_f64"] for r in out["regae_train"].values()),
                       "regae_worst_reg_params_vs_f64": max(r["reg_params_vs_f64"] for r in out["regae_train"].values()),
                       "regae_worst_reg_probe_vs_f64": max(r["reg_probe_vs_f64"] for r in out["regae_train"].values()),
                       "bench_size": {n: {k_: v for k_, v in r.items() if k_.endswith("_vs_f64")} for n, r in out["bench_size"].items()},
                       "config5_bench_batches": out.get("config5_bench_batches")}
    path = os.path.join(ROOT, "gpurun_out", f"{tag_out}_parity_errors.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["_summary"]))
    print(json.dumps(out["_summary_more"]))


if __name__ == "__main__":
    main()
