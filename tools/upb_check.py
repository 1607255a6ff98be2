#!/usr/bin/env python3
"""Runs ON THE GPU BOX: one generator-mode and one transfer-mode step at the config-3 shape with one unit per workgroup and with the
default (several), twice each: batch sums, loss vector and flat gradient must be equal BIT FOR BIT (the units of a workgroup are
independent; only the launch shape changes)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from colvarsfinder import core, nn, pp
from tests.synth import Traj, diag_coeff_for, make_molecule_traj

dev = torch.device("cuda:0")
out = {}
for lag in (0, 3):
    res = {}
    for upb in ("1", "1", "", "", "3"):
        if upb:
            os.environ["CVF_EF16_UPB"] = upb
        else:
            os.environ.pop("CVF_EF16_UPB", None)
        n_atoms, B, k = 22, 20000 + 37, 3
        traj, w, ref = make_molecule_traj(n_atoms, B + lag, seed=99)
        layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))])
        torch.manual_seed(5)
        model = nn.EigenFunctions([66, 20, 20, 20, 1], k)
        a = torch.tensor(diag_coeff_for(n_atoms, 3), dtype=torch.float32) if lag == 0 else None
        task = core.EigenFunctionTask(Traj(traj[:64], w[:64], 0.5), layer, model, "/tmp/cvf_upb", 20.0, [1.0, 0.7, 0.4], diag_coeff=a, beta=1.0,
                                      lag_tau=0.5 * lag, k=k, device=dev, verbose=False, save_model_every_step=0)
        X, W = torch.tensor(traj[:B]), torch.tensor(w[:B])
        Xl, Wl = (torch.tensor(traj[lag:lag + B]), torch.tensor(w[lag:lag + B])) if lag else (None, None)
        lv = task.loss_func(X, W, Xl, Wl)
        task.backward()
        ws = task._ws[B]
        res.setdefault(upb or "default", []).append((ws.stats.cpu().clone(), ws.loss_vec.cpu().clone(), task._flat.grad.cpu().clone()))
    base = res["1"][0]
    out["lag%d" % lag] = {name: [bool(all(torch.equal(a_, b_) for a_, b_ in zip(r, base))) for r in runs] for name, runs in res.items()}
print(json.dumps(out))
