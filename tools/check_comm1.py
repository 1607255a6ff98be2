#!/usr/bin/env python3
"""One-rank check of the data-parallel code path on ONE GPU with REAL RCCL collectives inside the captured hipGraphs: an
initialised one-rank `nccl` process group with CVF_FORCE_COLLECTIVES=1 makes EigenFunctionTask run its two all-reduces, unfused
Adam and the graph capture with the collectives inside - once through torch.distributed, once through the C ABI's cvf_comm_*
(CVF_COMM=abi) - and both must reproduce the plain single-process run (a one-rank sum changes nothing).
    python tools/check_comm1.py          (parent: runs the three variants as child processes, compares)
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)


def run(out_path):
    import torch
    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj, diag_coeff_for, make_molecule_traj
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    _dist.init_from_env("nccl")
    n_atoms = 22
    traj, w, ref = make_molecule_traj(n_atoms, 4000, seed=321)
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))])
    out = {"collectives": bool(_dist.collectives()), "abi": os.environ.get("CVF_COMM", "") == "abi"}
    for kind, lag in (("gen", 0), ("tr", 1.0)):
        torch.manual_seed(7)
        np.random.seed(11)
        model = nn.EigenFunctions([66, 20, 20, 20, 1], 3)
        a = torch.tensor(diag_coeff_for(n_atoms, 5), dtype=torch.float32)
        task = core.EigenFunctionTask(Traj(traj, w, 0.5), layer, model, "/tmp/cvf_comm1", 20.0, [1.0, 0.7, 0.4], diag_coeff=a, beta=1.0,
                                      lag_tau=lag, learning_rate=2e-3, k=3, batch_size=800, num_epochs=3, device=dev, verbose=False,
                                      save_model_every_step=0)
        task.train()
        out[kind] = {"loss": np.stack([e[0][:, 0].numpy() for e in task.loss_list]).tolist(), "graphs": bool(task._use_graphs),
                     "theta": task._flat.theta.cpu().numpy().tolist()}
    if out["abi"]:
        out["abi_comm_created"] = _dist._abi_comm is not None
    json.dump(out, open(out_path, "w"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return run(sys.argv[2])
    res = {}
    for name, env in (("plain", {}), ("torch", {"CVF_FORCE_COLLECTIVES": "1"}), ("abi", {"CVF_FORCE_COLLECTIVES": "1", "CVF_COMM": "abi"})):
        e = dict(os.environ)
        e.update(env)
        if env:
            e.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29871"})
        path = f"/tmp/cvf_comm1_{name}.json"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=e, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(json.dumps({"ok": False, "failed": name, "stderr": r.stderr[-1500:]}))
            return 1
        res[name] = json.load(open(path))
    rep = {"ok": True}
    for name in ("torch", "abi"):
        assert res[name]["collectives"], name
        for kind in ("gen", "tr"):
            dl = float(np.max(np.abs(np.array(res[name][kind]["loss"]) - np.array(res["plain"][kind]["loss"])) /
                              np.abs(np.array(res["plain"][kind]["loss"]))))
            dt = float(np.max(np.abs(np.array(res[name][kind]["theta"]) - np.array(res["plain"][kind]["theta"]))))
            rep[f"{name}_{kind}"] = {"max_rel_loss_diff": dl, "max_abs_param_diff": dt, "graphs": res[name][kind]["graphs"]}
            rep["ok"] = rep["ok"] and dl < 1e-5 and dt < 1e-4
    rep["abi_comm_created"] = res["abi"].get("abi_comm_created", False)
    rep["ok"] = rep["ok"] and rep["abi_comm_created"]
    print(json.dumps(rep))
    return 0 if rep["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
