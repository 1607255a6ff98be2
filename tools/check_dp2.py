#!/usr/bin/env python3
"""Data-parallel check on ONE GPU: the same EigenFunctionTask / AutoEncoderTask training run once in a single process and
once as two ranks (both on cuda:0, `gloo` process group, eager launches) - the two-rank run must reproduce the single-process
losses and parameters up to summation order.  Exercises the sharding, the two all-reduces and the split Adam on the real
kernels; the RCCL + hipGraph variant of the same code path is the driver's multi-GPU run.
    python tools/check_dp2.py            (parent: runs the reference run, spawns the two ranks, compares)
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)


def run(kind, out_path):
    import torch
    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj, diag_coeff_for, make_molecule_traj
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        _dist.init_from_env("gloo")
    n_atoms = 22
    traj, w, ref = make_molecule_traj(n_atoms, 5000, seed=123)
    mapped = kind.endswith("_mm")     # the trajectory behind a memory-mapped .npy file: a rank reads its own rows only
    kind = kind[:-3] if mapped else kind
    if mapped:
        from colvarsfinder.utils import MappedTrajectory
        path = f"/tmp/dp2_traj_{os.getpid()}.npy"
        np.save(path, traj.astype(np.float32))
        traj_obj = MappedTrajectory(path, weights=w, dt=0.5)
    else:
        traj_obj = Traj(traj, w, 0.5)
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))])
    torch.manual_seed(7)
    np.random.seed(11)
    if kind in ("gen", "tr"):
        model = nn.EigenFunctions([66, 20, 20, 20, 1], 3)
        a = torch.tensor(diag_coeff_for(n_atoms, 5), dtype=torch.float32)
        task = core.EigenFunctionTask(traj_obj, layer, model, "/tmp/cvf_dp2", 20.0, [1.0, 0.7, 0.4], diag_coeff=a, beta=1.0,
                                      lag_tau=0 if kind == "gen" else 1.0, learning_rate=2e-3, k=3, batch_size=1000, num_epochs=2,
                                      device=dev, verbose=False, save_model_every_step=0)
    elif kind == "regae":   # reconstruction + transfer-operator regulariser + latent penalties; the feature trajectory stays whole per rank
        model = nn.RegAutoEncoder([66, 20, 2], [2, 20, 66], [2, 20, 1], 2)
        task = core.RegAutoEncoderTask(traj_obj, layer, model, "/tmp/cvf_dp2", eig_weights=[1.0, 0.6], learning_rate=2e-3,
                                       batch_size=1000, num_epochs=2, alpha=1.0, gamma=[0.3, 0.2], eta=[0.0, 0.1, 0.1], lag_tau_ae=0.5,
                                       lag_tau_reg=1.0, device=dev, verbose=False, save_model_every_step=0)
    elif kind == "regae_gen":   # generator-mode regulariser (lag_tau_reg = 0) + the encoder's gradient-norm penalty eta[0]: both run on an
        # inner EigenFunctionTask whose batch sums join the cross-rank sums (VERDICT r3 item 7 / "missing" 3; core.py:899-916, 1008-1022)
        torch.manual_seed(7)
        model = nn.RegAutoEncoder([66, 20, 20, 2], [2, 20, 66], [2, 20, 1], 2)
        task = core.RegAutoEncoderTask(traj_obj, layer, model, "/tmp/cvf_dp2", eig_weights=[1.0, 0.6], learning_rate=2e-3,
                                       batch_size=1000, num_epochs=2, alpha=1.0, gamma=[0.3, 0.2], eta=[0.05, 0.1, 0.0], lag_tau_ae=0.5,
                                       lag_tau_reg=0, device=dev, verbose=False, save_model_every_step=0)
    else:
        model = nn.AutoEncoder([66, 20, 20, 2], [2, 10, 66])
        task = core.AutoEncoderTask(traj_obj, layer, model, "/tmp/cvf_dp2", learning_rate=2e-3, batch_size=1000, num_epochs=2,
                                    device=dev, verbose=False, save_model_every_step=0)
    task.train()
    torch.cuda.synchronize()
    if _dist.rank() == 0:
        losses = np.concatenate([np.asarray(e[0]).reshape(len(e[0]), -1) for e in task.loss_list])
        # (an eigenfunction's last bias has exact gradient 0 - the loss is shift-invariant - and random-walks on roundoff
        #  under Adam in any run: left out of the comparison, as in tests/test_gpu_parity.py)
        skip = (lambda n: n.endswith(".4.bias")) if kind in ("gen", "tr") else \
               (lambda n: n.startswith("reg.") and n.endswith(".2.bias")) if kind.startswith("regae") else (lambda n: False)
        params = np.concatenate([p.detach().cpu().numpy().reshape(-1) for n, p in model.named_parameters() if not skip(n)])
        np.savez(out_path, losses=losses, params=params)
    # frames this process keeps in HBM (the trajectory / feature rows and train()'s gathers), for the 1/world check
    with open(out_path.replace(".npz", f"_r{_dist.rank()}.json"), "w") as fh:
        json.dump(dict(resident_bytes=int(getattr(task, "resident_bytes", 0)), world=_dist.world(), frames=5000, n_atoms=n_atoms,
                       host_bytes_read=int(traj_obj.trajectory.bytes_read) if mapped else None), fh)
    if mapped:
        os.remove(path)
    if _dist.world() > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return run(sys.argv[2], sys.argv[3])
    report = {}
    ok = True
    for kind in ("gen", "tr", "ae", "gen_mm", "ae_mm", "regae", "regae_gen"):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        env["CVF_GRAPH"] = "0"      # eager in both runs: the comparison is about the data-parallel arithmetic
        subprocess.run([sys.executable, __file__, "worker", kind, f"/tmp/dp2_{kind}_w1.npz"], check=True, env=env, timeout=300)
        procs = []
        import socket
        with socket.socket() as sk:           # a free rendezvous port
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        for r in range(2):
            e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
            procs.append(subprocess.Popen([sys.executable, __file__, "worker", kind, f"/tmp/dp2_{kind}_w2.npz"], env=e))
        for p in procs:
            if p.wait(timeout=300) != 0:
                raise SystemExit(f"rank failed for {kind}")
        a, b = np.load(f"/tmp/dp2_{kind}_w1.npz"), np.load(f"/tmp/dp2_{kind}_w2.npz")
        dl = float(np.max(np.abs(a["losses"] - b["losses"]) / np.maximum(np.abs(a["losses"]), 1e-3)))
        dp = float(np.max(np.abs(a["params"] - b["params"])))
        # shard residency: a rank of the two-rank job holds about half of the frames the job touches - its slices
        # of the static batches (+ their lagged partners in transfer mode) - never the whole trajectory
        res = [json.load(open(f"/tmp/dp2_{kind}_w2_r{r}.json"))["resident_bytes"] for r in range(2)]
        per_frame = {"gen": 22 * 12 + 4, "tr": 2 * (22 * 12 + 4), "ae": 66 * 4, "regae": 0, "regae_gen": 0}[kind.replace("_mm", "")]
        whole = 5000 * per_frame
        report[kind] = dict(steps=int(a["losses"].shape[0]), max_rel_loss_diff=dl, max_abs_param_diff=dp,
                            resident_bytes_per_rank=res, whole_set_bytes=whole)
        ok = ok and dl < 2e-4 and dp < 2e-3 and all(0.35 * whole <= r <= 0.505 * whole for r in res)
        if kind.endswith("_mm"):
            # host side too: a rank of the two-rank job READS about half of the trajectory file (its slices of the batches),
            # the single process all of it, once
            host = [json.load(open(f"/tmp/dp2_{kind}_w2_r{r}.json"))["host_bytes_read"] for r in range(2)]
            host1 = json.load(open(f"/tmp/dp2_{kind}_w1_r0.json"))["host_bytes_read"]
            file_bytes = 5000 * 22 * 12
            report[kind].update(host_bytes_read_per_rank=host, host_bytes_read_single_process=host1, file_bytes=file_bytes)
            ok = ok and all(0.35 * file_bytes <= h <= 0.505 * file_bytes for h in host) and host1 == file_bytes
    print(json.dumps(dict(check="two ranks (gloo, one GPU) vs one process", ok=ok, **report)))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
