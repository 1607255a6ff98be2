#!/usr/bin/env python3
"""The one-shot peer-to-peer reduce (cvf_p2p_*, csrc/p2p.hip) with SEVERAL PROCESSES ON ONE GPU: every rank maps the others'
windows through HIP IPC handles, then sums vectors of the step's two sizes (fp64 batch sums, fp32 flat gradient) many times,
eagerly and from a captured hipGraph, and checks the result bit for bit against the sum formed in rank order from the gathered
inputs (what cvf_p2p_* promises on every rank) - and a short EigenFunctionTask training with CVF_COMM=p2p against the same
training over the process group's own all_reduce.   python tools/check_p2p.py [world]      (parent: spawns the ranks)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    sys.path.insert(0, p)


def worker(out_path):
    import torch
    import torch.distributed as dist
    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj, diag_coeff_for, make_molecule_traj
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    _dist.init_from_env("gloo")
    rank, world = _dist.rank(), _dist.world()
    rep = {}
    # ---- (1) raw all-reduces, both dtypes, several sizes, 40 rounds each (the parity flips every call), vs rank-order sums
    os.environ["CVF_COMM"] = "p2p"
    worst = 0
    for dtype, sizes in ((torch.float64, (13, 34, 70)), (torch.float32, (6603, 51366, 1))):
        for n in sizes:
            for it in range(40):
                g = torch.Generator().manual_seed(1000 * rank + 7 * it + n)
                x = (torch.randn(n, generator=g, dtype=torch.float64) * 10.0 ** float(it % 5 - 2)).to(dtype)
                parts = [torch.zeros_like(x) for _ in range(world)]
                dist.all_gather(parts, x)
                want = parts[0].clone()
                for r in range(1, world):
                    want = want + parts[r]            # rank order, same dtype: what every rank must hold
                t = x.to(dev)
                _dist.allreduce_sum_(t)
                got = t.cpu()
                if not torch.equal(got, want):
                    worst += 1
    rep["raw_mismatches"] = worst
    rep["raw_error_word"] = _dist.p2p_error()
    # ---- (2) from a captured hipGraph (the epoch lives on the device): replay 25 times with fresh inputs
    buf = torch.zeros(6603, device=dev)
    src = torch.zeros(6603, device=dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    dist.barrier()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        buf.copy_(src)
        _dist.allreduce_sum_(buf)       # warm-up on the side stream (capture needs one)
        torch.cuda.synchronize()
        dist.barrier()
        with torch.cuda.graph(g, stream=s):
            buf.copy_(src)
            _dist.allreduce_sum_(buf)
    bad = 0
    for it in range(25):
        gen = torch.Generator().manual_seed(5000 + 31 * it + rank)
        x = torch.randn(6603, generator=gen)
        parts = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        want = parts[0].clone()
        for r in range(1, world):
            want = want + parts[r]
        src.copy_(x)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        bad += 0 if torch.equal(buf.cpu(), want) else 1
    rep["graph_mismatches"] = bad
    rep["graph_error_word"] = _dist.p2p_error()
    # ---- (3) a short sharded training with the two sums over cvf_p2p_* vs the process group's all_reduce
    losses = {}
    for mode in ("p2p", "group"):
        os.environ["CVF_COMM"] = "p2p" if mode == "p2p" else ""
        os.environ["CVF_GRAPH"] = "0"
        n_atoms = 22
        traj, w, ref = make_molecule_traj(n_atoms, 4000, seed=321)
        layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, [("position", tuple(range(n_atoms)))])
        torch.manual_seed(3)
        np.random.seed(9)
        model = nn.EigenFunctions([66, 20, 20, 20, 1], 3)
        a = torch.tensor(diag_coeff_for(n_atoms, 5), dtype=torch.float32)
        task = core.EigenFunctionTask(Traj(traj, w, 0.5), layer, model, "/tmp/cvf_p2p", 20.0, [1.0, 0.7, 0.4], diag_coeff=a, beta=1.0,
                                      lag_tau=0, learning_rate=2e-3, k=3, batch_size=800, num_epochs=2, device=dev, verbose=False,
                                      save_model_every_step=0)
        task.train()
        torch.cuda.synchronize()
        losses[mode] = np.concatenate([np.asarray(e[0]).reshape(len(e[0]), -1) for e in task.loss_list])
    rep["train_max_rel_diff"] = float(np.max(np.abs(losses["p2p"] - losses["group"]) / np.maximum(np.abs(losses["group"]), 1e-3)))
    rep["train_error_word"] = _dist.p2p_error()
    rep["world"] = world
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(rep, fh)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker(sys.argv[2])
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = f"/tmp/check_p2p_w{world}.json"
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, __file__, "worker", out], env=e))
    rc = [p.wait(timeout=600) for p in procs]
    if any(rc):
        raise SystemExit(f"a rank failed: {rc}")
    rep = json.load(open(out))
    ok = (rep["raw_mismatches"] == 0 and rep["graph_mismatches"] == 0 and rep["raw_error_word"] == 0 and rep["graph_error_word"] == 0 and
          rep["train_error_word"] == 0 and rep["train_max_rel_diff"] < 1e-5)
    print(json.dumps(dict(check="one-shot P2P reduce, several ranks on one GPU", ok=ok, **rep)))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
