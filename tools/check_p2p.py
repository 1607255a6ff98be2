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
        os.environ["CVF_COMM"] = "p2p" if mode == "p2p" else "rccl"      # ("rccl": the process group's own all_reduce - gloo here)
        os.environ["CVF_FUSED_COMM"] = "0"
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
    # ---- (4) the data-parallel step in FOUR launches: both sums folded into the finishing launch and the slab
    #      reduction (cvf_ef16_finish_dp / cvf_ef_stats_dp / cvf_ef_loss_dp, cvf_slab_reduce_dp + Adam) - eagerly and replayed from
    #      the epoch hipGraphs - must reproduce the separate all-reduce launches over the same windows BIT FOR BIT (same rank-order
    #      sums, same Adam arithmetic), generator and transfer mode, the fast layout and a shape outside it
    fused = {}
    shapes = [("ef16_gen", [66, 12, 12, 1], 2, 0, True), ("ef16_tr", [66, 12, 12, 1], 2, 2, True), ("generic_gen", [30, 16, 16, 1], 2, 0, False)]
    if world <= 2:
        shapes.append(("c3_gen", [66, 20, 20, 20, 1], 3, 0, True))     # (207 workgroups of 16 waves per rank: two ranks fit one GPU)
    for tag, dims, k, lag, pos_only in shapes:
        rows = {}
        for mode, env in (("fused_graph", dict(CVF_COMM="p2p", CVF_FUSED_COMM="1", CVF_GRAPH="1")),
                          ("fused_eager", dict(CVF_COMM="p2p", CVF_FUSED_COMM="1", CVF_GRAPH="0")),
                          ("separate", dict(CVF_COMM="p2p", CVF_FUSED_COMM="0", CVF_GRAPH="0"))):
            os.environ.update(env)
            n_atoms = dims[0] // 3 if pos_only else 12
            traj, w, ref = make_molecule_traj(n_atoms, 3000 + lag, seed=77)
            feats = [("position", tuple(range(n_atoms)))] if pos_only else \
                [("position", (0, 2, 3, 5, 7, 9)), ("bond", (0, 1)), ("bond", (2, 7)), ("angle", (1, 2, 3)), ("dihedral", (0, 1, 2, 3)),
                 ("dihedral", (4, 5, 6, 7)), ("angle", (6, 8, 9)), ("bond", (3, 4)), ("bond", (5, 6)), ("angle", (9, 10, 11)), ("bond", (10, 11))]
            layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, feats)
            assert layer.d_r == dims[0], (layer.d_r, dims)
            torch.manual_seed(5)
            np.random.seed(11)
            model = nn.EigenFunctions(dims, k)
            a = torch.tensor(diag_coeff_for(n_atoms, 5), dtype=torch.float32) if lag == 0 else None
            task = core.EigenFunctionTask(Traj(traj, w, 0.5), layer, model, "/tmp/cvf_p2p", 20.0, [1.0, 0.7, 0.4][:k], diag_coeff=a, beta=1.0,
                                          lag_tau=0.5 * lag, learning_rate=2e-3, k=k, batch_size=600, num_epochs=3, device=dev, verbose=False,
                                          save_model_every_step=0)
            launches = {}
            task._events = launches if mode == "fused_eager" else None
            task.train()
            torch.cuda.synchronize()
            rows[mode] = (np.concatenate([np.asarray(e[0]).reshape(len(e[0]), -1) for e in task.loss_list]),
                          torch.cat([p_.detach().reshape(-1).cpu() for p_ in model.parameters()]).numpy())
            if mode == "fused_eager":
                rep[f"launches_{tag}"] = sorted(launches.keys())
        fused[tag] = dict(graph_equals_eager=bool(np.array_equal(rows["fused_graph"][0], rows["fused_eager"][0]) and
                                                  np.array_equal(rows["fused_graph"][1], rows["fused_eager"][1])),
                          fused_equals_separate=bool(np.array_equal(rows["fused_eager"][0], rows["separate"][0]) and
                                                     np.array_equal(rows["fused_eager"][1], rows["separate"][1])),
                          finite=bool(np.isfinite(rows["fused_graph"][0]).all()))
    rep["fused"] = fused
    rep["fused_error_word"] = _dist.p2p_error()
    rep["world"] = world
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(rep, fh)
    dist.barrier()
    dist.destroy_process_group()


def worker_timeout(out_path):
    """a peer that arrives later than the time-out must make the job fail LOUDLY.  Rank 1 sleeps 2.5 s before
    its second all-reduce with the time-out set to 0.4 s: rank 0's kernel gives up, fills its result with NaN and sets the
    host-visible error word; _dist.check_comm() - what the tasks call wherever they read results back - must raise there.  The
    same for the exchange folded into cvf_slab_reduce_dp."""
    import time
    import torch
    import torch.distributed as dist
    from colvarsfinder import _dist, _hip
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["CVF_COMM"] = "p2p"
    os.environ["CVF_P2P_TIMEOUT_MS"] = "400"
    _dist.init_from_env("gloo")
    rank = _dist.rank()
    rep = dict(rank=rank)
    which = os.environ.get("CVF_TIMEOUT_CASE", "flag")
    t = torch.full((100,), float(rank + 1), device=dev)
    _dist.allreduce_sum_(t)                       # brings the windows up; both ranks in step
    torch.cuda.synchronize()
    rep["first_ok"] = bool((t == 3.0).all()) and _dist.p2p_error() == 0
    dist.barrier()
    if rank == 1:
        time.sleep(2.5)
    t = torch.full((100,), float(rank + 1), device=dev)
    if which == "flag":
        _dist.allreduce_sum_(t)
    else:
        d = torch.ones(13, device=dev, dtype=torch.float64)
        _dist.allreduce_sum_(d)                  # (the statistics exchange whose number the gradient exchange carries; rank 0 times out here already)
        g = torch.empty_like(t)
        _hip.check(_hip.lib().cvf_slab_reduce_dp(_hip.ptr(t), 1, 100, _hip.ptr(g), None, _dist.fused_comm(), _hip.stream()), "cvf_slab_reduce_dp")
        t = g
    torch.cuda.synchronize()
    rep["result_is_nan"] = bool(torch.isnan(t).all())
    rep["error_word"] = _dist.p2p_error()
    try:
        _dist.check_comm()
        rep["raised"] = False
    except RuntimeError as exc:
        rep["raised"] = True
        rep["message"] = str(exc)[:120]
    with open(f"{out_path}.{rank}", "w") as fh:
        json.dump(rep, fh)
    dist.barrier()
    dist.destroy_process_group()


def main_timeout():
    import socket
    out_all = {}
    for case in ("flag", "fused"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = f"/tmp/check_p2p_timeout_{case}.json"
        procs = [subprocess.Popen([sys.executable, __file__, "worker_timeout", out],
                                  env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                                           CVF_TIMEOUT_CASE=case)) for r in range(2)]
        rc = [p.wait(timeout=300) for p in procs]
        if any(rc):
            raise SystemExit(f"a rank failed: {rc}")
        out_all[case] = [json.load(open(f"{out}.{r}")) for r in range(2)]
    ok = all(reps[0]["first_ok"] and reps[1]["first_ok"] and reps[0]["raised"] and reps[0]["result_is_nan"] and reps[0]["error_word"] != 0
             and not reps[1]["raised"] for reps in out_all.values())
    print(json.dumps(dict(check="a late peer makes the job fail loudly", ok=ok, **out_all)))
    if not ok:
        raise SystemExit(1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker(sys.argv[2])
    if len(sys.argv) > 1 and sys.argv[1] == "worker_timeout":
        return worker_timeout(sys.argv[2])
    if len(sys.argv) > 1 and sys.argv[1] == "timeout":
        return main_timeout()
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
          rep["train_error_word"] == 0 and rep["train_max_rel_diff"] < 1e-5 and rep["fused_error_word"] == 0 and
          all(v["graph_equals_eager"] and v["fused_equals_separate"] and v["finite"] for v in rep["fused"].values()))
    print(json.dumps(dict(check="one-shot P2P reduce, several ranks on one GPU", ok=ok, **rep)))
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
