#!/usr/bin/env python3
"""bench.py - MD frames/sec through the EigenFunctionTask train step on MI355X.

Workload (BASELINE.json config 3, SURVEY.md section 8d "C3"): alanine-dipeptide-shaped synthetic
trajectory, 22 atoms, alignment on all atoms + position features (d_r = 66), EigenFunctionTask in
generator mode with k = 3 nets [66,20,20,20,1], diffusion-weighted diag_coeff, alpha = 20, Adam.
One "step" = one full train step on one batch resident in HBM: align+features (K1), nets forward and
input gradients (K4a), q = J A J^T g (K2/K3), batch sums (K5), loss tail, parameter gradient (K4b),
Adam (K6).  With N > 1 (one process per GPU, launched by torch.distributed.run) every rank owns its own
shard of frames and the global batch is N x --batch frames (weak scaling); the two all-reduces of
SURVEY.md section 8e (batch sums before the backward pass, flat gradient after it) run over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement): value = frames of all ranks / second.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

N_ATOMS, K_NETS, LAYERS = 22, 3, [66, 20, 20, 20, 1]
ALPHA, EIG_W, BETA, LR = 20.0, [1.0, 0.75, 0.5], 1.0, 1e-3
# --workload c5: BASELINE config 5 shape (not the default line): 5000 atoms, 32 positions + 96 dihedrals + 96 distances
C5 = dict(n_atoms=5000, k=6, layers=[384, 20, 20, 20, 1], eig_w=[1.0, 0.9, 0.8, 0.7, 0.6, 0.5], batch=2000, frames=16000)
SEED = 20260103
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak

# algorithmic work per frame (DESIGN.md section "Kernels")
P_W = 66 * 20 + 20 * 20 + 20 * 20 + 20            # multiply-adds of one net's forward
K1_BYTES = 12 * N_ATOMS + 4 + 4 * 66              # 532 B/frame
FLOP_FWD = 2 * K_NETS * (P_W + (P_W - 66 * 20) + 66 * 20)            # forward + d-chain + g = W1^T d
FLOP_K1 = 2 * (12 + 12) * N_ATOMS + 1500              # covariance + aligned positions per atom, 3x3 eigen-solve
FLOP_METRIC = 2 * K_NETS * 33 * N_ATOMS          # three passes of q = J A J^T g: ~33 fused multiply-adds per atom and net
FLOP_BWD = 2 * K_NETS * (P_W + 800 + 2 * P_W)  # tangent chain, zbar chain, outer products (h and the d chain come from the forward kernel)


def make_shard(n_frames, rank, n_atoms=N_ATOMS, scale=2.0, sigma=0.3):
    from tests.synth import make_weights, random_rotations
    ref = np.random.RandomState(SEED).normal(scale=scale, size=(n_atoms, 3))
    rs = np.random.RandomState(SEED + 1 + rank)
    Q = random_rotations(rs, n_frames).astype(np.float32)
    t = rs.normal(size=(n_frames, 1, 3)).astype(np.float32)
    x = np.empty((n_frames, n_atoms, 3), dtype=np.float32)
    for s0 in range(0, n_frames, 2000):   # chunked: the config-5 shard is ~1 GB
        xi = rs.normal(scale=sigma, size=(min(2000, n_frames - s0), n_atoms, 3)).astype(np.float32)
        x[s0:s0 + len(xi)] = np.einsum("bij,baj->bai", Q[s0:s0 + len(xi)], ref[None].astype(np.float32) + xi) + t[s0:s0 + len(xi)]
    return x, make_weights(rs, n_frames), ref


def c5_features(n_atoms):
    rs = np.random.RandomState(SEED + 5)
    feats = [("position", tuple(int(i) for i in rs.choice(n_atoms, 32, replace=False)))]
    feats += [("dihedral", tuple(int(i) for i in rs.choice(n_atoms, 4, replace=False))) for _ in range(96)]
    feats += [("bond", tuple(int(i) for i in rs.choice(n_atoms, 2, replace=False))) for _ in range(96)]
    return feats


def cpu_baseline(x, w, ref, a, sd0, batch, budget_s):
    """The oracle's train step (pure PyTorch on the host cores, autograd through linalg.svd like the
    reference's CPU path) on the same first batch; bounded to ~budget_s seconds."""
    from oracle import losses
    from oracle.pp import AlignFeature
    ncores = min(len(os.sched_getaffinity(0)), 16)   # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(ncores)
    pp = AlignFeature(list(range(N_ATOMS)), ref, [("position", tuple(range(N_ATOMS)))])
    sd = {n: p.clone().requires_grad_(True) for n, p in sd0.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=LR)
    X0, w0 = torch.tensor(x[:batch]), torch.tensor(w[:batch], dtype=torch.float32)

    def step():
        X = X0.clone().requires_grad_(True)
        opt.zero_grad(set_to_none=True)
        loss = losses.ef_loss(sd, K_NETS, pp, X, w0, alpha=ALPHA, eig_w=EIG_W, diag_coeff=a, beta=BETA)[0]
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 100:
            break
    return dict(value=batch * n / el, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} train steps of the CPU oracle (PyTorch fp32, autograd through linalg.svd) on the same "
                       f"first batch of {batch} frames, {el:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=20000, help="frames per GPU per step (reference notebook: 20000)")
    ap.add_argument("--frames", type=int, default=100000, help="frames per GPU shard (config 3: 100k)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 disables it)")
    ap.add_argument("--workload", choices=["c3", "c5", "c2", "regae"], default="c3",
                    help="c3 = the benchmark line; c5 = config-5 shape, c2 = config-2 AutoEncoderTask (extra measurements)")
    args = ap.parse_args()
    if args.workload == "c5":
        return main_c5(args)
    if args.workload == "regae":
        return main_regae(args)
    if args.workload == "c2":
        return main_c2(args)

    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj, diag_coeff_for

    _dist.init_from_env("nccl")
    world, rank = _dist.world(), _dist.rank()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)

    x, w, ref = make_shard(args.frames, rank)
    a = torch.tensor(diag_coeff_for(N_ATOMS, SEED), dtype=torch.float32)
    torch.manual_seed(SEED)
    model = nn.EigenFunctions(LAYERS, K_NETS)
    sd0 = {n: p.detach().clone() for n, p in model.state_dict().items()}
    layer = pp.AlignFeatureLayer(N_ATOMS, list(range(N_ATOMS)), ref, [("position", tuple(range(N_ATOMS)))])
    task = core.EigenFunctionTask(Traj(x, w, 1.0), layer, model, "/tmp/cvf_bench", ALPHA, EIG_W, diag_coeff=a, beta=BETA,
                                  lag_tau=0, learning_rate=LR, k=K_NETS, batch_size=args.batch, device=dev, verbose=False,
                                  save_model_every_step=0)
    B = min(args.batch, args.frames)
    n_batches = args.frames // B
    X, Wt = task._traj, task._weights

    log = torch.zeros(n_batches, 3 + 2 * K_NETS, device=dev, dtype=torch.float64)

    pipeline = task._pipeline

    def step(i):
        # the product's own step path (EigenFunctionTask.train): whole-step hipGraph replay per (static) batch, and the
        # alignment kernel of batch i+1 - which does not depend on the parameters - running beside step i
        b = i % n_batches
        s = b * B
        if pipeline:
            s2 = ((i + 1) % n_batches) * B
            task._graph_step(("bench", b, i % 2, i > 0),
                             lambda: task.train_step(X[s:s + B], Wt[s:s + B], slot=i % 2, aligned=i > 0, prefetch=(X[s2:s2 + B], None)),
                             log[b])
        else:
            task._graph_step(("bench", b), lambda: task.train_step(X[s:s + B], Wt[s:s + B]), log[b])
        return log[b]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    graphs = task._use_graphs
    if not graphs:
        task._events = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss_vec = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if graphs:
        # graph replay hides the individual launches: time them with HIP events around every C-ABI call over
        # a second, eager pass of the same K steps (not part of `value`)
        task._use_graphs, task._events, task._last_call = False, {}, {}
        for i in range(args.steps):
            # park the GPU while the host queues this step's launches, so that the events bracket back-to-back
            # kernel executions rather than the host's launch latency
            torch.cuda._sleep(2_000_000)
            step(args.warmup + args.steps + i)
        barrier()
        task._use_graphs = True
    events, task._events = task._events, None
    last_calls, task._last_call = (task._last_call or {}), None
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax)
    final_loss = float(loss_vec[0])
    assert np.isfinite(final_loss), "training diverged"

    # per-kernel average launch duration from the HIP events recorded inside the timed region
    kern_ms = {name: float(np.mean([s_.elapsed_time(e_) for s_, e_ in ev])) for name, ev in events.items()}
    if rank != 0:
        return
    dom = max(kern_ms, key=kern_ms.get)
    # The dominant call once more, 40 launches back to back inside ONE event pair (the GPU parked while they are queued):
    # an event pair around a single launch also times the bracket itself (~3 us here), which is why the per-call averages
    # above sit that much over rocprofv3's kernel durations; this figure is the one `roofline` uses.
    dom_b2b_ms = None
    if dom in last_calls and dom != "cvf_ef_backward":      # (the backward call advances the optimiser's step counter)
        fn_, args_ = last_calls[dom]
        reps_ = 40
        torch.cuda.synchronize()
        torch.cuda._sleep(4_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps_):
            fn_(*args_)
        e1.record()
        torch.cuda.synchronize()
        dom_b2b_ms = e0.elapsed_time(e1) / reps_

    def pmc_traffic(call):
        """HBM bytes per launch of the kernel behind a C-ABI call, from the committed rocprofv3 PMC passes."""
        kernel = {"cvf_ef_backward": "ef_bwd_mfma_kernel", "cvf_ef_mlp_fwd": "ef_fwd_wg_kernel", "cvf_metric_apply": "metric_pure_kernel",
                  "cvf_ef_fwd_metric_stats": "ef_fwd_metric_kernel", "cvf_ef_align_fwd_metric_stats": "ef_fwd_metric_kernel",
                  "cvf_align_feature_fwd": "k1_align_quad_kernel", "cvf_align_feature_fwd@1M": "k1_stream_kernel"}.get(call)
        path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if kernel is None or not os.path.exists(path) or B != 20000 or args.workload != "c3":
            return None
        with open(path) as fh:
            c = json.load(fh)["kernels"].get(kernel)
        return None if c is None else (2.0 * c["FETCH_SIZE_KiB"] + c["WRITE_SIZE_KiB"]) * 1024.0

    if dom in ("cvf_ef_backward", "cvf_ef_mlp_fwd", "cvf_ef_fwd_metric_stats", "cvf_ef_align_fwd_metric_stats"):
        flop = {"cvf_ef_backward": FLOP_BWD, "cvf_ef_mlp_fwd": FLOP_FWD, "cvf_ef_fwd_metric_stats": FLOP_FWD + FLOP_METRIC,
                "cvf_ef_align_fwd_metric_stats": FLOP_FWD + FLOP_METRIC + FLOP_K1}[dom] * B
        dom_ms = dom_b2b_ms if dom_b2b_ms is not None else kern_ms[dom]
        ach = flop / (dom_ms * 1e-3) / 1e12
        roof = dict(kernel=dom, bound="mfma", achieved=ach, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP32_PEAK_TFLOPS,
                    traffic=pmc_traffic(dom), avg_launch_us=dom_ms * 1e3,
                    timing=("40 back-to-back launches in one HIP-event pair" if dom_b2b_ms is not None else
                            "HIP-event pair per launch (includes the bracket)"),
                    note="fp32 work (VALU chains + f32-input MFMA weight gradients); peak = fp32 vector = fp32 MFMA rate")
    else:
        ach = K1_BYTES * B / (kern_ms[dom] * 1e-3) / 1e9
        roof = dict(kernel=dom, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                    traffic=pmc_traffic(dom), avg_launch_us=kern_ms[dom] * 1e3)
    # The align+feature kernel alone.  Inside the step it is part of the fused launch (and at 20 000 frames any
    # stand-alone launch is latency-bound), so its HBM roofline is taken where BASELINE.json's north star puts it: one
    # launch over a 1 M-frame dipeptide trajectory resident in HBM (the shard repeated), HIP events per launch.
    from colvarsfinder import _hip
    lib, P = _hip.lib(), _hip.ptr

    def time_k1(xs, n, with_aux, reps):
        T_ = _hip.ntiles(n)
        f_tmp = torch.empty(T_ * 66 * 64, device=dev)
        a_tmp = torch.empty(T_ * 18 * 64, device=dev) if with_aux else None
        evs = []
        for _ in range(reps):
            torch.cuda._sleep(200_000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _hip.check(lib.cvf_align_feature_fwd(task._pp, P(xs), n, P(f_tmp), None, P(a_tmp), None, _hip.stream()), "k1")
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.mean([a_.elapsed_time(b_) for a_, b_ in evs[5:]]))

    k1_step = kern_ms["cvf_align_feature_fwd"] if "cvf_align_feature_fwd" in kern_ms else time_k1(X[:B].contiguous(), B, True, 30)
    N1M = 1_000_000
    x1m = X.repeat((N1M + X.shape[0] - 1) // X.shape[0], 1, 1)[:N1M].contiguous()
    k1 = time_k1(x1m, N1M, True, 25)
    k1_feat_only = time_k1(x1m, N1M, False, 25)
    del x1m
    k1_note = ("one launch over 1 000 000 frames (22 atoms) resident in HBM, features + the rotation/centroid/K^-1 rows of "
               "generator mode (72 B/frame that the 532 B/frame count leaves out); features_only = the same without those rows")
    k1_gbs = K1_BYTES * N1M / (k1 * 1e-3) / 1e9
    out = {
        "metric": "MD frames/sec through EigenFunctionTask train step",
        "value": world * B * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 3: alanine-dipeptide-shaped EigenFunctionTask, generator mode, k=3, "
                               "22 atoms, align+position features d_r=66, nets [66,20,20,20,1], diag_coeff, Adam",
                   "frames_per_gpu": args.frames, "batch_per_gpu": B, "global_batch": world * B,
                   "parallelism": f"dp{world} (frames sharded; all-reduce of batch sums + flat gradient)"},
        "roofline": roof,
        "roofline_align_feature": {"kernel": "cvf_align_feature_fwd", "bound": "hbm", "achieved": k1_gbs, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("cvf_align_feature_fwd@1M"),
                                   "avg_launch_us": k1 * 1e3,
                                   "bytes_per_frame": K1_BYTES, "frames_per_launch": N1M, "note": k1_note,
                                   "features_only": {"avg_launch_us": k1_feat_only * 1e3,
                                                     "achieved": K1_BYTES * N1M / (k1_feat_only * 1e-3) / 1e9,
                                                     "frac": K1_BYTES * N1M / (k1_feat_only * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                   "copy_ceiling_note": "a plain 16-byte copy kernel (tools/stream_probe.hip, 3 GB) moves 5.5 TB/s "
                                                        "read+write on this GPU: 1:1 read/write traffic cannot exceed ~69 % of 8 TB/s",
                                   "at_step_batch": {"frames_per_launch": B, "avg_launch_us": k1_step * 1e3,
                                                     "achieved": K1_BYTES * B / (k1_step * 1e-3) / 1e9,
                                                     "traffic": pmc_traffic("cvf_align_feature_fwd")}},
        "kernel_avg_us": {n: v * 1e3 for n, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])},
        "kernel_timing": ("HIP events around each C-ABI call over a second, eager pass of the same steps (the timed region "
                          "replays one hipGraph per step)") if graphs else "HIP events around each C-ABI call in the timed region",
        "hip_graph": bool(graphs),
        "pipelined_alignment": bool(pipeline),
        "traffic_note": ("roofline.traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B per launch from profiles/r1_pmc_traffic.json "
                         "(separate rocprofv3 --pmc passes of this command; gfx950 FETCH_SIZE halving corrected)"),
        "final_loss": final_loss,
    }
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(x, w, ref, a, sd0, B, args.cpu_seconds)
    print(json.dumps(out))


def main_c5(args):
    """Config-5 shape on the GPUs at hand (5000 atoms, k = 6, d_r = 384): exercises the streaming alignment kernel and the
    large-molecule derivative kernel inside the full train step.  Extra measurement, not the benchmark line."""
    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj, diag_coeff_for
    _dist.init_from_env("nccl")
    world, rank = _dist.world(), _dist.rank()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    n_atoms, k = C5["n_atoms"], C5["k"]
    frames = C5["frames"] if args.frames == 100000 else args.frames
    B = C5["batch"] if args.batch == 20000 else args.batch
    x, w, ref = make_shard(frames, rank, n_atoms, scale=2.0, sigma=0.05)   # nm-like units: 20 A / 0.5 A of SURVEY 8d (in A the tanh nets saturate at init)
    a = torch.tensor(diag_coeff_for(n_atoms, SEED), dtype=torch.float32)
    torch.manual_seed(SEED)
    model = nn.EigenFunctions(C5["layers"], k)
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, c5_features(n_atoms))
    task = core.EigenFunctionTask(Traj(x, w, 1.0), layer, model, "/tmp/cvf_bench", ALPHA, C5["eig_w"], diag_coeff=a, beta=BETA,
                                  lag_tau=0, learning_rate=LR, k=k, batch_size=B, device=dev, verbose=False, save_model_every_step=0)
    n_batches = frames // B
    X, Wt = task._traj, task._weights
    log = torch.zeros(n_batches, 3 + 2 * k, device=dev, dtype=torch.float64)

    def step(i):
        b = i % n_batches
        task._graph_step(("bench", b), lambda: task.train_step(X[b * B:(b + 1) * B], Wt[b * B:(b + 1) * B]), log[b])
        return log[b]

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        lv = step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    task._use_graphs, task._events = False, {}
    for i in range(min(args.steps, 20)):
        step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    kern = {n: float(np.mean([s_.elapsed_time(e_) for s_, e_ in ev])) * 1e3 for n, ev in task._events.items()}
    if rank == 0:
        k1 = kern["cvf_align_feature_fwd"]
        bpf = 12 * n_atoms + 4 + 4 * layer.d_r
        print(json.dumps({"workload": "config-5 shape: 5000 atoms, d_r=384, k=6, EigenFunctionTask generator step", "n_gpus": world,
                          "batch_per_gpu": B, "frames_per_gpu": frames, "value": world * B * args.steps / elapsed, "unit": "frames/s",
                          "ms_per_step": elapsed / args.steps * 1e3, "final_loss": float(lv[0]),
                          "kernel_avg_us": dict(sorted(kern.items(), key=lambda kv: -kv[1])),
                          "align_feature_GBps": bpf * B / (k1 * 1e-6) / 1e9, "align_feature_frac_of_8TBps": bpf * B / (k1 * 1e-6) / 8e12}))


def main_c2(args):
    """Config 2: alanine-dipeptide-shaped AutoEncoderTask, 22 atoms x 100k frames, bottleneck 2 (encoder [66,20,20,20,2],
    decoder [2,10,10,66]), B = 20 000.  Extra measurement, not the benchmark line: the step is one fused kernel pair
    (forward + weighted MSE + gradient, then slab sum + Adam) on the feature trajectory K1 produced once up front."""
    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj
    _dist.init_from_env("nccl")
    world, rank = _dist.world(), _dist.rank()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    x, w, ref = make_shard(args.frames, rank)
    torch.manual_seed(SEED)
    model = nn.AutoEncoder([66, 20, 20, 20, 2], [2, 10, 10, 66])
    layer = pp.AlignFeatureLayer(N_ATOMS, list(range(N_ATOMS)), ref, [("position", tuple(range(N_ATOMS)))])
    t0 = time.perf_counter()
    task = core.AutoEncoderTask(Traj(x, w, 1.0), layer, model, "/tmp/cvf_bench", learning_rate=LR, batch_size=args.batch,
                                device=dev, verbose=False, save_model_every_step=0)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    B = min(args.batch, args.frames)
    n_batches = args.frames // B
    idx = torch.arange(args.frames, device=dev, dtype=torch.long)
    Wt = task._weights
    inv = [1.0 / float(Wt[b * B:(b + 1) * B].sum(dtype=torch.float64)) for b in range(n_batches)]

    def step(i):
        b = i % n_batches
        return task._step(task._feature_traj, idx[b * B:(b + 1) * B], Wt[b * B:(b + 1) * B], True, inv[b], advance=True,
                          fuse_adam=(world == 1))

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"workload": "config 2: AutoEncoderTask [66,20,20,20,2]/[2,10,10,66], 22 atoms, B=20000", "n_gpus": world,
                          "value": world * B * args.steps / elapsed, "unit": "frames/s", "ms_per_step": elapsed / args.steps * 1e3,
                          "final_loss": float(loss), "init_seconds_incl_feature_trajectory": t_init}))


def main_regae(args):
    """RegAutoEncoderTask at the dipeptide shape of main.ipynb:452-458 scaled to 22 atoms: encoder
    [66,20,20,20,2], decoder [2,10,10,66], two regularisers [2,10,10,1], time-lagged reconstruction (lag 1 frame) +
    transfer-operator regulariser (lag 1 frame), B = 20 000, one GPU.  Extra measurement, not the benchmark line."""
    from colvarsfinder import core, nn, pp
    from tests.synth import Traj
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    x, w, ref = make_shard(args.frames, 0)
    torch.manual_seed(SEED)
    model = nn.RegAutoEncoder([66, 20, 20, 20, 2], [2, 10, 10, 66], [2, 10, 10, 1], 2)
    layer = pp.AlignFeatureLayer(N_ATOMS, list(range(N_ATOMS)), ref, [("position", tuple(range(N_ATOMS)))])
    task = core.RegAutoEncoderTask(Traj(x, w, 1.0), layer, model, "/tmp/cvf_bench", eig_weights=[1.0, 0.5], learning_rate=LR,
                                   batch_size=args.batch, alpha=1.0, gamma=[1.0, 10.0], lag_tau_ae=1.0, lag_tau_reg=1.0, device=dev,
                                   verbose=False, save_model_every_step=0)
    B = min(args.batch, args.frames - 1)
    n_batches = (args.frames - 1) // B
    idx = torch.arange(args.frames - 1, device=dev, dtype=torch.long)
    Wt = task._weights
    wl = Wt[1:].contiguous()
    wsum = [float(Wt[b * B:(b + 1) * B].sum(dtype=torch.float64)) for b in range(n_batches)]
    log = torch.zeros(n_batches, 4 + 2 + 3, device=dev, dtype=torch.float64)

    def step(i):
        b = i % n_batches
        sl = slice(b * B, (b + 1) * B)
        return task._step(task._feature_traj, idx[sl], Wt[sl], wl[sl], 1, 1, with_grad=True, advance=True, wsum=wsum[b], out=log[b])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        row = step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    task._events = {}
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    kern = {n: float(np.mean([a.elapsed_time(b) for a, b in ev[3:]])) * 1e3 for n, ev in task._events.items()}
    print(json.dumps({"workload": "RegAutoEncoderTask [66,20,20,20,2]/[2,10,10,66]/2x[2,10,10,1], 22 atoms, B=20000, lag 1/1", "n_gpus": 1,
                      "value": B * args.steps / elapsed, "unit": "frames/s", "ms_per_step": elapsed / args.steps * 1e3,
                      "call_avg_us": kern, "final_row": [float(v) for v in row.cpu()]}))


if __name__ == "__main__":
    main()
