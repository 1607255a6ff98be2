#!/usr/bin/env python3
"""bench.py - MD frames/sec through the EigenFunctionTask train step on MI355X.

Workload (BASELINE.json config 3, SURVEY.md section 8d "C3"): alanine-dipeptide-shaped synthetic
trajectory, 22 atoms, alignment on all atoms + position features (d_r = 66), EigenFunctionTask in
generator mode with k = 3 nets [66,20,20,20,1], diffusion-weighted diag_coeff, alpha = 20, Adam.
One "step" = one full train step on one batch resident in HBM: align+features (K1), nets forward and
input gradients (K4a), q = J A J^T g (K2/K3), batch sums (K5), loss tail, parameter gradient (K4b),
Adam (K6).  With N > 1 (one process per GPU, launched by torch.distributed.run) every rank owns its own
block of the frames and the two all-reduces of SURVEY.md section 8e (batch sums before the backward pass, flat gradient
after it) run over RCCL.  N > 1 defaults to STRONG scaling on BASELINE config 4: 1 M frames in all (1/N resident per GPU), a
fixed global batch (--global-batch, default 800 000 frames per step of the whole job = the whole training split of the
1 M-frame set at the reference's test_ratio 0.2, one step per epoch); --scaling weak keeps --batch frames per GPU.  Every line also carries `scaling_table`: the same job at global batches 20 000 / 160 000 / 800 000 and the weak line.

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
FLOP_OF_CALL = {"cvf_ef16_front": FLOP_FWD + FLOP_METRIC + FLOP_K1, "cvf_ef16_backward": FLOP_BWD, "cvf_ef_backward": FLOP_BWD, "cvf_ef_mlp_fwd": FLOP_FWD, "cvf_ef_fwd_metric_stats": FLOP_FWD + FLOP_METRIC,
                "cvf_ef_align_fwd_metric_stats": FLOP_FWD + FLOP_METRIC + FLOP_K1}
KERNEL_OF_CALL = {"cvf_ef16_front": "ef16_front_kernel", "cvf_ef16_backward": "ef16_back_kernel", "cvf_ef_backward": "ef_bwd_mfma_kernel", "cvf_ef_mlp_fwd": "ef_fwd_wide_kernel", "cvf_metric_apply": "metric_rows_kernel",
                  "cvf_ef_fwd_metric_stats": "ef_fwd_metric_kernel", "cvf_ef_align_fwd_metric_stats": "ef_fwd_metric_kernel",
                  "cvf_align_feature_fwd": "k1_align_quad_kernel", "cvf_align_feature_fwd@1M": "k1_stream_kernel",
                  "cvf_align_feature_fwd@c5": "k1_large_kernel"}   # (slice or pipelined kernel: the same traffic per frame; the PMC pass runs 20 000 frames)
PROFILE_TAG = "r4"   # profiles/<tag>_pmc_traffic.json is the committed PMC summary `roofline.traffic` is read from


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


def device_frames(n, ref, sigma, seed, dev, chunk=200_000):
    """Synthetic frames x_b = Q_b (ref + sigma xi_b) + t_b generated ON the device (a random rotation and
    translation of a noisy copy of the reference) + mean-normalised weights U(0.2, 2).  torch is plumbing here: it fills HBM."""
    g = torch.Generator(device=dev).manual_seed(int(seed))
    refd = torch.tensor(ref, device=dev, dtype=torch.float32)
    x = torch.empty(n, refd.shape[0], 3, device=dev, dtype=torch.float32)
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        q = torch.randn(m, 4, device=dev, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        w_, x_, y_, z_ = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_),
                         2 * (x_ * y_ + z_ * w_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_),
                         2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_), 1 - 2 * (x_ * x_ + y_ * y_)], dim=1).view(m, 3, 3)
        noisy = refd[None] + sigma * torch.randn(m, refd.shape[0], 3, device=dev, generator=g)
        x[s0:s0 + m] = torch.einsum("bij,baj->bai", R, noisy) + torch.randn(m, 1, 3, device=dev, generator=g)
    w = 0.2 + 1.8 * torch.rand(n, device=dev, generator=g)
    return x, (w / w.mean()).contiguous()


def note(msg):
    """Progress line on stderr (rank 0): a long run keeps talking, and a crash is located by the last line."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def run_steps(task, X, Wt, B, steps, warmup, tag, world, dev, max_batches=8, preheat_ms=0.0):
    """W untimed + EXACTLY K timed train steps of the product's own step path on this rank's resident rows, B frames per rank
    per step.  As EigenFunctionTask.train() replays one hipGraph per EPOCH (all its static batches), the steps run in chunks of
    C = the number of distinct resident batches (<= max_batches) through task._graph_call - one graph replay per chunk - and the
    remainder K mod C as single-step replays.  The timed region is bracketed by a barrier + device synchronisation on both
    sides.  `preheat_ms` > 0: after the W warm-up steps, further UNTIMED chunks of the same steps run for about that many
    milliseconds, so that the clock starts with the GPU at the frequency a training run of thousands of steps sees (the
    driver's default K = 20 steps last 1.5 ms in all - shorter than the power manager's ramp; the count is reported as
    `preheat_steps`).  Returns (max-over-ranks seconds, last loss vector, single-step function)."""
    n_batches = max(1, min(X.shape[0] // B, max_batches))
    log = torch.zeros(n_batches, 3 + 2 * task.k, device=dev, dtype=torch.float64)

    def one(b):
        s = b * B
        return task.train_step(X[s:s + B], Wt[s:s + B], out=log[b])

    def step(i):
        b = i % n_batches
        task._graph_step((tag, b), lambda o: one(b), log[b], takes_out=True)
        return log[b]

    def chunk():
        task._graph_call((tag, "chunk"), lambda: [one(b) for b in range(n_batches)])

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    n_chunks, rest = divmod(steps, n_batches)
    for i in range(max(1, (warmup + n_batches - 1) // n_batches) + 1):   # (captures the chunk graph before the clock starts)
        chunk()
    for i in range(rest if task._use_graphs else 0):                       # (and the single-step graphs the remainder uses)
        step(i)
        step(i)
    run_steps.preheat_steps = 0
    if preheat_ms > 0:
        barrier()
        t0 = time.perf_counter()
        chunk()
        barrier()
        per_chunk = max(time.perf_counter() - t0, 1e-5)
        n_pre = int(min(2000, max(0.0, preheat_ms * 1e-3 / per_chunk)))
        if world > 1:   # the same count on every rank (the chunks hold collectives)
            t = torch.tensor([n_pre], device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            n_pre = int(t)
        for _ in range(n_pre):
            chunk()
        # (the first graph launch after a long queue has drained costs the host 0.1-0.4 ms of clean-up - tools/timing_probe.py -
        # which must not land inside a 1.5 ms timed region: drain, replay once more, drain)
        barrier()
        chunk()
        run_steps.preheat_steps = (n_pre + 2) * n_batches
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_chunks):
        chunk()
    for i in range(rest):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    lv = log[(rest - 1) % n_batches if rest else n_batches - 1]
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax)
    assert np.isfinite(float(lv[0])), "training diverged"
    return elapsed, lv, step


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves as a CHILD process (never an exec: nothing in
    this process has touched the GPU yet, and it must stay that way) - `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`, the command the task statement
    names.  Rank 0's JSON line is the child's stdout and passes through; this process exits with the child's return code.
    CVF_BENCH_BACKEND=gloo + CVF_BENCH_ONE_GPU=1 put every rank on cuda:0 over gloo (the rehearsal a one-GPU box allows)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n)))
    print(f"[bench] --gpus {n} without RANK in the environment: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    rc = subprocess.run(cmd, env=env).returncode
    sys.exit(rc)


def bench_device():
    """(backend, device) of this rank: RCCL on its own GPU; CVF_BENCH_ONE_GPU=1 maps every rank to cuda:0 (with
    CVF_BENCH_BACKEND=gloo: the two-ranks-on-one-GPU rehearsal of tests/test_gpu_api_contract.py)."""
    backend = os.environ.get("CVF_BENCH_BACKEND", "nccl")
    local = 0 if os.environ.get("CVF_BENCH_ONE_GPU", "0") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    return backend, torch.device("cuda", local)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=20000,
                    help="N=1 line: frames per step (reference notebook: 20000); --scaling weak: frames per GPU per step")
    ap.add_argument("--frames", type=int, default=100000, help="N=1 line: frames resident (config 3: 100k)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="N>1: strong (default) = BASELINE config 4, --frames-total frames and --global-batch frames per step split "
                         "over the ranks; weak = --batch frames per GPU per step")
    ap.add_argument("--frames-total", type=int, default=1_000_000, help="config 4: frames of the whole job")
    ap.add_argument("--global-batch", type=int, default=800_000,
                    help="strong scaling: frames per step of the whole job (default: the training split of the 1 M-frame set)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 disables it)")
    ap.add_argument("--preheat-ms", type=float, default=75.0,
                    help="untimed steps run for about this long after the W warm-up steps, before the clock starts (GPU frequency ramp; 0 = off)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (other global batches, K1 rooflines)")
    ap.add_argument("--workload", choices=["c3", "c5", "c2", "regae", "transfer"], default="c3",
                    help="c3 = the benchmark line; c5 = config-5 shape, c2 = config-2 AutoEncoderTask (extra measurements)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)
    if args.workload == "transfer":
        return main_transfer(args)
    if args.workload == "c5":
        return main_c5(args)
    if args.workload == "regae":
        return main_regae(args)
    if args.workload == "c2":
        return main_c2(args)

    from colvarsfinder import _dist, _hip, core, nn, pp
    from tests.synth import Traj, diag_coeff_for

    backend, dev = bench_device()
    torch.cuda.set_device(dev)
    if backend == "nccl" and dev.index != int(os.environ.get("LOCAL_RANK", "0")):
        raise SystemExit("CVF_BENCH_ONE_GPU=1 needs CVF_BENCH_BACKEND=gloo (RCCL wants one GPU per rank)")
    _dist.init_from_env(backend)
    world, rank = _dist.world(), _dist.rank()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    scaling = args.scaling or ("strong" if world > 1 else "weak")

    ref = np.random.RandomState(SEED).normal(scale=2.0, size=(N_ATOMS, 3))
    a = torch.tensor(diag_coeff_for(N_ATOMS, SEED), dtype=torch.float32)
    torch.manual_seed(SEED)
    model = nn.EigenFunctions(LAYERS, K_NETS)
    sd0 = {n: p.detach().clone() for n, p in model.state_dict().items()}
    layer = pp.AlignFeatureLayer(N_ATOMS, list(range(N_ATOMS)), ref, [("position", tuple(range(N_ATOMS)))])
    # The task is built on a token trajectory: bench.py owns the resident frames (generated on the device) and drives the
    # task's own step path on them, exactly as EigenFunctionTask.train() does on its resident batches.
    tok = np.zeros((64, N_ATOMS, 3), dtype=np.float32) + ref[None].astype(np.float32)
    task = core.EigenFunctionTask(Traj(tok, np.ones(64), 1.0), layer, model, "/tmp/cvf_bench", ALPHA, EIG_W, diag_coeff=a, beta=BETA,
                                  lag_tau=0, learning_rate=LR, k=K_NETS, batch_size=args.batch, device=dev, verbose=False,
                                  save_model_every_step=0)

    def restart():
        """Same initial parameters and optimizer state for every measured configuration."""
        model.load_state_dict(sd0)
        task._flat.repack()
        task.optimizer.exp_avg.zero_(); task.optimizer.exp_avg_sq.zero_(); task.optimizer.step_count.zero_()

    # ---- the headline workload of this N
    if scaling == "weak":   # (N = 1 without --scaling: config 3; `--gpus 1 --scaling strong` runs the N = 1 point of the strong-scaling job)
        frames_rank, B = args.frames, min(args.batch, args.frames)
        workload = ("BASELINE config 3: alanine-dipeptide-shaped EigenFunctionTask, generator mode, k=3, 22 atoms, align+position "
                    "features d_r=66, nets [66,20,20,20,1], diag_coeff, Adam")
    else:
        assert args.global_batch % world == 0 and args.frames_total % world == 0
        frames_rank, B = args.frames_total // world, args.global_batch // world
        workload = (f"BASELINE config 4: config-3 model on {args.frames_total} frames sharded over {world} GPUs "
                    f"({frames_rank} resident per GPU), global batch {args.global_batch} fixed (strong scaling), all-reduce of the "
                    "batch sums + flat gradient per step")
    X, Wt = device_frames(frames_rank, ref, 0.3, SEED + 1 + rank, dev)
    note(f"headline: {frames_rank} frames resident per GPU, {B} per GPU per step, {world} GPU(s), {scaling} scaling")
    elapsed, loss_vec, step = run_steps(task, X, Wt, B, args.steps, args.warmup, "bench", world, dev, preheat_ms=args.preheat_ms)
    preheat_steps = run_steps.preheat_steps
    graphs = task._use_graphs
    final_loss = float(loss_vec[0])

    # ---- per-kernel durations: HIP events around every C-ABI call over a second, eager pass of the same steps (not part
    # of `value`); the GPU is parked while the host queues each step, so the events bracket back-to-back executions
    note(f"timed region done: {elapsed / args.steps * 1e3:.4f} ms/step; per-call HIP-event pass")
    task._use_graphs, task._events, task._last_call = False, {}, {}
    for i in range(min(args.steps, 50)):
        torch.cuda._sleep(2_000_000)
        step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    task._use_graphs = graphs
    events, task._events = task._events, None
    last_calls, task._last_call = (task._last_call or {}), None
    kern_ms = {name: float(np.mean([s_.elapsed_time(e_) for s_, e_ in ev])) for name, ev in events.items()}
    launches_per_step = {name: len(ev) / max(1, min(args.steps, 50)) for name, ev in events.items()}   # C-ABI calls + all-reduce launches of one step

    coll_ms = {n: v for n, v in kern_ms.items() if n.startswith("allreduce_")}    # the step's two cross-rank sums (N > 1)
    dom = max((n for n in kern_ms if n not in coll_ms), key=kern_ms.get)
    # The dominant call once more, 40 launches back to back inside ONE event pair (the GPU parked while they are queued):
    # an event pair around a single launch also times the bracket itself (~3 us here), which is why the per-call averages
    # above sit that much over rocprofv3's kernel durations; this figure is the one `roofline` uses.
    dom_b2b_ms = None
    if dom in last_calls and dom not in ("cvf_ef_backward", "cvf_ef16_backward") and world == 1:   # (before anything frees the buffers these launches point into)      # (the backward call advances the optimiser's step counter)
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


    extras = {}
    if not args.no_extras:
        # ---- secondary: the config-4 job (1 M frames in all) at other global batches, and the weak-scaling line
        combos = [("strong", g) for g in (20_000, 160_000, 800_000)] + [("weak", 20_000 * world)]
        rows = {}
        for kind, G in combos:
            if kind == "strong":
                fr = args.frames_total // world
            else:
                fr = 100_000
            b = G // world
            if b < 64 or b > fr or G % world:
                continue
            if kind == scaling and b == B and fr == frames_rank:
                rows[f"{kind}:{G}"] = dict(global_batch=G, batch_per_gpu=b, frames_per_gpu=fr, ms_per_step=elapsed / args.steps * 1e3,
                                           value=world * B * args.steps / elapsed)
                continue
            note(f"scaling table: {kind} global batch {G} ({b} per GPU, {fr} resident per GPU)")
            restart()
            if fr > X.shape[0]:
                X, Wt = device_frames(fr, ref, 0.3, SEED + 1 + rank, dev)
            k_steps = max(10, min(args.steps, int(4e6 // G)))
            el, _, _ = run_steps(task, X[:fr], Wt[:fr], b, k_steps, 5, f"{kind}{G}", world, dev, max_batches=4)
            rows[f"{kind}:{G}"] = dict(global_batch=G, batch_per_gpu=b, frames_per_gpu=fr, steps=k_steps, ms_per_step=el / k_steps * 1e3,
                                       value=G * k_steps / el)
            task._graphs.clear()
            task._ws.clear()
        extras["scaling_table"] = dict(
            note=("same model and step; 'strong:G' = BASELINE config 4 (frames_total frames split over the ranks, G frames per step "
                  "of the whole job, fixed as N grows); 'weak:G' = 20 000 frames per GPU per step.  value = frames/s of the whole job."),
            frames_total=args.frames_total, rows=rows)

    if rank != 0:
        return
    prof = PROFILE_TAG

    def pmc_traffic(call, scale=1):
        """HBM bytes per launch of the kernel behind a C-ABI call, from the committed rocprofv3 PMC passes (entries that name
        their frames_per_launch are scaled to `scale` frames)."""
        kernel = KERNEL_OF_CALL.get(call)
        path = os.path.join(ROOT, "profiles", f"{prof}_pmc_traffic.json")
        if not os.path.exists(path):      # (until this round's counter passes are committed: the previous round's)
            path = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")
        if kernel is None or not os.path.exists(path) or B != 20000 or world != 1:
            return None
        with open(path) as fh:
            c = json.load(fh)["kernels"].get(kernel)
        return None if c is None else (2.0 * c["FETCH_SIZE_KiB"] + c["WRITE_SIZE_KiB"]) * 1024.0 * scale / c.get("frames_per_launch", scale)

    if dom in FLOP_OF_CALL:
        flop = FLOP_OF_CALL[dom] * B
        dom_ms = dom_b2b_ms if dom_b2b_ms is not None else kern_ms[dom]
        ach = flop / (dom_ms * 1e-3) / 1e12
        roof = dict(kernel=dom, bound="mfma", achieved=ach, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP32_PEAK_TFLOPS,
                    traffic=pmc_traffic(dom), avg_launch_us=dom_ms * 1e3, flop_per_frame=FLOP_OF_CALL[dom], frames_per_launch=B,
                    timing=("40 back-to-back launches in one HIP-event pair" if dom_b2b_ms is not None else
                            "HIP-event pair per launch (includes the bracket)"),
                    note="fp32 work (VALU chains + f32-input MFMA weight gradients); peak = fp32 vector = fp32 MFMA rate")
    else:
        ach = K1_BYTES * B / (kern_ms[dom] * 1e-3) / 1e9
        roof = dict(kernel=dom, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                    traffic=pmc_traffic(dom), avg_launch_us=kern_ms[dom] * 1e3)
    out = {
        "metric": "MD frames/sec through EigenFunctionTask train step",
        "value": world * B * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preheat_steps": preheat_steps,
        "preheat_note": "untimed steps of the same kind run after the W warm-up steps (--preheat-ms) so that the timed K steps see the "
                        "GPU clock of a long training run; --preheat-ms 0 turns it off",
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "frames_per_gpu": frames_rank, "batch_per_gpu": B, "global_batch": world * B,
                   "parallelism": f"dp{world} (frames sharded; all-reduce of batch sums + flat gradient)"},
        "roofline": roof,
        "kernel_avg_us": {n: v * 1e3 for n, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])},
        "kernel_timing": ("HIP events around each C-ABI call over a second, eager pass of the same steps (the timed region "
                          "replays one hipGraph per step)") if graphs else "HIP events around each C-ABI call (eager pass)",
        "collective_avg_us": ({n: v * 1e3 for n, v in coll_ms.items()} if coll_ms else None),
        "collective_note": ("HIP events around each all-reduce in the eager per-call pass, rank 0 (includes the wait for the slowest rank: "
                            "arrival skew shows here); None on one GPU - the single-process step has no collective"),
        "launches_per_step": {"total": float(sum(launches_per_step.values())), "calls": launches_per_step,
                              "cross_rank_sums": (None if world == 1 and not _dist.collectives() else
                                                  "inside cvf_ef16_finish_dp / cvf_slab_reduce_dp (peer-to-peer windows, csrc/cvf_p2p.hpp)"
                                                  if _dist.fused_comm() is not None else "separate all-reduce launches"),
                              "comm_mode": _dist.comm_mode()},
        "hip_graph": bool(graphs),
        "graph_granularity": "one hipGraph replay per chunk of the resident static batches (as train() replays one per epoch)",
        "traffic_note": (f"roofline.traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B per launch from profiles/{prof}_pmc_traffic.json "
                         "(separate rocprofv3 --pmc passes of this command; gfx950 FETCH_SIZE halving corrected)"),
        "final_loss": final_loss,
    }
    if scaling == "strong":
        out["strong_scaling_note"] = (f"this line is the {args.frames_total}-frame job at a global batch of {world * B} frames per step on {world} GPU(s); "
                                      f"its one-GPU reference is scaling_table.rows['strong:{world * B}'] of the N = 1 line (the same job), not the "
                                      "N = 1 headline, which is BASELINE config 3 (100 000 frames, 20 000 per step)")
    out.update(extras)
    if world == 1 and not args.no_extras:
        del X, Wt
        task._graphs.clear(); task._ws.clear()
        torch.cuda.empty_cache()
        note("align+feature kernel alone, out of cache")
        out["roofline_align_feature"] = align_feature_roofline(task, ref, dev, pmc_traffic)
    if world == 1 and not args.no_extras:
        out["other_configs"] = other_configs()
    if world == 1 and args.cpu_seconds > 0:
        note("CPU baseline")
        xb, wb = device_frames(B, ref, 0.3, SEED + 1, dev)
        out["cpu_baseline"] = cpu_baseline(xb.cpu().numpy(), wb.cpu().numpy().astype(np.float64), ref, a, sd0, B, args.cpu_seconds)
    print(json.dumps(out))


def other_configs():
    """The other BASELINE configurations' steps on this GPU, each as a child `python bench.py --workload X` (its own process:
    fresh allocator, its own one JSON line): config 2 (AutoEncoderTask), the transfer-operator mode the shipped notebook runs,
    and the config-5 shape (5000 atoms, k = 6) with its position against the HBM roofline - SURVEY 8d: a generator step reads
    every frame twice, 2 (12 N + 4) + O(k) = 120 008 B per frame-step at N = 5000.  Extra measurements, not `value`."""
    import subprocess
    res = {}
    for wl, extra in (("c2", []), ("transfer", []), ("regae", []), ("c5", ["--batch", "2000"]), ("c5", ["--batch", "16000"])):
        key = wl if not extra else f"{wl}_batch{extra[1]}"
        note(f"other configurations: {key}")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "60", "--warmup", "10"] + extra,
                               capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
        except Exception as exc:   # (an extra measurement must not take the benchmark line down with it)
            res[key] = {"error": f"{type(exc).__name__}: {exc}"}
            continue
        row = {"workload": d.get("workload"), "us_per_step": d["ms_per_step"] * 1e3, "frames_per_s": d["value"],
               "call_avg_us": d.get("kernel_avg_us") or d.get("call_avg_us")}
        if "roofline" in d:
            row["roofline"] = d["roofline"]
        if wl == "c5":
            bpf = 2 * (12 * C5["n_atoms"] + 4) + 8 * C5["k"]
            gbs = bpf * d["batch_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9
            row["roofline_c5_step"] = {"bound": "hbm", "bytes_per_frame_step": bpf, "frames_per_step": d["batch_per_gpu"], "achieved": gbs,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                       "align_feature_frac_inside_the_step": d.get("align_feature_frac_of_8TBps")}
        res[key] = row
    return res


def gpu_clock_state():
    """What the power manager reports when the K1 timing loop starts (sysfs text, read in this process): the sclk / mclk tables with
    the active level starred, and the power cap.  Diagnostic only - lets a slow box be told from a slow kernel."""
    import glob
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        row = {}
        for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
            try:
                row[name] = [ln.strip() for ln in open(os.path.join(card, name)).read().splitlines() if "*" in ln]
            except OSError:
                pass
        for cap in glob.glob(os.path.join(card, "hwmon/hwmon*/power1_cap")):
            try:
                row["power1_cap_uW"] = int(open(cap).read().strip())
            except (OSError, ValueError):
                pass
        if row:
            out[card.split("/")[-2]] = row
    return out or None


def align_feature_roofline(task, ref, dev, pmc_traffic):
    """The align+feature kernel K1 alone, where BASELINE.json's north star puts its HBM roofline.  Inside the 20 000-frame step
    it is part of the fused launch (and any stand-alone launch of that size is latency-bound), so it is timed at shard size,
    OUT OF CACHE: (i) the dipeptide shape over 4 M frames (1.06 GB in, 1.06 GB out - four times the 256 MB Infinity Cache; the
    1 M-frame launch of round 1 is kept beside it); (ii) BASELINE config 5's shape, 5000 atoms x 100 k frames (6 GB in), which is
    the configuration BASELINE.json names 'HBM-bound align+feature path'.  HIP events per launch, on the launch stream.
    VERDICT r3 item 4: every case runs >= 50 ms of untimed launches first (the power manager's ramp), then >= 30 timed launches,
    and IN THE SAME LOOP a plain stream over the same footprint (cvf_probe_stream: a float4 copy of half the bytes = the same bytes
    moved 1:1, and a read-only sweep of the input) so that the line carries what the box delivers at that moment
    (`copy_GBps`, `read_GBps`) and the kernel's position against it (`k1_over_copy`, `k1_over_read`)."""
    from colvarsfinder import _hip, pp
    lib, P = _hip.lib(), _hip.ptr

    def timed(fn, reps, preheat_ms):
        t0 = time.perf_counter()
        while True:                               # untimed: at least preheat_ms of back-to-back launches
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if (time.perf_counter() - t0) * 1e3 >= preheat_ms:
                break
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = np.asarray([a_.elapsed_time(b_) for a_, b_ in evs])
        return float(np.mean(ms)), float(np.min(ms)), float(np.max(ms))

    def case(desc, xs, n, d_r, bpf, with_aux, scratch=None, reps=30, rows=False):
        """K1 on n frames; then, same loop shape, the two plain streams over the same bytes.  rows: the row-major output
        [frame][d_r] (what AutoEncoderTask's one-off feature trajectory is, core.py:635) instead of the tiled one."""
        T_ = _hip.ntiles(n)
        f_tmp = torch.empty(n * d_r if rows else T_ * d_r * 64, device=dev)
        a_tmp = torch.empty(T_ * 18 * 64, device=dev) if with_aux else None
        s = _hip.stream()
        k1 = lambda: _hip.check(lib.cvf_align_feature_fwd(desc, P(xs), n, None if rows else P(f_tmp), P(f_tmp) if rows else None,   # noqa: E731
                                                          P(a_tmp), P(scratch), s), "k1")
        mean, lo, hi = timed(k1, reps, 50.0)
        total = float(bpf) * n
        row = dict(avg_launch_us=mean * 1e3, min_launch_us=lo * 1e3, max_launch_us=hi * 1e3, launches_timed=reps, preheat_ms=50,
                   achieved=total / (mean * 1e-3) / 1e9, frac=total / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS)
        if not with_aux:
            flat = xs.reshape(-1)
            n4 = int(min(flat.numel() // 4, total // 32))          # copy: n4 pieces read + n4 written = `total` bytes moved
            dst = torch.empty(int(lib.cvf_probe_stream_out_floats(0, n4)), device=dev)
            cp = lambda: _hip.check(lib.cvf_probe_stream(0, P(dst), P(flat), n4, s), "probe copy")   # noqa: E731
            cmean = timed(cp, reps, 20.0)[0]
            nr = flat.numel() // 4
            dst_r = torch.empty(int(lib.cvf_probe_stream_out_floats(1, nr)), device=dev)
            rd = lambda: _hip.check(lib.cvf_probe_stream(1, P(dst_r), P(flat), nr, s), "probe read")   # noqa: E731
            rmean = timed(rd, reps, 20.0)[0]
            copy_gbs, read_gbs = 32.0 * n4 / (cmean * 1e-3) / 1e9, 16.0 * nr / (rmean * 1e-3) / 1e9
            row.update(copy_GBps=copy_gbs, read_GBps=read_gbs, k1_over_copy=row["achieved"] / copy_gbs, k1_over_read=row["achieved"] / read_gbs)
            del dst, dst_r
        return row

    clocks = gpu_clock_state()
    res = {}
    for label, n in (("dipeptide_4M", 4_000_000), ("dipeptide_1M", 1_000_000)):
        xs, _ = device_frames(n, ref, 0.3, SEED + 77, dev)
        feat = case(task._pp, xs, n, 66, K1_BYTES, False)
        feat_rows = case(task._pp, xs, n, 66, K1_BYTES, False, rows=True)
        feat["row_major_output"] = {k_: feat_rows[k_] for k_ in ("avg_launch_us", "frac", "k1_over_copy")}
        gen = case(task._pp, xs, n, 66, K1_BYTES, True)
        gen["note"] = "+ rotation/centroid/K^-1 rows (72 B/frame the 532 B/frame count leaves out)"
        del xs
        res[label] = dict(frames_per_launch=n, bytes_per_frame=K1_BYTES, footprint_MB=(264 + 264) * n / 1e6, features_only=feat,
                          generator_outputs=gen)
    # config-5 shape
    n5, na5 = 100_000, C5["n_atoms"]
    ref5 = np.random.RandomState(SEED).normal(scale=2.0, size=(na5, 3))
    layer5 = pp.AlignFeatureLayer(na5, list(range(na5)), ref5, c5_features(na5)).to(dev)
    d5 = layer5.pp_desc()
    x5, _ = device_frames(n5, ref5, 0.05, SEED + 78, dev, chunk=5000)
    bpf5 = 12 * na5 + 4 + 4 * layer5.d_r
    feat5 = case(d5, x5, n5, layer5.d_r, bpf5, False)
    # large batches take the resident role-split kernel (k1_large_pipe_kernel, round 4); the one-group-per-workgroup kernel it replaced
    # (k1_large_slice_kernel, still what small batches run) in the same lease
    def with_env(name, value, fn):
        os.environ[name] = value
        try:
            return fn()
        finally:
            del os.environ[name]
    feat5["kernel_ab_us"] = {"pipelined": feat5["avg_launch_us"],
                             "one_group_per_workgroup": with_env("CVF_K1_NOPIPE", "1", lambda: case(d5, x5, n5, layer5.d_r, bpf5, False, reps=15)["avg_launch_us"])}
    rows5 = case(d5, x5, n5, layer5.d_r, bpf5, False, rows=True)
    feat5["row_major_output"] = {k_: rows5[k_] for k_ in ("avg_launch_us", "frac", "k1_over_copy")}
    # where the launch's time goes (developer probes of csrc/k1_large.hip, results discarded): the streaming waves alone, and the launch
    # without its feature stores
    feat5["launch_decomposition"] = {
        "streaming_waves_only_us": with_env("CVF_K1_PIPE_PROBE", "1", lambda: case(d5, x5, n5, layer5.d_r, bpf5, False, reps=15)["avg_launch_us"]),
        "without_feature_stores_us": with_env("CVF_K1_PIPE_PROBE", "8", lambda: case(d5, x5, n5, layer5.d_r, bpf5, False, reps=15)["avg_launch_us"])}
    sc5 = _hip.align_scratch(d5, n5, dev)
    gen5 = case(d5, x5, n5, layer5.d_r, bpf5, True, scratch=sc5)
    # (with the generator-mode outputs the one-group-per-workgroup kernel is the default: the pipelined kernel, forced here, is between 1 % faster and 7 % slower by box)
    gen5["kernel_ab_us"] = {"one_group_per_workgroup": gen5["avg_launch_us"],
                            "pipelined": with_env("CVF_K1_PIPE_MIN_GROUPS", "1024", lambda: case(d5, x5, n5, layer5.d_r, bpf5, True, scratch=sc5, reps=15)["avg_launch_us"])}
    gen5["note"] = "+ rotation/centroid rows and the slot copy the derivative kernel reads"
    del x5, sc5
    res["config5_100k"] = dict(frames_per_launch=n5, n_atoms=na5, d_r=layer5.d_r, bytes_per_frame=bpf5, footprint_MB=bpf5 * n5 / 1e6,
                               features_only=feat5, generator_outputs=gen5)
    head = res["config5_100k"]["features_only"]
    return {"kernel": "cvf_align_feature_fwd", "bound": "hbm", "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": head["frac"], "avg_launch_us": head["avg_launch_us"], "bytes_per_frame": bpf5, "frames_per_launch": n5,
            "copy_GBps": head["copy_GBps"], "read_GBps": head["read_GBps"], "k1_over_copy": head["k1_over_copy"],
            "k1_over_read": head["k1_over_read"],
            "stream_note": "copy / read = cvf_probe_stream over the same bytes, timed in the same loop (>= 50 ms untimed launches, then 30 timed)",
            "clock_state_at_start": clocks,
            "traffic": pmc_traffic("cvf_align_feature_fwd@c5", n5),
            "workload": "BASELINE config 5 shape (5000 atoms, d_r=384), 100 000 frames resident = 6.2 GB per launch, out of cache",
            "cases": res}


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
    # nm-like units: 20 A / 0.5 A of SURVEY 8d (in A the tanh nets saturate at init); frames generated on the device
    ref = np.random.RandomState(SEED).normal(scale=2.0, size=(n_atoms, 3))
    a = torch.tensor(diag_coeff_for(n_atoms, SEED), dtype=torch.float32)
    torch.manual_seed(SEED)
    model = nn.EigenFunctions(C5["layers"], k)
    layer = pp.AlignFeatureLayer(n_atoms, list(range(n_atoms)), ref, c5_features(n_atoms))
    tok = np.zeros((64, n_atoms, 3), dtype=np.float32) + ref[None].astype(np.float32)
    task = core.EigenFunctionTask(Traj(tok, np.ones(64), 1.0), layer, model, "/tmp/cvf_bench", ALPHA, C5["eig_w"], diag_coeff=a, beta=BETA,
                                  lag_tau=0, learning_rate=LR, k=k, batch_size=B, device=dev, verbose=False, save_model_every_step=0)
    frames = max(frames, B)
    n_batches = frames // B
    X, Wt = device_frames(frames, ref, 0.05, SEED + 1 + rank, dev, chunk=2000)
    log = torch.zeros(n_batches, 3 + 2 * k, device=dev, dtype=torch.float64)

    def step(i):
        b = i % n_batches
        task._graph_step(("bench", b), lambda o: task.train_step(X[b * B:(b + 1) * B], Wt[b * B:(b + 1) * B], out=o), log[b], takes_out=True)
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


def main_transfer(args):
    """Config-3 shape in TRANSFER-OPERATOR mode (lag_tau > 0: the mode the shipped dipeptide notebook runs, main.ipynb:275,420):
    22 atoms, k = 3, nets [66,20,20,20,1], B = 20 000 frames + their lagged partners per step.  Extra measurement."""
    from colvarsfinder import _dist, core, nn, pp
    from tests.synth import Traj
    _dist.init_from_env("nccl")
    world, rank = _dist.world(), _dist.rank()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    ref = np.random.RandomState(SEED).normal(scale=2.0, size=(N_ATOMS, 3))
    torch.manual_seed(SEED)
    model = nn.EigenFunctions(LAYERS, K_NETS)
    layer = pp.AlignFeatureLayer(N_ATOMS, list(range(N_ATOMS)), ref, [("position", tuple(range(N_ATOMS)))])
    tok = np.zeros((64, N_ATOMS, 3), dtype=np.float32) + ref[None].astype(np.float32)
    task = core.EigenFunctionTask(Traj(tok, np.ones(64), 1.0), layer, model, "/tmp/cvf_bench", ALPHA, EIG_W, beta=BETA, lag_tau=3.0,
                                  learning_rate=LR, k=K_NETS, batch_size=args.batch, device=dev, verbose=False, save_model_every_step=0)
    B, lag = min(args.batch, args.frames), 3
    X, Wt = device_frames(args.frames + lag, ref, 0.3, SEED + 1 + rank, dev)
    n_batches = args.frames // B
    log = torch.zeros(n_batches, 3 + 2 * K_NETS, device=dev, dtype=torch.float64)

    def one(b):
        s_ = b * B
        return task.train_step(X[s_:s_ + B], Wt[s_:s_ + B], X[s_ + lag:s_ + lag + B], Wt[s_ + lag:s_ + lag + B], out=log[b])

    def chunk():
        task._graph_call(("transfer", "chunk"), lambda: [one(b) for b in range(n_batches)])

    n_chunks = max(1, args.steps // n_batches)
    for _ in range(max(2, args.warmup // n_batches + 1) + 100):      # (+ the preheat of the headline run)
        chunk()
    torch.cuda.synchronize()
    chunk()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_chunks):
        chunk()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    steps = n_chunks * n_batches
    task._use_graphs, task._events = False, {}
    for i in range(20):
        one(i % n_batches)
    torch.cuda.synchronize()
    kern = {n: float(np.mean([s_.elapsed_time(e_) for s_, e_ in ev])) * 1e3 for n, ev in task._events.items()}
    if rank == 0:
        print(json.dumps({"workload": "config-3 shape, transfer-operator mode (lag 3 frames): 22 atoms, k=3, nets [66,20,20,20,1], B=20000 + lagged partners",
                          "n_gpus": world, "value": world * B * steps / elapsed, "unit": "frames/s", "steps": steps,
                          "ms_per_step": elapsed / steps * 1e3, "final_loss": float(log[n_batches - 1, 0]),
                          "call_avg_us": dict(sorted(kern.items(), key=lambda kv: -kv[1])),
                          "call_timing": "HIP events around each C-ABI call, eager pass (includes launch overhead)"}))


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
    # per-call pass: HIP events around cvf_ae_step (one C call = ae16_kernel + the slab sum / Adam launch), the GPU parked while queued
    evs = []
    for i in range(30):
        torch.cuda._sleep(1_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step(args.warmup + args.steps + i)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    call_us = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evs])) * 1e3
    n_par = sum(p.numel() for p in model.parameters())
    flop = 6.0 * n_par * B          # forward, data gradient, weight gradient: 2 flop per weight and frame each (6 k P)
    ach = flop / (elapsed / args.steps) / 1e12
    if rank == 0:
        print(json.dumps({"workload": "config 2: AutoEncoderTask [66,20,20,20,2]/[2,10,10,66], 22 atoms, B=20000", "n_gpus": world,
                          "value": world * B * args.steps / elapsed, "unit": "frames/s", "ms_per_step": elapsed / args.steps * 1e3,
                          "batch_per_gpu": B, "call_avg_us": {"cvf_ae_step": call_us},
                          "roofline": {"kernel": "cvf_ae_step (ae16_kernel + slab_reduce_kernel)", "bound": "mfma", "flop_per_frame": 6.0 * n_par,
                                       "achieved": ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS,
                                       "note": "6 x 3088 weights x frames / step time; a latency-bound dependent chain, far from the fp32 peak"},
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
    try:   # (leave the process group in order: every rank is past its last collective here)
        import torch.distributed as _d
        if _d.is_available() and _d.is_initialized():
            _d.destroy_process_group()
    except Exception:
        pass
