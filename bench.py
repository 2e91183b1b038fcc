#!/usr/bin/env python3
"""bench.py — NFP forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one NFPPooling forward + one backward (given grad_out) over one batch of
synthetic feature maps already resident in HBM, through the product path (nn.Module ->
autograd -> C ABI -> HIP kernels).  Workload at every N: BASELINE.json configs[1],
NFP(cosine, k=3) on [B=64, C=512, 7x7] fp32 per GPU (weak scaling: every rank owns its
own batch; the path has no exchange step, so there is no collective in the timed region
beyond the bracketing barriers).  Prints ONE JSON line on rank 0.

The K timed steps are captured once into a HIP graph and replayed (`--launch eager`
times plain launches instead): at this size a step is ~10 us of GPU work, far below the
host cost of two Python->ctypes->hipLaunch round trips.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured-achievable copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--size", type=int, default=7)
    ap.add_argument("--radius", type=int, default=1)
    ap.add_argument("--measure", default="cosine")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def algorithmic_bytes(B, C, HW, N, e):
    """per output pixel fwd = C*e + N*e, bwd = 2*C*e + N*e."""
    px = B * HW
    return px * (C * e + N * e), px * (2 * C * e + N * e)


def time_kernel_graph(fn, reps, stream):
    """Average duration of `fn`'s kernel: `reps` back-to-back launches captured in one HIP
    graph, bracketed by events on the launch stream."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)  # us
    best.sort()
    return best[len(best) // 2]


def usable_cpus():
    """CPUs this process may actually run on (cgroup quota / affinity), not the box's logical count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args, budget_s):
    """The reference's CPU op sequence (oracle/unfold_torch.py) on this box's host cores,
    same [B,C,H,W] workload, bounded to ~budget_s seconds.  The thread count is the fastest of a
    few candidates (torch's default of one thread per logical CPU oversubscribes a shared box)."""
    from oracle.unfold_torch import UnfoldNFP
    B, C, S = args.batch, args.channels, args.size
    ctor = dict(R=args.radius, measure=args.measure, padding=args.radius)
    if args.measure == "norm":
        ctor["p"] = 2
    m = UnfoldNFP(C, **ctor)
    x = torch.randn(B, C, S, S, requires_grad=True)
    go = torch.randn(B, m.N, S, S)

    def one():
        x.grad = None
        t = time.perf_counter()
        m(x).backward(go)
        return time.perf_counter() - t

    ncpu = usable_cpus()
    best_thr, best_t = None, None
    for thr in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
        torch.set_num_threads(thr)
        one()
        t = min(one(), one())
        if best_t is None or t < best_t:
            best_thr, best_t = thr, t
    torch.set_num_threads(best_thr)
    one()
    n, t0 = 0, time.perf_counter()
    while True:
        x.grad = None
        m(x).backward(go)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 100:
            break
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B * S * S / dt / 1e6, 5), "unit": "Mpixels/s", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": round(dt * 1e3, 2),
            "sample": f"{n} fwd+bwd steps of the same [{B},{C},{S},{S}] fp32 workload, reference op sequence "
                      f"(pad -> one-hot depthwise convs -> cosine_similarity -> autograd) in PyTorch CPU, "
                      f"{torch.get_num_threads()} threads (fastest of 8/16/32/64; {ncpu} usable of "
                      f"{os.cpu_count()} logical cpus)"}


def unfold_on_gpu(args, dev, steps=50):
    """BASELINE.json configs[1] asks for "1xMI355X vs reference unfold path": the same restatement of the
    reference's op sequence (oracle/unfold_torch.py), run as ATen/MIOpen kernels on this GPU on the same
    workload.  A reported baseline like cpu_baseline, never the product path.  Timed both ways the product
    is timed: K steps in one HIP graph, and eager."""
    from oracle.unfold_torch import UnfoldNFP
    B, C, S = args.batch, args.channels, args.size
    ctor = dict(R=args.radius, measure=args.measure, padding=args.radius)
    if args.measure == "norm":
        ctor["p"] = 2
    m = UnfoldNFP(C, **ctor)
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    m.w_comp, m.w_centre = m.w_comp.to(dev, dtype), m.w_centre.to(dev, dtype)
    x = torch.randn(B, C, S, S, device=dev, dtype=dtype, requires_grad=True)
    go = torch.randn(B, m.N, S, S, device=dev, dtype=dtype)

    def step():
        return torch.autograd.grad(m(x), x, go)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for _ in range(5):
            step()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / steps * 1e3
    graph_ms = None
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(steps):
                step()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / steps * 1e3
    except Exception as exc:  # a library kernel that cannot be captured: keep the eager number
        sys.stderr.write(f"unfold_on_gpu: graph capture failed ({type(exc).__name__}); eager only\n")
        torch.cuda.synchronize()
    best = min(eager_ms, graph_ms) if graph_ms is not None else eager_ms
    return {"value": round(B * S * S / best / 1e3, 3), "unit": "Mpixels/s", "ms_per_step_graph":
            None if graph_ms is None else round(graph_ms, 4), "ms_per_step_eager": round(eager_ms, 4),
            "what": "reference op sequence (pad -> one-hot depthwise convs -> cosine_similarity/norm -> autograd) "
                    "as PyTorch-ROCm ATen/MIOpen kernels on the same GPU and workload"}


def workload_key(args):
    return f"{args.batch}x{args.channels}x{args.size}x{args.size} k{2 * args.radius + 1} {args.measure} {args.dtype}"


def pmc_traffic(kernel_prefix, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/traffic_latest.json, written by scripts/gpu_traffic.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    16-byte-per-lane streaming reads on gfx950).  None when no profile of THIS workload is committed."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    kernels = rec.get("workloads", {}).get(workload_key(args))
    if kernels is None and workload_key(args) == "64x512x7x7 k3 cosine f32":
        kernels = rec.get("kernels", {})   # first-generation file: headline workload only
    for k, v in (kernels or {}).items():
        if k.startswith(kernel_prefix):
            return v.get("hbm_bytes_per_launch")
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # NFP_BENCH_BACKEND=gloo rehearses the N>1 path with several ranks sharing one GPU (RCCL refuses that)
    backend = os.environ.get("NFP_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from neighbour_feature_pooling_amd import NFPPooling, _abi
    L = _abi.load()
    B, C, S, R = args.batch, args.channels, args.size, args.radius
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    ctor = dict(R=R, measure=args.measure, padding=R)
    if args.measure == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    N = m.out_channels
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, C, S, S, device=dev, generator=gen).to(dtype).requires_grad_(True)
    go = torch.randn(B, N, S, S, device=dev, generator=gen).to(dtype)

    def step():
        out = m(x)
        (gx,) = torch.autograd.grad(out, x, go)
        return out, gx

    def barrier():
        if dist is not None:
            dist.barrier()

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    n_before = L.nfp_launch_count()

    if args.launch == "graph":
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
        graph.replay()  # untimed: first replay uploads the graph
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        graph.replay()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
    else:
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
    assert L.nfp_launch_count() >= n_before + 2 * args.steps, "HIP kernels did not run"
    fwd_variant = bwd_variant = ""
    with torch.cuda.stream(stream):
        out = m(x)
        fwd_variant = L.nfp_last_variant().decode()
        torch.autograd.grad(out, x, go)
        bwd_variant = L.nfp_last_variant().decode()

    from neighbour_feature_pooling_amd.parallel import max_over_ranks
    elapsed = max_over_ranks(t1 - t0, device=dev if backend == "nccl" else "cpu")
    px_per_step = B * S * S
    value = world * px_per_step * args.steps / elapsed / 1e6

    res = None
    if rank == 0:
        # per-kernel timing on the launch stream (HIP events around graph-replayed back-to-back launches)
        e = 4 if args.dtype == "f32" else 2
        fb, bb = algorithmic_bytes(B, C, S * S, N, e)
        with torch.cuda.stream(stream):
            out = m(x)
            xd = x.detach()
            with torch.no_grad():
                t_fwd = time_kernel_graph(lambda: m(xd), 50, stream)
            t_fwd_saving = time_kernel_graph(lambda: m(x), 50, stream)
            t_bwd = time_kernel_graph(lambda: torch.autograd.grad(out, x, go, retain_graph=True), 50, stream)
        dom_bytes, dom_t, dom = (bb, t_bwd, "backward") if t_bwd >= t_fwd_saving else (fb, t_fwd_saving, "forward")
        achieved = dom_bytes / (dom_t * 1e-6) / 1e9
        res = {
            "metric": "NFP fwd+bwd Mpixels/s @ [B64,C512,7\u00d77,k3]; 1/2/4/8 GPU + %HBM roofline",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"NFP({args.measure},k={2 * R + 1},reflect pad {R}) fwd+bwd on "
                                   f"[{B},{C},{S},{S}] {args.dtype} NCHW per GPU (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "launch": args.launch,
                       "parallelism": f"batch-sharded replicas x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": f"{dom}:{bwd_variant if dom == 'backward' else fwd_variant}",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic((bwd_variant if dom == "backward" else fwd_variant)
                                                .split("<")[0].split("+")[0], args),
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": round(dom_t, 3)},
            "kernels": {"forward_us": round(t_fwd_saving, 3), "forward_nograd_us": round(t_fwd, 3),
                        "backward_us": round(t_bwd, 3), "forward_variant": fwd_variant,
                        "backward_variant": bwd_variant, "fwd_bytes": fb, "bwd_bytes": bb,
                        "fwd_GBs": round(fb / t_fwd_saving / 1e3, 1), "bwd_GBs": round(bb / t_bwd / 1e3, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["unfold_path_same_gpu"] = unfold_on_gpu(args, dev)
            res["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
