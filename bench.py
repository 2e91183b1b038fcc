#!/usr/bin/env python3
"""bench.py — NFP forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts N ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` as a
child process, one rank per GPU over RCCL); under an external launcher (WORLD_SIZE / RANK in the environment) it runs
as the rank it is told to be and checks that WORLD_SIZE equals N.

A step = one NFPPooling forward + one backward (given grad_out) over one batch of synthetic feature maps
already resident in HBM, through the product path (nn.Module -> autograd -> C ABI -> HIP kernels).
Workload at every N: BASELINE.json configs[1], NFP(cosine, k=3) on [B=64, C=512, 7x7] fp32 per GPU (weak
scaling: every rank owns its own batch; the path has no exchange step, so there is no collective in the
timed region beyond the bracketing barriers).  Prints ONE JSON line on rank 0.

Two legs, both timed over exactly K steps between barriers:
  * `value` / `ms_per_step`: K steps on ONE set of (x, grad_out) — 6.4 MB of x, served by L2 / the 256 MiB Infinity
    Cache after the first step, which is what a training step sees: the backbone has just written the feature map.
    (Round 1's protocol; every step still writes a fresh out / grad_x buffer.)
  * `cold` and `roofline`: the ROTATING leg — step i works on buffer set i mod S, S sets holding more than 256 MiB
    of x alone, a fresh grad_x / out per step, and a 512 MiB fill right before the clock starts: x, grad_out and
    grad_x are HBM traffic (within a step the backward re-reads the x its forward read microseconds before).  The
    HBM roofline fraction is quoted on THIS leg's kernel times, never on the cache-resident ones.
Beside them (`--no-extras` skips these): the same kernels at a saturating batch, and `large_maps` — the module on the
maps above 512 pixels MobileNetV3_MultiStageNFP feeds it at B = 256 (row-band kernels), per-kernel event times.
The K timed steps are captured once into a HIP graph and replayed (`--launch eager` times plain launches
instead): a step is ~11 us of GPU work, far below the host cost of two Python->ctypes->hipLaunch round trips.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured-achievable copy rate
INFINITY_CACHE_BYTES = 256 << 20


def parse(argv=None):
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
    ap.add_argument("--layout", choices=["nchw", "nhwc"], default="nchw")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the cache-resident, ReLU-input, saturating-batch, large-map and "
                    "live-traffic legs: every NFP launch of the process then belongs to the rotating leg, so a rocprofv3 "
                    "--kernel-trace --stats summary of the run averages exactly the launches `roofline` is quoted on")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # run under rocprofv3 --pmc by bench.py itself
    ap.add_argument("--print-launch", action="store_true", help="with --gpus N > 1 and no launcher: print the "
                    "torch.distributed.run command the N ranks would be started with, and exit")
    return ap.parse_args(argv)


def algorithmic_bytes(B, C, HW, N, e):
    """per output pixel fwd = C*e + N*e, bwd = 2*C*e + N*e."""
    px = B * HW
    return px * (C * e + N * e), px * (2 * C * e + N * e)


def time_graph(fns, stream, repeats=5):
    """Average duration (us) of one call: the calls `fns` (a list, run in order) are captured into one HIP graph
    and replayed, bracketed by events on the launch stream; median of `repeats` replays."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        keep = [fn() for fn in fns]   # (results stay alive inside the graph's pool: distinct output buffers)
    del keep
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / len(fns))
    ts.sort()
    return ts[len(ts) // 2]


def time_kernel_graph(fn, reps, stream):
    """One callable repeated `reps` times (scripts/ab_flags.py and friends use this form)."""
    return time_graph([fn] * reps, stream)


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


def module_ctor(args):
    ctor = dict(R=args.radius, measure=args.measure, padding=args.radius)
    if args.measure == "norm":
        ctor["p"] = 2
    return ctor


def cpu_baseline(args, budget_s):
    """The reference's CPU op sequence (oracle/unfold_torch.py) on this box's host cores,
    same [B,C,H,W] workload, bounded to ~budget_s seconds.  The thread count is the fastest of a
    few candidates (torch's default of one thread per logical CPU oversubscribes a shared box)."""
    from oracle.unfold_torch import UnfoldNFP
    B, C, S = args.batch, args.channels, args.size
    m = UnfoldNFP(C, **module_ctor(args))
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
    m = UnfoldNFP(C, **module_ctor(args))
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


def source_hash():
    """Hash of the kernel sources: a committed traffic profile is only reported for the code it was measured on
    (the GPU box has no .git, so the commit id is not available there)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neighbour_feature_pooling_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        h.update(f.encode())
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from profiles/traffic_latest.json (scripts/gpu_traffic.sh) — only when that
    file was measured on THESE kernel sources and this workload; otherwise None."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    except (OSError, ValueError):
        return None
    if rec.get("source_hash") != source_hash():
        return None
    for k, v in (rec.get("workloads", {}).get(workload_key(args)) or {}).items():
        if k.startswith(kernel):
            return v.get("hbm_bytes_per_launch")
    return None


def live_traffic(kernel, argv):
    """FETCH_SIZE and WRITE_SIZE of `kernel` measured NOW: this script re-runs itself (--pmc-child: a short rotating
    loop, nothing else) under `rocprofv3 --pmc`, one counter per pass as MI355X_MICROARCH.md prescribes (TCC has 4
    slots; FETCH_SIZE takes 3, WRITE_SIZE 2), kernel-trace only.  hbm bytes per launch = (2*FETCH_SIZE +
    WRITE_SIZE) KiB: on gfx950 FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes.
    Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="nfp_pmc_", dir="/tmp")
        cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child"] + argv
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=150, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            per = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                   if r["Counter_Name"] == counter and ("nfp::" + kernel) in r["Kernel_Name"]]
            if not per:
                return None, f"no {kernel} dispatch in the {counter} pass"
            vals[counter] = sum(per) / len(per)
        except Exception as exc:   # profiler missing / refused / timed out: the committed profile or null
            return None, f"{counter} pass failed: {type(exc).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    b = int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    return b, {"FETCH_SIZE_KB": round(vals["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(vals["WRITE_SIZE"], 1),
               "read_bytes": int(2 * vals["FETCH_SIZE"] * 1024), "write_bytes": int(vals["WRITE_SIZE"] * 1024)}


class KernelTimer:
    """Per-kernel durations the way a profiler's kernel trace measures them: libnfp_hip.so brackets a launch with two
    HIP events recorded at the kernel's own start and end (include/nfp.h: nfp_time_next_launch)."""

    def __init__(self, L):
        import ctypes
        self.ct, self.L = ctypes, L
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def _event(self):
        ev = self.ct.c_void_p()
        assert self.hip.hipEventCreate(self.ct.byref(ev)) == 0
        return ev

    def measure(self, launches):
        """launches: callables that each enqueue exactly ONE nfp kernel first.  Returns their durations in us."""
        pairs = []
        for fn in launches:
            e0, e1 = self._event(), self._event()
            self.L.nfp_time_next_launch(e0, e1)
            fn()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        out = []
        for e0, e1 in pairs:
            ms = self.ct.c_float()
            assert self.hip.hipEventElapsedTime(self.ct.byref(ms), e0, e1) == 0
            out.append(ms.value * 1e3)
            self.hip.hipEventDestroy(e0)
            self.hip.hipEventDestroy(e1)
        return out


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


class Workload:
    """S rotating sets of (x, grad_out) for one module; `step(i)` = product forward + backward on set i mod S."""

    def __init__(self, args, dev, rank, batch=None, sets=None, relu=False):
        from neighbour_feature_pooling_amd import NFPPooling
        self.B = batch or args.batch
        C, S_, R = args.channels, args.size, args.radius
        self.dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
        self.m = NFPPooling(C, **module_ctor(args))
        self.N = self.m.out_channels
        e = 4 if args.dtype == "f32" else 2
        x_bytes = self.B * C * S_ * S_ * e
        self.sets = sets if sets is not None else INFINITY_CACHE_BYTES // x_bytes + 2   # x alone exceeds the cache
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.x, self.go = [], []
        for _ in range(self.sets):
            x = torch.randn(self.B, C, S_, S_, device=dev, generator=gen)
            if relu:
                x = x.relu()
            x = x.to(self.dtype)
            if args.layout == "nhwc":
                x = x.contiguous(memory_format=torch.channels_last)
            self.x.append(x.requires_grad_(True))
            self.go.append(torch.randn(self.B, self.N, S_, S_, device=dev, generator=gen).to(self.dtype))
        self.fb, self.bb = algorithmic_bytes(self.B, C, S_ * S_, self.N, e)

    def step(self, i=0):
        k = i % self.sets
        out = self.m(self.x[k])
        (gx,) = torch.autograd.grad(out, self.x[k], self.go[k])
        return out, gx

    def kernel_events(self, timer, stream, rounds=2):
        """(forward us, backward us): each kernel of `rounds` passes over the sets (eager steps, forward then
        backward of the same set, as the timed region orders them) bracketed by its own pair of events; means."""
        tf, tb = [], []
        with torch.cuda.stream(stream):
            for i in range(rounds * self.sets):
                k = i % self.sets
                out = []
                tf += timer.measure([lambda: out.append(self.m(self.x[k]))])
                tb += timer.measure([lambda: torch.autograd.grad(out[0], self.x[k], self.go[k])])
        return sum(tf) / len(tf), sum(tb) / len(tb), median(tf), median(tb)

    def kernel_times(self, stream, reps=None, isolated_backward=False):
        """(forward us, backward us) per launch over the sets, HIP events around captured graphs on the launch
        stream.  The backward is timed IN the step sequence — a graph of `reps` steps minus a graph of the same
        `reps` forwards — i.e. as the timed region runs it: grad_out / out / grad_x are HBM traffic, x was read by
        the step's forward a few microseconds earlier.  isolated_backward=True times backward launches alone
        instead (x from HBM too when the sets rotate)."""
        reps = reps or max(self.sets, 40)
        with torch.cuda.stream(stream):
            tf = time_graph([(lambda k=i % self.sets: self.m(self.x[k])) for i in range(reps)], stream)
            if isolated_backward:
                outs = [self.m(self.x[k]) for k in range(self.sets)]
                tb = time_graph([(lambda k=i % self.sets: torch.autograd.grad(outs[k], self.x[k], self.go[k], retain_graph=True))
                                 for i in range(reps)], stream)
            else:
                tb = time_graph([(lambda k=i: self.step(k)) for i in range(reps)], stream) - tf
        return tf, tb


def pmc_child(args):
    """What rocprofv3 --pmc profiles: 24 rotating steps of the default workload, nothing else."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = Workload(args, dev, 0)
    for i in range(24):
        w.step(i)
    torch.cuda.synchronize()


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    backend = os.environ.get("NFP_BENCH_BACKEND", "nccl")
    from neighbour_feature_pooling_amd import parallel
    if args.gpus > 1 and not parallel.launched_by_torchrun():
        # `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) and only relays their
        # exit code — it has made no GPU call, and the ranks are children, never an exec of this process
        argv = [a for a in sys.argv[1:] if a != "--print-launch"]
        sys.exit(parallel.self_launch(args.gpus, os.path.abspath(__file__), argv, backend=backend,
                                      dry_run=args.print_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # NFP_BENCH_BACKEND=gloo rehearses the N>1 path with several ranks sharing one GPU (RCCL refuses that)
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

    from neighbour_feature_pooling_amd import _abi
    L = _abi.load()
    B, C, S, R = args.batch, args.channels, args.size, args.radius
    rot = Workload(args, dev, rank)               # the timed leg: rotating sets, every byte from HBM
    hot = Workload(args, dev, rank, sets=1)       # one resident set

    def barrier():
        if dist is not None:
            dist.barrier()

    stream = torch.cuda.Stream(device=dev)

    flush_buf = torch.empty(2 * INFINITY_CACHE_BYTES, dtype=torch.uint8, device=dev)

    def timed(w, cold=True):
        """K steps of workload `w` on `stream`: elapsed seconds.  cold: a 512 MiB fill runs (untimed) right before the
        timed region, so that nothing the steps read or write — x, grad_out, and the grad_x / out buffers the graph
        wrote on its previous replay — is still in the 256 MiB Infinity Cache when the clock starts."""
        with torch.cuda.stream(stream):
            for i in range(args.warmup):
                w.step(i)
        torch.cuda.synchronize()
        n0 = L.nfp_launch_count()
        if args.launch == "graph":
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                keep = [w.step(i) for i in range(args.steps)]
            del keep
            n_graph = L.nfp_launch_count() - n0
            torch.cuda.synchronize()
            graph.replay()  # untimed: first replay uploads the graph
            if cold:
                flush_buf.fill_(1)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()   # this rank's K steps; the closing barrier's own latency (tens of us over RCCL,
            barrier()                  # comparable to K = 20 steps) is not NFP work: MAX over ranks is taken below
            del graph
        else:
            if cold:
                flush_buf.fill_(1)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(stream):
                for i in range(args.steps):
                    w.step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            barrier()
            n_graph = L.nfp_launch_count() - n0
        assert n_graph >= 2 * args.steps, "HIP kernels did not run"
        return t1 - t0

    from neighbour_feature_pooling_amd.parallel import max_over_ranks
    red_dev = dev if backend == "nccl" else "cpu"
    elapsed_cold = max_over_ranks(timed(rot), device=red_dev)
    elapsed = max_over_ranks(timed(hot, cold=False), device=red_dev)
    extras = world == 1 and not args.no_extras
    with torch.cuda.stream(stream):
        out = hot.m(hot.x[0])
        fwd_variant = L.nfp_last_variant().decode()
        torch.autograd.grad(out, hot.x[0], hot.go[0])
        torch.cuda.synchronize()
        bwd_variant = L.nfp_last_variant().decode()
    px_per_step = B * S * S
    value = world * px_per_step * args.steps / elapsed / 1e6

    res = None
    if rank == 0:
        timer = KernelTimer(L)
        # Per-kernel durations of the TIMED REGION's launches: HIP events on the launch stream around graph-replayed
        # launches — a graph of forwards, and a graph of steps minus it.  (rocprofv3's kernel trace of a graph replay
        # stamps a kernel's start where its predecessor ends, so its per-kernel averages partition the step the same
        # way.)  Also reported: each launch bracketed by its own event pair in eager order (kernel start to kernel end
        # on an otherwise idle GPU, nfp_time_next_launch).
        tf, tb = rot.kernel_times(stream)
        tf_ev, tb_ev, tf_med, tb_med = rot.kernel_events(timer, stream)
        dom_bytes, dom_t, dom, dom_variant = ((rot.bb, tb, "backward", bwd_variant) if tb >= tf else
                                              (rot.fb, tf, "forward", fwd_variant))
        achieved = dom_bytes / (dom_t * 1e-6) / 1e9
        dom_kernel = dom_variant.split("<")[0].split("+")[0]
        traffic, traffic_src = None, None
        if extras:
            traffic, traffic_src = live_traffic(dom_kernel, [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)])
        if traffic is None:
            why = traffic_src
            traffic = committed_traffic(dom_kernel, args)
            traffic_src = ("profiles/traffic_latest.json (same kernel sources)" if traffic is not None
                           else f"none ({why or 'no live pass'}; no committed profile of these sources)")
        e = 4 if args.dtype == "f32" else 2
        res = {
            "metric": "NFP fwd+bwd Mpixels/s @ [B64,C512,7×7,k3]; 1/2/4/8 GPU + %HBM roofline",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"NFP({args.measure},k={2 * R + 1},reflect pad {R}) fwd+bwd on "
                                   f"[{B},{C},{S},{S}] {args.dtype} {args.layout.upper()} per GPU (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "global_batch": B * world, "launch": args.launch,
                       "buffers": "value / ms_per_step: one resident set of (x, grad_out) per GPU, a fresh out / grad_x buffer per "
                                  "step (round 1's protocol: the feature map has just been written by the backbone); "
                                  "`cold` and `roofline`: rotating sets, every byte from HBM",
                       "parallelism": f"batch-sharded replicas x{world}, no data-path collective"},
            "cold": {"what": f"the same K steps over {rot.sets} rotating sets of (x, grad_out) = "
                             f"{rot.sets * rot.B * C * S * S * e >> 20} MiB of x per GPU (> 256 MiB Infinity Cache), a fresh "
                             f"grad_x / out per step, 512 MiB fill right before the clock: every step reads and writes HBM",
                     "value": round(world * px_per_step * args.steps / elapsed_cold / 1e6, 3), "unit": "Mpixels/s",
                     "ms_per_step": round(elapsed_cold / args.steps * 1e3, 6)},
            "roofline": {"bound": "hbm", "leg": "cold: rotating buffers, kernel timed inside the step sequence (grad_out, out "
                                                  "and grad_x are HBM traffic; the backward re-reads the x its forward read "
                                                  "microseconds earlier)",
                         "kernel": f"{dom}:{dom_variant}",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": round(dom_t, 3)},
            "kernels": {"leg": "cold (rotating buffers)", "how": "forward_us: HIP events around a graph of rotating forwards / launches; backward_us: the same "
                               "around a graph of rotating steps, minus forward_us (launch boundaries included, as in "
                               "the timed region and in rocprofv3's trace of a graph replay); *_eager_event_us: mean / "
                               "median of each launch's own hipExtLaunchKernel event pair, eager order, idle GPU between",
                        "forward_us": round(tf, 3), "backward_us": round(tb, 3),
                        "forward_eager_event_us": round(tf_ev, 3), "backward_eager_event_us": round(tb_ev, 3),
                        "forward_eager_event_median_us": round(tf_med, 3), "backward_eager_event_median_us": round(tb_med, 3),
                        "forward_variant": fwd_variant, "backward_variant": bwd_variant,
                        "fwd_bytes": rot.fb, "bwd_bytes": rot.bb,
                        "fwd_GBs": round(rot.fb / tf / 1e3, 1), "bwd_GBs": round(rot.bb / tb / 1e3, 1)},
        }
        if extras:
            tf_hot, tb_hot = hot.kernel_times(stream, reps=50)
            res["cache_resident"] = {"what": "kernel times of the `value` leg (one buffer set: inputs from L2 / Infinity Cache)",
                                     "forward_us": round(tf_hot, 3), "backward_us": round(tb_hot, 3),
                                     "fwd_GBs": round(rot.fb / tf_hot / 1e3, 1), "bwd_GBs": round(rot.bb / tb_hot / 1e3, 1)}
            rl = Workload(args, dev, rank, relu=True)                       # what a ResNet trunk emits: x >= 0, ~50 % zeros
            rf, rbk = rl.kernel_times(stream)
            res["relu_input"] = {"what": "the same workload with x = relu(randn): non-negative, about half zeros",
                                 "forward_us": round(rf, 3), "backward_us": round(rbk, 3),
                                 "value": round(px_per_step / (rf + rbk), 3), "unit": "Mpixels/s (kernel time)"}
            del rl
            big = Workload(args, dev, rank, batch=4096, sets=2)            # 0.8 GB of x per set: far beyond every cache
            # (round 4, VERDICT r3: graph-replayed launches like every other leg — round 3 timed eager launches between
            # events here, with the GPU idling at lower clocks in between, and read 112 us for a kernel rocprofv3 saw at 92)
            bfe, bbe, _, _ = big.kernel_events(timer, stream, rounds=2)
            bf_, bb_ = big.kernel_times(stream, reps=6)
            with torch.cuda.stream(stream):
                o = big.m(big.x[0])
                bfv = L.nfp_last_variant().decode()
                torch.autograd.grad(o, big.x[0], big.go[0])
                torch.cuda.synchronize()
                bbv = L.nfp_last_variant().decode()
            res["saturating_batch"] = {"batch": 4096, "how": "graphs of 6 rotating launches (2 sets of 0.8 GB of x), as the `kernels` leg",
                                       "forward_us": round(bf_, 2), "backward_us": round(bb_, 2),
                                       "forward_eager_event_us": round(bfe, 2), "backward_eager_event_us": round(bbe, 2),
                                       "forward_variant": bfv, "backward_variant": bbv,
                                       "fwd_GBs": round(big.fb / bf_ / 1e3, 1), "bwd_GBs": round(big.bb / bb_ / 1e3, 1),
                                       "fwd_frac_of_peak": round(big.fb / bf_ / 1e3 / HBM_PEAK_GBS, 4),
                                       "bwd_frac_of_peak": round(big.bb / bb_ / 1e3 / HBM_PEAK_GBS, 4),
                                       "value": round(4096 * S * S / (bf_ + bb_), 1), "unit": "Mpixels/s (kernel time)"}
            del big, o
            # the maps above 512 pixels that MobileNetV3_MultiStageNFP feeds the same module (texture_pooling.py:211-268) at
            # B = 256: the row-band kernels (csrc/nfp_tile.h), graphs of 6 rotating launches over three input sets
            import copy
            lm = []
            for C_, S_ in ((16, 112), (24, 56), (40, 28)):
                a2 = copy.copy(args)
                a2.channels, a2.size = C_, S_
                wl = Workload(a2, dev, rank, batch=256, sets=3)
                lf, lb = wl.kernel_times(stream, reps=6)
                with torch.cuda.stream(stream):
                    o = wl.m(wl.x[0])
                    lfv = L.nfp_last_variant().decode()
                    torch.autograd.grad(o, wl.x[0], wl.go[0])
                    torch.cuda.synchronize()
                    lbv = L.nfp_last_variant().decode()
                lm.append({"shape": [256, C_, S_, S_], "forward_us": round(lf, 2), "backward_us": round(lb, 2),
                           "forward_variant": lfv, "backward_variant": lbv,
                           "fwd_frac_of_peak": round(wl.fb / lf / 1e3 / HBM_PEAK_GBS, 4),
                           "bwd_frac_of_peak": round(wl.bb / lb / 1e3 / HBM_PEAK_GBS, 4)})
                del wl, o
                torch.cuda.empty_cache()
            res["large_maps"] = lm
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
