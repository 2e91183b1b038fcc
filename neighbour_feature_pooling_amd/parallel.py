"""Batch sharding of the NFP path across GPUs (one process per GPU).

NFP is independent per image: ranks own disjoint slices of the batch and the data
path has NO collective.  The only cross-rank traffic is what a caller asks for explicitly:
`gather_batch` (to reassemble an output for checking) and `max_over_ranks` (bench timing).
With torch.distributed backend "nccl" these run over RCCL/xGMI; tests use "gloo" on CPU.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """[lo, hi) of the `n` items rank `rank` of `world` owns; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} / world {world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def nfp_sharded(module, x_full):
    """Run `module` (an NFPPooling) on this rank's slice of a batch every rank holds: returns
    (local_out, (lo, hi)).  No communication."""
    rank, world = world_info()
    lo, hi = shard_range(x_full.shape[0], rank, world)
    return module(x_full[lo:hi].contiguous()), (lo, hi)


def gather_batch(local, total):
    """All-gather ragged batch slices back into one [total, ...] tensor (test / debugging aid)."""
    rank, world = world_info()
    if world == 1:
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def max_over_ranks(value, device=None):
    """max of a python float over ranks (the bench's step time is the slowest rank's)."""
    rank, world = world_info()
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


# ---- one process per GPU, started by the program itself --------------------------------------------------------
# `python bench.py --gpus N` (and `python -m neighbour_feature_pooling_amd.train --gpus N`) must run N ranks with no
# launcher on the command line.  The parent starts `python -m torch.distributed.run` as a CHILD process and relays its
# exit code; it never launches GPU work itself (importing torch does not initialise the GPU; torch.cuda.device_count()
# counts through amdsmi on this image and does not either — on an image without that path it would call
# hipGetDeviceCount in the parent, which is still no exec from a process that has), and nothing is exec'ed.

def launched_by_torchrun():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    """A port that was free a moment ago.  (Closed again before torch.distributed.run binds it: two launches started
    at the same instant on one node can draw the same port — the loser fails at rendezvous, loudly, and is re-run.  The
    driver's own launcher passes --master-port itself.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(n, target, argv, port=None):
    """The child command for `n` ranks of `target` — a script path, or ("-m", "package.module")."""
    tgt = list(target) if isinstance(target, (tuple, list)) else [target]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr=127.0.0.1", f"--master-port={port or free_port()}", *tgt, *argv]


def self_launch(n, target, argv, backend="nccl", dry_run=False):
    """Start `n` ranks of `target argv` and return their exit code (0 only if every rank succeeded).  With the RCCL
    backend every rank needs a GPU of its own: fails loudly otherwise.  dry_run: print the command, start nothing."""
    cmd = self_launch_command(n, target, argv)
    if dry_run:
        print(" ".join(cmd), flush=True)
        return 0
    if backend == "nccl":
        have = torch.cuda.device_count()
        if n > have:
            raise SystemExit(f"--gpus {n}: this node has {have} GPU(s); RCCL needs one per rank "
                             f"(a gloo rehearsal may share them)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode
