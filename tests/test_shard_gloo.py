"""N>1 path on CPU: two gloo ranks shard a batch, run NFP on their slices with no communication,
and the reassembled result equals the single-process one bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neighbour_feature_pooling_amd.parallel import shard_range


def test_shard_range_covers_batch_exactly():
    for n in (0, 1, 5, 64, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neighbour_feature_pooling_amd import NFPPooling
        from neighbour_feature_pooling_amd.parallel import gather_batch, max_over_ranks, nfp_sharded
        from neighbour_feature_pooling_amd.synth import feature_map
        B = 5  # ragged: 3 + 2
        x = torch.from_numpy(feature_map((B, 16, 7, 7), 321)).requires_grad_(True)
        go = torch.from_numpy(feature_map((B, 8, 7, 7), 322))
        m = NFPPooling(16, R=1, measure="cosine", padding=1)
        out, (lo, hi) = nfp_sharded(m, x)
        out.backward(go[lo:hi])
        full_out = gather_batch(out.detach(), B)
        full_gx = gather_batch(x.grad[lo:hi], B)
        slow = max_over_ranks(1.0 + rank)
        if rank == 0:
            np.savez(os.path.join(out_dir, "sharded.npz"), out=full_out.numpy(), gx=full_gx.numpy(), slow=slow)
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    from neighbour_feature_pooling_amd import NFPPooling
    from neighbour_feature_pooling_amd.synth import feature_map
    x = torch.from_numpy(feature_map((5, 16, 7, 7), 321)).requires_grad_(True)
    go = torch.from_numpy(feature_map((5, 8, 7, 7), 322))
    out = NFPPooling(16, R=1, measure="cosine", padding=1)(x)
    out.backward(go)
    assert np.array_equal(got["out"], out.detach().numpy())
    assert np.array_equal(got["gx"], x.grad.numpy())
    assert float(got["slow"]) == 2.0
