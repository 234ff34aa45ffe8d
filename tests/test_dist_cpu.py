"""world_size-2 gloo test of the multi-GPU plumbing: clouds are sharded by
rank with no data-path collective; only the barrier and the MAX-over-ranks timing communicate."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from geot_amd import dist_utils
    from geot_amd.synth import make_batch
    from oracle import capi
    w, r, _ = dist_utils.init("gloo")
    assert (w, r) == (world, rank)
    lo, hi = dist_utils.cloud_range(rank, 2)
    xyz, _ = make_batch(hi - lo, 512, start_index=lo)
    idx = capi.fps_dense(xyz, 64, 512, True)           # per-rank work, no communication
    checksum = float(idx.astype(np.int64).sum())
    dist_utils.barrier()
    t_max = dist_utils.max_over_ranks(1.0 + rank)      # slowest rank defines the step time
    total = dist_utils.sum_over_ranks(hi - lo)
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([checksum], dtype=torch.float64))
    q.put((rank, lo, hi, t_max, total, [float(g) for g in gathered]))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, tot0, g0), (r1, lo1, hi1, t1, tot1, g1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 2, 2, 4)        # disjoint, contiguous cloud ranges
    assert t0 == t1 == 2.0 and tot0 == tot1 == 4.0
    assert g0 == g1 and g0[0] != g0[1]                 # different clouds -> different samples
    # single-process reference of the same shards
    from geot_amd.synth import make_batch
    from oracle import capi
    for r, (lo, hi) in enumerate([(0, 2), (2, 4)]):
        xyz, _ = make_batch(hi - lo, 512, start_index=lo)
        assert float(capi.fps_dense(xyz, 64, 512, True).astype(np.int64).sum()) == g0[r]
