"""world_size-2 gloo tests of the multi-GPU plumbing: clouds are sharded by
rank with no data-path collective; only the barrier and the MAX-over-ranks timing communicate.  The NTM
block's one optional exchange -- the per-class anchor rows, so that class_T / ema_t match what a single
process computes on the whole batch -- is an all-gather of 17 x 18 floats, checked here against the
single-process result."""
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


def _anchor_batch():
    rng = np.random.default_rng(5)
    logits = rng.standard_normal((4, 17, 300))
    e = np.exp(logits - logits.max(1, keepdims=True))
    eta = e / e.sum(1, keepdims=True)
    # exact ties across ranks: class 3's best probability appears in cloud 1 (rank 0) and cloud 2 (rank 1),
    # class 5's in clouds 2 and 3 (both rank 1) -- the first in flattened (b, n) order must win
    for cls, (b1, n1), (b2, n2) in ((3, (1, 250), (2, 10)), (5, (2, 200), (3, 7))):
        row = np.full(17, 0.001)
        row[cls] = 1 - 0.016
        eta[b1, :, n1] = row
        row2 = np.roll(row, 0).copy()
        row2[(cls + 1) % 17], row2[(cls + 2) % 17] = 0.0015, 0.0005     # same maximum, different row
        eta[b2, :, n2] = row2
    sigma = 0.5 + rng.random(17)
    ema = rng.random((17, 17)) + 0.1
    return eta, sigma, ema / ema.sum(1, keepdims=True)


def _anchor_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from geot_amd import dist_utils
    from geot_amd.ntm import class_transition
    dist_utils.init("gloo")
    eta, sigma, ema = _anchor_batch()
    lo, hi = dist_utils.cloud_range(rank, 2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    corr, nxt, cT, _ = class_transition(T(eta[lo:hi]), T(sigma), T(ema), group=dist.group.WORLD)
    local = class_transition(T(eta[lo:hi]), T(sigma), T(ema))[2]
    q.put((rank, corr.numpy(), nxt.numpy(), cT.numpy(), local.numpy()))
    dist.destroy_process_group()


def test_two_rank_anchor_rows_match_single_process():
    from geot_amd.ntm import class_transition
    from oracle import np_ntm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_anchor_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eta, sigma, ema = _anchor_batch()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    corr, nxt, cT, _ = class_transition(T(eta), T(sigma), T(ema))
    want = np_ntm.class_transition(eta, sigma, ema)
    assert np.array_equal(cT.numpy(), want["class_T"])
    for rank, r_corr, r_nxt, r_cT, r_local in res:
        assert np.array_equal(r_cT, cT.numpy()), "rank %d anchors differ from the single-process ones" % rank
        np.testing.assert_allclose(r_corr, corr.numpy(), rtol=1e-14)
        np.testing.assert_allclose(r_nxt, nxt.numpy(), rtol=1e-14)
    assert not np.array_equal(res[0][4], res[1][4])          # per-rank anchors do differ without the exchange
    assert np.array_equal(cT.numpy()[3], eta[1, :, 250]) and np.array_equal(cT.numpy()[5], eta[2, :, 200])


# ---- eight ranks (the node the scaling bench runs on), uneven shards -------------------------------------------------------
def _anchor_batch8(total=11):
    """11 unlabelled clouds over 8 ranks at 2 per rank: ranks 0-4 own two, rank 5 owns one, ranks 6-7 own NONE.  Exact
    ties between ranks for three classes: the first maximum in flattened (b, n) order -- the lowest rank's -- must win."""
    rng = np.random.default_rng(8)
    logits = rng.standard_normal((total, 17, 120))
    e = np.exp(logits - logits.max(1, keepdims=True))
    eta = e / e.sum(1, keepdims=True)
    ties = ((2, (1, 30), (9, 4)), (7, (4, 100), (10, 0)), (11, (6, 5), (8, 77)))       # class, first site, later site
    for cls, (b1, n1), (b2, n2) in ties:
        row = np.full(17, 0.001)
        row[cls] = 1 - 0.016
        eta[b1, :, n1] = row
        row2 = row.copy()
        row2[(cls + 1) % 17], row2[(cls + 2) % 17] = 0.0015, 0.0005
        eta[b2, :, n2] = row2
    sigma = 0.5 + rng.random(17)
    ema = rng.random((17, 17)) + 0.1
    return eta, sigma, ema / ema.sum(1, keepdims=True), ties


def _anchor_worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from geot_amd import dist_utils
    from geot_amd.ntm import class_transition
    dist_utils.init("gloo")
    eta, sigma, ema, _ = _anchor_batch8()
    lo, hi = dist_utils.cloud_range(rank, 2, total=eta.shape[0])
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))       # noqa: E731
    corr, nxt, cT, _ = class_transition(T(eta[lo:hi]), T(sigma), T(ema), group=dist.group.WORLD)
    owned = dist_utils.gather_over_ranks(hi - lo)
    q.put((rank, lo, hi, corr.numpy(), nxt.numpy(), cT.numpy(), owned, dist_utils.max_over_ranks(rank), dist_utils.sum_over_ranks(hi - lo)))
    dist.destroy_process_group()


def test_eight_rank_uneven_shards_and_anchor_exchange():
    """world 8 over gloo: cloud_range with a total (the last ranks own one / zero clouds), the control-plane reductions,
    and the anchor all-gather -- every rank, the empty ones included, ends with the single-process class_T / ema_t."""
    from geot_amd.ntm import class_transition
    from geot_amd import dist_utils
    assert [dist_utils.cloud_range(r, 2, total=11) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 11), (11, 11), (11, 11)]
    assert dist_utils.cloud_range(3, 8) == (24, 32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_anchor_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    eta, sigma, ema, ties = _anchor_batch8()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))       # noqa: E731
    corr, nxt, cT, _ = class_transition(T(eta), T(sigma), T(ema))
    for cls, (b1, n1), _late in ties:
        assert np.array_equal(cT.numpy()[cls], eta[b1, :, n1])
    for rank, lo, hi, r_corr, r_nxt, r_cT, owned, t_max, total in res:
        assert (lo, hi) == dist_utils.cloud_range(rank, 2, total=11)
        assert np.array_equal(r_cT, cT.numpy()), "rank %d anchors differ from the single-process ones" % rank
        np.testing.assert_allclose(r_corr, corr.numpy(), rtol=1e-14)
        np.testing.assert_allclose(r_nxt, nxt.numpy(), rtol=1e-14)
        assert owned == [2, 2, 2, 2, 2, 1, 0, 0] and t_max == 7.0 and total == 11.0


def _gradsync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from geot_amd import dist_utils, train_step as ts
    dist_utils.init("gloo")
    torch.manual_seed(100 + rank)                       # rank-different initial weights: both forms must start from rank 0's

    def net():
        return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                   torch.nn.Linear(16, 3))
    a, b = net(), net()
    b.load_state_dict(a.state_dict())
    ddp = torch.nn.parallel.DistributedDataParallel(a)              # broadcasts rank 0's parameters at construction
    sync = ts.GradSync([b], dist.group.WORLD)                       # ... as this does
    for p1, p2 in zip(a.parameters(), b.parameters()):
        assert torch.equal(p1, p2)
    oa, ob = torch.optim.SGD(a.parameters(), lr=0.1), torch.optim.SGD(b.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(7 + rank)                     # rank-different data
    worst = 0.0
    for _ in range(3):
        x, y = torch.randn(5 + rank, 6, generator=g), torch.randn(5 + rank, 3, generator=g)
        for model, opt, wrapped in ((a, oa, ddp), (b, ob, None)):
            opt.zero_grad(set_to_none=True)
            out = (wrapped or model)(x)
            torch.nn.functional.mse_loss(out, y).backward()
            if wrapped is None:
                sync()
            opt.step()
        for p1, p2 in zip(a.parameters(), b.parameters()):
            worst = max(worst, float((p1 - p2).abs().max()))
    gathered = [torch.zeros(3) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([p.detach().reshape(-1) for p in b.parameters()])[:3].contiguous())
    q.put((rank, worst, sync.collectives, [g_.tolist() for g_ in gathered]))
    dist.destroy_process_group()


def test_two_rank_flat_gradient_exchange_equals_ddp():
    """train_step.GradSync -- the ONE flat all-reduce a hipGraph-replayed step exchanges its gradients with (graph_step cannot
    hold DistributedDataParallel) -- against DistributedDataParallel itself, two ranks over gloo: the same parameters after
    three optimizer steps on rank-different data (mean of the ranks' gradients, divided before the sum as the reducer does),
    the same on both ranks, one collective per step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gradsync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, collectives, gathered in res:
        assert worst == 0.0, (rank, worst)              # two addends: the sum has one order
        assert collectives == 3
        assert gathered[0] == gathered[1]               # the ranks hold the same parameters
