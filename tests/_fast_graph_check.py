"""Child process of tests/test_graph_step_gpu.py: GEOT_GRAPH_LAUNCH=fast (the runtime's graph packet capture stays ON).
The two training steps must capture as graphs of kernel nodes alone, replay bit-equal to the eager steps with tens of
thousands of eager launches between replays (what corrupts memset / memcpy nodes in this mode), and a body that holds a
copy node must be refused.  Prints "fast ok <node counts>" at the end."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("GEOT_GRAPH_LAUNCH") == "fast" and "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ
import geot_amd  # noqa: E402
import torch  # noqa: E402
from test_graph_step_gpu import _sup_batches, _fix_batch, _state, _same, SMALL, DEV  # noqa: E402
from geot_amd import train_step as ts, graph_step as gs  # noqa: E402
from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T  # noqa: E402

assert not geot_amd.graph_replay_is_safe() and "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ
flood_buf = torch.randn(1 << 16, device=DEV)


def flood(n):
    for _ in range(n):
        flood_buf.mul_(1.0)


which = sys.argv[1]
counts = {}
if which == "supervised":
    batches = _sup_batches(2, 6000)
    torch.manual_seed(0)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    runs = {}
    for mode in ("eager", "graph"):
        m = PointTransformer_seg_T(**SMALL).to(DEV)
        m.load_state_dict(init)
        step = ts.SupervisedStep(m)
        call = gs.GraphedSupervisedStep(step, warmup=2) if mode == "graph" else step
        torch.manual_seed(7)
        losses = []
        for i in range(7):
            cur, nxt = batches[i % 3], batches[(i + 1) % 3]
            losses.append(call(cur[0], cur[1], cur[2], next_pos=nxt[0]).clone())
            if i in (3, 5):
                flood(40000)
        torch.cuda.synchronize()
        runs[mode] = (losses, _state(step))
    counts = call.node_types
    # a body with a device-to-device copy in it is refused
    m = PointTransformer_seg_T(**SMALL).to(DEV)

    class Copying(ts.SupervisedStep):
        def lookahead_work(self, pos):
            out = super().lookahead_work(pos)
            self.scratch = torch.empty_like(pos)
            self.scratch.copy_(pos)
            return out
    bad = gs.GraphedSupervisedStep(Copying(m), warmup=1)
    b = batches[0]
    bad(b[0], b[1], b[2])
    try:
        bad(b[0], b[1], b[2])
    except RuntimeError as e:
        assert "only kernel nodes" in str(e), e
    else:
        raise SystemExit("a graph with a copy node was accepted")
else:
    cfg = dict(ts.NTM_CFG, threed_k=8)
    batches = [_fix_batch(3), _fix_batch(400)]
    runs = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        step = ts.build_fixmatch(DEV, seg_cfg=SMALL, cfg=cfg, use_ddp=False)
        call = gs.GraphedFixMatchStep(step, warmup=2) if mode == "graph" else step
        torch.manual_seed(11)
        losses = []
        for i in range(6):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            res = call(cur[0], cur[1], next_batches=nxt)
            losses.append(torch.stack([res[k].detach().float().reshape(()) for k in sorted(res)]).clone())
            if i in (3, 4):
                flood(40000)
        torch.cuda.synchronize()
        runs[mode] = (losses, _state(step))
    counts = call.node_types
for i, (a, b) in enumerate(zip(runs["eager"][0], runs["graph"][0])):
    assert torch.equal(a, b), (i, a, b)
_same(runs["eager"][1], runs["graph"][1])
assert len(counts) >= 2 and all(set(v) == {"kernel"} for v in counts.values()), counts
print("fast ok", counts)
