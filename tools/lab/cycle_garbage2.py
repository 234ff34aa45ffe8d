"""The crash scenario of profiles/r04_graph_capture_notes.txt (7b) up to the capture: three eager steps with look-ahead, the
wrapper, two warm-up calls -- then what the cycle collector finds (collector disabled from the start)."""
import collections
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import geot_amd  # noqa: E402,F401
from geot_amd.synth import make_batch, region_labels  # noqa: E402


def main():
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
    from geot_amd import train_step as ts, graph_step as gs
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(dev)
    trainer = ts.SupervisedStep(model)
    B, N = 2, 8192
    batches = []
    for s in (0, 7):
        x_np, _ = make_batch(B, N, start_index=s)
        batches.append((torch.from_numpy(x_np).to(dev), torch.zeros(B, 1, dtype=torch.long, device=dev), torch.from_numpy(region_labels(x_np)).to(dev)))
    gc.collect()
    gc.disable()
    stage = os.environ.get("STAGE", "warm")
    for i in range(3):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        trainer(cur[0], cur[1], cur[2], next_pos=nxt[0])
    if stage != "eager":
        graphed = gs.GraphedSupervisedStep(trainer, warmup=5)
        for i in range(2):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            graphed(cur[0], cur[1], cur[2], next_pos=nxt[0])
    torch.cuda.synchronize()
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    kinds = collections.Counter(type(o).__module__ + "." + type(o).__qualname__ for o in gc.garbage)
    print(stage, ": unreachable objects:", n)
    for k, v in kinds.most_common(20):
        print("   %6d  %s" % (v, k))
    for o in gc.garbage:
        if callable(o) and hasattr(o, "__code__"):
            print("   function:", o.__code__.co_filename.split("/")[-1], o.__code__.co_firstlineno, o.__qualname__)
    tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
    print("   tensors in garbage:", len(tens), [(tuple(t.shape), t.grad_fn is not None) for t in tens[:10]])


if __name__ == "__main__":
    main()
