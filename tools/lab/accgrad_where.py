"""Where torch's 'AccumulateGrad node's stream does not match' warning is raised in the bench's sequence: eager steps on the
default stream, then the replay's warm-up / capture on its own stream (warnings turned into errors for the traceback)."""
import os
import sys
import traceback
import warnings

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
    B, N = int(os.environ.get("CLOUDS", "2")), 8192
    batches = []
    for s in (0, 7):
        x_np, _ = make_batch(B, N, start_index=s)
        batches.append((torch.from_numpy(x_np).to(dev), torch.zeros(B, 1, dtype=torch.long, device=dev), torch.from_numpy(region_labels(x_np)).to(dev)))
    warnings.simplefilter(os.environ.get("WARN", "error"))
    try:
        for i in range(3):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            trainer(cur[0], cur[1], cur[2], next_pos=nxt[0])
        print("eager steps: no warning")
        graphed = gs.GraphedSupervisedStep(trainer)
        for i in range(graphed.warmup + 3):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            graphed(cur[0], cur[1], cur[2], next_pos=nxt[0])
            print("graphed call", i, "captured" if graphed.captured else "warm-up", ": no warning")
            if os.environ.get("SCAN") == "1":
                import gc
                gc.collect()
                alive = []
                for obj in gc.get_objects():
                    try:
                        if isinstance(obj, torch.Tensor) and obj.grad_fn is not None:
                            alive.append((tuple(obj.shape), type(obj.grad_fn).__name__))
                    except Exception:  # noqa: BLE001
                        pass
                print("   tensors with a grad_fn alive:", len(alive), alive[:12])
    except Warning:
        traceback.print_exc()


if __name__ == "__main__":
    main()
