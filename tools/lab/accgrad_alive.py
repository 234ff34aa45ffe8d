"""Which objects keep AccumulateGrad nodes alive after a SupervisedStep iteration has returned (the source of torch's
'AccumulateGrad node's stream does not match' warning when the next iteration runs on another stream)."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import geot_amd  # noqa: E402,F401
from geot_amd.synth import make_batch, region_labels  # noqa: E402


def main():
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
    from geot_amd import train_step as ts
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = PointTransformer_seg_T(**TOOTH_SEG_CFG).to(dev)
    trainer = ts.SupervisedStep(model)
    B, N = 2, 8192
    xyz_np, _ = make_batch(B, N)
    xyz = torch.from_numpy(xyz_np).to(dev)
    tgt = torch.from_numpy(region_labels(xyz_np)).to(dev)
    cls = torch.zeros(B, 1, dtype=torch.long, device=dev)
    xyz2 = torch.from_numpy(make_batch(B, N, start_index=7)[0]).to(dev)
    for variant in ("plain", "lookahead"):
        loss = trainer(xyz, cls, tgt, next_pos=xyz2 if variant == "lookahead" else None)
        del loss
        torch.cuda.synchronize()
        gc.collect()
        with_accum = [p for p in model.parameters() if p.requires_grad]
        # a live AccumulateGrad node shows up as a graph node reachable from some tensor's grad_fn; scan python-visible tensors
        holders = []
        for obj in gc.get_objects():
            try:
                if isinstance(obj, torch.Tensor) and obj.grad_fn is not None:
                    holders.append((type(obj).__name__, tuple(obj.shape), type(obj.grad_fn).__name__))
            except Exception:  # noqa: BLE001
                pass
        print(variant, "tensors with a grad_fn alive after the step:", len(holders))
        for h in holders[:40]:
            print("   ", h)
        trainer._geometry = None


if __name__ == "__main__":
    main()
