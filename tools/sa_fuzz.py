"""Randomised check of the fused SetAbstraction kernel: random shapes (clouds, points, groups a multiple of 8 or
not, 8 / 16 / 32 / 64 samples, 0-8 or many feature channels, 1-4 layers up to 256 wide) against the unfused
composition group -> 1x1 conv stack (BN folded, ReLU) -> max in torch fp32, and the kernel's loop forms
(generic / register-pooled with one group per store / runs of 8 groups) against each other bit for bit.

    python tools/sa_fuzz.py [--cases 40] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.pointnet2 import pytorch_utils as pt_utils  # noqa: E402
from geot_amd.pointnet2 import pointnet2_utils as pu  # noqa: E402
from geot_amd.sa_fused import fused_group_mlp_max, fused_sa_available  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    worst = 0.0
    for case in range(a.cases):
        rng = np.random.default_rng(1000 * a.seed + case)
        torch.manual_seed(1000 * a.seed + case)
        b = int(rng.integers(1, 4))
        n = int(rng.integers(40, 3000))
        npoint = int(rng.choice([1, 7, 8, 64, 120, 333, 512, 1000]))
        ns = int(rng.choice([8, 16, 32, 32, 32, 64]))
        c_feat = int(rng.choice([0, 1, 3, 5, 8, 13, 40]))
        nl = int(rng.integers(1, 5))
        widths = [int(rng.choice([1, 16, 32, 33, 64, 100, 128, 200, 256])) for _ in range(nl)]
        mlp = pt_utils.SharedMLP([3 + c_feat] + widths, bn=True).to(DEV).eval()
        with torch.no_grad():
            for m in mlp.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.uniform_(-0.3, 0.3); m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
        if not fused_sa_available(mlp):       # the weights must fit the LDS next to the activation tiles
            continue
        xyz = torch.rand(b, n, 3, device=DEV)
        new_xyz = xyz[:, torch.randint(0, n, (npoint,), device=DEV)].contiguous()
        feats = torch.randn(b, c_feat, n, device=DEV) if c_feat else None
        idx = torch.randint(0, n, (b, npoint, ns), device=DEV, dtype=torch.int32)
        info = "case %d: b %d n %d npoint %d ns %d c_feat %d widths %s" % (case, b, n, npoint, ns, c_feat, widths)
        outs = {}
        with torch.no_grad():
            for name, env in (("generic", {"GEOT_SA_FAST": "0"}), ("run1", {"GEOT_SA_RUN": "1"}), ("run8", {"GEOT_SA_RUN": "8"}),
                              ("auto", {})):
                for k in ("GEOT_SA_FAST", "GEOT_SA_RUN"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                outs[name] = fused_group_mlp_max(xyz, new_xyz, feats, idx, mlp)
            for k in ("GEOT_SA_FAST", "GEOT_SA_RUN"):
                os.environ.pop(k, None)
            g_xyz = pu.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
            grouped = torch.cat([g_xyz, pu.grouping_operation(feats, idx)], 1) if c_feat else g_xyz
            ref = mlp(grouped).max(dim=3)[0]
        for name in ("run1", "run8", "auto"):
            if not torch.equal(outs["generic"], outs[name]):
                print("MISMATCH between loop forms generic /", name, info)
                sys.exit(1)
        err = float((outs["auto"] - ref).abs().max() / (ref.abs().max() + 1e-6))
        worst = max(worst, err)
        if err > 2e-5:   # worth a look: is it conditioning (fp64 agrees with neither more than with the other)?
            ref64 = mlp.double()(grouped.double()).max(dim=3)[0]
            mlp.float()
            e_fused = float((outs["auto"].double() - ref64).abs().max() / (ref64.abs().max() + 1e-6))
            e_torch = float((ref.double() - ref64).abs().max() / (ref64.abs().max() + 1e-6))
            print("  note: rel err %.2g vs torch fp32; vs fp64: fused %.2g, torch fp32 %.2g  (%s)" % (err, e_fused, e_torch, info), flush=True)
        if err > 2e-4:
            print("MISMATCH vs the unfused composition: rel err %.3g" % err, info)
            sys.exit(1)
        if case % 10 == 9:
            print("case %d ok (worst rel err so far %.2g)" % (case, worst), flush=True)
    print("sa_fuzz: %d cases, loop forms bit-identical, worst relative error vs the unfused composition %.2g" % (a.cases, worst))


if __name__ == "__main__":
    main()
