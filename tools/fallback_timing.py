"""What the fall-back paths behind the shape walls cost (developer tool, GPU box): each accelerated kernel has a size
range; outside it a plainer kernel or the composed ops run.  Times both sides of every wall with HIP events.

    FPS            n <= 24 576: bucket-pruned, register-resident      | n > 24 576: unpruned / streaming kernel
    EdgeConv tail  k <= 255, Nq <= 17 066 (geot_edgeconv_eligible)    | composed: grouping + GroupNorm + LeakyReLU + max
    fused SA body  nsample 8 / 16 / multiples of 32, eval mode        | composed: QueryAndGroup + SharedMLP + max_pool
    NTM kernels    C = 17: MFMA / LDS-tiled                           | other C: run-time-C kernels (ntm_generic.hip)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch  # noqa: E402


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd import ntm
    dev = "cuda"
    for n in (24576, 24577, 32768):
        x = torch.from_numpy(make_batch(1, n)[0]).to(dev)
        us = timed(lambda: p2.furthest_point_sampling(x, 4096), 3)
        print("FPS %6d -> 4096                         %9.1f us  %6.3f us per sample" % (n, us, us / 4096), flush=True)
    from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    x = torch.from_numpy(make_batch(1, 24000)[0]).to(dev)
    f = torch.randn(1, 3, 24000, device=dev)
    for ns in (32, 24):
        sa = PointnetSAModuleVotes(mlp=[3, 64, 64, 128], npoint=6000, radius=0.1, nsample=ns, use_xyz=True).to(dev).eval()
        inds = p2.furthest_point_sampling(x, 6000)
        with torch.no_grad():
            us = timed(lambda: sa(x, f, inds))
        print("SA body 6000 x nsample %2d (%s)      %9.1f us" % (ns, "fused MFMA kernel" if ns == 32 else "composed fallback ", us), flush=True)
    from geot_amd.openpoints.models.backbone.transformer import DGCNN_Propagation
    for k, label in ((4, "fused tail"), (4, "composed (GEOT_EDGE_TAIL=torch)")):
        if "composed" in label:
            os.environ["GEOT_EDGE_TAIL"] = "torch"
        mod = DGCNN_Propagation(k=k).to(dev)
        os.environ.pop("GEOT_EDGE_TAIL", None)
        coor_q = x[:, :8192].transpose(1, 2).contiguous()
        coor = x[:, :4096].transpose(1, 2).contiguous()
        fq = torch.randn(1, 384, 8192, device=dev, requires_grad=True)
        fk = torch.randn(1, 384, 4096, device=dev, requires_grad=True)

        def step():
            y = mod(coor, fk, coor_q, fq)
            y.sum().backward()
        us = timed(step, 5)
        print("DGCNN propagation 4096 -> 8192, fwd+bwd, %-32s %9.1f us" % (label, us), flush=True)
    for c in (17, 16, 20):
        p = torch.softmax(torch.randn(8, c, 24000, device=dev), 1)
        cm = torch.softmax(torch.randn(c, c, device=dev), 1)
        mod = ntm.sig_t_mean(c).to(dev)
        with torch.no_grad():
            us = timed(lambda: mod(p, cm))
        nbytes = 4.0 * 8 * 24000 * (c + c * c)
        print("sig_t_mean fwd, 8 x 24000 points, C = %2d (%s)   %9.1f us  %5.2f TB/s" %
              (c, "specialised" if c == 17 else "run-time C ", us, nbytes / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
