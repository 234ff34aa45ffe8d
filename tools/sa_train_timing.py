"""configs[1]'s SetAbstraction module in TRAINING mode (fwd + bwd, batch-statistics BatchNorm), factored vs composed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geot_amd.synth import make_batch
from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
B = int(os.environ.get("B", "8"))
xyz = torch.from_numpy(make_batch(B, 24000)[0]).cuda()
feats = torch.randn(B, 3, 24000, device="cuda", requires_grad=True)
for factored in (False, True):
    torch.manual_seed(0)
    sa = PointnetSAModuleVotes(mlp=[3, 64, 64, 128], npoint=6000, radius=0.1, nsample=32).cuda().train()
    sa.factored_train = factored
    for _ in range(3):
        sa(xyz, feats)[1].sum().backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sa(xyz, feats)[1].sum().backward()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("SA training fwd+bwd, %d clouds, %s: %.2f ms/step = %.0f clouds/s" % (B, "factored" if factored else "composed", ms, B / ms * 1e3))
