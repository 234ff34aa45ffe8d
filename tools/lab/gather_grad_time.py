"""Lab: time three_interpolate_grad at given sizes (B C N M env) -- used with -D knock-out builds on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd.synth import make_batch
from geot_amd.ext import pointnet2_ext as p2
B, C, N, M = (int(os.environ.get(k, d)) for k, d in (("B", "8"), ("C", "1536"), ("N", "8192"), ("M", "4096")))
xyz = torch.from_numpy(make_batch(B, 24000)[0]).cuda()
if os.environ.get("FPS", "0") == "1":          # the model's layout: the known points are an FPS prefix of the unknown ones
    sel = p2.furthest_point_sampling(xyz, N).long()
    pts = torch.gather(xyz, 1, sel.unsqueeze(-1).expand(-1, -1, 3))
    unknown, known = pts.contiguous(), pts[:, :M].contiguous()
else:
    unknown, known = xyz[:, :N].contiguous(), xyz[:, :M].contiguous()
_, i3 = p2.three_nn(unknown, known)
w = torch.rand(B, N, 3, device="cuda"); w = w / w.sum(2, keepdim=True)
g = torch.randn(B, C, N, device="cuda")
for _ in range(3):
    p2.three_interpolate_grad(g, i3, w, M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    p2.three_interpolate_grad(g, i3, w, M)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
nb = 4 * B * C * (N + M)
cnt = torch.bincount((i3.long() + (torch.arange(B, device="cuda") * M).view(B, 1, 1)).reshape(-1), minlength=B * M).view(B, M)
wave_max = cnt.view(B, -1, 64).max(-1)[0].float()
print("B=%d C=%d N=%d M=%d FPS=%s: %.1f us  %.2f TB/s; list length mean %.1f max %d, mean over waves of the wave's max %.1f"
      % (B, C, N, M, os.environ.get("FPS", "0"), us, nb / us / 1e6, cnt.float().mean(), int(cnt.max()), wave_max.mean()))
