import os, sys, time
import torch, torch.nn.functional as F
dev = "cuda"
B, H, N, D = 8, 4, 512, 96
q, k, v = (torch.randn(B, H, N, D, device=dev, requires_grad=True) for _ in range(3))
scale = D ** -0.5
def manual():
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    return attn @ v
def sdpa():
    return F.scaled_dot_product_attention(q, k, v)
for name, fn in (("manual", manual), ("sdpa", sdpa)):
    try:
        for _ in range(5):
            o = fn(); o.sum().backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            o = fn(); o.sum().backward()
        e1.record(); torch.cuda.synchronize()
        print(name, "fwd+bwd %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
    except Exception as e:
        print(name, "FAILED", repr(e)[:300])
a, b = manual(), sdpa()
print("max diff", float((a - b).abs().max()), float(a.abs().max()))
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    try:
        with sdpa_kernel(be):
            o = sdpa(); o.sum().backward()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                o = sdpa(); o.sum().backward()
            e1.record(); torch.cuda.synchronize()
        print(be, "ok %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3), float((o - a).abs().max()))
    except Exception as e:
        print(be, "unavailable:", repr(e)[:200])
