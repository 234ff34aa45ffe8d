"""Gather-gradient check (developer tool, GPU box): the balanced (sliced-index) kernel against the per-target walk
(GEOT_GATHER_IMPL=l) and a float64 scatter-add, on the model's shape and on adversarial index sets (hub targets with
hundreds of sources, targets without sources, ragged m / c), plus run-to-run bit reproducibility and timings."""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_case(name, b, c, n, m, idx, w, timing=False):
    from geot_amd.ext import pointnet2_ext as p2
    g = torch.randn(b, c, n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(b * 1000 + c))
    out = p2.three_interpolate_grad(g, idx, w, m)
    again = p2.three_interpolate_grad(g, idx, w, m)
    ref = torch.zeros(b, c, m, dtype=torch.float64, device="cuda")
    ref.scatter_add_(2, idx.long().reshape(b, 1, n * 3).expand(-1, c, -1),
                     (g.double().unsqueeze(-1) * w.double().unsqueeze(1)).reshape(b, c, n * 3))
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    line = "%-34s b=%d c=%d n=%d m=%d  max rel err %.2e  reproducible %s" % (name, b, c, n, m, err, bool(torch.equal(out, again)))
    if timing:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            p2.three_interpolate_grad(g, idx, w, m)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += "  %8.1f us  %.2f TB/s" % (us, (4.0 * b * c * (n + m) + 24.0 * b * n) / us / 1e6)
    print(line, flush=True)
    assert err < 2e-5, name
    return out


def main():
    from geot_amd.synth import make_batch
    from geot_amd.ext import pointnet2_ext as p2
    impl = os.environ.get("GEOT_GATHER_IMPL", "(default)")
    print("GEOT_GATHER_IMPL =", impl)
    outs = {}
    xyz = torch.from_numpy(make_batch(8, 24000)[0]).cuda()
    _, i3 = p2.three_nn(xyz, xyz[:, :8192].contiguous())
    w = torch.rand(8, 24000, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    w = w / w.sum(2, keepdim=True)
    torch.manual_seed(0)
    outs["prop0"] = run_case("prop0 (model shape)", 8, 1536, 24000, 8192, i3, w, timing=True)
    outs["prop0_c384"] = run_case("prop0, reference order", 8, 384, 24000, 8192, i3, w, timing=True)
    outs["ragged"] = run_case("ragged c, m", 2, 70, 24000, 8000, i3[:2].clamp(max=7999).contiguous(), w[:2].contiguous())
    hub = i3[:2].clone()
    hub[:, ::7, 0] = 5                 # ~3400 sources on target 5: far beyond the sliced layout's list cap
    hub[:, 1::50, 1] = 8191
    hub[0, :, 2] = hub[0, :, 2] % 64   # every source of cloud 0 also hits one of 64 targets (375 each)
    outs["hubs"] = run_case("hub targets / empty targets", 2, 64, 24000, 8192, hub.contiguous(), w[:2].contiguous())
    torch.save({k: v[:, ::17, ::5].cpu() for k, v in outs.items()}, os.path.join("/tmp", "gg_check_%s.pt" % impl.strip("()")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "both":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for impl in ("", "l"):
            env = dict(os.environ)
            if impl:
                env["GEOT_GATHER_IMPL"] = impl
            subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
        a = torch.load(os.path.join("/tmp", "gg_check_default.pt"))
        b = torch.load(os.path.join("/tmp", "gg_check_l.pt"))
        for k in a:
            d = float((a[k] - b[k]).abs().max() / b[k].abs().max())
            print("balanced vs per-target walk, %-12s max rel diff %.2e" % (k, d))
            assert d < 2e-6, k
    else:
        main()
