"""sig_t_mean at run-time class counts (csrc/ntm_generic.hip): the MFMA form against the lane-per-point form
(GEOT_NTM_GENERIC=rows), forward and d raw, 8 x 24000 points; bytes = 4 B N (C + C^2) forward, 4 B N (C + 2 C^2) for d raw.
Developer tool, GPU box:  python tools/lab/ntm_generic_time.py [C ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def timed(fn, it=20):
    for _ in range(3):
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
    from geot_amd import ntm
    dev, B, N = "cuda", 8, 24000
    counts = [int(a) for a in sys.argv[1:]] or [17, 5, 8, 12, 16, 20, 21, 32]
    for c in counts:
        torch.manual_seed(c)
        p = torch.softmax(torch.randn(B, c, N, device=dev), 1)
        cm = torch.softmax(torch.randn(c, c, device=dev), 1)
        mod = ntm.sig_t_mean(c).to(dev)
        W = torch.stack([l.weight for l in mod.fc]).detach().contiguous()
        g = torch.randn(B * N, c, c, device=dev)
        fb, bb = 4.0 * B * N * (c + c * c), 4.0 * B * N * (c + 2 * c * c)
        row = "C = %2d" % c
        outs = {}
        for impl in (("mfma", "rows") if c != 17 else ("specialised",)):
            if impl != "specialised":
                os.environ["GEOT_NTM_GENERIC"] = impl
            with torch.no_grad():
                uf = timed(lambda: mod(p, cm))
                ub = timed(lambda: ntm.sig_t_mean_grad_raw(p, cm, W, g))
                outs[impl] = (mod(p, cm), ntm.sig_t_mean_grad_raw(p, cm, W, g))
            os.environ.pop("GEOT_NTM_GENERIC", None)
            row += "   %s: fwd %7.1f us %5.2f TB/s, d raw %7.1f us %5.2f TB/s" % (impl, uf, fb / uf / 1e6, ub, bb / ub / 1e6)
        if "rows" in outs:
            a, b = outs["mfma"], outs["rows"]
            row += "   max |mfma - rows| fwd %.2e d raw %.2e (rel to max |d raw|)" % (
                float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max() / b[1].abs().max()))
        print(row, flush=True)


if __name__ == "__main__":
    main()
