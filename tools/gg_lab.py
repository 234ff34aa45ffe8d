"""(Builds the LAB COPY tools/lab/kernels/gather_group.hip: the product source carries no knock-out switches since round 5.)
Gather-gradient lab (developer tool): builds gather_group.hip variants with -DGEOT_GG_LAB_* switches that remove
one phase of the reverse-index gather at a time and times the prop0 interpolation gradient
(B x 1536 x 24000 -> 8192).  Results of the variants are garbage by construction; only the timing matters.

    python tools/gg_lab.py build     # here
    python tools/gg_lab.py run       # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "_lab")
VARIANTS = {
    "base": [],
    "nostage": ["-DGEOT_GG_LAB_NOSTAGE"],
    "nowalk": ["-DGEOT_GG_LAB_NOWALK"],
    "noidx": ["-DGEOT_GG_LAB_NOIDX"],
    "noldsread": ["-DGEOT_GG_LAB_NOLDSREAD"],
    "walk_noidx_nolds": ["-DGEOT_GG_LAB_NOSTAGE", "-DGEOT_GG_LAB_NOIDX", "-DGEOT_GG_LAB_NOLDSREAD"],
    "sell_nostage": ["-DGEOT_SELL_LAB_NOSTAGE"],
    "sell_nowalk": ["-DGEOT_SELL_LAB_NOWALK"],
    "sell_neither": ["-DGEOT_SELL_LAB_NOSTAGE", "-DGEOT_SELL_LAB_NOWALK"],
}
ONLY = os.environ.get("VARIANTS")
if ONLY:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in ONLY.split(",")}
for extra in sys.argv[2:]:
    if "=" in extra:
        k, v = extra.split("=", 1)
        VARIANTS[k] = v.split(",") if v else []


def build():
    os.makedirs(LAB, exist_ok=True)
    for name, flags in VARIANTS.items():
        out = os.path.join(LAB, "libgg_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "geot_amd", "csrc")] + flags + \
              [os.path.join(ROOT, "tools", "lab", "kernels", "gather_group.hip"), "-o", out]
        subprocess.check_call(cmd)
        print("built", out)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from geot_amd.synth import make_batch
    from geot_amd.ext import pointnet2_ext as p2
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    B, C, N, M = int(os.environ.get("B", "8")), int(os.environ.get("C", "1536")), 24000, 8192
    xyz = torch.from_numpy(make_batch(B, N)[0]).cuda()
    known = xyz[:, :M].contiguous()
    _, i3 = p2.three_nn(xyz, known)
    w = torch.rand(B, N, 3, device="cuda")
    w = w / w.sum(2, keepdim=True)
    go = torch.randn(B, C, N, device="cuda")
    out = torch.empty(B, C, M, device="cuda")
    ws = torch.empty(B * C * M, device="cuda")
    nbytes = 4 * B * (C * N + C * M) + 24 * B * N
    for name in VARIANTS:
        path = os.path.join(LAB, "libgg_%s.so" % name)
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        f = lib.geot_three_interpolate_grad_out
        f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 6

        def call():
            err = f(B, C, N, M, go.data_ptr(), i3.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
            assert err == 0, err
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print("%-18s B=%d C=%d  %8.1f us  %6.2f TB/s (of the full kernel's bytes)" % (name, B, C, us, nbytes / us / 1e6), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
