"""(Builds the LAB COPY tools/lab/kernels/sa_mlp.hip: the product source carries no knock-out switches since round 5.)
SA-body kernel lab (developer tool): builds sa_mlp.hip variants with -DGEOT_SA_LAB_* switches that remove one
phase at a time (gather / epilogue stores / MFMA k-loop) and times them, to see where the non-MFMA half of the
launch goes.  Results of the variants are garbage by construction; only the timing matters.

    python tools/sa_lab.py build     # here
    python tools/sa_lab.py run       # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "_lab")
VARIANTS = {
    "base": [],
    "nogather": ["-DGEOT_SA_LAB_NOGATHER"],
    "nostore": ["-DGEOT_SA_LAB_NOSTORE"],
    "nopool": ["-DGEOT_SA_LAB_NOPOOL"],
    "noout": ["-DGEOT_SA_LAB_NOOUT"],
    "nomfma": ["-DGEOT_SA_LAB_NOMFMA"],
    "mfma_only": ["-DGEOT_SA_LAB_NOGATHER", "-DGEOT_SA_LAB_NOSTORE", "-DGEOT_SA_LAB_NOPOOL", "-DGEOT_SA_LAB_NOOUT"],
}
for extra in sys.argv[2:]:
    if "=" in extra:
        k, v = extra.split("=", 1)
        VARIANTS[k] = v.split(",")


def build():
    os.makedirs(LAB, exist_ok=True)
    for name, flags in VARIANTS.items():
        out = os.path.join(LAB, "libsa_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "geot_amd", "csrc")] + flags + \
              [os.path.join(ROOT, "tools", "lab", "kernels", "sa_mlp.hip"), "-o", out]
        subprocess.check_call(cmd)
        print("built", out)


def run():
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    B = int(os.environ.get("B", "16"))
    n, npoint, ns, cf = 24000, 6000, 32, 3
    widths = (ctypes.c_int * 3)(64, 64, 128)
    g = torch.Generator(device="cuda").manual_seed(0)
    xyz = torch.rand(B, n, 3, device="cuda", generator=g)
    new_xyz = xyz[:, :npoint].contiguous()
    feats = torch.randn(B, cf, n, device="cuda", generator=g)
    idx = torch.randint(0, n, (B, npoint, ns), device="cuda", generator=g, dtype=torch.int32)
    out = torch.empty(B, 128, npoint, device="cuda")
    flop = 2.0 * B * npoint * ns * (6 * 64 + 64 * 64 + 64 * 128)
    for name in VARIANTS:
        path = os.path.join(LAB, "libsa_%s.so" % name)
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        lib.geot_sa_param_floats.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        nparam = lib.geot_sa_param_floats(cf, 3, widths)
        params = torch.randn(nparam, device="cuda", generator=g) * 0.1
        f = lib.geot_sa_group_mlp_max
        f.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

        def go():
            err = f(B, n, npoint, ns, cf, xyz.data_ptr(), new_xyz.data_ptr(), feats.data_ptr(), idx.data_ptr(), 1.0, 3, widths,
                    7, params.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert err == 0, err
        go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-12s B=%d  %8.3f ms  %6.1f TFLOP/s (of the full kernel's flops)  %5.1f %% of 157.3" %
              (name, B, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
