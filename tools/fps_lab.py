"""(Builds the LAB COPY tools/lab/kernels/fps.hip: the product source carries no knock-out switches since round 5.)
FPS kernel lab (developer tool): builds fps.hip variants with -DGEOT_LAB_* switches,
times them on the GPU and prints pruning statistics.  Not part of the product.

    python tools/fps_lab.py build     # here (no GPU needed)
    python tools/fps_lab.py run       # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "_lab")
VARIANTS = {
    "base": [],
    "stats": ["-DGEOT_LAB_STATS"],
    "stamps": ["-DGEOT_LAB_STAMPS"],
    "t6": ["-DGEOT_FP_TMAX=6"],
}
for extra in sys.argv[2:]:
    if "=" in extra:
        k, v = extra.split("=", 1)
        VARIANTS[k] = v.split(",")


def build():
    os.makedirs(LAB, exist_ok=True)
    for name, flags in VARIANTS.items():
        out = os.path.join(LAB, "libfps_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "geot_amd", "csrc")] + flags + \
              [os.path.join(ROOT, "tools", "lab", "kernels", "fps.hip"), "-o", out]
        subprocess.check_call(cmd)
        print("built", out)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from geot_amd.synth import make_batch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    NPTS = int(os.environ.get("NPTS", "24000"))
    xyz_np, _ = make_batch(1, NPTS)
    xyz = torch.from_numpy(xyz_np).cuda()
    print("N =", NPTS)
    for name in VARIANTS:
        path = os.path.join(LAB, "libfps_%s.so" % name)
        if not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        f = lib.geot_furthest_point_sampling
        f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
        for m in (512, 2048, 8192):
            out = torch.zeros(1, m, dtype=torch.int32, device="cuda")

            def go():
                tmp = torch.full((1, NPTS), 1e10, device="cuda")
                err = f(1, NPTS, m, xyz.data_ptr(), tmp.data_ptr(), out.data_ptr(), 1024, 0,
                        torch.cuda.current_stream().cuda_stream)
                assert err == 0, err
            go()
            torch.cuda.synchronize()
            if name == "stamps":
                st = (ctypes.c_ulonglong * 8)()
                lib.geot_lab_read_stats(st, 1)
                go()
                torch.cuda.synchronize()
                lib.geot_lab_read_stats(st, 1)
                nw = st[5] / (m - 1)          # waves
                rounds = max(st[6], 1)
                wr = nw * rounds              # wave-rounds
                print("  stamps m=%d: %.2f samples/round; cycles per wave-round: apply %.0f  search %.0f  "
                      "publish+barrier %.0f  resolve %.0f  total %.0f  => %.0f cycles/sample"
                      % (m, (m - 1) / rounds, st[0] / wr, st[1] / wr, st[2] / wr, st[3] / wr, sum(st[:4]) / wr,
                         sum(st[:4]) / wr * rounds / (m - 1)), flush=True)
                nred, wred = st[7] & 0xFFFFFF, st[7] >> 24
                print("      slowest wave's apply+search per round %.0f cycles; waves that re-searched: %.2f per round, "
                      "their apply+search %.0f cycles" % (st[4] / rounds, nred / rounds, wred / max(nred, 1)), flush=True)
                continue
            if name == "stats":
                st = (ctypes.c_ulonglong * 8)()
                lib.geot_lab_read_stats(st, 1)
                go()
                torch.cuda.synchronize()
                lib.geot_lab_read_stats(st, 1)
                waves_rounds = st[1]
                print("  stats m=%d: active slots per wave-test %.3f; samples committed per round %.2f; "
                      "prefix cut by (A) distance %.2f / (B) runner-up bound %.2f of rounds"
                      % (m, st[0] / waves_rounds, st[2] / max(st[3], 1), st[4] / max(st[3], 1), st[5] / max(st[3], 1)),
                      flush=True)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                go()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print("%-14s m=%5d  %8.3f ms  %6.3f us/round" % (name, m, ms, 1e3 * ms / (m - 1)), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
