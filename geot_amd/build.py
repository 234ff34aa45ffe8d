"""Build recipe for libgeot_hip.so (hipcc, gfx950 only, in-tree).

``python -m geot_amd.build`` or ``geot_amd.build.build()``.  hipcc cross-compiles
without a GPU, so this runs in the build container; the resulting
``geot_amd/libgeot_hip.so`` travels to the GPU box with the repo snapshot.
No torch headers are involved: the library is a plain C-ABI (include/geot_hip.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgeot_hip.so")
# squared-distance arithmetic variants (csrc/geot_common.h GEOT_DISTANCE_MODE): name -> (macro value, library)
VARIANTS = {"exact": (0, LIB), "fma": (1, os.path.join(HERE, "libgeot_hip_fma.so")),
            "fma_xy": (2, os.path.join(HERE, "libgeot_hip_fma_xy.so"))}

SOURCES = ["fps.hip", "neighbors.hip", "knn_grid.hip", "gather_group.hip", "tile_scatter.hip", "ntm.hip", "ntm_generic.hip", "sa_mlp.hip", "dataprep.hip", "edgeconv.hip", "bnrelu.hip", "channels_last.hip", "loss.hip", "layernorm.hip"]
HEADERS = ["geot_common.h", "ntm_generic.h", "tile_scatter.h", os.path.join(ROOT, "include", "geot_hip.h")]

# -ffp-contract=off: squared distances must be un-contracted IEEE fp32 so that
# integer outputs match the CPU oracle bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant="exact"):
    """Compile the library (mtime-incremental).  variant: "exact" (default) | "fma" | "fma_xy" -- see VARIANTS."""
    mode, lib_path = VARIANTS[variant]
    obj_dir = OBJ if variant == "exact" else os.path.join(OBJ, variant)
    flags = FLAGS + ["-DGEOT_DISTANCE_MODE=%d" % mode] + os.environ.get("GEOT_EXTRA_HIPCC_FLAGS", "").split()   # (lab sweeps)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    # lab sweeps (tools/lab/*.sh): GEOT_LAB_KERNELS=tools/lab/kernels takes a file from there when it exists -- the copies that
    # still carry the knock-out / tuning switches the product sources no longer have
    lab_dir = os.environ.get("GEOT_LAB_KERNELS")
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        if lab_dir and os.path.exists(os.path.join(ROOT, lab_dir, s)):
            src = os.path.join(ROOT, lab_dir, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(lib_path, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


def build_all(force=False, verbose=False, variants=("exact", "fma")):
    """The default library and the contracted-distance build the parity tests exercise ("fma_xy" on request)."""
    return [build(force, verbose, v) for v in variants]


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in VARIANTS] or ["exact"]
    for v in which:
        print(build(force="--force" in sys.argv, verbose=True, variant=v))
