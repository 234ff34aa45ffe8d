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

SOURCES = ["fps.hip", "neighbors.hip", "knn_grid.hip", "gather_group.hip", "ntm.hip", "sa_mlp.hip", "dataprep.hip", "edgeconv.hip"]
HEADERS = ["geot_common.h", os.path.join(ROOT, "include", "geot_hip.h")]

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


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
