"""TEST INFRASTRUCTURE ONLY.  Builds the one part of the reference's hot-path neighbourhood that compiles
from its own sources with g++ alone: the dataloader's grid_subsampling.

  /root/reference/openpoints/cpp/subsampling/grid_subsampling/grid_subsampling.cpp
  /root/reference/openpoints/cpp/subsampling/cpp_utils/cloud/cloud.cpp
  + oracle/grid_ref_driver.cpp (flat-array harness in place of the CPython/numpy wrapper)
  -> oracle/_ref/libgrid_subsampling_ref.so

The sources are compiled where they lie; nothing is copied into the repo and the output directory is
git-ignored.  (The CUDA extensions are unbuildable here -- no nvcc, no CUDA torch -- see DESIGN.md §4.)
The flags mirror the reference's own setup.py (-std=c++11, default x86-64 code generation: no FMA
contraction), so the float arithmetic is the reference's.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/openpoints/cpp/subsampling"
OUT = os.path.join(HERE, "_ref", "libgrid_subsampling_ref.so")


def build():
    if not os.path.isdir(REF):
        return None
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(REF, "grid_subsampling", "grid_subsampling.cpp"),
            os.path.join(REF, "cpp_utils", "cloud", "cloud.cpp"),
            os.path.join(HERE, "grid_ref_driver.cpp")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-fPIC", "-shared", "-fvisibility=hidden", "-w",
                           "-I" + REF, "-o", OUT] + srcs)
    return OUT


if __name__ == "__main__":
    print(build())
    sys.exit(0)
