"""The contracted-distance build (GEOT_DISTANCE=fma: fma(dz,dz,fma(dy,dy,dx*dx)), the arithmetic nvcc -fmad=true most
likely gave the authors' binaries -- SURVEY.md App. A) against its oracle twin, bit for bit.  The
library is chosen once per process, so the checks run in ONE child process with the switch set."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_contracted_distance_build_matches_its_oracle_twin():
    from geot_amd import build as hip_build
    hip_build.build(variant="fma")
    env = dict(os.environ, GEOT_DISTANCE="fma")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_contracted_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "contracted parity ok: fma" in r.stdout


def test_distance_variants_build_and_export_their_mode():
    """CPU: both libraries load, export every declared symbol and report their arithmetic."""
    import ctypes
    from geot_amd import build as hip_build, _lib
    for variant, (mode, _) in hip_build.VARIANTS.items():
        if variant == "fma_xy":
            continue                      # built on request only
        path = hip_build.build(variant=variant)
        lib = ctypes.CDLL(path)
        assert lib.geot_distance_mode() == mode and lib.geot_abi_version() == _lib.ABI_VERSION
        for sym in _lib.exported_symbols():
            assert hasattr(lib, sym), (variant, sym)
    r = subprocess.run([sys.executable, "-c", "import geot_amd._lib"], env=dict(os.environ, GEOT_DISTANCE="bogus"),
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "GEOT_DISTANCE" in r.stderr


def test_oracle_twin_differs_only_in_rounding():
    """CPU: the fma oracle and the exact oracle agree on indices for well-separated points and differ by <= 1 ulp-ish
    in the squared distances somewhere."""
    import numpy as np
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from oracle import capi; from geot_amd.synth import make_batch;"
            "x = make_batch(1, 2048, start_index=3)[0]; d, i = capi.three_nn(x, x[:, :512].copy());"
            "np.save(sys.argv[1], d); np.save(sys.argv[2], i)" % ROOT)
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("exact", "fma"):
            a, b = os.path.join(tmp, mode + "_d.npy"), os.path.join(tmp, mode + "_i.npy")
            subprocess.run([sys.executable, "-c", code, a, b], env=dict(os.environ, GEOT_DISTANCE=mode), check=True, cwd=ROOT)
            out[mode] = (np.load(a), np.load(b))
    d0, i0 = out["exact"]
    d1, i1 = out["fma"]
    assert (i0 == i1).mean() > 0.999
    assert (d0 != d1).any() and np.abs(d0 - d1).max() <= 4e-7 * max(d0.max(), 1.0)
