"""Hook for vectors produced by the reference's own CUDA build (tools/emit_reference_vectors.py ->
tests/golden/external/*.npz).  None can be produced in the build container (no nvcc / NVIDIA device), so these
tests skip until a maintainer drops a file in; from then on the oracle -- in whichever squared-distance arithmetic
reproduces the file: exact | fma | fma_xy -- and the HIP library of that arithmetic are held to it bit for bit."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "external", "*.npz")))
MODES = ("exact", "fma", "fma_xy")


def _run(path, mode, gpu):
    if mode != "exact":
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libgeot_oracle_%s.so" % mode])
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_external_check.py"), path] + (["--gpu"] if gpu else [])
    return subprocess.run(cmd, env=dict(os.environ, GEOT_DISTANCE=mode), capture_output=True, text=True, timeout=1200)


def matching_mode(path):
    for mode in MODES:
        if _run(path, mode, False).returncode == 0:
            return mode
    return None


@pytest.mark.skipif(not FILES, reason="no reference-CUDA vectors under tests/golden/external/ (see its README)")
@pytest.mark.parametrize("path", FILES)
def test_oracle_reproduces_external_reference_vectors(path):
    mode = matching_mode(path)
    assert mode is not None, "no squared-distance arithmetic of the oracle reproduces %s" % path
    print("reference CUDA vectors reproduced with GEOT_DISTANCE=%s" % mode)


@pytest.mark.gpu
@pytest.mark.skipif(not FILES, reason="no reference-CUDA vectors under tests/golden/external/ (see its README)")
@pytest.mark.parametrize("path", FILES)
def test_hip_library_reproduces_external_reference_vectors(path):
    from geot_amd import build as hip_build
    mode = matching_mode(path)
    assert mode is not None
    hip_build.build(variant=mode)
    r = _run(path, mode, True)
    assert r.returncode == 0 and "external ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_the_hook_itself_on_vectors_made_by_the_oracle(tmp_path, oracle):
    """The consumer is exercised end to end with a stand-in file written by the oracle in fma mode: the selector
    must come back with a mode that reproduces it (exact may tie on easy clouds; a mismatching file must fail)."""
    import numpy as np
    from geot_amd.synth import make_batch
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); from oracle import capi; from geot_amd.synth import make_batch\n"
        "b, n, m = 1, 3000, 700; xyz = make_batch(b, n, start_index=500, dup_frac=0.01)[0]; out = {'cases': np.array(['t']), 't_xyz': xyz, 't_m': np.int32(m)}\n"
        "k1 = capi.fps_dense(xyz, m, 512, True); out['t_fps_k1'] = k1; out['t_fps_k1p'] = capi.fps_dense(xyz, m, 1024, False)\n"
        "off, noff = np.array([n], np.int32), np.array([m], np.int32)\n"
        "out['t_fps_k2_xyz'] = xyz.reshape(-1, 3)[capi.fps_offset(xyz.reshape(-1, 3), off, noff)].reshape(b, m, 3)\n"
        "c = np.take_along_axis(xyz, k1[..., None].astype(np.int64).repeat(3, -1), 1)\n"
        "out['t_ball_r0.1_ns32'] = capi.ball_query(c, xyz, 0.1, 32); out['t_three_nn_idx'] = capi.three_nn(xyz, c)[1]\n"
        "out['t_knn5_idx'] = capi.knnquery_heap(5, xyz.reshape(-1, 3), c.reshape(-1, 3), off, noff)[0].reshape(b, m, 5)\n"
        "np.savez(sys.argv[1], **out)\n" % ROOT)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libgeot_oracle_fma.so"])
    path = str(tmp_path / "standin.npz")
    subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, GEOT_DISTANCE="fma"), check=True, cwd=ROOT)
    assert _run(path, "fma", False).returncode == 0
    assert matching_mode(path) in ("exact", "fma")
    g = dict(np.load(path))
    g["t_fps_k1"] = g["t_fps_k1"][:, ::-1].copy()         # a file no arithmetic reproduces
    bad = str(tmp_path / "bad.npz")
    np.savez(bad, **g)
    assert matching_mode(bad) is None
