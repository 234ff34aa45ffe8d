"""GEMM solution selection for the model's dense layers (stock rocBLAS / hipBLASLt kernels behind torch.bmm / Linear).

The hot path of this package is hand-written HIP; the dense layers of ``PointTransformer_seg_T`` are library GEMMs,
half of the step's kernel time.  PyTorch's TunableOp times every rocBLAS / hipBLASLt solution for a GEMM shape once and
remembers the fastest; ``tunableop_gfx950.csv`` holds that choice for the shapes of BASELINE configs[2] / [4] (8 and
6 + 2 clouds of 24 000 points, in both layouts of the FP stages, and the same batches at the authors' 16 000 points), recorded on MI355X with the ROCm 7.2 image (the file's ``Validator`` rows pin PyTorch,
HIP, rocBLAS, hipBLASLt and ``gfx950``: on any other stack TunableOp ignores the file and the default solutions run).
Same fp32 arithmetic, different tiling: 45.0 -> 41.0 ms per configs[2] step.

"Fastest" is measured alone on the chip -- ``tunableop_gfx950_alone.csv`` -- and for most shapes that is a launch of exactly 256
workgroups, one per CU, which waits for its slowest workgroup: beside the look-ahead's 8192-sample FPS (8 CUs, 4.7 ms) such a
launch takes 50-85 % longer (profiles/r04_fps_beside.txt).  The default file therefore holds, for the 25 shapes of the second half
of the 8-cloud supervised iteration (backward of the transformer blocks and the patch encoder: what the look-ahead runs beside),
the solution that is fastest WITH such an FPS running on a side stream (tools/tune_lookahead_gemms.py; 21 of the 25 differ),
and the `alone` choice for everything else: 32.4 -> 31.7 ms per step with look-ahead (34.8 -> 35.2 without it;
profiles/r04_gemm_selection_beside_fps.txt).  GEOT_TUNABLEOP_FILE=<path> selects another file (A/B runs).

``enable()`` switches TunableOp on with these results and tuning OFF (no timing runs at start-up; unknown shapes use the
library default); ``enable(tune=True, path=...)`` records a new file (minutes)."""
import os

DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable(tune=False, path=None):
    """-> the results file in use, or None when TunableOp is unavailable."""
    import torch
    try:
        import torch.cuda.tunable as tunable
    except ImportError:
        return None
    path = path or os.environ.get("GEOT_TUNABLEOP_FILE") or DEFAULT      # (the env: A/B runs against another selection)
    work = path
    if not tune:
        # TunableOp owns the file it is pointed at (it may rewrite it): every process works on a private copy, so N
        # ranks never write one file and the recorded selection in the tree stays byte-identical.  mkstemp: an
        # unpredictable name created O_EXCL (the old /tmp/geot_tunableop_<pid>.csv could be pre-planted); removed at exit
        import atexit
        import shutil
        import tempfile
        try:
            fd, work = tempfile.mkstemp(prefix="geot_tunableop_", suffix=".csv")
            os.close(fd)
            shutil.copyfile(path, work)
        except OSError:
            return None
        atexit.register(_remove, work)
    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    tunable.set_filename(work, insert_device_ordinal=False)     # one selection for every rank: the shapes are per rank
    if not tune:
        if hasattr(tunable, "write_file_on_exit"):              # (not in every torch build: the private copy absorbs a
            tunable.write_file_on_exit(False)                   #  rewrite at exit and is removed afterwards either way)
        ok = False
        try:
            ok = bool(tunable.read_file(work))                  # False on a Validator mismatch (other torch / ROCm / GPU)
        except Exception:                                       # noqa: BLE001
            ok = False
        if not ok:
            # the recorded selection does not apply to this stack: say so (None) and run plain library defaults,
            # instead of reporting a selection that is not in effect
            tunable.enable(False)
            return None
    return path


def _remove(path):
    try:
        os.remove(path)
    except OSError:
        pass
