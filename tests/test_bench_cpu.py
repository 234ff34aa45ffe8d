"""CPU tests of the measurement plumbing: the N-rank self-launcher of bench.py (no GPU needed: the children are a
stub), the steady-state window tool, the PMC-traffic lookup and the GEMM-selection switch."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("geot_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_imports_and_parses_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--workload" in r.stdout


def test_spawn_ranks_sets_the_rendezvous_and_reports_the_worst_exit(tmp_path, monkeypatch):
    """`python bench.py --gpus N` without a launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    started by a parent that has not touched the GPU; a failing rank takes the others down and its code is returned."""
    bench = _load_bench()
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent("""
        import json, os, sys, time
        rank = int(os.environ["RANK"])
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
        open(os.path.join(os.path.dirname(__file__), "rank%d.json" % rank), "w").write(json.dumps(rec))
        if "--fail" in sys.argv and rank == 1:
            sys.exit(7)
        if "--fail" in sys.argv:
            time.sleep(60)          # would hang in a collective: the launcher must terminate it
    """))
    monkeypatch.setattr(bench, "__file__", str(stub))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])

    class A:
        gpus = 3
    assert bench.spawn_ranks(A) == 0
    recs = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert all(r["WORLD_SIZE"] == "3" and r["MASTER_ADDR"] == "127.0.0.1" for r in recs)
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and recs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--fail"])
    import time
    t0 = time.time()
    assert bench.spawn_ranks(A) == 7          # the failing rank's code, not the SIGTERM of the ranks it took down
    assert time.time() - t0 < 30


def test_also_legs_schema_and_failure_handling():
    """The default one-GPU run appends configs[1] (SetAbstraction forward) and configs[4] (FixMatch+NTM iteration) under
    `also`: each leg = the child's own JSON line cut down to the contract's fields; a failing child is recorded, not fatal."""
    bench = _load_bench()
    seen = []

    class R:
        def __init__(self, rc, out, err=""):
            self.returncode, self.stdout, self.stderr = rc, out, err

    def fake_run(cmd, env=None, **kw):
        seen.append((cmd, env))
        wl = cmd[cmd.index("--workload") + 1]
        if wl == "fixmatch":
            return R(3, "", "boom")
        line = {"metric": "point-clouds/sec", "value": 281.0, "unit": "clouds/s", "steps": 20, "warmup": 3, "ms_per_step": 3.56,
                "dtype": "f32", "data": "synthetic", "config": {"workload": "configs[1] ..."},
                "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.006, "traffic": None},
                "cpu_baseline": {"value": 2.1, "unit": "clouds/s", "cores": 16, "kind": "port", "sample": "12 steps"},
                "n_gpus": 1, "hot_path": {"dropped": True}}
        return R(0, "noise\n" + json.dumps(line) + "\n")

    class A:
        steps, warmup, no_cpu_baseline = 20, 5, False
    os.environ["RANK"] = "0"
    try:
        out = bench.also_legs(A, run=fake_run)
    finally:
        del os.environ["RANK"]
    assert set(out) == {"sa", "fixmatch"}
    sa = out["sa"]
    for k in ("steps", "ms_per_step", "value", "roofline", "cpu_baseline", "config", "metric", "unit"):
        assert k in sa, k
    assert "hot_path" not in sa and sa["roofline"]["frac"] == 0.006 and sa["command"].startswith("python bench.py --gpus 1 --no-also")
    assert out["fixmatch"]["error"].startswith("exit 3")
    for cmd, env in seen:
        assert "--no-also" in cmd and "RANK" not in env          # a child never recurses and never inherits a rendezvous
    fm = [c for c, _ in seen if "fixmatch" in c][0]
    assert fm[fm.index("--steps") + 1] == "10" and fm[fm.index("--warmup") + 1] == "3"


def test_trace_window_summarises_only_the_steady_state(tmp_path):
    rows = ["Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    t = 0
    for step in range(6):                                     # warm-up steps carry a slow tuning kernel
        rows.append("K,0,0,0,0,0,0,void geot::fps_pruned_kernel<768>,0,%d,%d" % (t, t + 4_000_000))
        if step < 2:
            rows.append("K,0,0,0,0,0,0,tuning_kernel,0,%d,%d" % (t + 5_000_000, t + 50_000_000))
        rows.append("K,0,0,0,0,0,0,gemm,0,%d,%d" % (t + 60_000_000, t + 62_000_000))
        t += 100_000_000
    trace = tmp_path / "kt.csv"
    trace.write_text("\n".join(rows) + "\n")
    out = tmp_path / "w.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_window.py"), str(trace), "--skip", "2", "--steps", "3",
                        "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    assert "tuning_kernel" not in text and "fps_pruned_kernel" in text and "gemm" in text
    assert "wall 100.000 ms/step" in text and "summed kernel time 6.000 ms/step" in text


def test_pmc_traffic_lookup_reports_its_source(tmp_path, monkeypatch):
    bench = _load_bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    for v, traffic in ((1, 111), (2, 222)):
        (prof / ("r02_bench_model_v%d_pmc_traffic.json" % v)).write_text(json.dumps(
            {"commit": "abc%d" % v, "kernels": {"geot::fps_pruned_kernel<768, 32, false, 8>": {"traffic_bytes": traffic}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, src = bench.pmc_traffic("fps_pruned_kernel<768, 32, false", "bench_model")
    assert t == 222 and src == {"file": os.path.join("profiles", "r02_bench_model_v2_pmc_traffic.json"), "commit": "abc2",
                                "kernel_source_unchanged": None}          # (a profile from before the source hashes)
    assert bench.pmc_traffic("no_such_kernel", "bench_model") == (None, None)
    # a profile that recorded the hash of the kernel's source: unchanged / changed since
    import hashlib
    csrc = tmp_path / "geot_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "fps.hip").write_text("__global__ void fps_pruned_kernel() {}\n")
    sha = hashlib.sha256((csrc / "fps.hip").read_bytes()).hexdigest()[:16]
    (prof / "r02_bench_model_v3_pmc_traffic.json").write_text(json.dumps(
        {"commit": "abc3", "csrc_sha16": {"fps.hip": sha, "geot_common.h": "0" * 16},
         "kernels": {"geot::fps_pruned_kernel<768, 32, false, 8>": {"traffic_bytes": 333}}}))
    t, src = bench.pmc_traffic("fps_pruned_kernel<768, 32, false", "bench_model")
    assert t == 333 and src["kernel_source_unchanged"] is True
    (csrc / "fps.hip").write_text("__global__ void fps_pruned_kernel() { /* edited */ }\n")
    assert bench.pmc_traffic("fps_pruned_kernel<768, 32, false", "bench_model")[1]["kernel_source_unchanged"] is False


def test_recorded_gemm_selection_is_a_wellformed_tunableop_file():
    from geot_amd import tuning
    lines = open(tuning.DEFAULT).read().strip().splitlines()
    validators = [l for l in lines if l.startswith("Validator,")]
    assert any("GCN_ARCH_NAME,gfx950" in l for l in validators) and any("ROCBLAS_VERSION" in l for l in validators)
    entries = [l.split(",") for l in lines if l.startswith("GemmTunableOp") or "TunableOp_float" in l.split(",")[0]]
    assert len(entries) > 50 and all(len(e) >= 3 for e in entries)
    assert all("float" in e[0] for e in entries)              # fp32 GEMMs only: the selection never changes precision


def test_lookahead_gemm_selection_differs_from_the_alone_file_only_where_it_was_retuned():
    """geot_amd/tuning: the default selection = the `alone` file (every shape's fastest solution alone on the chip) with the
    shapes that run beside the look-ahead re-tuned under that contention (tools/tune_lookahead_gemms.py): same validators, the
    same shapes (+ at most a few new ones), a different solution for a few dozen 8-cloud shapes of the blocks' / encoder's
    backward and for nothing else."""
    import os
    from geot_amd import tuning

    def load(path):
        head, rows = [], {}
        for line in open(path).read().strip().splitlines():
            p = line.split(",")
            if p[0] == "Validator":
                head.append(line)
            else:
                rows[(p[0], p[1])] = p[2]
        return head, rows
    alone_path = os.path.join(os.path.dirname(tuning.DEFAULT), "tunableop_gfx950_alone.csv")
    h0, alone = load(alone_path)
    h1, cur = load(tuning.DEFAULT)
    assert h0 == h1
    assert set(alone) <= set(cur) and len(set(cur) - set(alone)) <= 4
    changed = [k for k in alone if alone[k] != cur[k]]
    assert 10 <= len(changed) <= 40
    # 8-cloud supervised shapes only: 4096 tokens (8 x 512 groups) or 131072 group points (8 x 512 x 32) in the parameters
    assert all(any(tok in k[1].split("_") for tok in ("4096", "131072")) or "_B_32_" in k[1] for k in changed), changed


def test_replay_guard_reports_the_eager_leg_when_a_replay_stalls(tmp_path):
    """N > 1: a replay that never returns must not cost the run -- after the deadline rank 0 prints the line of the eager leg it
    timed first and every rank leaves with 0; a replay that finishes cancels the guard."""
    prog = textwrap.dedent("""
        import importlib.util, sys, time
        spec = importlib.util.spec_from_file_location("geot_bench", %r)
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        mode = sys.argv[1]
        g = bench.ReplayGuard(0.3, '{"metric": "eager leg"}' if mode != "other_rank" else None)
        if mode == "finishes":
            g.cancel()
            time.sleep(0.8)
            print("done")
            sys.exit(0)
        time.sleep(30)              # the stalled replay
        print("late")
    """ % os.path.join(ROOT, "bench.py"))
    script = tmp_path / "guard.py"
    script.write_text(prog)
    env = dict(os.environ, GEOT_BUILD_ON_IMPORT="0")
    r = subprocess.run([sys.executable, str(script), "stalls"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.strip().splitlines() == ['{"metric": "eager leg"}'] and "did not finish" in r.stderr
    r = subprocess.run([sys.executable, str(script), "other_rank"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, str(script), "finishes"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.strip() == "done"
