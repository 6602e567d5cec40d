"""bench.py's N-rank launcher: `--gpus N` means N ranks or a refusal, never a silent single-rank run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UAMMD_BENCH_SAME_DEVICE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=600)


def test_more_gpus_than_the_machine_has_is_refused():
    """No WORLD_SIZE in the environment and fewer visible GPUs than --gpus: exit code 2 and a message, no JSON line."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 2" in r.stderr and "GPU" in r.stderr
    assert '"metric"' not in r.stdout


def test_world_size_must_match_gpus():
    """Started by a launcher with another world size than --gpus says: refused as well (the JSON's n_gpus would lie)."""
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert '"metric"' not in r.stdout


def test_a_stage_that_never_finishes_ends_in_a_json_error_line_not_a_hang():
    """bench.py at N > 1: every stage (process group + communicator, preflight, timed workloads) has a wall-clock budget; when one runs
    out — a collective that never completes — rank 0 prints ONE JSON line with an `error` field and every rank exits.  Here: a stage
    with a one-second budget that sleeps."""
    import json
    code = ("import sys, time; sys.path.insert(0, %r); import bench; bench._start_watchdog(0, 2, 2); "
            "bench.stage('a collective that never completes', 1.0); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4, (r.returncode, r.stderr[-500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and "did not finish" in line["error"] and line["stage"] == "a collective that never completes" and line["n_gpus"] == 2


def test_an_optional_stage_that_hangs_keeps_the_result_so_far():
    """The strong-scaling LJ line runs LAST and as an optional stage (bench.py: `_WATCHDOG["fallback"]`): when it never finishes, rank 0
    prints the run's complete result so far — with the failure noted under the stage's key — and the processes exit 0."""
    import json
    code = ("import sys, time; sys.path.insert(0, %r); import bench; bench._start_watchdog(0, 2, 2); "
            "bench._WATCHDOG['fallback'] = {'metric': 'm', 'value': 123.0, 'n_gpus': 2}; bench._WATCHDOG['fallback_key'] = 'lj_strong'; "
            "bench.stage('strong-scaling LJ line (optional)', 1.0); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 123.0 and "did not finish" in line["lj_strong"]["error"] and "error" not in line
