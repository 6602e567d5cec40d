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
