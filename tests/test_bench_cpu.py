"""bench.py's launcher half without a GPU: `python bench.py --gpus 2` with no WORLD_SIZE in the environment starts two ranks of
its own (torch.distributed.run on 127.0.0.1), they rendezvous (gloo here), time the same K steps between barriers, take the max
over ranks, and rank 0 prints ONE JSON line that says n_gpus 2.  The workload is a stub (`launcher_selftest`: no kernels, never a
measurement); what is tested is the path the driver's `--gpus N` takes (the entry point must spawn its own ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout)


def test_gpus_2_without_world_size_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "launcher_selftest", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_in_group"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert "starting 2 ranks" in r.stderr


def test_gpus_1_is_a_single_process_without_a_group():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--workload", "launcher_selftest", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["ranks_in_group"] == 1
    assert "starting" not in r.stderr


def test_more_ranks_than_gpus_is_refused():
    """One rank per GPU: on a node with fewer GPUs than --gpus (none here) the launcher says so instead of oversubscribing."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "this node has" in (r.stderr + r.stdout)


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2", "--workload", "launcher_selftest", "--backend", "gloo"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
