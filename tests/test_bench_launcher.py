"""CPU tests of bench.py's launch path: `--gpus N` without a launcher spawns N fresh rank processes (before anything
touches a GPU), the ranks rendezvous on 127.0.0.1, gather their samples to rank 0 and sum their statistics, and rank 0
prints ONE JSON line with n_gpus = N.  --dry replaces the engines by a stand-in and RCCL by gloo, so it runs here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", *args], capture_output=True, text=True, timeout=600, env=env)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_reports_the_node():
    r = _run("--gpus", "2", "--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 1 and j["scaling"] == "weak"
    # whole-job aggregate: both ranks' games; the stand-in engine finishes S/4 games per step
    assert j["config"]["games_in_window"] == 2 * 5 * 4096 // 4
    # the one exchange step: rank r contributes 8 + r rows per step, all of them arrive on rank 0
    assert j["config"]["samples_gathered"] == 5 * (8 + 9) == j["config"]["samples_in_window"]
    # the statistics all-reduce: the stand-in vector is arange(16) * (rank + 1)
    assert j["config"]["node_stats"]["sum"] == [3.0 * i for i in range(16)]


def test_single_process_default():
    r = _run("--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["samples_gathered"] == 0


def test_a_window_without_enough_games_is_an_error_not_a_number():
    r = _run("--steps", "3", "--warmup", "0")        # 3 steps x S/4 games < S
    assert r.returncode != 0 and "INVALID RUN" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_must_match_the_launcher():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "launcher started 1 ranks" in r.stderr


def test_gpus_8_dry_run_of_the_whole_node():
    """the driver's N = 8 case, dry: eight rank processes rendezvous on 127.0.0.1 over gloo, every rank's samples reach rank 0,
    the statistics are summed over the node, one JSON line comes out"""
    r = _run("--gpus", "8", "--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak"
    assert j["config"]["games_in_window"] == 8 * 5 * 4096 // 4
    assert j["config"]["samples_gathered"] == 5 * sum(8 + k for k in range(8)) == j["config"]["samples_in_window"]
    assert j["config"]["node_stats"]["sum"] == [36.0 * i for i in range(16)]
