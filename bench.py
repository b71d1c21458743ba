#!/usr/bin/env python3
"""Self-play throughput bench — BASELINE.json metric:
"self-play games/sec (whole node), Connect4 @ 800 MCTS sims".

Workload = BASELINE configs[1]: Connect4, 4096 concurrent games per GPU, 800 simulations on every move (playout cap
off), 6-block/64-channel ResNet (k3, 32 head channels) with random-init weights, the reference's self-play flags
(game_runner.py:2018-2041), positions self-generated (synthetic).  Everything lives in HBM; no host round-trip happens
inside the timed region.

A *step* is a fixed block of `--rounds-per-step` engine rounds (default 2048): in one round every game slot of every
engine shard advances by its next simulations (backup of the previous leaf, maybe a move, descent + expansion of the
next leaf, position-cache probe) and the net evaluates the shard's leaf batch.  The run is a stream of games
(play_manager_bench.cc:171-181: finished slots start the next game at once):

  pre-roll   untimed rounds until 2 x concurrent_games games have finished, so that the slots are de-phased
             (the games all start together, and opening / endgame rounds cost differently)
  warm-up    W steps, untimed
  window     K steps, timed between two barriers; games/s = games finished in the window / its wall time
             (network_pareto.py:401-444 measures the same two counters over windows)

The run FAILS (non-zero exit, no JSON line) if fewer than concurrent_games games finished inside the window or a slot's
game stream ran dry: such a number would not be the metric.

Multi-GPU: one process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the launcher; `--gpus N` without a launcher spawns the
N processes itself, before anything touches a GPU), every rank runs the same workload on its own engines and weights
(weak scaling), the window's samples are gathered to rank 0 over RCCL and the statistics are summed with one all-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_EVAL = 37.7e6     # Connect4 6b64c k3 (conv + linear MACs x 2)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
B_SIM = 1300.0             # algorithmic bytes per simulation of the tree step


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; a step = --rounds-per-step engine rounds")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps before the window (after the de-phasing pre-roll)")
    ap.add_argument("--rounds-per-step", type=int, default=None,
                    help="lock-step driver: engine rounds per step (default 2048); pipeline driver: epochs per step (default 80)")
    ap.add_argument("--driver", choices=["auto", "pipeline", "rounds"], default="auto",
                    help="pipeline = asynchronous tree / net pipeline (azmi_run_pipeline: Connect4, PUCT); rounds = lock-step rounds "
                         "(azmi_run_rounds); auto = the pipeline where it applies")
    ap.add_argument("--sims-per-epoch", type=int, default=None, help="pipeline driver: simulations per epoch (default 256 x concurrent games)")
    ap.add_argument("--game", choices=["connect4", "tawlbwrdd", "stargambit"], default="connect4",
                    help="connect4 = BASELINE configs[1] (the headline); tawlbwrdd = configs[2] (2048 games, 400 sims, YAML net); "
                         "stargambit = configs[4] per GPU (star_gambit_unified, 1024 games, 800 sims, 200000-entry device cache)")
    ap.add_argument("--games", type=int, default=None, help="concurrent games per GPU (4096 / 2048)")
    ap.add_argument("--sims", type=int, default=None, help="simulations per move (800 / 400)")
    ap.add_argument("--engines", type=int, default=None, help="engine shards (HIP streams) per GPU (default 4)")
    ap.add_argument("--cache", type=int, default=None,
                    help="max_cache_size per GPU (reference default 200000, config.py:197; sized up for 288 GB of HBM)")
    ap.add_argument("--inline", type=int, default=0, help="max simulations finished per slot per round without the net (0 = engine default)")
    ap.add_argument("--hwq", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (0 = leave the runtime default of 4)")
    ap.add_argument("--playout-cap", action="store_true", help="playout-cap randomisation at the TrainConfig defaults (config.py:86,100)")
    ap.add_argument("--gumbel", action="store_true", help="Gumbel AlphaZero search (configs/tawlbwrdd.yaml:24-25: gumbel_enabled, capped searches PUCT)")
    ap.add_argument("--net", choices=["hip", "torch"], default=None)
    ap.add_argument("--precision", choices=["bf16", "bf16x3", "fp32"], default="bf16",
                    help="HIP leaf net tier: bf16 (the reference's autocast arithmetic, the headline), bf16x3 (split bf16 operands: the north star's 1e-5 on the matrix cores), fp32 (plain kernels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the two short secondary measurements reported in config")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="wall time of the CPU-baseline sample")
    ap.add_argument("--preroll-factor", type=float, default=2.0, help="pre-roll until this many x concurrent_games games have finished")
    ap.add_argument("--profile-window", action="store_true",
                    help="for rocprofv3 runs: a short window is allowed (no validity gate); the line is marked profile_window and its value is not the metric")
    ap.add_argument("--no-tawlbwrdd", action="store_true",
                    help="the default run (1 GPU, Connect4) also measures BASELINE configs[2] - Tawlbwrdd 2048 games x 400 sims, PUCT and Gumbel, with its "
                         "CPU baseline - and reports it as the `tawlbwrdd` block of the line; this flag skips that")
    ap.add_argument("--no-stargambit", action="store_true",
                    help="... and BASELINE configs[4] per GPU - star_gambit_unified 1024 games x 800 sims, 200 k-entry device cache - as the `stargambit` block; this flag skips that")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="driver auto: when the pipeline probe fails on this box, measure the lock-step driver instead of failing (the line then carries fallbacks = 1)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)    # a measurement process started by the default run
    ap.add_argument("--probe", action="store_true", help=argparse.SUPPRESS)     # child process of the default driver choice: exit 0 when the pipeline runs here
    ap.add_argument("--dry", action="store_true",
                    help="CPU dry run of the launch / distributed plumbing (gloo, no device, a stand-in engine): used by the tests")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without torchrun
def spawn_ranks(args):
    """Starts one fresh process per GPU (RANK/LOCAL_RANK/WORLD_SIZE set, rendezvous on 127.0.0.1) BEFORE this process has
    made any GPU call, relays rank 0's JSON line, and fails if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is read by a thread; the ranks are polled together: when one of them dies (before or at the rendezvous, in a
    # collective) the others would wait for it for ever, so they are terminated and the run fails instead of hanging
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            failed = bad
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(p.wait())
    reader.join(10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if failed or any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}" + (f" (first failure: rank, code = {failed})" if failed else "") + "\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------------------------------
def selfplay_params(az, games, sims, stream_games, cache=0, playout_cap=False, gumbel=False):
    """self_play() settings, game_runner.py:2018-2041 with TrainConfig defaults (config.py:79-139,235-236);
    playout-cap randomisation is OFF so that every move is a full 800-simulation search."""
    pp = az.PlayParams()
    pp.concurrent_games = games
    pp.games_to_play = stream_games
    pp.max_batch_size = games
    pp.mcts_visits = [sims, sims]
    pp.cpuct = 1.25
    pp.fpu_reduction = 0.25
    pp.start_temp = 1.0
    pp.final_temp = 0.2
    pp.temp_decay_half_life = 10.0
    pp.history_enabled = True
    pp.self_play = True
    pp.tree_reuse = True
    pp.epsilon = 0.25
    pp.mcts_root_temp = 1.25
    pp.root_fpu_zero = True
    pp.shaped_dirichlet = True
    pp.policy_target_pruning = True
    pp.playout_cap_randomization = bool(playout_cap)
    if playout_cap:                      # config.py:86,100 and game_runner.py:2018-2041
        pp.playout_cap_depth = 25
        pp.playout_cap_percent = 0.75
    pp.resign_percent = 0.02
    pp.resign_playthrough_percent = 0.20
    pp.max_cache_size = cache
    pp.gumbel_enabled = bool(gumbel)
    pp.model_groups = [0, 0]      # one network on both seats: set_model_groups(), game_runner.py:773-787, 2054
    # experiment hook (never set for a reported number): AZMI_BENCH_OVERRIDES="history_enabled=0,epsilon=0.0"
    for item in filter(None, os.environ.get("AZMI_BENCH_OVERRIDES", "").split(",")):
        k, v = item.split("=")
        setattr(pp, k, type(getattr(pp, k))(float(v)))
    return pp


FALLBACKS = []        # driver fallbacks of this run (--allow-fallback only; otherwise the run fails): reported as the scalar `fallbacks`
PIPE_ERRORS = []      # pipeline errors met (and recovered from) during this run: reported in config.pipeline_errors, never hidden


def pipe_call(az, pm, net, n, spe, stream):
    """azmi_run_pipeline through the recoverable-error contract of round 4: a pipeline error (a spin that hit the stall cap: the host
    was held up between the two launches, another tenant took CUs) is reported once, leaves the engine whole, and the call is made
    again - at most three such errors per run, each one recorded in the bench line."""
    for _ in range(3):
        try:
            if isinstance(net, (list, tuple)):
                return az.run_pipeline_groups(pm, list(net), n, spe, stream)
            return az.run_pipeline(pm, net, n, spe, stream)
        except RuntimeError as e:
            if "pipeline error mask" not in str(e) or len(PIPE_ERRORS) >= 3:
                raise
            PIPE_ERRORS.append(str(e)[:160])
            sys.stderr.write("bench.py: pipeline error (recovered, call repeated): %s\n" % str(e)[:160]); sys.stderr.flush()
    raise RuntimeError("pipeline failed three times in a row")


def fixture_numerics(az, torch_net, game, precision, device_index):
    """max |delta| of THIS build's leaf net (the precision tier the run uses) against the reference NNArch's fp32 outputs, measured
    live on the committed fixture of the game's net (tests/golden/nn_*.npz, made by tests/golden/make_nn_fixture.py from the
    reference's neural_net.py:448-510): scalars `max_abs_dpi`, `max_abs_dv` for the line (outside the timed region)."""
    import numpy as np
    import torch
    fname, spec_fn = {"connect4": ("nn_connect4_6b64c.npz", "connect4_spec"), "tawlbwrdd": ("nn_tawlbwrdd_4b64c.npz", "tawlbwrdd_spec"),
                      "stargambit": ("nn_stargambit_4b64c.npz", "stargambit_spec")}[game]
    path = os.path.join(ROOT, "tests", "golden", fname)
    if not os.path.exists(path):
        return {}
    fx = np.load(path)
    spec = getattr(torch_net, spec_fn)()
    net = torch_net.LeafNet(spec)
    net.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd.")})
    hip = az.HipLeafNet(net.eval(), spec, device=device_index, precision=precision)
    v, pi = hip.process(torch.from_numpy(fx["input"]).to(torch.device("cuda", device_index)))
    torch.cuda.synchronize()
    return {"max_abs_dpi": float(np.abs(pi.cpu().numpy() - fx["pi"]).max()), "max_abs_dv": float(np.abs(v.cpu().numpy() - fx["v"]).max()),
            "numerics_fixture": "tests/golden/" + fname}


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:                                   # a cgroup CPU quota is the real core count of a container
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(az, sims, seconds, S, hip_net, cache, threads=None, game="connect4", sims_per_game=None):
    """the reference's architecture on this box's host cores, timed beside the GPU engine — the oracle
    (CPU restatement of the reference PlayManager, `kind` "port") with `cores - 1` worker threads (the reference's default,
    config.py:439-441), the 4096 concurrent games split over the workers, and the leaf net served by the MI355X through
    HOST buffers (build_batch -> process -> update_inferences: canonical rows gathered on the host, copied to the GPU,
    evaluated by the same HIP net, copied back; azmi_net_eval_host, no interpreter in the loop).  Each worker is one
    single-threaded PlayManager over its share of the slots with a position cache of the reference's default size
    (200 000 entries, config.py:197, split like the slots), so nothing is shared between workers: an upper bound for the
    reference, whose workers contend on queue and cache mutexes (play_manager.h:66-72).  Bounded sample: `seconds` of
    wall time of the game stream, the first `seconds / 4` discarded as warm-up of trees and caches."""
    import ctypes as C
    import threading
    import oracle_api as orc
    from alphazero._capi import lib
    threads = threads or max(1, host_cores() - 1)
    fn = C.cast(lib.azmi_net_eval_host, C.c_void_p).value
    out = {"unit": "games/s", "cores": threads, "kind": "port"}
    gid = orc.GAME_TAWLBWRDD if game == "tawlbwrdd" else orc.GAME_CONNECT4
    gname = {"tawlbwrdd": "Tawlbwrdd", "stargambit": "star_gambit_unified"}.get(game, "Connect4")
    netname = {"tawlbwrdd": "configs/tawlbwrdd.yaml net", "stargambit": "configs/star_gambit_unified.yaml net"}.get(game, "6b64c net")

    def leg(eval_kind, slots_total, nsims, secs, cache_total, nthreads):
        """one timed run: `nthreads` oracle PlayManagers side by side; returns (games/s, sims/s, evals/s) after warm-up"""
        per = max(1, slots_total // nthreads)
        pms = []
        for t in range(nthreads):
            pp = selfplay_params(az, per, nsims, 1 << 30, cache=(cache_total // nthreads if eval_kind == "nn" else 0))
            pp.max_batch_size = per
            pp.history_enabled = True
            if eval_kind == "random":
                pp.eval_type = [1, 1]
            if game == "stargambit":      # configs/star_gambit_unified.yaml: the four variants at equal probability, temperatures as the GPU run's
                pp.start_temp, pp.final_temp, pp.temp_decay_half_life_by_variant = 1.2, 0.2, [3.0, 4.0, 5.0, 8.0]
            pms.append(orc.PlayManager(orc.Game.sg_unified() if game == "stargambit" else gid, pp, 1000 + 7919 * t, per_slot_rng=False, record_moves=False))
        marks = []

        def snap():
            return (time.perf_counter(), sum(o.games_completed() for o in pms), sum(o.counters()["sims"] for o in pms),
                    sum(o.counters()["evals"] for o in pms))

        def worker(o, limit):
            o.set_time_limit(limit)
            if eval_kind == "nn":
                o.run_native(fn, hip_net._h.value)
            else:
                o.run()

        for phase, limit in (("warm", secs / 4.0), ("timed", secs)):
            ths = [threading.Thread(target=worker, args=(o, limit)) for o in pms]
            a = snap()
            for th in ths: th.start()
            for th in ths: th.join()
            b = snap()
            marks.append((a, b))
        (t0, g0, s0, e0), (t1, g1, s1, e1) = marks[1]
        dt = t1 - t0
        return (g1 - g0) / dt, (s1 - s0) / dt, (e1 - e0) / dt, g1 - g0, dt

    if hip_net is not None and game == "stargambit":
        # (round 6: a short leg for configs[4] - the oracle port plays a StarGambit game in seconds, so the rate is the sample's
        # simulations/s over the GPU run's simulations per game; a leg that cannot run leaves the RANDOM-evaluator figure below)
        try:
            g, s, e, n, dt = leg("nn", S, sims, seconds, cache, threads)
        except Exception as ex:      # noqa: BLE001 (recorded in the line)
            out["nn_leg_error"] = str(ex)[:160]
            hip_net = None
    if hip_net is not None:
        if game != "stargambit":
            g, s, e, n, dt = leg("nn", S, sims, seconds, cache, threads)
        out["value"] = g
        if n < 64 and sims_per_game:      # a long game does not finish inside a bounded sample: the rate follows from the simulations
            out["value"] = s / sims_per_game
            out["value_note"] = (f"only {n} games finished inside the sample; value = the sample's simulations/s / {sims_per_game:.0f} simulations per game "
                                 f"(measured on the GPU run of the same workload)")
        out["sample"] = (f"{n} {gname} self-play games x {sims} sims finished in {dt:.1f}s (after {seconds / 4:.0f}s warm-up) by the oracle PlayManager: "
                         f"{threads} worker threads (host cores - 1) x {max(1, S // threads)} concurrent games, {netname} served by the MI355X "
                         f"through host buffers (azmi_net_eval_host), position cache {cache} entries; {s / 1e6:.3f} Msims/s, {e / 1e6:.3f} M net evaluations/s")
        out["sims_per_s"] = s
        out["evals_per_s"] = e
        out["cache_entries"] = cache
    g, s, _, n, dt = leg("random", S, sims, max(4.0, seconds / 4), 0, threads)
    out["tree_only"] = {"games_per_s": g, "sims_per_s": s, "note": f"EvalType.RANDOM (no net), {threads} threads, {n} games in {dt:.1f}s"}
    out["tree_only_sims_per_s"] = s
    if "value" not in out:
        out["value"] = g if (n >= 64 or not sims_per_game) else s / sims_per_game
        out["sims_per_s"] = s
        out["sample"] = out["tree_only"]["note"] + (f"; value = simulations/s / {sims_per_game:.0f} simulations per game of the GPU run" if (n < 64 and sims_per_game) else "")
    if game != "connect4":
        return out
    # BASELINE configs[0]: Connect4, 64 concurrent games, 100 sims, one worker thread — the reference's own CPU-runnable case
    g, s, _, n, dt = leg("random", 64, 100, 3.0, 0, 1)
    out["configs0"] = {"workload": "Connect4, 64 concurrent games, 100 sims, 1 worker thread", "random_eval_games_per_s": g, "random_eval_sims_per_s": s}
    if hip_net is not None:
        g, s, e, n, dt = leg("nn", 64, 100, 3.0, 200_000, 1)
        out["configs0"].update({"gpu_served_net_games_per_s": g, "gpu_served_net_sims_per_s": s})
    return out


# ------------------------------------------------------------------------------------------------------------------
class DryEngine:
    """Stand-in for the engines in --dry mode (CPU, gloo): deterministic counters, a few fake sample rows per step."""

    def __init__(self, rank, S):
        import torch
        self.torch, self.rank, self.S = torch, rank, S
        self.games = self.sims = self.evals = 0
        self.rows = []

    def step(self, rounds):
        self.games += self.S // 4
        self.sims += self.S * rounds
        self.evals += self.S * rounds // 3
        n = 8 + self.rank
        self.rows.append((self.torch.full((n, 4, 6, 7), float(self.rank)), self.torch.zeros(n, 3), self.torch.zeros(n, 7)))

    def take_rows(self):
        t = self.torch
        out = tuple(t.cat([r[i] for r in self.rows], 0) for i in range(3)) if self.rows else (t.zeros(0, 4, 6, 7), t.zeros(0, 3), t.zeros(0, 7))
        self.rows = []
        return out


def run_worker(extra, timeout):
    """one measurement in a child process of its own (started before this process has touched a GPU): returns the parsed JSON
    line, or {"error": ...}"""
    cmd = [sys.executable, os.path.abspath(__file__), "--worker"] + extra
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout}s", "cmd": " ".join(extra)}
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr.decode()[-600:], "cmd": " ".join(extra)}
    return json.loads(lines[-1])


def orchestrate(args):
    """The default run: the headline (Connect4, BASELINE configs[1]) and, beside it in the same line, BASELINE configs[2] -
    Tawlbwrdd 11x11, 2048 concurrent games, 400 sims, the configs/tawlbwrdd.yaml net - with PUCT and with Gumbel
    (configs/tawlbwrdd.yaml:24-25) and its own CPU baseline (play_manager_bench.cc:171-181 methodology).  Each measurement is a
    process of its own, one after the other on the one GPU; this process never touches it."""
    t0 = time.perf_counter()
    passthrough = [a for a in sys.argv[1:] if a not in ("--no-tawlbwrdd", "--no-stargambit")]
    head = run_worker(passthrough, 900)
    if "error" in head:
        sys.stderr.write("bench.py: the headline measurement failed: " + json.dumps(head) + "\n")
        return 3
    # windows long enough for more than 2048 games to finish (the validity gate of a run): ~200 / ~145 games per 2048-round step
    common = ["--game", "tawlbwrdd", "--warmup", "1", "--no-secondary", "--preroll-factor", "1.0", "--cpu-seconds", "12"]
    if args.no_cpu_baseline:
        common.append("--no-cpu-baseline")
    puct = run_worker(common + ["--steps", "13"], 600)
    gumbel = run_worker(common + ["--steps", "17", "--gumbel", "--no-cpu-baseline"], 600)

    def trim(d):
        if "error" in d:
            return d
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "roofline_tree", "cpu_baseline")
        out = {k: d[k] for k in keep if k in d}
        out["config"] = {k: v for k, v in d["config"].items() if k != "node_stats"}
        return out
    head["tawlbwrdd"] = trim(puct)
    head["tawlbwrdd"]["gumbel"] = trim(gumbel)
    # BASELINE configs[4] (per GPU): star_gambit_unified, 1024 concurrent games, 800 sims, device-side S3-FIFO of 200 000 entries
    # (configs/star_gambit_unified.yaml:5-15,23-24); round 6: a 10 s CPU leg (its rate = the sample's simulations/s over the GPU run's simulations per game)
    if not args.no_stargambit:
        sgb = run_worker(["--game", "stargambit", "--warmup", "1", "--no-secondary", "--preroll-factor", "0.5", "--steps", "100", "--cpu-seconds", "10"]
                         + (["--no-cpu-baseline"] if args.no_cpu_baseline else []), 700)
        head["stargambit"] = trim(sgb)
    head["config"]["bench_wall_s"] = time.perf_counter() - t0
    flatten_line(head)
    print(json.dumps(head))
    return 0


def flatten_line(head):
    """a parsed line keeps the contract's keys and the SCALARS inside `config`, `roofline`, `cpu_baseline`; nested
    blocks and extra top-level keys are dropped.  Every figure DESIGN 8.1 quotes is therefore copied into `config` as a scalar (the
    nested blocks stay in the line for whoever reads the full JSON)."""
    cfg = head["config"]
    for k in ("pipeline_errors", "lost_total", "pipeline_freezes", "tile_us_per_board", "fallbacks"):
        if k in head:
            cfg[k if k != "pipeline_errors" else "pipeline_errors_count"] = head[k]
    rt = head.get("roofline_tree") or {}
    for k_src, k_dst in (("frac", "tree_hbm_frac_per_launch"), ("aggregate_frac", "tree_hbm_frac_aggregate"), ("traffic", "tree_traffic_bytes_per_launch"),
                         ("traffic_bytes_per_simulation", "tree_traffic_bytes_per_simulation"), ("traffic_over_algorithmic", "tree_traffic_over_algorithmic")):
        if rt.get(k_src) is not None:
            cfg[k_dst] = rt[k_src]
    tw = head.get("tawlbwrdd") or {}
    if "value" in tw:
        cfg["tawlbwrdd_games_per_s"] = tw["value"]
        cfg["tawlbwrdd_mfma_frac"] = (tw.get("roofline") or {}).get("aggregate_frac")
        cfg["tawlbwrdd_mfma_frac_per_launch"] = (tw.get("roofline") or {}).get("frac")
        cfg["tawlbwrdd_tree_traffic_bytes_per_shard_round"] = (tw.get("roofline_tree") or {}).get("traffic")
        cfg["tawlbwrdd_cpu_baseline_games_per_s"] = (tw.get("cpu_baseline") or {}).get("value")
        cfg["tawlbwrdd_max_abs_dpi"] = (tw.get("config") or {}).get("max_abs_dpi")
        cfg["tawlbwrdd_max_abs_dv"] = (tw.get("config") or {}).get("max_abs_dv")
    elif "error" in tw:
        cfg["tawlbwrdd_error"] = str(tw["error"])[:100]
    tg = tw.get("gumbel") or {}
    if "value" in tg:
        cfg["tawlbwrdd_gumbel_games_per_s"] = tg["value"]
        cfg["tawlbwrdd_gumbel_mfma_frac"] = (tg.get("roofline") or {}).get("aggregate_frac")
    elif "error" in tg:
        cfg["tawlbwrdd_gumbel_error"] = str(tg["error"])[:100]
    sgb = head.get("stargambit") or {}
    if "value" in sgb:
        cfg["stargambit_games_per_s"] = sgb["value"]
        cfg["stargambit_mfma_frac"] = (sgb.get("roofline") or {}).get("aggregate_frac")
        cfg["stargambit_mfma_frac_per_launch"] = (sgb.get("roofline") or {}).get("frac")
        cfg["stargambit_net_traffic_bytes_per_launch"] = (sgb.get("roofline") or {}).get("traffic")
        cfg["stargambit_tree_traffic_bytes_per_shard_round"] = (sgb.get("roofline_tree") or {}).get("traffic")
        cfg["stargambit_cache_hit_rate"] = (sgb.get("config") or {}).get("cache_hit_rate")
        cfg["stargambit_cpu_baseline_games_per_s"] = (sgb.get("cpu_baseline") or {}).get("value")
    elif "error" in sgb:
        cfg["stargambit_error"] = str(sgb["error"])[:100]
    cb = head.get("cpu_baseline")
    if cb and "games_per_s_cache_200k" in cfg:
        # the like-for-like pair: the CPU sample runs the reference's default cache (200 000 entries); the GPU
        # engine at THAT cache size is config.games_per_s_cache_200k, not `value` (whose cache is sized for 288 GB of HBM)
        cb["gpu_same_cache_games_per_s"] = cfg["games_per_s_cache_200k"]
        cb["note"] = ("cache_entries = %d (reference default, config.py:197); the GPU engine at the same cache size: gpu_same_cache_games_per_s "
                      "(hit rate %.2f) - compare THAT with value; the headline's cache is %d entries" % (cb.get("cache_entries", 200000), cfg.get("hit_rate_cache_200k", 0.0), cfg.get("max_cache_size", 0)))


def _pipeline_probe(device_index, slots):
    """three short calls of the asynchronous pipeline on a throw-away engine of `slots` games (50 simulations per move, small cache);
    False when the engine does not qualify or the library reports a pipeline error"""
    try:
        import alphazero as az
        from alphazero import torch_net
        spec = torch_net.connect4_spec()
        hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec, device=device_index)
        pp = selfplay_params(az, slots, 50, 1 << 30, cache=1 << 20)
        pm = az.PlayManager(az.Connect4GS(), pp, seed=1, device=device_index)
        if not az.pipeline_supported(pm, hip):
            return False
        for _ in range(3):
            az.run_pipeline(pm, hip, 2, 64 * slots)
        return True
    except RuntimeError as e:
        sys.stderr.write("bench.py: pipeline probe: %s\n" % (str(e)[:300],))
        return False


def main():
    args = parse()
    if args.probe:
        sys.exit(0 if _pipeline_probe(int(os.environ.get("LOCAL_RANK", "0")), args.games or 4096) else 7)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if (not args.dry and not args.worker and not args.no_tawlbwrdd and args.gpus == 1 and "RANK" not in os.environ and args.game == "connect4"
            and not args.gumbel and not args.playout_cap and not args.profile_window):
        sys.exit(orchestrate(args))
    if args.hwq:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hwq)   # must be set before the HIP runtime starts
    # The probe of the pipeline (see use_pipe below) runs in a process of its own, BEFORE this one starts the HIP runtime: run here, its
    # streams stayed mapped to hardware queues and the lock-step secondaries of the same process (4 shards = 4 streams on the
    # runtime's 4 queues) ran at half speed.
    probe_ok = None
    if (args.driver == "auto" and not args.dry and args.game == "connect4" and (args.net or "hip") == "hip"):
        try:
            for _attempt in range(2):      # (one repeat: a probe that met a busy box once is not a verdict on the box)
                probe_ok = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe", "--games", str(args.games or 4096)],
                                          env=dict(os.environ), timeout=600).returncode == 0
                if probe_ok:
                    break
        except subprocess.TimeoutExpired:
            probe_ok = False
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and "RANK" in os.environ:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks\n")
        sys.exit(2)
    # AZMI_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL init, barriers, sample gather, reductions) with whatever world size
    # the launcher gave, 1 included: lets a one-GPU box exercise it under torchrun
    use_dist = world > 1 or os.environ.get("AZMI_BENCH_FORCE_DIST") == "1"
    dry = args.dry
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=600))

    sg = args.game == "stargambit"
    tafl = args.game in ("tawlbwrdd", "stargambit")      # the wide-game engine (one wavefront per slot, spatial-head net)
    if args.games is None: args.games = 1024 if sg else 2048 if tafl else 4096
    if args.sims is None: args.sims = 800 if sg else 400 if tafl else 800
    # the pipeline drives ONE engine with every slot (one GPU-wide position cache); the lock-step driver wants 4 shards
    use_pipe = args.driver == "pipeline" or (args.driver == "auto" and not tafl and (args.net or "hip") == "hip" and not args.dry)
    if use_pipe and args.driver == "auto":
        # a safety net, not a tuning knob: the pipeline's persistent kernels rely on how this part places workgroups.
        # A short probe at the bench's slot count runs before anything is sized; if it raises on this box (census, time cap), every
        # rank falls back to the lock-step driver - a slower line beats none - and the line's config says which driver ran.
        ok = bool(probe_ok)
        if use_dist:
            flag = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item() > 0.5)
        if not ok:
            # round 6: a headline that silently ran on the lock-step fallback would read as a slow run, not as a
            # failure.  The run FAILS here unless --allow-fallback is given; with it the line carries fallbacks = 1 and names the driver.
            if not args.allow_fallback:
                if rank == 0:
                    sys.stderr.write("bench.py: INVALID RUN: the pipeline probe failed on this box (twice) and --allow-fallback was not given; "
                                     "`--driver rounds` measures the lock-step driver on purpose\n")
                if use_dist:
                    dist.destroy_process_group()
                sys.exit(4)
            if rank == 0:
                sys.stderr.write("bench.py: the pipeline probe failed on this box: falling back to --driver rounds (--allow-fallback)\n")
            use_pipe = False
            FALLBACKS.append("probe")
    if args.engines is None: args.engines = 1 if use_pipe else 4      # lock-step, measured: Connect4 1/2/4/8 shards and Tawlbwrdd 2/4/8 shards both peak at 4
    if args.rounds_per_step is None: args.rounds_per_step = 80 if use_pipe else 2048
    if args.cache is None: args.cache = 200_000 if sg else 0 if tafl else 128_000_000      # Tawlbwrdd: measured 5 % hit rate with 2 M entries and 16 % fewer games/s, so off; StarGambit: configs[4] / config.py:197
    # SURVEY §8d; StarGambit net (configs/star_gambit_unified.yaml, 36 x 13 x 13): stem 7.0 + 8 trunk convs 99.7 + head 1x1s 2.8 + two
    # head convs 24.9 + policy 1x1 0.2 + value / global FCs 0.7 = 135.3 MFLOP per position
    flop_per_eval = 135.3e6 if sg else 93.1e6 if tafl else FLOP_PER_EVAL
    S, sims, K, R = args.games, args.sims, args.engines, args.rounds_per_step
    assert S % K == 0
    Se = S // K                       # slots per engine shard

    def barrier():
        if use_dist:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    if dry:
        eng = DryEngine(rank, S)
        hip_net = None
        net_kind = "dry"

        def run_rounds(n, ev=None):
            eng.step(n)

        def totals():
            return eng.games, eng.sims, eng.evals, 0, 0

        def live_slots():
            return S

        def take_rows():
            return eng.take_rows()

        def stat_vector():
            return torch.arange(16, dtype=torch.float64) * (rank + 1)
    else:
        import alphazero as az
        from alphazero import torch_net
        Game = az.StarGambitUnifiedGS if sg else az.TawlbwrddGS if tafl else az.Connect4GS
        # the game stream never runs dry: finished slots restart at once (play_manager_bench.cc:171-181 sizes its pool as
        # 8 x concurrent for the same purpose); the finished samples leave the engines' ring every step (hist_saver's job)
        STREAM = 1 << 30

        def make_engines(cache, playout_cap, seed0):
            out = []
            for i in range(K):
                pp = selfplay_params(az, Se, sims, STREAM, cache=cache // K, playout_cap=playout_cap, gumbel=args.gumbel)
                if sg:   # configs/star_gambit_unified.yaml: self_play_temp 1.2, final_temp 0.2, per-variant half lives 3 / 4 / 5 / 8 turns
                    pp.start_temp, pp.final_temp, pp.temp_decay_half_life_by_variant = 1.2, 0.2, [3.0, 4.0, 5.0, 8.0]
                out.append(az.PlayManager(Game(), pp, seed=seed0 + 7919 * rank + 104729 * i, device=local_rank, max_inline=args.inline,
                                          history_capacity=Se * (300 if sg else 400 if tafl else 42) * 4))
            return out

        # K engine shards of S/K slots, one HIP stream each: while one shard's leaf batch is on the matrix
        # cores another shard's tree kernel runs on the CUs the net leaves free.
        pms = make_engines(args.cache, args.playout_cap, 20240601)
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        sps = [st.cuda_stream for st in streams]
        io = [pm.io_tensors() for pm in pms]
        spec = torch_net.stargambit_spec() if sg else torch_net.tawlbwrdd_spec() if tafl else torch_net.connect4_spec()
        net = torch_net.random_init(spec, seed=0).to(dev)
        net_kind = args.net or "hip"
        hip_net = az.HipLeafNet(net, spec, device=local_rank, precision=args.precision) if net_kind == "hip" else None
        if net_kind == "torch":
            net = net.to(memory_format=torch.channels_last)

        def evaluate(group, i):
            canon, v_buf, pi_buf = io[i]
            if hip_net is not None:
                group[i].net_forward(hip_net, sps[i])
            else:
                with torch.cuda.stream(streams[i]):
                    v, pi = net.process(canon, amp_dtype=torch.bfloat16)
                    v_buf.copy_(v)
                    pi_buf.copy_(pi)

        spe = args.sims_per_epoch or 256 * S
        pipe_acc = {"net_us": 0.0, "tree_us": 0.0, "epochs": 0, "tiles": 0, "boards": 0, "late": 0, "host_us": 0.0}

        def run_rounds_on(group, n, ev=None):
            """n rounds of every shard. With the HIP net the loop is the native driver (azmi_run_rounds);
            every 64th round is launched from here with HIP events around the two kernels of shard 0.
            Pipeline driver: n EPOCHS of the asynchronous pipeline (azmi_run_pipeline) on the one engine; the durations of
            its net / tree kernels (HIP events on their own streams, inside the library) are summed when `ev` is given."""
            if use_pipe:
                for pm_ in group:
                    st_ = pipe_call(az, pm_, hip_net, n, spe, sps[0])
                    if ev is not None:
                        pipe_acc["net_us"] += st_["net_kernel_us"]; pipe_acc["tree_us"] += st_["tree_kernel_us"]; pipe_acc["epochs"] += st_["epochs"]; pipe_acc["host_us"] += st_["host_enqueue_us"]
                        pipe_acc.setdefault("tiles0", pipe_acc.get("tiles_now", 0)); pipe_acc.setdefault("boards0", pipe_acc.get("boards_now", 0))
                        pipe_acc["late"] = max(pipe_acc["late"], st_["tree_latest_start_us"], st_["net_latest_start_us"])
                        pipe_acc["net_wgs"], pipe_acc["tree_wgs"] = st_["net_wgs"], st_["tree_wgs"]
                        pipe_acc["lost_total"] = st_.get("lost_total", 0); pipe_acc["freezes"] = st_.get("freezes", 0)
                    pipe_acc["tiles_now"], pipe_acc["boards_now"] = st_["tiles"], st_["tile_boards"]
                    pipe_acc["tiles"] = pipe_acc["tiles_now"] - pipe_acc.get("tiles0", 0); pipe_acc["boards"] = pipe_acc["boards_now"] - pipe_acc.get("boards0", 0)
                return
            done = 0
            while done < n:
                if ev is not None:
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    for i in range(K):
                        if i == 0:
                            e0.record(streams[0])
                        if hip_net is not None:
                            group[i].round_net(hip_net, sps[i], part=1)    # the launches of the native loop, in two halves
                        else:
                            group[i].round(sps[i])
                        if i == 0:
                            e1.record(streams[0])
                        if hip_net is not None:
                            group[i].round_net(hip_net, sps[i], part=2)
                        else:
                            evaluate(group, i)
                        if i == 0:
                            e2.record(streams[0])
                    ev.append((e0, e1, e2))
                    done += 1
                chunk = min(64 if ev is not None else 256, n - done)
                if chunk <= 0:
                    continue
                if hip_net is not None:
                    az.run_rounds(group, hip_net, chunk, sps)
                else:
                    for _ in range(chunk):
                        for i in range(K):
                            group[i].round(sps[i])
                            evaluate(group, i)
                done += chunk

        def run_rounds(n, ev=None):
            run_rounds_on(pms, n, ev)

        def totals_of(group):
            done = sum(pm.poll()[0] for pm in group)
            cs = [pm.counters() for pm in group]
            return (done, sum(c["sims"] for c in cs), sum(c["evals"] for c in cs), sum(c["cache_hits"] for c in cs),
                    sum(c["cache_misses"] for c in cs))

        def totals():
            return totals_of(pms)

        def live_slots():
            return sum(pm.poll()[1] for pm in pms)

        def take_rows_of(group):
            parts = [pm.take_history_device(dev) for pm in group]
            return tuple(torch.cat([p[i] for p in parts], 0) for i in range(3))

        def take_rows():
            return take_rows_of(pms)

        def stat_vector():
            """what self_play() reads from the PlayManager (game_runner.py:2073-2145) as raw sums: scores [P+1], resign scores
            [P+1], then the ten accumulators of azmi_pm_stat_sums (game length, games, moves, full / fast moves, leaf depth,
            entropy, fast leaf depth, fast entropy, valid moves) — one all-reduce sums them over the node"""
            import numpy as np
            sc = np.sum([pm.scores() for pm in pms], 0)
            rs = np.sum([pm.resign_scores() for pm in pms], 0)
            keys = ("game_length", "games", "moves", "full_moves", "fast_moves", "leaf_depth", "entropy", "fast_leaf_depth", "fast_entropy", "valid_moves")
            sums = [sum(pm.stat_sums()[k] for pm in pms) for k in keys]
            # per-variant tables of a game with variants (play_manager.h:218-275): games and scores per variant ride in the same vector
            var = []
            for vid in range(pms[0].num_tracked_variants()):
                var += [float(sum(pm.variant_games_completed(vid) for pm in pms))] + list(np.sum([pm.variant_scores(vid) for pm in pms], 0))
            return torch.tensor(list(sc) + list(rs) + sums + var, dtype=torch.float64)

    def preroll(run, tot, target_games, what):
        """untimed rounds until `target_games` games have finished (slots de-phased); returns rounds used"""
        used = 0
        while tot()[0] < target_games:
            run(R)
            used += R
            if used > 400 * R:
                raise RuntimeError(f"{what}: pre-roll did not reach {target_games} finished games in {used} rounds")
        return used

    if rank == 0 and not dry:
        sys.stderr.write(f"bench.py: {args.game} S={S} sims={sims} driver={'pipeline' if use_pipe else 'rounds'}: pre-roll ...\n"); sys.stderr.flush()
    # ---- pre-roll (de-phasing), warm-up, timed window --------------------------------------------------------------
    pre_rounds = preroll(run_rounds, totals, int(args.preroll_factor * S), "headline")
    for _ in range(args.warmup):
        run_rounds(R)
        take_rows()
    barrier()
    done0, sims0, evals0, h0, m0 = totals()
    events = [] if not dry else None
    window_rows = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_rounds(R, events)
        window_rows.append(take_rows())          # the step's finished samples leave the engines (device to device)
    done1, sims1, evals1, h1, m1 = totals()
    live = live_slots()
    # the one exchange step: the window's samples go to rank 0 over RCCL/xGMI
    gathered_rows = 0
    gather_kind = None
    if use_dist:
        from alphazero import gather
        parts = [torch.cat([w[i] for w in window_rows], 0) for i in range(3)]
        if dry:
            res = gather.gather_rows_to_rank0(parts, rank, world)      # gloo (CPU rehearsal)
            gather_kind = "torch.distributed (gloo)"
        else:
            # the native exchange behind the C ABI (azmi_gather_rows over librccl: unpadded rows, no torch collective in the data path).
            # Every rank must take the same road: a rank whose communicator cannot be made (librccl missing, ...) says so, and then
            # all of them gather through torch.distributed instead - the line names the road taken and why
            ng, why = None, ""
            try:
                ng = gather.NativeGather(rank, world, local_rank)
            except Exception as e:          # noqa: BLE001 (reported in the line, not swallowed)
                why = str(e)[:160]
            ok = torch.tensor([1.0 if ng is not None else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() > 0:
                res = ng.gather_rows_to_rank0(parts)
                torch.cuda.synchronize()
                gather_kind = "azmi_gather_rows (librccl: ncclAllGather of counts + grouped ncclSend/ncclRecv of unpadded rows)"
            else:
                res = gather.gather_rows_to_rank0(parts, rank, world)
                torch.cuda.synchronize()
                gather_kind = "torch.distributed (nccl = RCCL); the native communicator could not be made on every rank" + (": " + why if why else "")
        if rank == 0:
            gathered_rows = int(res[0].shape[0])
    barrier()
    dt = time.perf_counter() - t0
    local_rows = sum(int(w[0].shape[0]) for w in window_rows)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    cnt = torch.tensor([float(done1 - done0), float(sims1 - sims0), float(evals1 - evals0), float(h1 - h0), float(m1 - m0),
                        float(local_rows), float(S - live)], device=dev, dtype=torch.float64)
    stats = stat_vector().to(dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)      # scores, game length, depth / entropy sums, ... of the whole node
        mn = torch.tensor([float(done1 - done0)], device=dev, dtype=torch.float64)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        min_rank_games = float(mn.item())
    else:
        min_rank_games = float(done1 - done0)
    dt = float(tmax.item())
    n_games, n_sims, n_evals, n_hits, n_miss, n_rows, dead = (float(x) for x in cnt.tolist())
    failure = None
    if args.profile_window:
        pass
    elif dead > 0:
        failure = f"a game stream ran dry inside the timed region ({int(dead)} slots idle)"
    elif min_rank_games < S:
        failure = (f"only {int(min_rank_games)} games finished inside the window on some rank (< {S} concurrent games): "
                   f"raise --steps or --rounds-per-step")
    if failure:
        if rank == 0:
            sys.stderr.write("bench.py: INVALID RUN: " + failure + "\n")
        if use_dist:
            dist.destroy_process_group()
        sys.exit(3)

    if rank == 0:
        hit_rate = n_hits / max(1.0, n_hits + n_miss)
        st = stats.tolist()
        nsc = 3
        games_total = st[2 * nsc + 1]
        node_stats = {"scores": st[:nsc], "resign_scores": st[nsc:2 * nsc],
                      "avg_game_length": st[2 * nsc] / games_total if games_total else 0.0,
                      "avg_leaf_depth": st[2 * nsc + 5] / st[2 * nsc + 3] if st[2 * nsc + 3] else 0.0,
                      "avg_search_entropy": st[2 * nsc + 6] / st[2 * nsc + 3] if st[2 * nsc + 3] else 0.0,
                      "avg_valid_moves": st[2 * nsc + 9] / st[2 * nsc + 2] if st[2 * nsc + 2] else 0.0,
                      "games_since_start": games_total} if not dry else {"sum": st}
        if not dry and len(st) > 2 * nsc + 10:       # StarGambit: per-variant games and win / loss / draw counts over the node
            vs = st[2 * nsc + 10:]
            node_stats["variants"] = {("skirmish", "showdown", "clash", "battle")[i]: {"games": vs[4 * i], "scores": vs[4 * i + 1:4 * i + 4]}
                                      for i in range(len(vs) // 4)}
        out = {
            "metric": "self-play games/sec (whole node), Connect4 @ 800 MCTS sims" if not tafl else f"self-play games/sec (whole node), {'star_gambit_unified' if sg else 'Tawlbwrdd'} @ {sims} MCTS sims",
            "value": n_games / dt,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            **({"profile_window": True} if args.profile_window else {}),
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16": "bf16", "bf16x3": "bf16x3 (split bf16 operands, fp32 accumulate: within 1e-5 of the reference's fp32 outputs)", "fp32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": (f"star_gambit_unified (four variants, 13x13 canvas, 1709 moves), {S} concurrent games/GPU, {sims} sims/move, 4-block/64-ch ResNet (configs/star_gambit_unified.yaml net, spatial + global policy head), {'Gumbel' if args.gumbel else 'PUCT'}, "
                             if sg else f"Tawlbwrdd 11x11, {S} concurrent games/GPU, {sims} sims/move, 4-block/64-ch ResNet (configs/tawlbwrdd.yaml net, spatial head), {'Gumbel' if args.gumbel else 'PUCT'}, "
                             if tafl else f"Connect4, {S} concurrent games/GPU, {sims} sims/move, 6-block/64-ch ResNet (k3, 32 head ch), ")
                            + f"self-play flags of game_runner.py:2018-2041 with playout-cap {'ON (25 sims on 75% of moves)' if args.playout_cap else 'off'}, random-init weights",
                "concurrent_games_per_gpu": S, "engine_shards": K, "sims_per_move": sims, "net": net_kind,
                "driver": ("pipeline (azmi_run_pipeline): a step = %d epochs of <= %d simulations; persistent tree wavefronts + persistent net workgroups, "
                           "moves / game ends / cache inserts between epochs" % (R, spe)) if use_pipe else "rounds (azmi_run_rounds): a step = %d lock-step rounds of every shard" % R,
                "rounds_per_step": R, "preroll_rounds": pre_rounds, "ms_per_round": dt / (args.steps * R) * 1e3,
                "max_cache_size": args.cache, "cache_hit_rate": hit_rate,
                "net_numerics": ("bf16 MFMA operands, fp32 accumulation / residual stream / heads: measured max |delta| against the reference NNArch's fp32 outputs "
                                 "(tests/golden fixtures): pi 2.2e-4, v 5.2e-5 on the random-init Connect4 net (torch bf16 autocast, the reference's own "
                                 "inference arithmetic: 6.4e-4); the 1e-5 tier is the bf16x3 net (split bf16 operands, 4.3e-7), see tier_1e5") if not tafl else
                                "bf16 MFMA operands, fp32 accumulation; measured max |delta| vs the reference NNArch fp32 outputs: pi 6.2e-7, v 3.8e-6 (Tawlbwrdd)",
                "sims_per_s": n_sims / dt, "leaf_evals_per_s": n_evals / dt,
                "games_in_window": n_games, "samples_in_window": n_rows, "samples_gathered": gathered_rows, "sample_gather": gather_kind,
                "node_stats": node_stats,
            },
        }
        if not dry:
            launches = args.steps * R * K
            if use_pipe:        # one launch of each persistent kernel per epoch: the library's own HIP events
                launches = max(1, pipe_acc["epochs"])
                tree_ms = pipe_acc["tree_us"] / launches / 1e3
                nn_ms = pipe_acc["net_us"] / launches / 1e3
            else:
                tree_ms = sum(a.elapsed_time(b) for a, b, _ in events) / max(1, len(events))
                nn_ms = sum(b.elapsed_time(c) for _, b, c in events) / max(1, len(events))
            # a k_leafnet launch evaluates the rows of its shard's eval list (leaves that missed the cache and are not
            # terminal) = the `evals` counter; the torch path and the spatial kernel evaluate the whole slot-indexed batch
            # (the spatial kernels take the eval list too whenever the position cache is on: the listed rows are packed at the front of
            # a dense batch and the workgroups past the live count exit at once)
            # (round 4: the `evals` counter for every HIP net - without a cache the spatial kernel runs every slot's row, terminal leaves
            # included, but only the leaves the search asked for are algorithmic work: counting all slots read 1.7 % high)
            rows_evaluated = n_evals / world if hip_net is not None else float(Se) * launches
            achieved = flop_per_eval * rows_evaluated / dt / 1e12
            per_launch = (flop_per_eval * rows_evaluated / launches) / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
            out["config"].update({"tree_kernel_ms": tree_ms, "net_ms": nn_ms})
            if hip_net is not None:
                out["config"].update(fixture_numerics(az, torch_net, args.game, args.precision, local_rank))
            if use_pipe:
                out["config"]["pipeline_errors"] = list(PIPE_ERRORS)      # recovered pipeline errors of this process (normally none)
                # top-level scalars (a parsed line keeps them): errors met and recovered from in this process, requests
                # the net side gave up on and the boundary sent again
                out["pipeline_errors"] = len(PIPE_ERRORS)
                out["fallbacks"] = len(FALLBACKS)            # 0: every step of this line ran on the driver config.driver names
                out["lost_total"] = int(pipe_acc.get("lost_total", 0))
                out["pipeline_freezes"] = int(pipe_acc.get("freezes", 0))      # polling wavefronts that stood still > 2 ms (the GPU's scheduler; credited, not errors)
                # workgroup-time the net side spends per board in the mix: its workgroups x the net kernel's time / boards evaluated
                # (idle polls included; the same tile alone on the chip: roofline.tiles_alone)
                if pipe_acc["boards"]:
                    out["tile_us_per_board"] = pipe_acc.get("net_wgs", 0) * pipe_acc["net_us"] / pipe_acc["boards"]
            if FALLBACKS and not use_pipe:
                out["fallbacks"] = len(FALLBACKS)
            if use_pipe:      # the host's share: enqueueing an epoch's launches (it runs ahead of the GPU; one synchronisation per step)
                out["config"]["host_enqueue_us_per_epoch"] = pipe_acc["host_us"] / launches
            out["roofline"] = {
                "bound": "mfma", "achieved": per_launch, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": per_launch / MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": ("k_pipe_net (persistent leaf-net workgroups of the asynchronous pipeline: 3- / 6-board tiles of k_leafnet_c4 pulled off the request ring): "
                           "%.0f positions (avg) x %.1f MFLOP per launch = one epoch, %d net workgroups beside %d tree workgroups, %.2f boards per tile"
                           % (rows_evaluated / launches, flop_per_eval / 1e6, pipe_acc.get("net_wgs", 0), pipe_acc.get("tree_wgs", 0), pipe_acc["boards"] / max(1, pipe_acc["tiles"]))) if use_pipe else
                          "%s: %.0f positions (avg) x %.1f MFLOP per launch, %d overlapping launches per round" % ("k_leafnet_sp (+ k_heads_fc)" if tafl else "k_net_move (the leaf-net tiles of k_leafnet_c4 + the round's move step in one launch)", rows_evaluated / launches, flop_per_eval / 1e6, K),
                "per_launch_event_ms": nn_ms,
                "aggregate_achieved": achieved, "aggregate_frac": achieved / MFMA_PEAK_TFLOPS,
                "definition": "achieved = algorithmic FLOPs of ONE launch (positions it evaluated x FLOP per position) / its average duration, "
                              "HIP events on its stream over the timed region (the rocprof average in profiles/ agrees); K launches overlap "
                              "and share the chip with each other and with the tree kernels, so the whole-GPU rate is aggregate_achieved = "
                              "FLOPs of all launches of the region / wall time of the region (rank 0's GPU)",
            }
            if use_pipe and hip_net is not None and world == 1:
                # the same tiles ALONE on the chip (no tree wavefronts beside them, every workgroup with work from the first cycle): 3072
                # rows = 512 six-board tiles, two per CU, 200 launches timed with events - what the kernel itself does, beside what it does
                # in the mix (profiles/r3_leafnet_pmc_3072.csv has the counters of this launch)
                xr = (torch.rand((3072,) + tuple(spec.in_shape), device=dev) < 0.3).float()
                vr = torch.empty((3072, spec.num_players + 1), device=dev); pr = torch.empty((3072, spec.num_moves), device=dev)
                for _ in range(20):
                    hip_net.forward(xr, vr, pr)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
                for _ in range(200):
                    hip_net.forward(xr, vr, pr)
                eb.record(); torch.cuda.synchronize()
                us = ea.elapsed_time(eb) * 1e3 / 200
                tf = 3072 * flop_per_eval / us / 1e6
                out["roofline"]["tiles_alone"] = {"rows": 3072, "us_per_launch": us, "achieved": tf, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                                                  "note": "k_leafnet_c4 (the tile the persistent net workgroups run) as a plain launch of 512 six-board tiles, alone on the chip"}
                out["roofline"]["tiles_alone_frac"] = tf / MFMA_PEAK_TFLOPS      # (flat copies: a parsed line keeps scalars only)
                out["roofline"]["tiles_alone_us_per_launch"] = us
                del xr, vr, pr
                # round 6: the same tile SUSTAINED - 18432 rows = 3072 six-board tiles = six rounds of workgroups per launch: the ramp
                # (every workgroup's first weight chunks from a cold L2 at once) and the tail (a CU idles from its first finished tile to
                # its last) of a one-round launch are amortised, as they are for the pipeline's persistent workgroups - DESIGN 4.5 (v)
                xs = (torch.rand((18432,) + tuple(spec.in_shape), device=dev) < 0.3).float()
                vs = torch.empty((18432, spec.num_players + 1), device=dev); ps_ = torch.empty((18432, spec.num_moves), device=dev)
                for _ in range(5):
                    hip_net.forward(xs, vs, ps_)
                ea.record()
                for _ in range(60):
                    hip_net.forward(xs, vs, ps_)
                eb.record(); torch.cuda.synchronize()
                us2 = ea.elapsed_time(eb) * 1e3 / 60
                tf2 = 18432 * flop_per_eval / us2 / 1e6
                out["roofline"]["tiles_alone_sustained_frac"] = tf2 / MFMA_PEAK_TFLOPS
                out["roofline"]["tiles_alone_sustained_us_per_launch"] = us2
                out["roofline"]["tiles_alone_sustained_rows"] = 18432
                del xs, vs, ps_
            if True:
                # the second kernel of the path, the tree step (HBM side): algorithmic bytes per simulation from SURVEY §8d
                # (select + backup + expand + state + canonical + eval rows: Connect4 1.3 KB with the measured depth 3.5 / 6.8
                # children; Tawlbwrdd 18 KB with depth 1.7 / 113 children, dominated by the dense pi[2662] row)
                sims_rank = n_sims / world
                # StarGambit: 24.3 KB canonical write + 6.8 KB pi row + ~1.7 KB tree (depth ~4, ~20 children) + 0.2 KB state
                b_sim = 33000.0 if sg else 18000.0 if tafl else B_SIM
                tree_launch = (b_sim * sims_rank / launches) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
                out["roofline_tree"] = {
                    "kernel": ("k_pipe_tree<Connect4> (persistent tree wavefronts of the asynchronous pipeline, one launch per epoch)" if use_pipe else "k_cache_insert + k_round_big_sim / _move<StarGambit> + k_compact (one shard-round)" if sg else "k_round_big_sim<Tawlbwrdd> + k_round_big_move + k_compact (one shard-round)" if tafl else "k_cache_insert + k_sim<Connect4> (one shard-round; the move step rides in the net launch)"),
                    "bound": "hbm", "achieved": tree_launch, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": tree_launch / HBM_PEAK_GBS, "traffic": None, "per_launch_event_ms": tree_ms,
                    "aggregate_achieved": b_sim * sims_rank / dt / 1e9, "aggregate_frac": b_sim * sims_rank / dt / 1e9 / HBM_PEAK_GBS,
                    "bytes_per_simulation": b_sim,
                    "note": "latency-bound, not bandwidth-bound: one simulation is a chain of dependent memory round trips; "
                            "the figure to watch is the per-launch time (profiles/)"}
            # HBM-side traffic: not measurable from inside the process; taken from the committed rocprofv3 --pmc summaries of THIS
            # round (profiles/r5_*: every row carries the commit it was collected at; a file of another round is not read), per
            # launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950 (the tree
            # kernel's 32-byte record reads are calibrated instead: FETCH_SIZE = TCC_EA0_RDREQ x 64 B there, profiles/r4_pmc_tree.csv)
            def newest(name):      # this round's file when it exists, else last round's (every row carries its commit: traffic_commit)
                for rnd in ("r6", "r5"):
                    if os.path.exists(os.path.join(ROOT, "profiles", f"{rnd}_{name}")):
                        return rnd
                return "r6"
            suffix = "_stargambit" if sg else "_tawlbwrdd" if tafl else ""
            RND = newest(f"pmc_traffic{suffix}.csv")
            pmc_net = os.path.join(ROOT, "profiles", f"{RND}_pmc_traffic{suffix}.csv")
            pmc_tree = os.path.join(ROOT, "profiles", f"{newest('pmc_tree.csv')}_pmc_tree.csv")
            if use_pipe and os.path.exists(pmc_net):
                # the pipeline's two kernels of an epoch must be co-resident and counter collection serialises dispatches
                # (profiles/r3_pmc_pipeline_probe.txt), so the counters are taken on k_pipe_net ALONE draining a pre-filled ring
                # (scripts/pipe_net_pmc.py): bytes per position there x the positions of an average launch here
                pts = []
                for line in open(pmc_net):
                    f = line.strip().split(",")
                    if "k_pipe_net" in f[0] and float(f[2]) > 0:
                        pts.append((float(f[2]), (2.0 * float(f[3]) + float(f[4])) * 1024.0, f[5]))
                if len(pts) >= 2:
                    (n1, b1, commit), (n2, b2, _) = pts[0], pts[-1]
                    per_pos = (b2 - b1) / (n2 - n1)
                    fixed = b1 - per_pos * n1
                    out["roofline"]["traffic"] = fixed + per_pos * rows_evaluated / launches
                    out["roofline"]["traffic_commit"] = commit
                    out["roofline"]["traffic_note"] = ("L2-to-fabric bytes per k_pipe_net launch = %.2f MB once (the weight image into each XCD's L2) + %.0f B per position x the positions of "
                                                       "an average launch; both from 2 x FETCH_SIZE + WRITE_SIZE of the net kernel ALONE draining %d and %d pre-filled requests "
                                                       "(profiles/%s_pmc_traffic.csv, commit %s): rocprofv3 --pmc serialises dispatches, the co-resident pair cannot be counted"
                                                       % (fixed / 1e6, per_pos, int(n1), int(n2), RND, commit))
            if use_pipe and os.path.exists(pmc_tree):
                # the tree kernel ALONE (every seat EvalType.RANDOM, scripts/pipe_tree_only.py under rocprofv3 --pmc, separate passes):
                # bytes per simulation at the L2-to-fabric counters x the simulations of an average launch here.  That run has no
                # cache probes and no requests (the mix adds ~0.6 KB of shard keys + payload per probe), so it is a floor.
                vals, commit = {}, "?"
                for line in open(pmc_tree):
                    f = line.strip().split(",")
                    if len(f) >= 7 and f[1] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                        vals[f[1]] = float(f[4]); commit = f[6]
                if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                    per_sim = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
                    rt = out["roofline_tree"]
                    rt["traffic"] = per_sim * sims_rank / launches
                    rt["traffic_commit"] = commit
                    rt["traffic_bytes_per_simulation"] = per_sim
                    rt["traffic_over_algorithmic"] = per_sim / 620.0
                    rt["traffic_note"] = ("k_pipe_tree ALONE with EvalType.RANDOM seats (no net kernel beside it, so rocprofv3 --pmc can count it): FETCH_SIZE + WRITE_SIZE = %.0f B per "
                                          "simulation x the simulations of an average launch here; the algorithmic bytes of THAT mode are ~620 B (no canonical planes, no eval rows, no cache probe), "
                                          "ratio %.2f.  Same passes: L2 hit rate %.2f, L1 -> L2 read latency %.0f cycles average, wavefront cycles %.0f %% parked on waits / %.0f %% issuing "
                                          "(profiles/%s_pmc_tree.csv, commit %s)" % (per_sim, per_sim / 620.0,
                                          vals.get("TCC_HIT_sum", 0) / max(1e-9, vals.get("TCC_HIT_sum", 0) + vals.get("TCC_MISS_sum", 0)),
                                          vals.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(1e-9, vals.get("TCP_TCC_READ_REQ_sum", 0)),
                                          100.0 * vals.get("SQ_WAIT_ANY", 0) / max(1e-9, vals.get("SQ_WAVE_CYCLES", 0)), 100.0 * vals.get("SQ_ACTIVE_INST_ANY", 0) / max(1e-9, vals.get("SQ_WAVE_CYCLES", 0)), newest("pmc_tree.csv"), commit))
            elif hip_net is not None and not use_pipe and os.path.exists(pmc_net):
                # the tree phase of a wide game's shard-round is three launches (k_round_big_sim + k_round_big_move + k_compact; the
                # one-kernel form k_round_big / k_round_big_o2 where the split is off): their bytes are SUMMED (the last
                # matching row used to win, which was the move step alone)
                tree_rows = ("k_round_big_sim<", "k_round_big_sim1<", "k_round_big_move<", "k_round_big<", "k_round_big_o2<", "k_compact<", "k_cache_insert<") if tafl else ("k_sim<", "k_round<azmi::Connect4")
                tree_sum, tree_seen = 0.0, []
                for line in open(pmc_net):
                    f = line.strip().split(",")
                    hit = [t for t in tree_rows if t in f[0]]
                    if hit:
                        tree_sum += (2.0 * float(f[2]) + float(f[3])) * 1024.0
                        tree_seen.append(hit[0].rstrip("<"))
                        out["roofline_tree"]["traffic"] = tree_sum
                        out["roofline_tree"]["traffic_commit"] = f[4] if len(f) > 4 else None
                        out["roofline_tree"]["traffic_note"] = "sum over the tree phase's launches of one shard-round (" + " + ".join(tree_seen) + "): 2 x FETCH_SIZE + WRITE_SIZE each, " + os.path.relpath(pmc_net, ROOT)
                    if "k_leafnet" in line or "k_net_move" in line:
                        out["roofline"]["traffic"] = (2.0 * float(f[2]) + float(f[3])) * 1024.0
                        out["roofline"]["traffic_commit"] = f[4] if len(f) > 4 else None
                        out["roofline"]["traffic_note"] = ("bytes per k_leafnet launch at the L2-to-fabric counters (2 x FETCH_SIZE + WRITE_SIZE, "
                                                           + os.path.relpath(pmc_net, ROOT) + ")")
            if world == 1 and hip_net is not None and not args.no_secondary and not tafl and not args.playout_cap:
                # short secondary measurements of the same workload, reported beside the headline, never as it:
                #  (a) playout-cap randomisation at the reference's self-play defaults (fast_mcts_visits 25 on 75 % of
                #      moves, config.py:86,100): what self_play() runs by default
                #  (b) the position cache at the reference's default size (200 000 entries, config.py:197), which separates
                #      the share of the headline that comes from the cache being sized for 288 GB of HBM
                #  (c) 16384 concurrent games instead of 4096: what the chip does when the slots do not limit it (a slot's
                #      simulations are sequential, so at 4096 slots the path is bound by their latency, not by the chip)
                #  (d) the bf16x3 leaf net (split bf16 operands: the 1e-5 tier of the north star) on the lock-step driver
                pms.clear()            # frees the headline engines' HBM before the secondary engines are built
                hip_f32 = None
                for name, S2, cache2, cap2, kind, note in (
                        ("playout_cap_on", S, args.cache, True, "same", "25 sims on 75% of moves, 800 on the rest"),
                        ("cache_200k", S, 200_000, False, "same", "max_cache_size = 200000 (reference default), 800 sims on every move"),
                        ("cache_32m", S, 32_000_000, False, "same", "max_cache_size = 32 M entries (the headline's setting in rounds 1-2 and in the first half of round 3), 800 sims on every move"),
                        ("slots_16384", 16384, args.cache, False, "same", "16384 concurrent games (4 x the headline's), 800 sims on every move"),
                        ("tier_1e5", S, args.cache, False, "x3", "the bf16x3 leaf net (precision='bf16x3': bf16 high + low parts of weights and activations, three MFMAs per product; "
                                                                    "max |delta| vs the reference NNArch's fp32 outputs 4.3e-7 on the random-init fixture, 5.4e-6 on the peaked one: the north star's 1e-5 tier), "
                                                                    "on the asynchronous pipeline (round 4: k_pipe_net<.., X3>, one net workgroup per CU beside the tree workgroups; round 3: lock-step, 1384 games/s), the headline's cache size; "
                                                                    "the plain-fp32 kernels (precision='fp32', any net shape, 7.5e-8) run this workload at 15 games/s"),
                        ("gumbel", S, args.cache, False, "gumbel", "Gumbel AlphaZero roots (gumbel_enabled, m = 16: mcts.cc:233-342), 800 sims on every move; round 4: the pipeline's Gumbel build of the tree kernel "
                                                                   "(lock-step rounds, 4 shards, same box: see DESIGN 8)"),
                        ("two_nets", S, args.cache, False, "two", "the gating shape (play_past, game_runner.py:2184-2332): two different nets behind two model groups, seats swapped by the permutations, one request ring and "
                                                                  "one S3-FIFO (half the entries each) per group; round 4: azmi_run_pipeline_groups")):
                    if os.environ.get("AZMI_BENCH_SECONDARY") and name not in os.environ["AZMI_BENCH_SECONDARY"].split(","):
                        continue
                    sys.stderr.write(f"bench.py: secondary {name} ...\n"); sys.stderr.flush()
                    pipe2 = use_pipe            # (round 4: the bf16x3 tier runs on the pipeline too: k_pipe_net<.., X3>, one net workgroup per CU)
                    K2 = 1 if pipe2 else 4
                    if kind == "x3":
                        hip_f32 = az.HipLeafNet(net, spec, device=local_rank, precision="bf16x3")
                    net2 = hip_f32 if kind == "x3" else hip_net
                    if kind == "two":
                        net2 = [hip_net, az.HipLeafNet(torch_net.random_init(spec, seed=1), spec, device=local_rank)]
                    pms2 = []
                    for i in range(K2):
                        pp2 = selfplay_params(az, S2 // K2, sims, STREAM, cache=cache2 // K2, playout_cap=cap2, gumbel=(kind == "gumbel"))
                        if kind == "two":
                            pp2.model_groups, pp2.seat_perms = [0, 1], [[0, 1], [1, 0]]
                        pms2.append(az.PlayManager(Game(), pp2, seed=977 + 104729 * i, device=local_rank, max_inline=args.inline, history_capacity=(S2 // K2) * 42 * 4))
                    R2 = R if pipe2 else 2048
                    if len(streams) < K2:
                        streams.extend(torch.cuda.Stream(device=dev) for _ in range(K2 - len(streams)))
                    sps2 = [st_.cuda_stream for st_ in streams[:K2]]

                    def run2(n, pms2=pms2, pipe2=pipe2, net2=net2, S2=S2, sps2=sps2):
                        if pipe2:
                            pipe_call(az, pms2[0], net2, n, 256 * S2, sps2[0])
                        else:
                            done2 = 0
                            while done2 < n:
                                if isinstance(net2, list):
                                    az.run_rounds_groups(pms2, net2, min(256, n - done2), sps2[:len(pms2)])
                                else:
                                    az.run_rounds(pms2, net2, min(256, n - done2), sps2[:len(pms2)])
                                done2 += 256
                        for pm2 in pms2:
                            pm2.take_history_device(dev)
                    tot2 = lambda pms2=pms2: totals_of(pms2)
                    short = False
                    pre2 = 0 if short else preroll(run2, tot2, int((1.0 if S2 > S else args.preroll_factor) * S2), name)
                    run2((1 if short else 2) * R2)
                    torch.cuda.synchronize()
                    d0, s0, e0, hh0, mm0 = tot2()
                    t2 = time.perf_counter()
                    k2 = 2 if short else 6
                    for _ in range(k2):
                        run2(R2)
                    d1, s1, e1, hh1, mm1 = tot2()
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t2
                    out["config"][name] = {"games_per_s": (d1 - d0) / dt2, "sims_per_s": (s1 - s0) / dt2, "leaf_evals_per_s": (e1 - e0) / dt2,
                                           "cache_hit_rate": (hh1 - hh0) / max(1, (hh1 - hh0) + (mm1 - mm0)), "steps": k2, "preroll_rounds": pre2,
                                           "games_in_window": d1 - d0, "live_slots": sum(pm.poll()[1] for pm in pms2), "note": note + "; secondary figure"}
                    out["config"]["games_per_s_" + name] = out["config"][name]["games_per_s"]      # (flat copy: a parsed line keeps scalars only)
                    out["config"]["hit_rate_" + name] = out["config"][name]["cache_hit_rate"]
                    if short:      # too short for games to finish from a cold start: the rate follows from the simulations
                        out["config"][name]["games_per_s"] = ((s1 - s0) / dt2) / (n_sims / n_games)
                        out["config"][name]["note"] += f"; games/s = this window's simulations/s / the headline's {n_sims / n_games:.0f} simulations per game (a cold-start window of {dt2:.1f}s)"
                    del pms2, tot2, run2
            if world == 1 and not args.no_cpu_baseline and not args.gumbel:
                sys.stderr.write("bench.py: cpu baseline ...\n"); sys.stderr.flush()
                # the reference's own cache size: 200 000 entries (config.py:197); Tawlbwrdd: off, as in its GPU run
                out["cpu_baseline"] = cpu_baseline(az, sims, args.cpu_seconds, S, hip_net, 200_000 if (sg or not tafl) else 0, game=args.game,
                                                   sims_per_game=(n_sims / n_games) if n_games else None)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
