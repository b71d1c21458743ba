#!/usr/bin/env python3
"""Self-play throughput bench — BASELINE.json metric:
"self-play games/sec (whole node), Connect4 @ 800 MCTS sims".

A *step* is one engine round: every one of the `concurrent_games` slots on this GPU finishes one
MCTS simulation (backup of the previous leaf, maybe a move, descent + expansion of the next leaf:
k_round) and the 6-block/64-channel ResNet evaluates the resulting leaf batch.  Inputs live in HBM;
no host round-trip happens inside the timed region.  Workload = BASELINE configs[1]
(Connect4, 4096 concurrent games, 800 sims, 6b64c net, random-init weights, synthetic = self-generated
positions), one engine + one weight replica per GPU (weak scaling), samples gathered to rank 0 with
RCCL at the end of the timed region.

Prints ONE JSON line on rank 0 (see README of the task contract).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "alphazero-pybind11_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_EVAL = 37.7e6     # Connect4 6b64c k3 (conv + linear MACs x 2)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30000, help="timed rounds; the default spans > 2 game lengths (13 k rounds each) so that the opening/endgame phase mix of the initially synchronised games averages out")
    ap.add_argument("--warmup", type=int, default=30000)
    ap.add_argument("--game", choices=["connect4", "tawlbwrdd"], default="connect4",
                    help="connect4 = BASELINE configs[1] (the headline); tawlbwrdd = configs[2] (2048 games, 400 sims, YAML net)")
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short playout-cap-on measurement reported in config")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


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


def cpu_baseline(az, sims, seconds, threads=None):
    """The oracle (CPU restatement of the reference PlayManager) on a bounded sample of the same search: Connect4,
    800 sims/move, same self-play flags, EvalType.RANDOM (no net), on `threads` host threads — the reference's default
    worker count is cores - 1 (config.py:439-441).  Each thread runs independent one-slot PlayManagers back to back
    (the C call releases the GIL), which is an upper bound for the reference's tree side: no queue, no mutex, no net."""
    import threading
    import oracle_api as orc
    if threads is None:
        cores = len(os.sched_getaffinity(0))
        try:                                   # a cgroup CPU quota is the real core count of a container
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = min(cores, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
        threads = max(1, cores - 1)
    pp = selfplay_params(az, 1, sims, 1)
    pp.eval_type = [1, 1]
    pp.history_enabled = True
    totals = [[0, 0] for _ in range(threads)]
    t0 = time.perf_counter()

    def worker(t):
        k = 0
        while time.perf_counter() - t0 < seconds:
            o = orc.PlayManager(orc.GAME_CONNECT4, pp, 1000 + 7919 * t + k, per_slot_rng=False, record_moves=False)
            o.run()
            totals[t][0] += o.games_completed()
            totals[t][1] += o.counters()["sims"]
            k += 1

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    games = sum(a for a, _ in totals); n_sims = sum(b for _, b in totals)
    return {"value": games / dt, "unit": "games/s", "cores": threads, "kind": "port",
            "sample": f"{games} Connect4 self-play games x {sims} sims, oracle PlayManager, EvalType.RANDOM evaluator "
                      f"(no net), {threads} threads (host cores - 1), {dt:.1f}s; {n_sims / dt / 1e6:.3f} Msims/s total, "
                      f"{n_sims / dt / threads / 1e6:.3f} Msims/s per thread",
            "sims_per_s": n_sims / dt, "per_thread_games_per_s": games / dt / threads}


def main():
    args = parse()
    if args.hwq:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hwq)   # must be set before the HIP runtime starts
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # AZMI_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL init, barriers, sample gather, reductions) with whatever world size
    # the launcher gave, 1 included: lets a one-GPU box exercise it under torchrun
    use_dist = world > 1 or os.environ.get("AZMI_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import alphazero as az
    from alphazero import torch_net

    tafl = args.game == "tawlbwrdd"
    if args.games is None: args.games = 2048 if tafl else 4096
    if args.sims is None: args.sims = 400 if tafl else 800
    if args.engines is None: args.engines = 4      # measured: Connect4 1/2/4/8 shards and Tawlbwrdd 2/4/8 shards both peak at 4
    if args.cache is None: args.cache = 0 if tafl else 32_000_000      # Tawlbwrdd: measured 5 % hit rate with 2 M entries and 16 % fewer games/s, so off
    Game = az.TawlbwrddGS if tafl else az.Connect4GS
    flop_per_eval = 93.1e6 if tafl else FLOP_PER_EVAL
    S, sims, K = args.games, args.sims, args.engines
    assert S % K == 0
    Se = S // K                       # slots per engine shard
    # stream pool (play_manager_bench.cc:171-181: games_to_play = 8 x concurrent), widened when the run is
    # long enough that a slot could finish more than 8 games: a dry stream would idle slots and void the number
    rounds_per_game = 1000 if args.playout_cap else 4000        # conservative lower bounds (measured 2900 / 13000)
    if tafl: rounds_per_game = 8000
    stream_games = Se * max(8, -(-(args.warmup + args.steps) // rounds_per_game))
    # K engine shards of S/K slots, one HIP stream each: while one shard's leaf batch is on the matrix
    # cores another shard's tree kernel runs on the CUs the net leaves free.
    pms, streams = [], []
    for i in range(K):
        pp = selfplay_params(az, Se, sims, stream_games, cache=args.cache // K, playout_cap=args.playout_cap, gumbel=args.gumbel)
        pms.append(az.PlayManager(Game(), pp, seed=20240601 + 7919 * rank + 104729 * i, device=local_rank, max_inline=args.inline,
                                  history_capacity=(400_000 // K if tafl else 0)))
        streams.append(torch.cuda.Stream(device=dev))
    sps = [st.cuda_stream for st in streams]
    io = [pm.io_tensors() for pm in pms]

    spec = torch_net.tawlbwrdd_spec() if tafl else torch_net.connect4_spec()
    net = torch_net.random_init(spec, seed=0).to(dev)
    net_kind = args.net or "hip"
    hip_net = az.HipLeafNet(net, spec, device=local_rank) if net_kind == "hip" else None
    if net_kind == "torch":
        net = net.to(memory_format=torch.channels_last)

    def evaluate(i):
        canon, v_buf, pi_buf = io[i]
        if hip_net is not None:
            pms[i].net_forward(hip_net, sps[i])
        else:
            with torch.cuda.stream(streams[i]):
                v, pi = net.process(canon, amp_dtype=torch.bfloat16)
                v_buf.copy_(v)
                pi_buf.copy_(pi)

    def run_rounds(n, ev=None):
        """n rounds of every shard. With the HIP net the loop is the native driver (azmi_run_rounds);
        every 64th round is launched from here with HIP events around the two kernels of shard 0."""
        done = 0
        while done < n:
            if ev is not None:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                for i in range(K):
                    if i == 0:
                        e0.record(streams[0])
                    pms[i].round(sps[i])
                    if i == 0:
                        e1.record(streams[0])
                    evaluate(i)
                    if i == 0:
                        e2.record(streams[0])
                ev.append((e0, e1, e2))
                done += 1
            chunk = min(64 if ev is not None else 256, n - done)
            if chunk <= 0:
                continue
            if hip_net is not None:
                az.run_rounds(pms, hip_net, chunk, sps)
            else:
                for _ in range(chunk):
                    for i in range(K):
                        pms[i].round(sps[i])
                        evaluate(i)
            done += chunk

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def totals():
        done = sum(pm.poll()[0] for pm in pms)
        cs = [pm.counters() for pm in pms]
        return done, sum(c["sims"] for c in cs), sum(c["evals"] for c in cs), sum(c["cache_hits"] for c in cs), sum(c["cache_misses"] for c in cs)

    run_rounds(args.warmup)
    barrier()
    done0, sims0, evals0, h0, m0 = totals()
    events = []
    t0 = time.perf_counter()
    run_rounds(args.steps, events)
    # the one exchange step: finished samples of this window go to rank 0 over RCCL/xGMI
    done1, sims1, evals1, h1, m1 = totals()
    live = sum(pm.poll()[1] for pm in pms)
    if live != S:
        raise RuntimeError(f"game stream ran dry inside the timed region ({live} of {S} slots live): raise stream_games")
    hit_rate = (h1 - h0) / max(1, (h1 - h0) + (m1 - m0))
    gathered_rows = 0
    if use_dist:
        from alphazero import gather
        for pm in pms:
            gathered_rows += gather.gather_history_to_rank0(pm, dev, rank, world)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    games = torch.tensor([float(done1 - done0), float(sims1 - sims0), float(evals1 - evals0)], device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(games, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    n_games, n_sims, n_evals = (float(x) for x in games.tolist())

    if rank == 0:
        tree_ms = sum(a.elapsed_time(b) for a, b, _ in events) / max(1, len(events))
        nn_ms = sum(b.elapsed_time(c) for _, b, c in events) / max(1, len(events))
        # k_leafnet evaluates the Se rows of one shard per launch; K launches (one per shard) are in flight
        # at once and share the chip with each other and with the tree kernels, so a per-launch event
        # interval double-counts shared CUs.  `achieved` is therefore the algorithmic FLOPs of ALL net
        # launches of the timed region / the region's wall time (a lower bound on the kernel's own rate);
        # the per-launch HIP-event interval is reported next to it.
        launches = args.steps * K
        # Connect4 + HIP net: a launch evaluates only the rows of the engine's eval list (leaves that missed the
        # cache and are not terminal) = the `evals` counter; other paths evaluate the whole slot-indexed batch
        rows_evaluated = n_evals if (hip_net is not None and not tafl) else float(Se) * launches
        achieved = flop_per_eval * rows_evaluated / dt / 1e12
        per_launch = (flop_per_eval * rows_evaluated / launches) / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else achieved
        out = {
            "metric": "self-play games/sec (whole node), Connect4 @ 800 MCTS sims" if not tafl else f"self-play games/sec (whole node), Tawlbwrdd @ {sims} MCTS sims",
            "value": n_games / dt,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": (f"Tawlbwrdd 11x11, {S} concurrent games/GPU, {sims} sims/move, 4-block/64-ch ResNet (configs/tawlbwrdd.yaml net, spatial head), {'Gumbel' if args.gumbel else 'PUCT'}, "
                             if tafl else f"Connect4, {S} concurrent games/GPU, {sims} sims/move, 6-block/64-ch ResNet (k3, 32 head ch), ")
                            + f"self-play flags of game_runner.py:2018-2041 with playout-cap {'ON (25 sims on 75% of moves)' if args.playout_cap else 'off'}, random-init weights",
                "concurrent_games_per_gpu": S, "engine_shards": K, "sims_per_move": sims, "net": net_kind,
                "max_cache_size": args.cache, "cache_hit_rate": hit_rate,
                "sims_per_s": n_sims / dt, "leaf_evals_per_s": n_evals / dt,
                "tree_kernel_ms": tree_ms, "net_ms": nn_ms, "samples_gathered": gathered_rows,
                "games_in_window": n_games,
            },
            "roofline": {
                "bound": "mfma", "achieved": per_launch, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": per_launch / MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": "%s: %.0f positions (avg) x %.1f MFLOP per launch, %d overlapping launches per round" % ("k_leafnet_spatial" if tafl else "k_leafnet", rows_evaluated / launches, flop_per_eval / 1e6, K),
                "per_launch_event_ms": nn_ms,
                "aggregate_achieved": achieved, "aggregate_frac": achieved / MFMA_PEAK_TFLOPS,
                "definition": "achieved = algorithmic FLOPs of ONE launch (positions it evaluated x FLOP per position) / its average duration, "
                              "HIP events on its stream over the timed region (the rocprof average in profiles/ agrees); K launches overlap "
                              "and share the chip with each other and with the tree kernels, so the whole-GPU rate is aggregate_achieved = "
                              "FLOPs of all launches of the region / wall time of the region",
            },
        }
        # HBM-side traffic of the dominant kernel: not measurable from inside the process; taken from the committed
        # rocprofv3 --pmc summary of this same command (scripts/gpu_pmc.sh -> profiles/r1_pmc_traffic.csv), per launch,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_traffic.csv")
        if not tafl:
            # the second kernel of the path, the tree step (HBM side): algorithmic bytes per simulation from SURVEY §8d
            # (select + backup + expand + state + canonical + eval rows = 1.3 KB with the measured depth 3.5 / 6.8 children)
            b_sim = 1300.0
            tree_launch = (b_sim * n_sims / launches) / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0
            out["roofline_tree"] = {
                "kernel": "k_cache_insert + k_round<Connect4> (one shard-round)", "bound": "hbm", "achieved": tree_launch, "peak": 8000.0, "unit": "GB/s",
                "frac": tree_launch / 8000.0, "traffic": None, "per_launch_event_ms": tree_ms,
                "aggregate_achieved": b_sim * n_sims / dt / 1e9, "aggregate_frac": b_sim * n_sims / dt / 1e9 / 8000.0,
                "note": "latency-bound, not bandwidth-bound: one simulation is a chain of ~8 dependent memory round trips; "
                        "the figure to watch is the per-launch time (profiles/r1_kernel_stats.csv)"}
        if not tafl and hip_net is not None and os.path.exists(pmc):
            for line in open(pmc):
                if "k_round<azmi::Connect4" in line:
                    f = line.strip().split(",")
                    out["roofline_tree"]["traffic"] = (2.0 * float(f[-2]) + float(f[-1])) * 1024.0
                if line.startswith("k_leafnet"):
                    f = line.strip().split(",")
                    out["roofline"]["traffic"] = (2.0 * float(f[-2]) + float(f[-1])) * 1024.0
                    out["roofline"]["traffic_note"] = ("bytes per k_leafnet launch at the L2-to-fabric counters (2 x FETCH_SIZE + WRITE_SIZE, "
                                                       "profiles/r1_pmc_traffic.csv); ~8 x the 0.94 MB weight image: each of the 8 XCD L2s "
                                                       "fetches it once per launch, served from Infinity Cache")
        if world == 1 and hip_net is not None and not args.playout_cap and not args.no_secondary and not tafl:
            # the same workload with playout-cap randomisation at the reference's self-play defaults
            # (fast_mcts_visits 25 on 75 % of moves, config.py:86,100): reported beside the headline, never as it
            w2, k2 = 12000, 6000
            pms2 = []
            for i in range(K):
                pp = selfplay_params(az, Se, sims, Se * 32, cache=args.cache // K, playout_cap=True)
                pms2.append(az.PlayManager(az.Connect4GS(), pp, seed=977 + i, device=local_rank, max_inline=args.inline))
            az.run_rounds(pms2, hip_net, w2, sps)
            torch.cuda.synchronize()
            d0 = sum(pm.poll()[0] for pm in pms2); s0 = sum(pm.counters()["sims"] for pm in pms2)
            t2 = time.perf_counter()
            az.run_rounds(pms2, hip_net, k2, sps)
            d1 = sum(pm.poll()[0] for pm in pms2); s1 = sum(pm.counters()["sims"] for pm in pms2)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            live2 = sum(pm.poll()[1] for pm in pms2)
            out["config"]["playout_cap_on"] = {"games_per_s": (d1 - d0) / dt2, "sims_per_s": (s1 - s0) / dt2, "steps": k2, "warmup": w2,
                                               "live_slots": live2, "note": "25 sims on 75% of moves, 800 on the rest; secondary figure"}
            del pms2
        if world == 1 and not args.no_cpu_baseline and not tafl:
            out["cpu_baseline"] = cpu_baseline(az, sims, args.cpu_seconds)
            # the same 6b64c net on the host cores (fp32, PyTorch's default intra-op threads, batch 256): the leaf
            # evaluations a CPU-only run of this workload would also have to pay for
            cpu_net = torch_net.random_init(spec, seed=0).eval()
            xb = torch.zeros((256,) + tuple(int(d) for d in io[0][0].shape[1:]))
            torch.set_num_threads(max(1, out["cpu_baseline"]["cores"]))     # the same cores the tree sample used
            with torch.no_grad():
                cpu_net.process(xb)
                t3, reps = time.perf_counter(), 0
                while time.perf_counter() - t3 < 3.0:
                    cpu_net.process(xb); reps += 1
            evals_cpu = reps * 256 / (time.perf_counter() - t3)
            evals_per_game = n_evals / max(n_games, 1.0)
            cb = out["cpu_baseline"]
            tree_rate = cb["value"]
            net_rate = evals_cpu / evals_per_game
            # the SAME workload on the host cores = the tree search AND its leaf evaluations: both pieces are measured
            # (bounded samples), the combination is serial time per game on the same cores
            cb["tree_only_games_per_s"] = tree_rate
            cb["net_on_cpu"] = {"evals_per_s": evals_cpu, "threads": torch.get_num_threads(), "evals_per_game": evals_per_game,
                                "games_per_s_bound": net_rate}
            cb["value"] = 1.0 / (1.0 / tree_rate + 1.0 / net_rate)
            cb["sample"] += (f"; leaf net: 6b64c fp32 forward on the same host, batch 256, 3 s = {evals_cpu:.0f} evals/s, "
                             f"{evals_per_game:.0f} net evaluations per game (the GPU run's count, cache included); "
                             f"value = 1 / (1/{tree_rate:.0f} + {evals_per_game:.0f}/{evals_cpu:.0f}) games/s")
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
