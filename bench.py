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
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--warmup", type=int, default=36000)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--net", choices=["hip", "torch"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def selfplay_params(az, games, sims, stream_games):
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
    pp.playout_cap_randomization = False
    pp.resign_percent = 0.02
    pp.resign_playthrough_percent = 0.20
    return pp


def cpu_baseline(az, sims, seconds):
    """The oracle (CPU restatement of the reference PlayManager, one thread) on a bounded sample of the
    same search: Connect4, 800 sims/move, same self-play flags, EvalType.RANDOM (no net) -> an upper
    bound for the reference's tree side on one host core."""
    import oracle_api as orc
    pp = selfplay_params(az, 1, sims, 1)
    pp.eval_type = [1, 1]
    pp.history_enabled = True
    games, t0, n_sims = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        o = orc.PlayManager(orc.GAME_CONNECT4, pp, 1000 + games, per_slot_rng=False, record_moves=False)
        o.run()
        games += o.games_completed()
        n_sims += o.counters()["sims"]
    dt = time.perf_counter() - t0
    return {"value": games / dt, "unit": "games/s", "cores": 1, "kind": "port",
            "sample": f"{games} Connect4 self-play games x {sims} sims, oracle PlayManager, EvalType.RANDOM evaluator "
                      f"(no net), 1 thread, {dt:.1f}s; {n_sims / dt / 1e6:.3f} Msims/s",
            "sims_per_s": n_sims / dt}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import alphazero as az
    from alphazero import torch_net

    S, sims = args.games, args.sims
    total_rounds = args.warmup + args.steps
    stream_games = max(S, 8 * S)  # stream pool, play_manager_bench.cc:171-181
    pp = selfplay_params(az, S, sims, stream_games)
    pm = az.PlayManager(az.Connect4GS(), pp, seed=20240601 + 7919 * rank, device=local_rank)
    canon, v_buf, pi_buf = pm.io_tensors()

    spec = torch_net.connect4_spec()
    net = torch_net.random_init(spec, seed=0).to(dev)
    net_kind = args.net or "hip"
    if net_kind == "hip":
        hip_net = az.HipLeafNet(net, spec, max_batch=S, device=local_rank)

        def evaluate(stream_ptr):
            hip_net.forward(canon, v_buf, pi_buf, stream_ptr)
    else:
        net = net.to(memory_format=torch.channels_last)

        def evaluate(stream_ptr):
            v, pi = net.process(canon, amp_dtype=torch.bfloat16)
            v_buf.copy_(v)
            pi_buf.copy_(pi)

    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    def run_rounds(n, ev=None):
        for i in range(n):
            if ev is not None and (i & 15) == 0:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(stream); pm.round(sp); e1.record(stream); evaluate(sp); e2.record(stream)
                ev.append((e0, e1, e2))
            else:
                pm.round(sp)
                evaluate(sp)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_rounds(args.warmup)
    barrier()
    done0, _ = pm.poll(sp)
    c0 = pm.counters()
    events = []
    t0 = time.perf_counter()
    run_rounds(args.steps, events)
    # the one exchange step: finished samples of this window go to rank 0 over RCCL/xGMI
    done1, live = pm.poll(sp)
    gathered_rows = 0
    if world > 1:
        from alphazero import gather
        gathered_rows = gather.gather_history_to_rank0(pm, dev, rank, world)
    barrier()
    dt = time.perf_counter() - t0
    c1 = pm.counters()
    tmax = torch.tensor([dt], device=dev)
    games = torch.tensor([float(done1 - done0), float(c1["sims"] - c0["sims"]), float(c1["evals"] - c0["evals"])], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(games, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    n_games, n_sims, n_evals = (float(x) for x in games.tolist())

    if rank == 0:
        tree_ms = sum(a.elapsed_time(b) for a, b, _ in events) / max(1, len(events))
        nn_ms = sum(b.elapsed_time(c) for _, b, c in events) / max(1, len(events))
        flops = FLOP_PER_EVAL * S  # the net evaluates all S slot rows every round (fixed-shape batch)
        achieved = flops / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
        out = {
            "metric": "self-play games/sec (whole node), Connect4 @ 800 MCTS sims",
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
                "workload": f"Connect4, {S} concurrent games/GPU, {sims} sims/move, 6-block/64-ch ResNet (k3, 32 head ch), "
                            f"self-play flags of game_runner.py:2018-2041 with playout-cap off, random-init weights",
                "concurrent_games_per_gpu": S, "sims_per_move": sims, "net": net_kind,
                "sims_per_s": n_sims / dt, "leaf_evals_per_s": n_evals / dt,
                "tree_kernel_ms": tree_ms, "net_ms": nn_ms, "samples_gathered": gathered_rows,
                "games_in_window": n_games,
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": "leaf-net forward, %d positions x %.1f MFLOP per launch" % (S, FLOP_PER_EVAL / 1e6),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(az, sims, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
