"""-m gpu: tier T3 of SURVEY §8c — NN-in-the-loop tree parity.  The device fast path (k_sim / k_round -> eval list -> leaf net
kernels writing the slot-indexed (v, pi) rows in place -> position cache insert of device rows -> next round) plays, with the
HIP leaf net as its evaluator, exactly the games the ORACLE PlayManager plays when its evaluator callback sends each
leaf's canonical planes through the very same HIP net (neural_net.py:800-823 `process`, one leaf at a time): move for move
and visit count for visit count.  That holds bit for bit because a position's (v, pi) does not depend on the batch it is
evaluated in (the kernels reduce in a batch-independent order), so any difference is a tree / hand-off / cache bug, not
bf16 noise.  Cases: Connect4 (split rounds, fused net + move step, device cache on), Tawlbwrdd (wide-game engine, spatial
net), two different nets routed by model group on a Tafl game (the row list of the spatial kernels), and the fp32 net with
the cache on and one inline simulation per round (a cache hit must not be overwritten by a whole-batch recompute)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net_eval(hip):
    dev = torch.device("cuda", 0)

    def f(canon):
        v, pi = hip.process(torch.from_numpy(np.ascontiguousarray(canon)).to(dev))
        torch.cuda.synchronize()
        return v.cpu().numpy(), pi.cpu().numpy()
    return f


def _device_games(az, game, pp, seed, nets, max_inline=0):
    pm = az.PlayManager(game, pp, seed=seed, log_moves=True, max_inline=max_inline)
    st = torch.cuda.Stream()
    many = isinstance(nets, (list, tuple))
    while pm.remaining_games() > 0:
        if many:
            az.run_rounds_groups([pm], list(nets), 64, [st.cuda_stream])
        else:
            az.run_rounds([pm], nets, 64, [st.cuda_stream])
        if pm.poll(st.cuda_stream)[1] == 0:
            break
    torch.cuda.synchronize()
    return pm, pm.move_log()


def _check_slots(az, oracle, gid, pp, seed, rows, counts, slots, evaluator=None, group_evaluator=None):
    checked = 0
    for s in slots:
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games, one.max_batch_size = 1, 1, 1
        o = oracle.PlayManager(gid, one, oracle.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
        if group_evaluator is not None:
            o.run_groups(group_evaluator)
        else:
            o.run(evaluator)
        orows, ocounts = o.moves()
        sel = (rows[:, 0] == s) & (rows[:, 1] == 0)                 # the slot's first game (a one-slot oracle plays one game)
        assert sel.sum() == len(orows) > 0, (s, int(sel.sum()), len(orows))
        assert np.array_equal(rows[sel][:, 2:6], orows[:, 2:6]), f"slot {s}: moves differ from the oracle driven by the same net"
        assert np.array_equal(counts[sel], ocounts), f"slot {s}: visit counts differ"
        assert np.array_equal(rows[sel][:, 6:], orows[:, 6:]), f"slot {s}: pcg32 stream position differs"
        checked += len(orows)
    return checked


def _selfplay_params(az, S, sims, cache):
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = S, S, S
    pp.mcts_visits = [sims, sims]
    pp.model_groups = [0, 0]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.epsilon, pp.mcts_root_temp, pp.root_fpu_zero, pp.shaped_dirichlet = 0.25, 1.25, True, True
    pp.policy_target_pruning, pp.history_enabled = True, True
    pp.start_temp, pp.final_temp, pp.temp_decay_half_life = 1.0, 0.2, 10.0
    pp.max_cache_size = cache
    return pp


def test_connect4_fast_path_equals_the_oracle_driven_by_the_same_net(oracle):
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.connect4_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=11), spec)
    S, seed = 96, 4242
    pp = _selfplay_params(az, S, 120, cache=1 << 16)
    pm, (rows, counts) = _device_games(az, az.Connect4GS(), pp, seed, hip)
    assert pm.games_completed() == S
    c = pm.counters()
    assert c["cache_hits"] > 0 and 0 < c["evals"] < c["sims"]        # the cache took part; the net saw only the misses
    n = _check_slots(az, oracle, oracle.GAME_CONNECT4, pp, seed, rows, counts, (0, 1, 17, 40, 95), evaluator=_net_eval(hip))
    assert n > 40


def test_connect4_fast_path_without_cache_and_with_playout_cap(oracle):
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.connect4_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=12), spec)
    S, seed = 40, 77
    pp = _selfplay_params(az, S, 80, cache=0)
    pp.playout_cap_randomization, pp.playout_cap_depth, pp.playout_cap_percent = True, 12, 0.5
    pm, (rows, counts) = _device_games(az, az.Connect4GS(), pp, seed, hip)
    assert pm.games_completed() == S
    _check_slots(az, oracle, oracle.GAME_CONNECT4, pp, seed, rows, counts, (0, 7, 39), evaluator=_net_eval(hip))


def test_tawlbwrdd_fast_path_equals_the_oracle_driven_by_the_same_net(oracle):
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.tawlbwrdd_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=13), spec)
    S, seed = 6, 99
    pp = _selfplay_params(az, S, 24, cache=1 << 12)
    pm, (rows, counts) = _device_games(az, az.TawlbwrddGS(), pp, seed, hip)
    assert pm.games_completed() == S
    _check_slots(az, oracle, oracle.GAME_TAWLBWRDD, pp, seed, rows, counts, (0, 5), evaluator=_net_eval(hip))


def test_tawlbwrdd_with_the_bf16x3_net_equals_the_oracle_driven_by_the_same_net(oracle):
    """round 4: the spatial tile's bf16x3 tier (csrc/leafnet_sp.h, Geo<.., SPLIT>) as the engine's evaluator"""
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.tawlbwrdd_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=13), spec, precision="bf16x3")
    S, seed = 6, 99
    pp = _selfplay_params(az, S, 24, cache=1 << 12)
    pm, (rows, counts) = _device_games(az, az.TawlbwrddGS(), pp, seed, hip)
    assert pm.games_completed() == S
    _check_slots(az, oracle, oracle.GAME_TAWLBWRDD, pp, seed, rows, counts, (0, 5), evaluator=_net_eval(hip))


def test_two_nets_routed_by_model_group_on_a_tafl_game(oracle):
    """gating shape (play_past, game_runner.py:2184-2332) on Brandubh: two DIFFERENT spatial nets, both seatings.  Each net
    must only write the rows of its own model group's leaves (the spatial kernels used to evaluate the whole
    slot-indexed batch, so the last net overwrote the other group's answers)."""
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.brandubh_spec()
    nets = [az.HipLeafNet(torch_net.random_init(spec, seed=21), spec), az.HipLeafNet(torch_net.random_init(spec, seed=22), spec)]
    evs = [_net_eval(n) for n in nets]
    S, seed = 8, 5
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = S, S, S
    pp.mcts_visits = [20, 28]
    pp.model_groups, pp.seat_perms = [0, 1], [[0, 1], [1, 0]]
    pp.cpuct, pp.fpu_reduction, pp.history_enabled = 1.25, 0.25, True
    pp.start_temp = pp.final_temp = 0.5
    pm, (rows, counts) = _device_games(az, az.BrandubhGS(), pp, seed, nets)
    assert pm.games_completed() == S
    _check_slots(az, oracle, oracle.GAME_BRANDUBH, pp, seed, rows, counts, (0, 1, 6, 7), group_evaluator=lambda g, c: evs[g](c))
    # and the two nets really disagree on these positions: swapping them changes the games
    pm2, (rows2, _) = _device_games(az, az.BrandubhGS(), pp, seed, nets[::-1])
    assert not np.array_equal(rows[:, 2], rows2[:, 2]) or len(rows) != len(rows2)


def test_fp32_net_with_cache_and_one_inline_simulation(oracle):
    """with the position cache on, a cache hit that ends a round (max_inline = 1) leaves its (pi, v) in
    the slot's rows; the fp32 net path must not recompute those rows from the slot's stale planes."""
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.connect4_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=14), spec, precision="fp32")
    S, seed = 24, 31
    pp = _selfplay_params(az, S, 60, cache=1 << 14)
    pm, (rows, counts) = _device_games(az, az.Connect4GS(), pp, seed, hip, max_inline=1)
    assert pm.games_completed() == S and pm.counters()["cache_hits"] > 0
    _check_slots(az, oracle, oracle.GAME_CONNECT4, pp, seed, rows, counts, (0, 11, 23), evaluator=_net_eval(hip))


def test_bf16x3_net_in_the_loop_equals_the_oracle_driven_by_the_same_net(oracle):
    """the 1e-5 tier on the matrix cores (precision="bf16x3", csrc/leafnet_c4.h SPLIT) as the engine's evaluator: lock-step rounds
    with the eval list (the split tiles are not fused with the move step), device cache on; the games are the oracle's when its
    evaluator sends each leaf through the same net."""
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.connect4_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=15), spec, precision="bf16x3")
    S, seed = 64, 2024
    pp = _selfplay_params(az, S, 80, cache=1 << 14)
    pm, (rows, counts) = _device_games(az, az.Connect4GS(), pp, seed, hip)
    assert pm.games_completed() == S and pm.counters()["cache_hits"] > 0
    assert az.pipeline_supported(az.PlayManager(az.Connect4GS(), pp, seed=1), hip)      # round 4: the pipeline runs this tier too (test_gpu_pipeline.py)
    _check_slots(az, oracle, oracle.GAME_CONNECT4, pp, seed, rows, counts, (0, 31, 63), evaluator=_net_eval(hip))
