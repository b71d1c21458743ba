"""-m gpu: alphazero.selfplay.self_play — the device counterpart of GameRunner.run + self_play()'s read-out:
for a given PlayParams and evaluator, the multiset of (canonical, v, pi) rows and the counters equal the oracle's."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows_key(c, v, p):
    rows = np.concatenate([c.reshape(len(c), -1), v, p], 1)
    return sorted(r.tobytes() for r in np.ascontiguousarray(rows, np.float32))


def test_random_evaluator_rows_and_counters_equal_the_oracle(oracle):
    import alphazero as az
    from alphazero import selfplay
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 12, 12, 12
    pp.mcts_visits = [40, 40]
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.history_enabled = True
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.epsilon, pp.mcts_root_temp, pp.shaped_dirichlet, pp.policy_target_pruning = 0.25, 1.25, True, True
    pp.playout_cap_randomization, pp.playout_cap_depth, pp.playout_cap_percent = True, 10, 0.5
    seed, K = 555, 3
    res, (c, v, p) = selfplay.self_play(az.Connect4GS, pp, engines=K, seed=seed)
    assert res.games == 12 and res.samples == len(c) and abs(sum(res.win_rates) - 1) < 1e-6
    want, scores, length, depth, ent, fdepth, fent, full, fast, moves, valid = [], np.zeros(3), 0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0.0
    for k in range(K):
        lo, hi = 12 * k // K, 12 * (k + 1) // K
        for s in range(hi - lo):
            one = az.PlayParams(); one.__dict__.update(pp.__dict__)
            one.games_to_play, one.concurrent_games = 1, 1
            o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(selfplay.shard_seed(seed, k), s), per_slot_rng=False)
            o.run()
            oc, ov, op = o.history()
            want += _rows_key(oc, ov, op)
            scores += o.scores()
            rows, _ = o.moves()
            nfast = int(rows[:, 5].sum()); nfull = len(rows) - nfast
            length += len(rows); full += nfull; fast += nfast; moves += len(rows)
            st = o.stats()      # avg_game_length, avg_leaf_depth, avg_search_entropy, fast leaf depth, fast entropy, moves/turn, valid moves
            depth += float(st[1]) * nfull; ent += float(st[2]) * nfull
            fdepth += float(st[3]) * nfast; fent += float(st[4]) * nfast
            valid += float(st[6]) * len(rows)
    assert sorted(want) == _rows_key(c, v, p)                          # the multiset of training rows, byte for byte
    # the same run written to disk in the reference's layout and read back (float16 storage)
    import tempfile
    from alphazero import history_io
    with tempfile.TemporaryDirectory() as d:
        selfplay.self_play(az.Connect4GS, pp, engines=K, seed=seed, data_folder=d, iteration=5, data_save_size=100)
        triples = history_io.glob_file_triples(d)
        assert sum(t[3] for t in triples) == len(c) and all(os.path.basename(t[0]).startswith("0005-") for t in triples)
        back = [np.concatenate([history_io.load_compressed(t[i]).float().numpy() for t in triples], 0) for i in range(3)]
        assert np.array_equal(back[0], c) and np.array_equal(back[1], v) and np.abs(back[2] - p).max() < 1e-3
    assert np.allclose(res.win_rates, scores / scores.sum())
    assert res.game_length == pytest.approx(length / 12)
    assert res.avg_leaf_depth == pytest.approx(depth / full, rel=1e-5) and res.avg_search_entropy == pytest.approx(ent / full, rel=1e-5)
    assert res.fast_avg_leaf_depth == pytest.approx(fdepth / fast, rel=1e-5) and res.fast_avg_search_entropy == pytest.approx(fent / fast, rel=1e-5)
    assert res.avg_valid_moves == pytest.approx(valid / moves, rel=1e-5) and res.avg_moves_per_turn == pytest.approx(1.0)
    assert res.hit_rate == 0 and res.leaf_evaluations == 0


def test_net_run_fills_the_history_and_the_shards_do_not_change_the_games():
    """with the HIP net: K = 1 and K = 4 shards of the same 64 slots play different games (seeds are per shard) but the same
    number of them with the same number of rows per move; a second call with the same arguments reproduces the first."""
    import torch
    import alphazero as az
    from alphazero import selfplay, torch_net
    net = az.HipLeafNet(torch_net.random_init(torch_net.connect4_spec(), seed=0), torch_net.connect4_spec())
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 64, 64, 64
    pp.mcts_visits = [50, 50]
    pp.model_groups = [0, 0]
    pp.history_enabled = True
    pp.cpuct, pp.fpu_reduction, pp.epsilon, pp.mcts_root_temp = 1.25, 0.25, 0.25, 1.25
    pp.max_cache_size = 65536
    a, (ca, va, pa) = selfplay.self_play(az.Connect4GS, pp, net, engines=4, seed=7)
    b, (cb, vb, pb) = selfplay.self_play(az.Connect4GS, pp, net, engines=4, seed=7)
    assert a.games == 64 and a.samples == ca.shape[0] > 64 * 7 and ca.is_cuda
    assert torch.equal(ca, cb) and torch.equal(va, vb) and torch.equal(pa, pb)        # reproducible
    assert abs(float(pa.sum(1).min()) - 1) < 1e-4 and bool((va.sum(1) == 1).all())
    assert a.hit_rate > 0 and a.leaf_evaluations < a.simulations and 0 < a.cache_saturation <= 1
    one, (c1, _, _) = selfplay.self_play(az.Connect4GS, pp, net, engines=1, seed=7)
    assert one.games == 64 and abs(one.game_length - a.game_length) < 6             # same distribution, other seeds


def test_gating_match_between_two_models():
    """play_past's shape (game_runner.py:2184-2332): new model vs past model on both seatings, per-permutation score tables
    folded into nn_rate / draw_rate / integer win counts; and new model vs RandPlayer (past_iter == 0)."""
    import alphazero as az
    from alphazero import selfplay, torch_net
    spec = torch_net.connect4_spec()
    new = az.HipLeafNet(torch_net.random_init(spec, seed=1), spec)
    past = az.HipLeafNet(torch_net.random_init(spec, seed=2), spec)
    pp = az.PlayParams()
    pp.concurrent_games, pp.games_to_play, pp.max_batch_size = 32, 64, 32        # bs * cb * n_perms
    pp.mcts_visits = [30, 30]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.start_temp = pp.final_temp = 0.5                                           # eval_temp
    pp.max_cache_size = 16384
    r = selfplay.gating_match(az.Connect4GS, pp, new, past, engines=2, seed=3, driver="rounds")
    assert r.n_games == 64 and r.nn_wins + r.past_wins + r.n_draws == 64
    assert len(r.perm_scores) == 2 and sum(sum(ps) for ps in r.perm_scores) == 64
    assert 0 <= r.nn_rate <= 1 and 0 <= r.draw_rate <= 1 and r.hit_rate > 0
    assert abs(r.nn_rate - (r.perm_scores[0][0] + r.perm_scores[1][1]) / 64) < 0.05      # both seatings weigh the same
    again = selfplay.gating_match(az.Connect4GS, pp, new, past, engines=2, seed=3, driver="rounds")
    assert again == r                                                               # reproducible
    default = selfplay.gating_match(az.Connect4GS, pp, new, past, engines=2, seed=3)
    assert default == r                                                             # the default driver is the reproducible one
    # driver="auto": one engine on the asynchronous pipeline, one net per model group (azmi_run_pipeline_groups); which slot a
    # restarted game lands in depends on the order the games end in, so the totals are checked, not the very games
    pipe = selfplay.gating_match(az.Connect4GS, pp, new, past, seed=3, driver="auto")
    assert pipe.n_games == 64 and pipe.nn_wins + pipe.past_wins + pipe.n_draws == 64 and pipe.hit_rate > 0
    assert sum(sum(ps) for ps in pipe.perm_scores) == 64 and abs(pipe.nn_rate - r.nn_rate) < 0.35
    with pytest.raises(RuntimeError, match="pipeline"):
        selfplay.gating_match(az.TawlbwrddGS, pp, new, past, driver="pipeline")
    rnd = selfplay.gating_match(az.Connect4GS, pp, new, None, engines=2, seed=3)    # vs RandPlayer: only group 0 reaches a net
    assert rnd.n_games == 64 and rnd.nn_wins + rnd.past_wins + rnd.n_draws == 64


def test_native_gather_behind_the_c_abi_world_size_one():
    """azmi_comm_* / azmi_gather_counts / azmi_gather_rows (csrc/gather.hip over librccl) at world size 1 - what a one-GPU box can run of
    SURVEY 8e's exchange: the counts all-gather, rank 0's own rows by a device copy, empty and ragged inputs."""
    import torch
    from alphazero import gather
    ng = gather.NativeGather(0, 1, 0)
    dev = torch.device("cuda", 0)
    for n in (0, 1, 777):
        parts = [torch.rand((n, 4, 6, 7), device=dev), torch.rand((n, 3), device=dev), torch.rand((n, 7), device=dev)]
        out = ng.gather_rows_to_rank0(parts)
        torch.cuda.synchronize()
        assert ng.last_counts == [n]
        assert all(o.shape == p.shape and torch.equal(o, p) for o, p in zip(out, parts))
