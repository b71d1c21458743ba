"""-m gpu: BASELINE.json configs[1] at FULL size (Connect4, 4096 concurrent games, 800 MCTS simulations per move) through
size-independent properties — the oracle cannot replay 80 M simulations, so it replays a sample of slots (a slot's game
does not depend on how many other slots run beside it) and the rest is checked through invariants:
every game finishes, scores add up, every search has its full budget, history rows are distributions, the run is a
function of the seed alone (independent of how many simulations a round may finish inline), and with the HIP leaf net the
position cache changes which leaves reach the net but not one move of one game."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S, SIMS = 4096, 800


def _params(az, evals=None, cache=0):
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = S, S, S
    pp.mcts_visits = [SIMS, SIMS]
    pp.model_groups = [0, 0]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25          # config.py:79-80
    pp.history_enabled = True
    pp.max_cache_size = cache
    if evals is not None:
        pp.eval_type = evals
    return pp


def _digest(rows, counts):
    """order-free checksum of a move log: rows sorted by (slot, game, turn); columns slot, game, move, turn, player + counts"""
    order = np.lexsort((rows[:, 3], rows[:, 1], rows[:, 0]))
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(rows[order][:, :5]).tobytes())
    h.update(np.ascontiguousarray(counts[order]).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def random_run():
    import alphazero as az
    pm = az.PlayManager(az.Connect4GS(), _params(az, [az.EvalType.RANDOM, az.EvalType.RANDOM]), seed=20240601, log_moves=True)
    pm.play()
    return pm, pm.move_log()


def test_full_size_every_game_finishes_with_full_searches(random_run):
    pm, (rows, counts) = random_run
    assert pm.games_completed() == S and pm.remaining_games() == 0
    sc = pm.scores()
    assert sc.sum() == S and (sc >= 0).all()
    assert np.array_equal(np.unique(rows[:, 0]), np.arange(S))           # every slot played
    # every move was chosen after a full search: the root has >= 800 visits, one of them its own evaluation
    assert (counts.sum(1) >= SIMS - 1).all()
    first = rows[:, 3] == 0
    assert first.sum() == S and (counts[first].sum(1) == SIMS - 1).all()   # no reused subtree at the first move
    # the log is a set of complete games: turns 0..len-1 per slot, players alternate, every move was legal when played
    order = np.lexsort((rows[:, 3], rows[:, 0]))
    r = rows[order]
    start = np.r_[True, r[1:, 0] != r[:-1, 0]]
    assert (r[start, 3] == 0).all() and (np.diff(r[:, 3])[~start[1:]] == 1).all()
    assert (r[:, 4] == r[:, 3] % 2).all()
    assert (counts[order][np.arange(len(r)), r[:, 2]] > 0).all()           # the played move had visits
    lens = np.bincount(r[:, 0], minlength=S)
    assert lens.min() >= 7 and lens.max() <= 42 and abs(pm.avg_game_length() - lens.mean()) < 1e-3
    c = pm.counters()
    assert c["sims"] >= int(lens.sum()) * (SIMS - 42) and c["evals"] == 0


def test_full_size_history_rows_are_distributions(random_run):
    pm, (rows, _) = random_run
    n = pm.hist_count()
    assert n == len(rows)                                  # one sample per move (no playout cap)
    canon = np.zeros((n, 4, 6, 7), np.float32); v = np.zeros((n, 3), np.float32); pi = np.zeros((n, 7), np.float32)
    assert pm.build_history_batch(canon, v, pi) == n
    assert np.abs(pi.sum(1) - 1).max() < 1e-5 and (pi >= 0).all()
    assert (v.sum(1) == 1).all() and np.isin(v, (0.0, 1.0)).all()          # game outcomes, one-hot (win/loss/draw)
    stones = canon[:, :2].sum((1, 2, 3))
    assert (canon[:, 2:].reshape(n, 2, -1).min(2).sum(1) == 1).all()       # exactly one player plane is all ones
    to_move = canon[:, 3, 0, 0].astype(np.int64)
    assert (stones.astype(np.int64) % 2 == to_move).all()                  # stones on the board = turn parity
    assert (pi[np.arange(n)][canon[:, :2].sum(1)[:, 0, :] > 0] == 0).all()  # no target mass on a full column


def test_full_size_sampled_slots_equal_the_oracle(random_run, oracle):
    import alphazero as az
    pm, (rows, counts) = random_run
    for s in np.random.default_rng(5).choice(S, 6, replace=False):
        one = _params(az, [az.EvalType.RANDOM, az.EvalType.RANDOM])
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(20240601, int(s)), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s           # moves, turns, players, pcg32 positions
        assert np.array_equal(counts[sel], ocounts), s


def test_full_size_run_is_a_function_of_the_seed(random_run):
    """the same seed with another inline budget per round (a scheduling knob) gives the same 4096 games, move for move"""
    import alphazero as az
    pm, (rows, counts) = random_run
    other = az.PlayManager(az.Connect4GS(), _params(az, [az.EvalType.RANDOM, az.EvalType.RANDOM]), seed=20240601, log_moves=True,
                           max_inline=3)
    other.play()
    r2, c2 = other.move_log()
    assert _digest(rows, counts) == _digest(r2, c2)
    assert np.array_equal(pm.scores(), other.scores())


def test_full_size_cache_is_transparent_with_the_hip_net():
    """6b64c HIP net, 4096 games x 800 sims: the run with a 32 M-entry position cache plays exactly the games of the run
    without one (so cached answers equal recomputed ones bit for bit: the net is batch-invariant at this size), while
    sending far fewer leaves to the net."""
    import torch
    import alphazero as az
    from alphazero import torch_net
    net = torch_net.random_init(torch_net.connect4_spec(), seed=0)
    hip = az.HipLeafNet(net, torch_net.connect4_spec())
    st = torch.cuda.Stream()
    out = []
    for cache in (0, 32_000_000):
        pm = az.PlayManager(az.Connect4GS(), _params(az, cache=cache), seed=77, log_moves=True)
        while pm.poll(st.cuda_stream)[1] > 0:
            az.run_rounds([pm], hip, 512, [st.cuda_stream])
        assert pm.games_completed() == S
        rows, counts = pm.move_log()
        out.append((_digest(rows, counts), pm.counters(), pm.scores()))
    (d0, c0, s0), (d1, c1, s1) = out
    assert d0 == d1 and np.array_equal(s0, s1)
    assert c0["sims"] == c1["sims"] and c0["cache_hits"] == 0
    assert c1["cache_hits"] > 0.4 * c1["sims"] and c1["evals"] < 0.6 * c0["evals"]


def test_full_size_tawlbwrdd_prefix_equals_the_oracle(oracle):
    """BASELINE configs[2] at full size (Tawlbwrdd, 2048 concurrent games, 400 simulations): a bounded number of rounds, then
    every sampled slot's moves so far equal the first moves of the oracle's game for that slot, and every logged search
    had its full budget."""
    import alphazero as az
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 2048, 2048, 2048
    pp.mcts_visits = [400, 400]
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    seed = 20240601
    pm = az.PlayManager(az.TawlbwrddGS(), pp, seed=seed, log_moves=True, max_inline=8)
    for _ in range(400):                     # 8 inline simulations per round: ~8 moves per slot
        pm.round()
    rows, counts = pm.move_log()
    assert len(rows) >= 4 * 2048 and np.array_equal(np.unique(rows[:, 0]), np.arange(2048))
    assert (counts.sum(1) >= 399).all()
    for s in np.random.default_rng(9).choice(2048, 3, replace=False):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(oracle.GAME_TAWLBWRDD, one, oracle.slot_seed(seed, int(s)), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        n = int(sel.sum())
        assert n >= 4
        assert np.array_equal(rows[sel][:, 1:], orows[:n, 1:]), s
        assert np.array_equal(counts[sel], ocounts[:n]), s


@pytest.mark.parametrize("playout_cap", [False, True])
def test_full_size_benchmark_flags_sampled_slots_equal_the_oracle(oracle, playout_cap):
    """the benchmarked configuration itself — bench.py's self-play flags (shaped Dirichlet noise, root temperature, FPU rule,
    temperature decay, target pruning, resign with play-through; with and without playout-cap randomisation) at 4096 x 800
    with the RANDOM evaluator: sampled slots replayed by the oracle, coin stream included."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import alphazero as az
    import bench
    pp = bench.selfplay_params(az, S, SIMS, S, cache=0, playout_cap=playout_cap)
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    seed = 424242
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
    pm.play()
    assert pm.games_completed() == S and pm.scores().sum() == S
    rows, counts = pm.move_log()
    capped = rows[:, 5] == 1
    assert capped.any() == playout_cap
    assert (counts[~capped].sum(1) >= SIMS - 1).all()
    if playout_cap:
        frac = capped.mean()
        assert 0.70 < frac < 0.80                       # playout_cap_percent 0.75 of the moves search 25 simulations
        assert pm.hist_count() == int((~capped).sum())  # only full searches are recorded
    for s in np.random.default_rng(3).choice(S, 4, replace=False):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(seed, int(s)), per_slot_rng=False)
        o.run()
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]), s
        assert np.array_equal(counts[sel], ocounts), s


def test_full_size_tawlbwrdd_with_the_net_plays_every_game_to_the_end(oracle):
    """BASELINE configs[2] at full size AND full length with the net in the loop: Tawlbwrdd, 2048 concurrent games, 400
    simulations per move, the configs/tawlbwrdd.yaml net on the spatial MFMA kernels, four engine shards as the bench runs
    them.  Size-independent properties over all games, and one sampled slot replayed by the oracle with the same net as
    its evaluator (tier T3): move for move, count for count."""
    import torch
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.tawlbwrdd_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
    K, Se, sims, seed = 4, 512, 400, 20240601
    pms, streams = [], []
    for k in range(K):
        pp = az.PlayParams()
        pp.games_to_play, pp.concurrent_games, pp.max_batch_size = Se, Se, Se
        pp.mcts_visits = [sims, sims]
        pp.model_groups = [0, 0]
        pp.cpuct, pp.fpu_reduction = 1.25, 0.25
        pp.epsilon, pp.mcts_root_temp, pp.root_fpu_zero, pp.shaped_dirichlet = 0.25, 1.25, True, True
        pp.start_temp, pp.final_temp, pp.temp_decay_half_life = 1.0, 0.2, 10.0
        pp.history_enabled = True
        pms.append(az.PlayManager(az.TawlbwrddGS(), pp, seed=seed + 104729 * k, log_moves=True))
        streams.append(torch.cuda.Stream())
    sps = [s.cuda_stream for s in streams]
    live = list(range(K))
    rounds = 0
    while live:
        az.run_rounds([pms[i] for i in live], hip, 512, [sps[i] for i in live])
        rounds += 512
        live = [i for i in live if pms[i].poll(sps[i])[1] > 0]
        assert rounds < 400_000
    torch.cuda.synchronize()
    total_rows = 0
    for k, pm in enumerate(pms):
        assert pm.games_completed() == Se and pm.scores().sum() == Se
        rows, counts = pm.move_log()
        assert np.array_equal(np.unique(rows[:, 0]), np.arange(Se))
        assert (counts.sum(1) >= sims - 1).all()                                  # every move after a full search
        assert (counts[np.arange(len(rows)), rows[:, 2]] > 0).all()               # the played move had visits
        lens = np.bincount(rows[:, 0], minlength=Se)
        assert lens.min() >= 4 and lens.max() <= 400 and abs(pm.avg_game_length() - lens.mean()) < 1e-3
        c = pm.counters()
        assert c["evals"] > 0.9 * c["sims"] * 0.5 and c["evals"] <= c["sims"]    # Tafl leaves are rarely terminal
        n = pm.hist_count()
        assert n == len(rows)
        total_rows += n
    # history rows of one shard: distributions over legal moves, one-hot outcomes
    pm = pms[1]
    n = pm.hist_count()
    canon = np.zeros((n, 7, 11, 11), np.float32); v = np.zeros((n, 3), np.float32); pi = np.zeros((n, 2662), np.float32)
    assert pm.build_history_batch(canon, v, pi) == n
    assert np.abs(pi.sum(1) - 1).max() < 1e-4 and (pi >= 0).all()
    assert (v.sum(1) == 1).all() and np.isin(v, (0.0, 1.0)).all()
    assert (canon[:, 0].reshape(n, -1).sum(1) <= 1).all()                          # at most one king
    # one slot of shard 2 against the oracle driven by the same net
    dev = torch.device("cuda", 0)

    def evaluator(c):
        vv, pp_ = hip.process(torch.from_numpy(np.ascontiguousarray(c)).to(dev))
        torch.cuda.synchronize()
        return vv.cpu().numpy(), pp_.cpu().numpy()
    k, s = 2, 137
    rows, counts = pms[k].move_log()
    one = az.PlayParams(); one.__dict__.update(pms[k]._params.__dict__)
    one.games_to_play, one.concurrent_games, one.max_batch_size = 1, 1, 1
    o = oracle.PlayManager(oracle.GAME_TAWLBWRDD, one, oracle.slot_seed(seed + 104729 * k, s), per_slot_rng=False)
    o.run(evaluator)
    orows, ocounts = o.moves()
    sel = rows[:, 0] == s
    assert sel.sum() == len(orows)
    assert np.array_equal(rows[sel][:, 1:], orows[:, 1:]) and np.array_equal(counts[sel], ocounts)


def test_full_size_pipeline_plays_the_lockstep_engines_games(oracle):
    """the benched configuration through the benched driver: bench.py's self-play flags, 4096 x 800, the 6b64c HIP net, one
    engine with a 32 M-entry cache driven by the asynchronous pipeline (persistent tree / mover / net wavefronts, 12 ms epochs)
    - all 4096 games equal, move for move and visit count for visit count, the games of the lock-step engine with the same
    seed; one sampled slot is replayed by the oracle with the same net as its evaluator."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import alphazero as az
    import bench
    from alphazero import torch_net
    spec = torch_net.connect4_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=0), spec)
    st = torch.cuda.Stream()
    seed = 31415
    pp = bench.selfplay_params(az, S, SIMS, S, cache=32_000_000)
    runs = []
    for driver in ("pipeline", "rounds"):
        pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True, history_capacity=S * 42 * 2)
        assert az.pipeline_supported(pm, hip)
        guard = 0
        while pm.poll(st.cuda_stream)[1] > 0 and guard < 4000:
            if driver == "pipeline":
                az.run_pipeline(pm, hip, 4, 256 * S, st.cuda_stream)
            else:
                az.run_rounds([pm], hip, 512, [st.cuda_stream])
            guard += 1
        assert pm.games_completed() == S and pm.scores().sum() == S
        rows, counts = pm.move_log()
        runs.append((pm, rows, counts))
    (pa, ra, ca), (pb, rb, cb) = runs
    assert _digest(ra, ca) == _digest(rb, cb)
    assert np.array_equal(pa.scores(), pb.scores()) and pa.counters()["sims"] == pb.counters()["sims"]
    assert pa.counters()["cache_hits"] > 0.4 * pa.counters()["sims"]
    dev = torch.device("cuda", 0)

    def net_eval(canon):
        v, pi = hip.process(torch.from_numpy(np.ascontiguousarray(canon)).to(dev))
        torch.cuda.synchronize()
        return v.cpu().numpy(), pi.cpu().numpy()
    s = 2718
    one = az.PlayParams(); one.__dict__.update(pp.__dict__)
    one.games_to_play, one.concurrent_games, one.max_batch_size = 1, 1, 1
    o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
    o.run(net_eval)
    orows, ocounts = o.moves()
    sel = ra[:, 0] == s
    assert np.array_equal(ra[sel][:, 1:], orows[:, 1:]) and np.array_equal(ca[sel], ocounts)
