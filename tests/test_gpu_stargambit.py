"""-m gpu: StarGambit on the device (csrc/dev_stargambit.h: lane-resident units, occupancy boards, wave-cooperative rules) against
the oracle (oracle/az_stargambit.hpp: the reference's unit list restated) - rules tier T0, the reference's own rule tests on
the device objects, pickles, PlayManager tier T2 (moves, visit counts, RNG positions, history rows with relative value targets,
per-variant tables), Gumbel / noise / playout-cap tiers, the stand-alone MCTS object."""
import ctypes as C
import pickle
import struct

import numpy as np
import pytest

import stargambit_cases as sgc

pytestmark = pytest.mark.gpu


def _replay(az, inits, moves, canonical=True, flags=0):
    """azmi_game_replay_ex over rows of (start image, move list)"""
    from alphazero import lib, check
    n, length = moves.shape
    stride = max(len(b) for b in inits)
    init = np.zeros((n, stride), np.uint8)
    for i, b in enumerate(inits):
        init[i, :len(b)] = np.frombuffer(b, np.uint8)
    out = dict(valid=np.zeros((n, 1709), np.uint8), scores=np.zeros((n, 3), np.float32), player=np.zeros(n, np.uint32),
               turn=np.zeros(n, np.uint32), key=np.zeros(n, np.uint64), status=np.zeros(n, np.int32))
    canon = np.zeros((n, 36, 13, 13), np.float32) if canonical else None
    check(lib.azmi_game_replay_ex(4, 0, init.ctypes.data, stride, moves.ctypes.data, n, length, out["valid"].ctypes.data,
                                  out["scores"].ctypes.data, None if canon is None else canon.ctypes.data, out["player"].ctypes.data,
                                  out["turn"].ctypes.data, out["key"].ctypes.data, out["status"].ctypes.data, flags))
    out["canonical"] = canon
    return out


def test_rules_random_playouts_every_prefix(oracle):
    """T0: random games of all four variants; after EVERY action the device's legal moves, scores, player and turn equal the
    oracle's, the canonical planes at every 5th action and at the end"""
    import alphazero as az
    rng = np.random.default_rng(42)
    n_term = 0
    for variant in range(4):
        start = az.StarGambitUnifiedGS(variant).to_bytes()
        records = []
        for g in range(10):
            game = oracle.Game.sg_unified(pinned=variant)
            seq, snaps = [], []
            while True:
                sc = game.scores()
                snaps.append((game.valid(), sc, game.player(), game.turn(), game.canonical() if (len(seq) % 5 == 0 or sc is not None) else None))
                if sc is not None or len(seq) >= 700:
                    break
                m = int(rng.choice(np.flatnonzero(snaps[-1][0])))
                game.play(m)
                seq.append(m)
            records.append((seq, snaps))
            n_term += snaps[-1][1] is not None
        length = max(len(r[0]) for r in records) + 1
        rows, want = [], []
        for seq, snaps in records:
            for k, snap in enumerate(snaps):
                row = -np.ones(length, np.int32)
                row[:k] = seq[:k]
                rows.append(row); want.append(snap)
        moves = np.stack(rows)
        out = _replay(az, [start] * len(rows), moves, canonical=False)
        assert (out["status"] == 0).all()
        for i, (va, sc, pl, tu, _) in enumerate(want):
            assert np.array_equal(out["valid"][i], va), (variant, i, np.flatnonzero(out["valid"][i] != va))
            assert (out["player"][i], out["turn"][i]) == (pl, tu), (variant, i)
            if sc is None:
                assert (out["scores"][i] == -1).all(), (variant, i)
            else:
                assert np.array_equal(out["scores"][i], sc), (variant, i)
        sel = [i for i, w in enumerate(want) if w[4] is not None]
        outc = _replay(az, [start] * len(sel), moves[sel])
        for j, i in enumerate(sel):
            assert np.array_equal(outc["canonical"][j], want[i][4]), (variant, i, np.argwhere(outc["canonical"][j] != want[i][4])[:5])
    assert n_term >= 30


def test_illegal_moves_are_reported(oracle):
    import alphazero as az
    start = az.StarGambitUnifiedGS(0).to_bytes()
    out = _replay(az, [start] * 3, np.array([[1708, -1], [5, -1], [1690 + 1, 1690 + 1]], np.int32))   # end turn on turn 1; a spatial action; fine
    assert out["status"].tolist() == [-1, -1, 0]     # (player 1's canonical NE is a legal deploy)
    # player 1's canonical SW is world NE - not one of its deploy facings: refused when checked, played when the
    # reference's unchecked play_move is asked for (its own test does that, star_gambit_gs_test.cc:818-821)
    out = _replay(az, [start], np.array([[1690 + 1, 1690 + 4]], np.int32))
    assert out["status"][0] == -1
    out = _replay(az, [start], np.array([[1690 + 1, 1690 + 4]], np.int32), flags=1)
    assert out["status"][0] == 0 and out["turn"][0] == 3


class _Dev:
    """the device objects already speak the reference's API; get_units() entries are UnitInfo objects"""


@pytest.mark.parametrize("case", sgc.ALL_CASES, ids=lambda f: f.__name__)
def test_reference_rule_cases_on_the_device_objects(case):
    import alphazero as az
    plain = (az.StarGambitSkirmishGS, az.StarGambitShowdownGS, az.StarGambitClashGS, az.StarGambitBattleGS)
    case(lambda v: plain[v](), lambda v: az.StarGambitUnifiedGS(v))


def test_pickle_is_the_reference_byte_layout(oracle):
    """StarGambitUnifiedGS::to_bytes / from_bytes (star_gambit_gs.cc:2446-2516 around :2246-2338): the device object's image
    equals the oracle's byte for byte after the same moves, reads an image the oracle wrote, survives pickle / copy"""
    import alphazero as az
    rng = np.random.default_rng(9)
    for variant in (0, 3):
        g = az.StarGambitUnifiedGS(variant, (0.1, 0.2, 0.3, 0.4))
        o = oracle.Game.sg_unified(variant, (0.1, 0.2, 0.3, 0.4))
        assert g.to_bytes() == o.sg_to_bytes()
        for _ in range(45):
            va = np.flatnonzero(o.valid())
            if va.size == 0:
                break
            m = int(rng.choice(va))
            g.play_move(m); o.play(m)
        assert g.to_bytes() == o.sg_to_bytes()
        h = az.StarGambitUnifiedGS.from_bytes(o.sg_to_bytes())
        assert h == g and h.to_bytes() == g.to_bytes() and h.get_variant_id() == variant
        assert np.array_equal(h.canonicalized(), o.canonical()) and np.array_equal(h.valid_moves(), o.valid())
        k = pickle.loads(pickle.dumps(g))
        assert type(k) is az.StarGambitUnifiedGS and k == g and k.to_bytes() == g.to_bytes()
        va = np.flatnonzero(o.valid())
        if va.size:
            m = int(va[0])
            k.play_move(m); o.play(m)
            assert np.array_equal(k.canonicalized(), o.canonical()) and k.to_bytes() == o.sg_to_bytes()
    p = pickle.loads(pickle.dumps(az.StarGambitClashGS()))
    assert type(p) is az.StarGambitClashGS and p.num_moves() == 1229
    with pytest.raises(RuntimeError):
        az.StarGambitUnifiedGS.from_bytes(g.to_bytes()[:-3])


def _pp(az, **kw):
    pp = az.PlayParams()
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    pp.games_to_play, pp.concurrent_games = 8, 8
    pp.mcts_visits = [24, 24]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.history_enabled = True
    pp.__dict__.update(kw)
    return pp


def _compare(az, orc, base_args, pp, seed):
    """device run vs one oracle PlayManager per slot (T2): rows of the move log, visit counts, scores, per-variant tables, history"""
    S = pp.concurrent_games
    pm = az.PlayManager(az.StarGambitUnifiedGS(*base_args), pp, seed=seed, log_moves=True)
    pm.play()
    rows, counts = pm.move_log()
    games = pm.slot_games()
    tot = np.zeros(3, np.float32)
    vgames = np.zeros(4, np.int64); vscores = np.zeros((4, 3), np.float32); vsums = np.zeros((4, 10))
    hist = []
    for s in range(S):
        if games[s] == 0:
            continue
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = int(games[s]), 1
        o = orc.PlayManager(orc.Game.sg_unified(*base_args), one, orc.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
        o.run()
        orows, ocounts = o.moves()
        sel = (rows[:, 0] == s) & (rows[:, 1] < games[s])
        d, dc = rows[sel], counts[sel]
        assert d.shape == orows.shape, (s, d.shape, orows.shape)
        bad = np.flatnonzero((d[:, 1:] != orows[:, 1:]).any(1))
        assert bad.size == 0, (s, bad[:3], d[bad[:3]], orows[bad[:3]])
        assert np.array_equal(dc, ocounts), s
        tot += o.scores()
        for v in range(4):
            dv = o.variant(v)
            vgames[v] += dv["games"]; vscores[v] += dv["scores"]
        hist.append(o.history())
    assert np.array_equal(pm.scores(), tot)
    assert pm.num_tracked_variants() == 4
    for v in range(4):
        assert pm.variant_games_completed(v) == vgames[v], (v, pm.variant_games_completed(v), vgames)
        assert np.array_equal(pm.variant_scores(v), vscores[v])
    return pm, hist


def _rows_multiset(c, v, p):
    return sorted(np.concatenate([c[i].reshape(-1), v[i], p[i]]).tobytes() for i in range(len(c)))


def test_playmanager_exact_tier(oracle):
    """T2, all four variants drawn from the coin stream, RANDOM evaluator: moves, visit counts, pcg32 positions, scores,
    per-variant tables; history rows (planes, RELATIVE value targets, visit targets) equal the oracle's as a multiset, byte for byte"""
    import alphazero as az
    pp = _pp(az)
    pm, hist = _compare(az, oracle, (-1, (0.25, 0.25, 0.25, 0.25)), pp, seed=2024)
    hc, hv, hp = pm.history()
    oc = np.concatenate([h[0] for h in hist]); ov = np.concatenate([h[1] for h in hist]); op = np.concatenate([h[2] for h in hist])
    assert len(hc) == len(oc) > 100
    assert _rows_multiset(hc, hv, hp) == _rows_multiset(oc, ov, op)
    assert sum(pm.variant_games_completed(v) > 0 for v in range(4)) >= 2       # the draw really mixes variants
    dec = hv[hv[:, 2] == 0]
    assert dec[:, 0].sum() > 0 and dec[:, 1].sum() > 0                          # both the winner's and the loser's view occur


def test_playmanager_option_tiers(oracle):
    """shaped Dirichlet noise + root temperature + pruning + per-variant temperature decay; Gumbel with its improved-policy
    targets; playout cap randomisation; no tree reuse - each against the oracle slot by slot (pinned and mixed variants)"""
    import alphazero as az
    tiers = [
        ((2, (0.25,) * 4), dict(epsilon=0.25, shaped_dirichlet=True, mcts_root_temp=1.25, root_fpu_zero=True, policy_target_pruning=True,
                                start_temp=1.2, final_temp=0.2, temp_decay_half_life_by_variant=[3.0, 4.0, 5.0, 8.0])),
        ((-1, (0.4, 0.1, 0.4, 0.1)), dict(gumbel_enabled=True, gumbel_m=8, mcts_visits=[32, 32], start_temp=1.2, final_temp=0.2,
                                          temp_decay_half_life=4.0)),
        ((3, (0.25,) * 4), dict(playout_cap_randomization=True, playout_cap_depth=8, playout_cap_percent=0.6, epsilon=0.25, games_to_play=4,
                                concurrent_games=4)),
        ((0, (0.25,) * 4), dict(tree_reuse=False, gumbel_enabled=True, gumbel_m=4, fast_search_uses_gumbel=True, playout_cap_randomization=True,
                                playout_cap_depth=10)),
    ]
    for i, (base, kw) in enumerate(tiers):
        pm, hist = _compare(az, oracle, base, _pp(az, **kw), seed=500 + i)
        hc, hv, hp = pm.history()
        oc = np.concatenate([h[0] for h in hist]); ov = np.concatenate([h[1] for h in hist]); op = np.concatenate([h[2] for h in hist])
        assert _rows_multiset(hc, hv, hp) == _rows_multiset(oc, ov, op), i


def test_variant_statistics_match_the_oracle(oracle):
    import alphazero as az
    pp = _pp(az, games_to_play=12, concurrent_games=4, mcts_visits=[12, 12])
    pm = az.PlayManager(az.StarGambitUnifiedGS(), pp, seed=31)
    pm.play()
    games = pm.slot_games()
    sums = np.zeros((4, 10))
    for s in range(4):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = int(games[s]), 1
        # a slot restarts with games_started (a run-wide counter) so the perm / variant draws stay per slot; 1 perm here
        o = oracle.PlayManager(oracle.Game.sg_unified(), one, oracle.slot_seed(31, s), per_slot_rng=False)
        o.run()
        for v in range(4):
            d = o.variant(v)
            if d["games"]:
                sums[v, 1] += d["games"]; sums[v, 0] += d["stats"][0] * d["games"]
    for v in range(4):
        dv = pm.variant_sums(v)
        assert dv[1] == sums[v, 1]
        assert abs(dv[0] - sums[v, 0]) < 1e-3 * max(1.0, sums[v, 0])
        if dv[1]:
            assert pm.variant_avg_game_length(v) == pytest.approx(dv[0] / dv[1])
            assert pm.variant_avg_moves_per_turn(v) >= 1.0 and pm.variant_avg_valid_moves(v) >= 1.0
    assert pm.games_completed() == 12


def test_mcts_object_on_stargambit(oracle):
    """the stand-alone MCTS class (py_wrapper.cc:192-220) with relative values: call by call against the oracle's Mcts"""
    import alphazero as az
    g = az.StarGambitUnifiedGS(2)
    o = oracle.Game.sg_unified(2)
    for m in (1690 + 2, 1690 + 6 + 1):     # deploy a fighter NW; player 1 deploys a cruiser (canonical NE)
        g.play_move(m); o.play(m)
    dm = az.MCTS(1.25, 2, 1709, fpu_reduction=0.25, relative_values=True, seed=77, game=az.StarGambitUnifiedGS)
    om = oracle.Mcts(1.25, 2, 1709, fpu_reduction=0.25, relative_values=True, seed=77)
    rng = np.random.default_rng(3)
    for sim in range(120):
        dl, ol = dm.find_leaf(g), om.find_leaf(o)
        assert np.array_equal(dl.canonicalized(), ol.canonical()), sim
        pi = rng.random(1709).astype(np.float32); pi /= pi.sum()
        v = rng.random(3).astype(np.float32); v /= v.sum()
        dm.process_result(g, v.copy(), pi); om.process_result(v.copy(), pi)
    assert np.array_equal(dm.counts(), om.counts())
    assert np.array_equal(dm.root_q_values(), om.root_q())
    assert np.array_equal(dm.root_value(), om.root_value())
    with pytest.raises(RuntimeError):      # relative_values is the game's property on the device
        az.MCTS(1.25, 2, 1709, relative_values=False, seed=1, game=az.StarGambitUnifiedGS)


def test_playout_seats_and_playout_eval(oracle):
    """EvalType.PLAYOUT on StarGambit: the rollout continues the descent's position history; playout_eval agrees with the oracle"""
    import alphazero as az
    g = az.StarGambitUnifiedGS(1)
    o = oracle.Game.sg_unified(1)
    for m in (1690 + 1, 1690 + 1):
        g.play_move(m); o.play(m)
    for seed in (5, 6, 7):
        v, pi = az.playout_eval(g, seed=seed)
        ov, opi = oracle.playout_eval(o, seed)
        assert np.array_equal(v, ov) and np.array_equal(pi, opi), seed
    pp = _pp(az, eval_type=[az.EvalType.PLAYOUT, az.EvalType.RANDOM], games_to_play=4, concurrent_games=4, mcts_visits=[10, 10])
    _compare(az, oracle, (0, (0.25,) * 4), pp, seed=91)


# ---- NN in the loop (tier T3) and BASELINE configs[4] at full size -------------------------------------------------------------
def _net_eval(hip):
    import torch
    dev = torch.device("cuda", 0)

    def f(canon):
        v, pi = hip.process(torch.from_numpy(np.ascontiguousarray(canon)).to(dev))
        torch.cuda.synchronize()
        return v.cpu().numpy(), pi.cpu().numpy()
    return f


def _run_with_net(az, pm, hip):
    import torch
    st = torch.cuda.Stream()
    while pm.remaining_games() > 0:
        az.run_rounds([pm], hip, 64, [st.cuda_stream])
        if pm.poll(st.cuda_stream)[1] == 0:
            break
    torch.cuda.synchronize()


def test_nn_in_the_loop_equals_the_oracle_driven_by_the_same_net(oracle):
    """T3 on StarGambit (configs/star_gambit_unified.yaml search: Gumbel, improved-policy targets, per-variant temperature decay):
    the device fast path with the HIP net (k_leafnet_sp on the 13x13 tile, relative values rotated to absolute in process_result)
    plays exactly the games of the oracle PlayManager whose evaluator sends each leaf's planes through the same net"""
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.stargambit_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=17), spec)
    S, seed = 6, 321
    pp = _pp(az, eval_type=[], games_to_play=S, concurrent_games=S, max_batch_size=S, mcts_visits=[20, 20], model_groups=[0, 0],
             gumbel_enabled=True, gumbel_m=8, start_temp=1.2, final_temp=0.2, temp_decay_half_life_by_variant=[3.0, 4.0, 5.0, 8.0])
    pm = az.PlayManager(az.StarGambitUnifiedGS(), pp, seed=seed, log_moves=True)
    _run_with_net(az, pm, hip)
    assert pm.games_completed() == S
    rows, counts = pm.move_log()
    ev = _net_eval(hip)
    for s in (0, 3, 5):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games, one.max_batch_size = 1, 1, 1
        o = oracle.PlayManager(oracle.Game.sg_unified(), one, oracle.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
        o.run(ev)
        orows, ocounts = o.moves()
        sel = (rows[:, 0] == s) & (rows[:, 1] == 0)
        assert sel.sum() == len(orows) > 0, (s, int(sel.sum()), len(orows))
        assert np.array_equal(rows[sel][:, 2:], orows[:, 2:]), f"slot {s}: moves / stream positions differ from the oracle driven by the same net"
        assert np.array_equal(counts[sel], ocounts), f"slot {s}: visit counts differ"


def test_full_size_configs4_random_evaluator_equals_the_oracle(oracle):
    """BASELINE configs[4] per GPU - star_gambit_unified, 1024 concurrent games, 800 simulations per move - with the RANDOM
    evaluator, every game to its end: size-independent properties over all moves + sampled slots equal the oracle move for move"""
    import alphazero as az
    S, sims, seed = 1024, 800, 8088
    pp = _pp(az, games_to_play=S, concurrent_games=S, mcts_visits=[sims, sims], epsilon=0.25, shaped_dirichlet=True, mcts_root_temp=1.25,
             root_fpu_zero=True, policy_target_pruning=True, start_temp=1.2, final_temp=0.2, temp_decay_half_life_by_variant=[3.0, 4.0, 5.0, 8.0])
    pm = az.PlayManager(az.StarGambitUnifiedGS(), pp, seed=seed, log_moves=True, history_capacity=S * 400)
    pm.play()
    assert pm.games_completed() == S
    rows, counts = pm.move_log()
    assert (counts.sum(1) >= sims - 1).all()          # every search had its full budget (a reused subtree brings its earlier visits along)
    assert (pm.scores().sum() == S) and pm.scores()[2] < S
    assert sum(pm.variant_games_completed(v) for v in range(4)) == S and all(pm.variant_games_completed(v) > 150 for v in range(4))
    hc, hv, hp = pm.history()
    assert len(hc) == len(rows) and np.allclose(hp.sum(1), 1.0, atol=1e-4) and (hv.sum(1) == 1.0).all()
    assert (hc[:, 0].reshape(len(hc), -1).sum(1) >= 91).all()                            # channel 0 = the variant's board mask
    assert ((hc[:, 32:36].reshape(len(hc), 4, -1).sum(2) > 0).sum(1) == 1).all()         # exactly one variant plane is lit
    assert pm.avg_moves_per_turn() > 1.5                                                  # several actions per turn
    for s in (0, 511, 1023):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games = 1, 1
        o = oracle.PlayManager(oracle.Game.sg_unified(), one, oracle.slot_seed(seed, s), per_slot_rng=False, perm_base=s)
        o.run()
        orows, ocounts = o.moves()
        sel = (rows[:, 0] == s) & (rows[:, 1] == 0)
        assert sel.sum() == len(orows) > 0 and np.array_equal(rows[sel][:, 2:], orows[:, 2:]) and np.array_equal(counts[sel], ocounts), s


def test_full_size_configs4_with_the_hip_net_and_the_device_cache():
    """configs[4] with its evaluator and its cache (200 000 entries, config.py:197): 1024 games x 800 sims on four engine shards,
    a bounded number of rounds; the cache takes part, the net sees only the misses, every sample row is well formed"""
    import torch
    import alphazero as az
    from alphazero import torch_net
    spec = torch_net.stargambit_spec()
    hip = az.HipLeafNet(torch_net.random_init(spec, seed=3), spec)
    K, S, sims = 4, 1024, 800
    pms = []
    for i in range(K):
        pp = _pp(az, eval_type=[], games_to_play=1 << 30, concurrent_games=S // K, max_batch_size=S // K, mcts_visits=[sims, sims],
                 model_groups=[0, 0], epsilon=0.25, shaped_dirichlet=True, mcts_root_temp=1.25, root_fpu_zero=True, policy_target_pruning=True,
                 start_temp=1.2, final_temp=0.2, temp_decay_half_life_by_variant=[3.0, 4.0, 5.0, 8.0], max_cache_size=200000 // K)
        pms.append(az.PlayManager(az.StarGambitUnifiedGS(), pp, seed=77 + i, history_capacity=(S // K) * 400))
    streams = [torch.cuda.Stream() for _ in range(K)]
    sps = [s.cuda_stream for s in streams]
    for _ in range(6):
        az.run_rounds(pms, hip, 1024, sps)
    torch.cuda.synchronize()
    c = [pm.counters() for pm in pms]
    sims_done, evals, hits = (sum(x[k] for x in c) for k in ("sims", "evals", "cache_hits"))
    assert sims_done > 4e6 and hits > 0 and 0 < evals < sims_done
    for pm in pms:
        pm.poll()
        hc, hv, hp = pm.history()
        if len(hc):
            assert np.allclose(hp.sum(1), 1.0, atol=1e-4) and (hv.sum(1) == 1.0).all() and np.isfinite(hc).all()


def test_symmetries_on_the_device_equal_the_oracle(oracle):
    """GameState::symmetries of StarGambitUnifiedGS (star_gambit_gs.cc:2623-2727): the device gather (csrc/symmetries.hip kind 2)
    against the oracle's restatement of the reference's scatter, on real history rows of all four variants"""
    import alphazero as az
    pp = _pp(az, games_to_play=4, concurrent_games=4, mcts_visits=[12, 12])
    pm = az.PlayManager(az.StarGambitUnifiedGS(), pp, seed=5)
    pm.play()
    hc, hv, hp = pm.history()
    sel = np.linspace(0, len(hc) - 1, 40).astype(int)
    oc, ov, op = az.symmetries_batch(az.StarGambitUnifiedGS, hc[sel], hv[sel], hp[sel])
    assert oc.shape == (40, 2, 36, 13, 13) and op.shape == (40, 2, 1709)
    for j, i in enumerate(sel):
        wc, wv, wp = oracle.symmetries(oracle.SYM_STARGAMBIT, hc[i], hv[i], hp[i])
        assert np.array_equal(oc[j], wc) and np.array_equal(ov[j], wv) and np.array_equal(op[j], wp), i
    g = az.StarGambitUnifiedGS(1)
    out = g.symmetries(az.PlayHistory(hc[0], hv[0], hp[0]))
    assert len(out) == 2 and np.array_equal(out[1].pi(), op[0][1])


def test_self_play_harness_on_stargambit(oracle):
    """alphazero.selfplay.self_play (the device counterpart of GameRunner.run + self_play(), game_runner.py:2057-2160) on the
    unified game: two engine shards, RANDOM evaluator; the sample rows equal the union of the oracle's per-slot runs byte for
    byte and the per-variant read-out (game_runner.py:2136-2145) adds up"""
    import alphazero as az
    from alphazero import selfplay
    pp = _pp(az, games_to_play=8, concurrent_games=8, mcts_visits=[16, 16], temp_decay_half_life_by_variant=[3.0, 4.0, 5.0, 8.0], final_temp=0.2)
    res, (c, v, p) = selfplay.self_play(az.StarGambitUnifiedGS(), pp, net=None, engines=2, seed=99)
    assert res.games == 8 and res.samples == len(c) and sum(res.variant_game_counts.values()) == 8
    rows = []
    for k in range(2):
        for s in range(4):
            one = az.PlayParams(); one.__dict__.update(pp.__dict__)
            one.games_to_play, one.concurrent_games = 1, 1
            o = oracle.PlayManager(oracle.Game.sg_unified(), one, oracle.slot_seed(selfplay.shard_seed(99, k), s), per_slot_rng=False, perm_base=s)
            o.run()
            rows.append(o.history())
    oc = np.concatenate([h[0] for h in rows]); ov = np.concatenate([h[1] for h in rows]); op = np.concatenate([h[2] for h in rows])
    assert _rows_multiset(c, v, p) == _rows_multiset(oc, ov, op)
    for vid, n in res.variant_game_counts.items():
        if n:
            assert abs(sum(res.variant_win_rates[vid]) - 1.0) < 1e-6 and res.variant_metrics[vid]["avg_mpt"] >= 1.0


def test_game_data_gs_of_a_running_slot(oracle):
    """game_data(i).gs() (py_wrapper.cc:265-288) on StarGambit: the slot's position as a game object, history included - it plays on
    exactly like the oracle game that replays the slot's logged moves"""
    import alphazero as az
    pp = _pp(az, games_to_play=2, concurrent_games=2, mcts_visits=[16, 16])
    pm = az.PlayManager(az.StarGambitUnifiedGS(2), pp, seed=12, log_moves=True)
    for _ in range(40):
        pm.round()
    pm.poll()
    rows, _ = pm.move_log()
    games = pm.slot_games()
    for slot in (0, 1):
        g = pm.game_data(slot).gs()
        mine = rows[(rows[:, 0] == slot) & (rows[:, 1] == games[slot])]     # the moves of the slot's running game
        o = oracle.Game.sg_unified(2)
        for m in mine[:, 2]:
            o.play(int(m))
        assert g.get_variant_id() == 2 and g.current_turn() == o.turn() and g.current_player() == o.player()
        assert g.to_bytes()[25:] == o.sg_to_bytes()[25:]
        assert np.array_equal(g.valid_moves(), o.valid()) and np.array_equal(g.canonicalized(), o.canonical())
