"""-m gpu: the device S3-FIFO cache against the oracle's restatement of s3fifo_cache.h, op by op, and the
engine with the cache switched on (search results must not depend on the cache)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pv(key, np_=7, nv=3):
    r = np.random.default_rng(int(key) & 0xFFFFFFFF)
    return r.random(np_, dtype=np.float32), r.random(nv, dtype=np.float32)


@pytest.mark.parametrize("max_size,shards,ghost", [(16, 1, 14), (64, 4, 56), (4, 1, 4), (1, 1, 1), (0, 1, 0), (256, 8, 0), (96, 3, 80),
                                                    (64, 1, 57), (256, 4, 228), (128, 2, 0), (640, 10, 570)])  # last four: 64-entry wave-resident shards
def test_device_cache_matches_oracle_op_by_op(oracle, max_size, shards, ghost):
    """Random interleaving of insert batches and find batches over a small key universe, so that
    promotion S->M, ghost re-admission, second-chance sweeps and the 2-bit frequency cap all occur
    (s3fifo_cache_test.cc:28-600 behaviours).  Finds inside one batch use distinct keys (concurrent
    finds of one key are order-free on the device and serial in the oracle)."""
    import alphazero as az
    dev = az.ShardedS3FIFOCache(max_size, shards, ghost, 7, 3)
    orc = oracle.Cache(max_size, shards, ghost, 7, 3)
    rng = np.random.default_rng(max_size * 1000 + shards)
    universe = rng.integers(1, 2 ** 63, size=max(16, 3 * max_size + 8), dtype=np.uint64)
    for step in range(200):
        if rng.random() < 0.5:
            keys = rng.choice(universe, size=rng.integers(1, 12))
            pol = np.stack([_pv(k)[0] for k in keys]); val = np.stack([_pv(k)[1] for k in keys])
            dev.insert_many(keys, pol, val)
            for k, p, v in zip(keys, pol, val):
                orc.insert(int(k), p, v)
        else:
            keys = rng.choice(universe, size=rng.integers(1, 12), replace=False)
            hit, p, v = dev.find_many(keys)
            for i, k in enumerate(keys):
                want = orc.find(int(k))
                assert bool(hit[i]) == (want is not None), (step, i, int(k))
                if want is not None:
                    assert np.array_equal(p[i], want[0]) and np.array_equal(v[i], want[1])
        st = orc.stats()
        got = dict(hits=dev.hits(), misses=dev.misses(), evictions=dev.evictions(), reinserts=dev.reinserts(), size=dev.size(), max_size=dev.max_size())
        assert got == st, (step, got, st)


def test_reference_python_surface():
    """S3FIFOCache(max_size, ghost_size, num_policy, num_value).find/insert (py_wrapper.cc:222-248)."""
    import alphazero as az
    c = az.S3FIFOCache(4, 4, 2, 1)
    assert c.find(123, 2, 1) is None and c.misses() == 1
    c.insert(123, [0.25, 0.75], [0.5])
    p, v = c.find(123, 2, 1)
    assert p.tolist() == [0.25, 0.75] and v.tolist() == [0.5] and c.hits() == 1 and c.size() == 1 and c.max_size() == 4


def _evaluator(canon):
    n = canon.shape[0]
    flat = canon.reshape(n, -1)
    w = np.linspace(0.5, 1.5, flat.shape[1], dtype=np.float32)
    s = flat @ w
    v = np.stack([0.3 + 0.1 * np.sin(s), 0.3 - 0.1 * np.sin(s), np.full(n, 0.4)], 1).astype(np.float32)
    pi = np.abs(np.sin(s[:, None] * np.arange(1, 8, dtype=np.float32))) + 0.05
    return v, (pi / pi.sum(1, keepdims=True)).astype(np.float32)


def test_engine_with_cache_is_transparent(oracle):
    """With a deterministic evaluator the cache changes which leaves are evaluated, never the search:
    moves / visit counts / RNG positions equal the oracle run WITHOUT a cache, and hits happen."""
    import alphazero as az
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 12, 12, 12
    pp.model_groups = [0, 0]
    pp.mcts_visits = [40, 40]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.history_enabled = True
    pp.max_cache_size = 4096
    seed = 31
    pm = az.PlayManager(az.Connect4GS(), pp, seed=seed, log_moves=True)
    batch = np.zeros((12, 4, 6, 7), np.float32)
    while pm.remaining_games() > 0:
        idx = pm.build_batch(0, batch)
        if not idx:
            continue
        v, pi = _evaluator(batch[: len(idx)])
        pm.update_inferences(0, idx, v, pi)
    rows, counts = pm.move_log()
    c = pm.counters()
    assert c["cache_hits"] > 0 and c["cache_misses"] > 0
    assert c["evals"] < c["sims"]
    for s in range(12):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games, one.max_cache_size = 1, 1, 0
        o = oracle.PlayManager(oracle.GAME_CONNECT4, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run(_evaluator)
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 2:], orows[:, 2:]), s
        assert np.array_equal(counts[sel], ocounts), s


def test_wide_game_engine_with_cache_is_transparent(oracle):
    """Tawlbwrdd (one wavefront per slot, 10.6 KB of policy per cache entry): with a deterministic evaluator the cache
    changes which leaves are evaluated, never the search."""
    import alphazero as az

    def evaluator(canon):
        n = canon.shape[0]
        flat = canon.reshape(n, -1)
        # batch-invariant by construction (a BLAS float32 gemv may sum in a batch-size dependent order, and a cached
        # answer must equal a recomputed one bit for bit): accumulate in float64, round once
        w = np.linspace(0.5, 1.5, flat.shape[1], dtype=np.float64)
        s = (flat.astype(np.float64) * w).sum(1).astype(np.float32)
        v = np.stack([0.3 + 0.1 * np.sin(s), 0.3 - 0.1 * np.sin(s), np.full(n, 0.4)], 1).astype(np.float32)
        pi = np.abs(np.sin(s[:, None] * np.float32(0.01) * np.arange(1, 2663, dtype=np.float32))) + np.float32(0.05)
        return v, (pi / pi.sum(1, keepdims=True)).astype(np.float32)

    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = 3, 3, 3
    pp.model_groups = [0, 0]
    pp.mcts_visits = [24, 24]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    pp.max_cache_size = 2048
    seed = 37
    pm = az.PlayManager(az.TawlbwrddGS(), pp, seed=seed, log_moves=True)
    batch = np.zeros((3, 7, 11, 11), np.float32)
    while pm.remaining_games() > 0:
        idx = pm.build_batch(0, batch)
        if not idx:
            continue
        v, pi = evaluator(batch[: len(idx)])
        pm.update_inferences(0, idx, v, pi)
    rows, counts = pm.move_log()
    c = pm.counters()
    assert c["cache_hits"] > 0 and c["cache_misses"] > 0 and c["evals"] < c["sims"]
    for s in range(3):
        one = az.PlayParams(); one.__dict__.update(pp.__dict__)
        one.games_to_play, one.concurrent_games, one.max_cache_size = 1, 1, 0
        o = oracle.PlayManager(oracle.GAME_TAWLBWRDD, one, oracle.slot_seed(seed, s), per_slot_rng=False)
        o.run(evaluator)
        orows, ocounts = o.moves()
        sel = rows[:, 0] == s
        assert np.array_equal(rows[sel][:, 2:], orows[:, 2:]), s
        assert np.array_equal(counts[sel], ocounts), s


# ---- PlayManager(gs, params, caches=[...]) — py_wrapper.cc:355-360, play_manager.cc:644-649; reference tests
# test_cache.py:385-470 ------------------------------------------------------------------------------------------
def _play_with_evaluator(az, pp, caches, seed, log=True):
    pm = az.PlayManager(az.Connect4GS(), pp, caches=caches, seed=seed, log_moves=log)
    batch = np.zeros((int(pp.concurrent_games), 4, 6, 7), np.float32)
    while pm.remaining_games() > 0:
        idx = pm.build_batch(0, batch)
        if not idx:
            continue
        v, pi = _evaluator(batch[: len(idx)])
        pm.update_inferences(0, idx, v, pi)
    return pm


def _c4_params(az, games=8, visits=30):
    pp = az.PlayParams()
    pp.games_to_play, pp.concurrent_games, pp.max_batch_size = games, games, games
    pp.model_groups = [0, 0]
    pp.mcts_visits = [visits, visits]
    pp.cpuct, pp.fpu_reduction = 1.25, 0.25
    return pp


def test_playmanager_cache_counters_and_no_cache():
    """test_cache.py:385-411: the internal cache reports max_size / size; no cache reports zeros."""
    import alphazero as az
    pp = _c4_params(az); pp.max_cache_size = 1024
    pm = _play_with_evaluator(az, pp, None, 5)
    assert pm.cache_max_size() == 1024 and 0 < pm.cache_size() <= 1024
    assert pm.cache_hits() > 0 and pm.cache_misses() > 0
    assert pm.cache_hits() == pm.counters()["cache_hits"]
    pp = _c4_params(az); pp.max_cache_size = 0
    pm = _play_with_evaluator(az, pp, None, 5)
    assert (pm.cache_max_size(), pm.cache_hits(), pm.cache_misses(), pm.cache_size(), pm.cache_evictions(), pm.cache_reinserts()) == (0,) * 6


def test_playmanager_external_cache_is_used_and_transparent():
    """test_cache.py:414-431 + the transparency property: the same games as a run with the engine's own cache."""
    import alphazero as az
    pp = _c4_params(az); pp.max_cache_size = 0
    ext = az.ShardedS3FIFOCache.for_engine(4096, 7, 3)
    pm = _play_with_evaluator(az, pp, [ext], 9)
    assert pm.cache_max_size() == 4096 and ext.size() > 0 and ext.misses() > 0
    assert pm.cache_size() == ext.size() and pm.cache_hits() == ext.hits()
    pp2 = _c4_params(az); pp2.max_cache_size = 4096
    own = _play_with_evaluator(az, pp2, None, 9)
    assert np.array_equal(pm.move_log()[0], own.move_log()[0]) and np.array_equal(pm.move_log()[1], own.move_log()[1])
    assert (pm.cache_hits(), pm.cache_misses()) == (own.cache_hits(), own.cache_misses())


def test_playmanager_external_cache_shared_across_instances():
    """test_cache.py:434-460: a second PlayManager on the same cache starts warm."""
    import alphazero as az
    shared = az.ShardedS3FIFOCache.for_engine(8192, 7, 3)
    pp = _c4_params(az)
    a = _play_with_evaluator(az, pp, [shared], 11)
    size1, hits1, evals1 = shared.size(), shared.hits(), a.counters()["evals"]
    assert size1 > 0
    b = _play_with_evaluator(az, pp, [shared], 11)       # the same games again: every leaf is already cached
    assert shared.hits() > hits1
    assert b.counters()["evals"] < evals1
    def by_slot(pm):    # log rows are in time order, which depends on how many leaves hit the cache: order them per slot
        rows = pm.move_log()[0]
        return rows[np.lexsort((rows[:, 3], rows[:, 1], rows[:, 0]))][:, :5]
    assert np.array_equal(by_slot(a), by_slot(b))
    del a, b
    assert shared.size() >= size1                        # the cache outlives the engines that borrowed it


def test_playmanager_external_cache_none_entries_and_errors():
    """test_cache.py:463-476 (None entries for groups that never reach a net) + argument checks."""
    import alphazero as az
    pp = _c4_params(az)
    pp.model_groups = [0, 1]
    pp.eval_type = [az.EvalType.RANDOM, az.EvalType.RANDOM]
    nn_cache = az.ShardedS3FIFOCache.for_engine(1024, 7, 3)
    pm = az.PlayManager(az.Connect4GS(), pp, caches=[nn_cache, None], seed=3)
    pm.play()
    assert pm.games_completed() == 8 and pm.cache_max_size() == 1024
    used = az.ShardedS3FIFOCache(1000, 1, 900, 7, 3)
    used.insert(5, np.zeros(7, np.float32), np.zeros(3, np.float32))
    with pytest.raises(RuntimeError, match="another shard layout"):          # a cache that already holds entries cannot be re-laid out
        az.PlayManager(az.Connect4GS(), pp, caches=[used, None])
    with pytest.raises(RuntimeError, match="num_policy"):
        az.PlayManager(az.Connect4GS(), pp, caches=[az.ShardedS3FIFOCache.for_engine(1024, 9, 3), None])
    with pytest.raises(RuntimeError, match="model groups"):
        az.PlayManager(az.Connect4GS(), pp, caches=[nn_cache])


def test_reference_cache_utils_style_caches_attach_to_a_playmanager():
    """test_cache.py:259-300, 473-488: caches as the reference's helpers create them — any shard count, 1 by default
    (cache_utils.create_sharded_cache) — work as PlayManager caches; cache_max_size() is the sum of the requested sizes."""
    import alphazero as az
    c = az.ShardedS3FIFOCache(1000, 2, 900, 7, 3)
    assert (c.size(), c.max_size(), c.hits(), c.misses(), c.evictions(), c.reinserts()) == (0, 1000, 0, 0, 0, 0)
    cache1 = az.ShardedS3FIFOCache(3000, 1, 2700, 7, 3)          # create_sharded_cache(Game, 3000)
    cache2 = az.ShardedS3FIFOCache(5000, 1, 4500, 7, 3)
    pp = _c4_params(az, games=4)
    pp.model_groups = [0, 1]
    pm = _play_two_groups(az, pp, [cache1, cache2])
    assert pm.games_completed() == 4
    assert pm.cache_max_size() == 3000 + 5000 and cache1.max_size() == 3000
    assert cache1.size() > 0 and cache2.size() > 0 and cache1.misses() > 0
    again = _play_two_groups(az, pp, [cache1, cache2])                         # the same objects again: warm
    assert again.counters()["evals"] < pm.counters()["evals"]


def _play_two_groups(az, pp, caches):
    pm = az.PlayManager(az.Connect4GS(), pp, caches=caches, seed=21)
    batch = np.zeros((int(pp.concurrent_games), 4, 6, 7), np.float32)
    while pm.remaining_games() > 0:
        for g in range(2):
            idx = pm.build_batch(g, batch)
            if idx:
                v, pi = _evaluator(batch[: len(idx)])
                pm.update_inferences(g, idx, v, pi)
    return pm


def test_reference_s3fifo_cases_on_the_device_cache():
    """s3fifo_cache_test.cc's single-thread cases (tests/s3fifo_cases.py) on the device S3FIFOCache / ShardedS3FIFOCache."""
    import alphazero as az
    import s3fifo_cases

    class Dev:
        def __init__(self, mx, gh, np_, nv, shards=1):
            self.c = az.ShardedS3FIFOCache(mx, shards, gh, np_, nv)
        def find(self, key): return self.c.find(key)
        def insert(self, key, p, v): self.c.insert(key, p, v)
        def stats(self):
            return dict(hits=self.c.hits(), misses=self.c.misses(), evictions=self.c.evictions(), reinserts=self.c.reinserts(),
                        size=self.c.size(), max_size=self.c.max_size())
    assert len(s3fifo_cases.run_all(Dev)) == 21
