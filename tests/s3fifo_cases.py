"""The reference's S3-FIFO test cases (s3fifo_cache_test.cc:28-597) as one script over an abstract cache, run against the
oracle (CPU, tests/test_oracle_pinned.py) and against the device cache (GPU, tests/test_gpu_cache.py).  `make(max_size,
ghost_size, num_policy, num_value, shards=1)` returns an object with find(key) -> (pi, v) | None, insert(key, pi, v) and
hits / misses / evictions / reinserts / size / max_size as attributes of stats()."""
import numpy as np


def _pol(n, base):   # make_policy / make_value, s3fifo_cache_test.cc:12-24: base + i
    return (np.arange(n) + base).astype(np.float32)


def run_all(make):
    ran = []

    def case(fn):
        fn(); ran.append(fn.__name__)

    def insert_and_find():                                      # :28-38
        c = make(10, 5, 3, 2); pol, val = _pol(3, 1.0), _pol(2, 0.5)
        c.insert(42, pol, val)
        p, v = c.find(42)
        assert np.array_equal(p, pol) and np.array_equal(v, val)
    case(insert_and_find)

    def find_miss():                                            # :40-45
        c = make(10, 5, 3, 2)
        assert c.find(99) is None and c.stats()["misses"] == 1
    case(find_miss)

    def duplicate_insert_is_noop():                             # :47-63
        c = make(10, 5, 3, 2)
        c.insert(42, _pol(3, 1.0), _pol(2, 0.5)); c.insert(42, _pol(3, 100.0), _pol(2, 100.0))
        p, v = c.find(42)
        assert np.array_equal(p, _pol(3, 1.0)) and np.array_equal(v, _pol(2, 0.5)) and c.stats()["size"] == 1
    case(duplicate_insert_is_noop)

    def stats_tracking():                                       # :65-78
        c = make(10, 5, 3, 2); c.insert(1, _pol(3, 1.0), _pol(2, 0.5))
        c.find(1); c.find(1); c.find(2)
        s = c.stats(); assert (s["hits"], s["misses"], s["size"]) == (2, 1, 1)
    case(stats_tracking)

    def eviction_keeps_size():                                  # :84-94
        c = make(10, 5, 2, 1)
        for i in range(15): c.insert(i, _pol(2, 0), _pol(1, 0))
        s = c.stats(); assert s["size"] == 10 and s["evictions"] == 5
    case(eviction_keeps_size)

    def scan_resistance():                                      # :96-120
        c = make(10, 5, 2, 1)
        for i in range(10): c.insert(i, _pol(2, 0), _pol(1, 0))
        c.find(0); c.find(0)
        for i in range(10, 20): c.insert(i, _pol(2, 0), _pol(1, 0))
        assert c.find(0) is not None
    case(scan_resistance)

    def small_to_main_promotion():                              # :122-146
        c = make(5, 3, 2, 1)
        for i in range(5): c.insert(i, _pol(2, 0), _pol(1, 0))
        c.find(0); c.insert(100, _pol(2, 0), _pol(1, 0))
        assert c.find(0) is not None and c.stats()["size"] == 5
    case(small_to_main_promotion)

    def ghost_admits_to_main():                                 # :148-174
        c = make(3, 5, 2, 1)
        for i in range(3): c.insert(i, _pol(2, 0), _pol(1, 0))
        c.insert(3, _pol(2, 0), _pol(1, 0))
        assert c.find(0) is None
        c.insert(0, _pol(2, 0), _pol(1, 0)); assert c.find(0) is not None
        c.insert(10, _pol(2, 0), _pol(1, 0)); c.insert(11, _pol(2, 0), _pol(1, 0))
        assert c.find(0) is not None
    case(ghost_admits_to_main)

    def main_clock_sweep():                                     # :176-203
        c = make(5, 3, 2, 1)
        for i in range(5): c.insert(i, _pol(2, 0), _pol(1, 0))
        for i in range(5): c.find(i)
        for i in range(100, 105): c.insert(i, _pol(2, 0), _pol(1, 0))
        s = c.stats(); assert s["size"] == 5 and s["evictions"] > 0
    case(main_clock_sweep)

    def eviction_cascade():                                     # :205-223
        c = make(5, 0, 2, 1)
        for i in range(5): c.insert(i, _pol(2, 0), _pol(1, 0)); c.find(i)
        c.insert(99, _pol(2, 0), _pol(1, 0))
        assert c.stats()["size"] == 5
    case(eviction_cascade)

    def ghost_reinsert_on_find():                               # :229-248
        c = make(3, 5, 2, 1)
        for i in range(3): c.insert(i, _pol(2, 0), _pol(1, 0))
        c.insert(3, _pol(2, 0), _pol(1, 0))
        assert c.find(0) is None and c.stats()["reinserts"] == 1
        assert c.find(0) is None and c.stats()["reinserts"] == 2
    case(ghost_reinsert_on_find)

    def ghost_consumed_on_insert():                             # :250-268
        c = make(3, 5, 2, 1)
        for i in range(3): c.insert(i, _pol(2, 0), _pol(1, 0))
        c.insert(3, _pol(2, 0), _pol(1, 0)); c.insert(0, _pol(2, 0), _pol(1, 0))
        assert c.find(0) is not None
    case(ghost_consumed_on_insert)

    def ghost_overflow():                                       # :270-291
        c = make(2, 2, 2, 1)
        for k in (1, 2, 3, 4, 5): c.insert(k, _pol(2, 0), _pol(1, 0))
        c.find(1); assert c.stats()["reinserts"] == 0
        c.find(2); assert c.stats()["reinserts"] == 1
    case(ghost_overflow)

    def ghost_disabled():                                       # :293-306
        c = make(3, 0, 2, 1)
        for k in (1, 2, 3, 4): c.insert(k, _pol(2, 0), _pol(1, 0))
        c.find(1); assert c.stats()["reinserts"] == 0
    case(ghost_disabled)

    def flat_storage_integrity():                               # :312-339
        c = make(10, 5, 4, 3)
        data = {1: (_pol(4, 1.0), _pol(3, 0.1)), 2: (_pol(4, 10.0), _pol(3, 1.0)), 3: (_pol(4, 100.0), _pol(3, 10.0))}
        for k, (p, v) in data.items(): c.insert(k, p, v)
        for k, (p, v) in data.items():
            gp, gv = c.find(k); assert np.array_equal(gp, p) and np.array_equal(gv, v)
    case(flat_storage_integrity)

    def sharded_insert_distributes_and_stats_aggregate():       # :345-391
        c = make(100, 50, 3, 2, shards=4)
        for i in range(20): c.insert(i, _pol(3, float(i)), _pol(2, float(i)))
        assert c.stats()["size"] == 20
        for i in range(20):
            p, v = c.find(i); assert np.array_equal(p, _pol(3, float(i))) and np.array_equal(v, _pol(2, float(i)))
        c = make(20, 10, 2, 1, shards=4)
        for i in range(20): c.insert(i, _pol(2, 0), _pol(1, 0))
        for i in range(20): c.find(i)
        for i in range(100, 105): c.find(i)
        s = c.stats(); assert (s["hits"], s["misses"], s["size"], s["max_size"]) == (20, 5, 20, 20)
    case(sharded_insert_distributes_and_stats_aggregate)

    def no_growth_after_fill():                                 # :452-473 (1000 cycles instead of 10000)
        c = make(100, 50, 4, 2)
        for i in range(100): c.insert(i, _pol(4, 0), _pol(2, 0))
        for i in range(100, 1100): c.insert(i, _pol(4, 0), _pol(2, 0)); c.find(i)
        assert c.stats()["size"] == 100
    case(no_growth_after_fill)

    def capacity_one():                                         # :479-497
        c = make(1, 1, 2, 1)
        c.insert(1, _pol(2, 1.0), _pol(1, 0.5)); assert np.array_equal(c.find(1)[0], _pol(2, 1.0))
        c.insert(2, _pol(2, 10.0), _pol(1, 5.0))
        assert c.find(1) is None and np.array_equal(c.find(2)[0], _pol(2, 10.0)) and c.stats()["size"] == 1
    case(capacity_one)

    def all_hits_all_misses():                                  # :510-532
        c = make(5, 3, 2, 1); c.insert(1, _pol(2, 1.0), _pol(1, 0.5))
        for _ in range(100): assert c.find(1) is not None
        s = c.stats(); assert (s["hits"], s["misses"]) == (100, 0)
        c = make(5, 3, 2, 1)
        for i in range(100): assert c.find(i + 1000) is None
        s = c.stats(); assert (s["hits"], s["misses"]) == (0, 100)
    case(all_hits_all_misses)

    def two_bit_freq_counter():                                 # :534-573
        c = make(4, 4, 2, 1); z = (_pol(2, 0), _pol(1, 0))
        for i in range(10, 14): c.insert(i, *z)
        for i in range(20, 24): c.insert(i, *z)
        c.insert(10, *z); c.insert(11, *z)
        for _ in range(10): c.find(10)
        c.find(11); c.find(22); c.find(23)
        c.insert(30, *z); c.find(30); c.insert(31, *z); c.find(31); c.insert(32, *z)
        assert c.find(10) is not None and c.find(11) is None
    case(two_bit_freq_counter)

    def eviction_from_empty_small():                            # :575-595
        c = make(3, 3, 2, 1)
        for i in range(3): c.insert(i, _pol(2, 0), _pol(1, 0)); c.find(i)
        for i in range(3, 9): c.insert(i, _pol(2, 0), _pol(1, 0))
        s = c.stats(); assert s["size"] == 3 and s["evictions"] > 0
    case(eviction_from_empty_small)
    return ran
