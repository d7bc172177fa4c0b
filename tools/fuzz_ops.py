"""Long runs of the randomised op tests of tests/test_ops_gpu.py over seeds the suite does not cover (python tools/fuzz_ops.py FIRST LAST):
grid subsample + both kinds of radius search on random shapes, and the radius-search configuration fuzz, bit-exact vs the C++ oracle."""
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_ops_gpu as T  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
t0, bad, n = time.time(), [], 0
for seed in range(first, last):
    for fn in (T.test_random_clouds_subsample_and_search, T.test_radius_search_fuzz_against_the_oracle):
        try:
            fn(seed)
            n += 1
        except pytest.skip.Exception:
            pass
        except Exception as e:                                    # noqa: BLE001
            bad.append((fn.__name__, seed))
            print("FAIL", fn.__name__, seed, repr(e)[:200])
    if len(bad) >= 5:
        break
print("op fuzz: %d cases, seeds %d..%d, failures %s, %.0f s" % (n, first, seed, bad, time.time() - t0))
