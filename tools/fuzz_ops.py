"""Long runs of the randomised op tests of tests/test_ops_gpu.py over seeds the suite does not cover
(python tools/fuzz_ops.py FIRST LAST [--json profiles/rNN_fuzz_ops.jsonl]): grid subsample + both kinds of radius search on random shapes,
and the radius-search configuration fuzz, bit-exact vs the C++ oracle.  Every failure is printed with its assertion text and the run goes
on to LAST (a generator-premise failure must not hide the seeds behind it); one JSON line per run is appended to --json."""
import argparse
import json
import os
import sys
import time
import traceback

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_ops_gpu as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("first", type=int)
ap.add_argument("last", type=int)
ap.add_argument("--json", default=None, help="append a one-line JSON record of the run to this file")
ap.add_argument("--max-seconds", type=float, default=0.0, help="stop cleanly (and say where) after this much wall time; 0 = no limit")
args = ap.parse_args()

t0, bad, n, skipped, seed = time.time(), [], 0, 0, args.first - 1
for seed in range(args.first, args.last):
    for fn in (T.test_random_clouds_subsample_and_search, T.test_radius_search_fuzz_against_the_oracle):
        try:
            fn(seed)
            n += 1
        except pytest.skip.Exception:
            skipped += 1
        except AssertionError as e:
            where = traceback.extract_tb(e.__traceback__)[-1]
            bad.append({"test": fn.__name__, "seed": seed, "assert": str(e)[:200] or "(no message)", "line": "%s:%d" % (os.path.basename(where.filename), where.lineno)})
            print("FAIL", bad[-1], flush=True)
        except Exception as e:                                    # noqa: BLE001
            bad.append({"test": fn.__name__, "seed": seed, "error": repr(e)[:200]})
            print("ERROR", bad[-1], flush=True)
    if args.max_seconds and time.time() - t0 > args.max_seconds:
        break
rec = {"tool": "fuzz_ops", "first": args.first, "last_done": seed, "requested_last": args.last - 1, "cases_exact": n, "skipped_degenerate": skipped,
       "failures": bad, "seconds": round(time.time() - t0, 1)}
print("op fuzz: " + json.dumps(rec))
if args.json:
    with open(args.json, "a") as f:
        f.write(json.dumps(rec) + "\n")
