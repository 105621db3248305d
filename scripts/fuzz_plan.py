"""randomised parity sweep of the TT path (tests/fuzz_cases.py: run_plan_cases) for a time budget.
usage: python scripts/fuzz_plan.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fbtt-embedding_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import fuzz_cases

t0 = time.time()
n, routes = fuzz_cases.run_plan_cases(seed=int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                                      budget=float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
print(f"{n} cases ok in {time.time() - t0:.0f} s; routes {routes}")
