import sys, os
sys.path.insert(0, '/root/repo/tools')
import numpy as np
import fuzz_parity as fp
seed=int(sys.argv[1]); cases=int(sys.argv[2])
rng, rng2, rng3 = np.random.RandomState(seed), np.random.RandomState(seed + 7919), np.random.RandomState(seed + 104729)
for k in range(cases):
    c = fp.draw_case(rng, rng2, None, rng3)
    print("case", k, fp.describe(c), flush=True)
    ok = fp.run_case(c)
    print("   ->", ok, flush=True)
