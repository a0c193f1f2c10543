"""One case of tools/fuzz_parity.py by number, nothing else run before it:  python tools/fuzz_one.py <seed> <case>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import fuzz_parity as fp
seed, case = int(sys.argv[1]), int(sys.argv[2])
rng, rng2, rng3 = np.random.RandomState(seed), np.random.RandomState(seed + 7919), np.random.RandomState(seed + 104729)
for k in range(case + 1):
    c = fp.draw_case(rng, rng2, None, rng3)
print("case", case, fp.describe(c), flush=True)
print("   ->", fp.run_case(c, verbose=True), flush=True)
