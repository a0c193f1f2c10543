#!/bin/bash
# Old and new build of the Riccati operator on one box: bitwise comparison of every output + timing -> gpurun_out/dare_ab.txt
cd /root/repo
{
for so in variants/*.so; do
  echo "== $so"
  LQRRT_LIB=$PWD/$so python tools/dare_dump.py gpurun_out/dare_$(basename $so .so).npz
done
python - <<'PY'
import numpy as np, glob
fs = sorted(glob.glob('gpurun_out/dare_*.npz'))
a = np.load(fs[0])
for f in fs[1:]:
    b = np.load(f)
    bad = [k for k in a.files if not (a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)))]
    print(fs[0], 'vs', f, ': bitwise identical' if not bad else ': DIFFER in %s' % bad)
    for k in bad:
        print('   ', k, 'max abs diff', np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float))))
PY
} 2>&1 | tee gpurun_out/dare_ab.txt
