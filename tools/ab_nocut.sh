#!/bin/bash
# parity + bench A/B on the GPU box: default build (round-2 rule + final-prefix convergence) vs LQRRT_NOCUT=1
R=/root/repo
O=$R/gpurun_out/nocut
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=" $O/pytest.log | tail -5
LQRRT_NOCUT=1 timeout 900 python -m pytest tests/test_hip_vs_coracle.py tests/test_fuzz_gpu.py tests/test_hip_parity.py -m gpu -x -q > $O/pytest_nocut.log 2>&1; echo "pytest nocut rc=$?" >> $O/pytest_nocut.log
grep -E "passed|failed|error|rc=" $O/pytest_nocut.log | tail -5
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/bench_cut_$i.json 2> $O/bench_cut_$i.err
  LQRRT_NOCUT=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $O/bench_nocut_$i.json 2> $O/bench_nocut_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('/root/repo/gpurun_out/nocut/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), {k: round(d[k], 2) for k in d if k.endswith('1024')})
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-600:])
PY
