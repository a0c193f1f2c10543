#!/bin/bash
# The round's evidence, collected on the GPU box (gpurun -- 'bash tools/profile_round.sh'):
#   1. plain bench line                                   -> gpurun_out/bench_plain.json
#   2. rocprofv3 --kernel-trace --stats of the same cmd   -> gpurun_out/prof_r1/bench_kernel_stats.csv (+ bench line under rocprof)
#   3. PMC pass FETCH_SIZE  (own run, kernel-trace only)  -> gpurun_out/prof_r1_fetch/bench_counter_collection.csv
#   4. PMC pass WRITE_SIZE  (own run)                     -> gpurun_out/prof_r1_write/bench_counter_collection.csv
# then, back in the container: python tools/summarize_profiles.py rNN
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err < /dev/null
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extras"
rm -rf $O/prof_r1 $O/prof_r1_fetch $O/prof_r1_write
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r1 -o bench -- $CMD > $O/bench_under_rocprof.json 2> $O/prof_r1.log < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_r1_fetch -o bench -- $CMD > /dev/null 2> $O/prof_r1_fetch.log < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_r1_write -o bench -- $CMD > /dev/null 2> $O/prof_r1_write.log < /dev/null
# keep only what the summary needs (the traces are large)
for d in prof_r1 prof_r1_fetch prof_r1_write; do
  find $O/$d -type f ! -name "bench_kernel_stats.csv" ! -name "bench_counter_collection.csv" -delete
done
find $O/prof_r1 -name "bench_kernel_stats.csv" -exec mv {} $O/prof_r1/ \; 2>/dev/null
find $O/prof_r1_fetch -name "bench_counter_collection.csv" -exec mv {} $O/prof_r1_fetch/ \; 2>/dev/null
find $O/prof_r1_write -name "bench_counter_collection.csv" -exec mv {} $O/prof_r1_write/ \; 2>/dev/null
ls -la $O/prof_r1 $O/prof_r1_fetch $O/prof_r1_write | head -20
tail -c 300 $O/bench_plain.json
