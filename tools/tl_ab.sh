cd /root/repo
bash tools/timeline.sh; cp gpurun_out/kt_tail.csv gpurun_out/kt_nocut.csv; python tools/timeline_report.py gpurun_out/kt_nocut.csv
LQRRT_NOCUT=0 bash tools/timeline.sh; cp gpurun_out/kt_tail.csv gpurun_out/kt_cut.csv; python tools/timeline_report.py gpurun_out/kt_cut.csv
python - <<'PY'
import csv, collections
for tag in ('nocut', 'cut'):
    rows = list(csv.DictReader(open('/root/repo/gpurun_out/kt_%s.csv' % tag)))
    d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'k_steer' in r['Kernel_Name']][300:]
    h = collections.Counter(int(x // 4) * 4 for x in d)
    print(tag, 'k_steer launches', len(d), 'mean %.1f us' % (sum(d) / len(d)), 'hist(4us):', sorted(h.items()))
PY
