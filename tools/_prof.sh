set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p -o pf --output-format csv -- python3 $R/tools/prefill_only.py > $R/gpurun_out/prefill.log 2> $R/gpurun_out/prefill.err
cd $R
cat gpurun_out/prefill.log
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_p/pf_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.1f}us total={float(r['TotalDurationNs'])/1e6:8.2f}ms {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
