cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pw -- python3 $GRAFT_REPO_ROOT/tools/bench_policy_widths.py > $GRAFT_REPO_ROOT/gpurun_out/pw.log 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob('/tmp/kt_pw/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%")
PY
