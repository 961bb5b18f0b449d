#!/bin/bash
# rocprofv3 --kernel-trace --stats of one gx_rollout configuration; prints the per-kernel averages
# usage: tools/prof_rollout_stats.sh <tag> [profile_step.py args]
tag=$1; shift
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python3 $out/../tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --repeat 20 "$@" > $out/kt_$tag.log 2>&1
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
cp $f $out/kt_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tape_kernel' in r['Name'] or 'rollout' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
