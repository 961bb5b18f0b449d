#!/bin/bash
# kernel stats + SQ counters of the inline layout sampler for the synthetic config 5 (18 objects)
out=$PWD/gpurun_out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_c5 /tmp/pmc_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_c5 -- python3 $repo/tools/profile_reset.py --task Ant_8Hazards_8Pillars_synthetic > $out/c5_kt.log 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob('/tmp/kt_c5/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'sample' in r['Name'] or 'scan' in r['Name'] or 'fake' in r['Name']: print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_c5 -- python3 $repo/tools/profile_reset.py --task Ant_8Hazards_8Pillars_synthetic > $out/c5_pmc.log 2>&1
python3 $repo/tools/pmc_means.py $(find /tmp/pmc_c5 -name "*counter_collection.csv" | head -1) > $out/c5_sampler_pmc_SQ.csv
grep -E "phase" $out/c5_sampler_pmc_SQ.csv | cut -d, -f1-3,7-10 | sed 's/gx::\(sample_phase._kernel\)[^"]*/\1/'
