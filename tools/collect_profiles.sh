#!/bin/bash
# Collect every rocprofv3 summary bench.py / DESIGN.md quote, on the GPU box (through gpurun, from the repo root):
#   tools/collect_profiles.sh r05      -> gpurun_out/r05_*.csv, r05_build_id.txt  (copy them into profiles/)
# Counter passes are separate runs with --pmc only (no --kernel-trace/--stats mixed in).
set -e
tag=${1:-r05}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 -c "
from guardx_amd import _native
l = _native.load(); print(l.gx_build_id().decode()); print(l.gx_build_compiler().decode())" > $out/${tag}_build_id.txt
kt() { # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_${tag}_$name -- "$@" > $out/${tag}_${name}_kt.log 2>&1
  cp $(find /tmp/kt_${tag}_$name -name "*kernel_stats.csv" | head -1) $out/${tag}_${name}_kernel_stats.csv
}
pmc() { # name, counters-label, counters..., -- command...
  local name=$1 label=$2; shift 2
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d /tmp/pmc_${tag}_${name}_$label -- "$@" > $out/${tag}_${name}_pmc_$label.log 2>&1
  python3 tools/pmc_means.py $(find /tmp/pmc_${tag}_${name}_$label -name "*counter_collection.csv" | head -1) > $out/${tag}_${name}_pmc_$label.csv
}
ROLL="python3 tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --repeat 20"
STEP="python3 tools/profile_step.py --mode step --env-num 4194304 --launches 20"
FUSE="python3 tools/profile_step.py --mode rollout --env-num 4194304 --launches 16"
RST="python3 tools/profile_reset.py"
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU"
kt rollout_N2000_T200 $ROLL
pmc rollout_N2000_T200 FETCH_SIZE FETCH_SIZE -- $ROLL
pmc rollout_N2000_T200 WRITE_SIZE WRITE_SIZE -- $ROLL
pmc rollout_N2000_T200 SQ $SQ -- $ROLL
kt step_N4194304 $STEP
pmc step_N4194304 FETCH_SIZE FETCH_SIZE -- $STEP
pmc step_N4194304 WRITE_SIZE WRITE_SIZE -- $STEP
pmc thread_rollout_N4194304_K16 FETCH_SIZE FETCH_SIZE -- $FUSE
pmc thread_rollout_N4194304_K16 WRITE_SIZE WRITE_SIZE -- $FUSE
for rb in swimmer ant walker; do   # the two-kernel rollout of the other robots (round 3: lane-group dynamics tape for Ant / Walker)
  RB="python3 tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --repeat 10 --robot xmls/$rb.xml"
  kt ${rb}_rollout_N2000_T200 $RB
  pmc ${rb}_rollout_N2000_T200 SQ $SQ -- $RB
done
kt sampler $RST
pmc sampler SQ SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES -- $RST
kt bench_noextras python3 bench.py --no-cpu-baseline --no-extras
# round 4: one GPU playing rank 0 of 8 in the default multi-GPU epoch (per-kernel times of a rank's epoch), the step-wise
# policy kernels of the wider networks, the 18-object sampler of the synthetic config 5
kt rehearsal_rank0_of_8 python3 tools/rehearse_rank.py --world 8 --epochs 30
kt policy_widths python3 tools/bench_policy_widths.py
kt sampler_config5 python3 tools/profile_reset.py --task Ant_8Hazards_8Pillars_synthetic
pmc sampler_config5 SQ SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES -- python3 tools/profile_reset.py --task Ant_8Hazards_8Pillars_synthetic
# round 5: the exhaustive reciprocal probe (tools/probes/rcp_exact_probe.hip), rank rehearsals of the Point and the Ant
(cd tools/probes && hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o rcp_exact_probe rcp_exact_probe.hip) && ./tools/probes/rcp_exact_probe > $out/${tag}_rcp_exact_probe.log 2>&1
python3 tools/rehearse_rank.py --world 8 --epochs 30 --json $out/${tag}_rehearsal_rank0_of_8.json > /dev/null 2>&1
python3 tools/rehearse_rank.py --world 8 --epochs 30 --robot xmls/ant.xml --json $out/${tag}_rehearsal_rank0_of_8_ant.json > /dev/null 2>&1
python3 tools/rehearse_rank.py --world 8 --epochs 30 --robot xmls/swimmer.xml --json $out/${tag}_rehearsal_rank0_of_8_swimmer.json > /dev/null 2>&1
python3 tools/rehearse_rank.py --world 8 --epochs 20 --robot xmls/walker.xml --json $out/${tag}_rehearsal_rank0_of_8_walker.json > /dev/null 2>&1
(cd tools/probes && hipcc -O3 --offload-arch=gfx950 -o mfma_rate_probe mfma_rate_probe.hip) && ./tools/probes/mfma_rate_probe > $out/${tag}_mfma_rate_probe.log 2>&1
rm -f $out/${tag}_*_kt.log $out/${tag}_*_pmc_*.log
ls $out/${tag}_* | wc -l
