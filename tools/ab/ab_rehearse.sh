for m in aux torch aux torch; do
GX_HANDOFF_STREAM=$m python tools/rehearse_rank.py --world 8 --epochs 30 --json gpurun_out/reh_$m.json > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/reh_$m.json")); print("$m point", d["one_gpu_own_sampler"]["ms_per_epoch"], d["expand_all"]["ms_per_epoch"], d["expand_local"]["ms_per_epoch"])
PY
done
for m in aux torch; do
GX_HANDOFF_STREAM=$m python tools/rehearse_rank.py --world 8 --epochs 30 --robot xmls/ant.xml --json gpurun_out/reh_ant_$m.json > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/reh_ant_$m.json")); print("$m ant", d["one_gpu_own_sampler"]["ms_per_epoch"], d["expand_all"]["ms_per_epoch"], d["expand_local"]["ms_per_epoch"])
PY
done
