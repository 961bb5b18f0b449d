for m in 1 2; do
python tools/rehearse_rank.py --world 8 --epochs 30 --json gpurun_out/reh_p.json > /dev/null 2> gpurun_out/reh_err.log || tail -5 gpurun_out/reh_err.log
python - <<PY
import json
d=json.load(open("gpurun_out/reh_p.json")); print("point", d["one_gpu_own_sampler"]["ms_per_epoch"], d["expand_all"]["ms_per_epoch"], d["expand_local"]["ms_per_epoch"], d["expand_all"]["model"])
PY
done
python tools/rehearse_rank.py --world 8 --epochs 30 --robot xmls/ant.xml --json gpurun_out/reh_a.json > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/reh_a.json")); print("ant", d["one_gpu_own_sampler"]["ms_per_epoch"], d["expand_all"]["ms_per_epoch"], d["expand_local"]["ms_per_epoch"], d["expand_all"]["model"])
PY
