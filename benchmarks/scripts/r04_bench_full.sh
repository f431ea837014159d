#!/bin/bash
# the driver's command: python bench.py (headline + roofline + other_workloads + cpu_baseline)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_full
( time python bench.py ) > gpurun_out/r04_full/bench.json 2> gpurun_out/r04_full/bench.err
tail -4 gpurun_out/r04_full/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_full/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('other_workloads'), indent=1)[:1500])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
