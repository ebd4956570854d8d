#!/bin/bash
# usage (on the GPU box): tools/gpu_check.sh <tag> -- GPU parity tests, per-kernel tables of cfg2/cfg3 steps, the bounded side configs
TAG=${1:-chk}
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
tail -2 gpurun_out/${TAG}_tests.log
tools/prof_steps.sh $TAG cfg2 cfg3 || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_bench_cfg2.json 2> gpurun_out/${TAG}_bench_cfg2.err || { tail -20 gpurun_out/${TAG}_bench_cfg2.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_cfg2.json"))
print("cfg2 main ms/step", d["ms_per_step"])
for o in d.get("other_configs", []):
    print(o["workload"][:5], "graph" if o["hip_graph"] else "eager", o.get("ms_per_step"))
PY
if [ "$2" = "full" ]; then
  timeout -k 10 400 python -u bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("cfg4 ms/step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
for o in d.get("other_configs", []):
    print(o["workload"][:5], "graph" if o["hip_graph"] else "eager", o.get("ms_per_step"))
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
fi
