#!/bin/bash
# persistent fp32 grouped-MLP launches (unit queue): parity first, then the bench line
mkdir -p gpurun_out/r03j
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03j
timeout 900 python -m pytest tests/test_gpu_policy.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 3 --all-slots-steps 0 > $O/bench.log 2>&1
python - "$O/bench.log" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("NO LINE"); print(open(sys.argv[1]).read()[-1500:]); sys.exit(0)
d = json.loads(l[-1]); k = d["kernels_ms"]
print("step %.2f ms  sa1 %.3f  sa2 %.3f  fps %.3f  bq %.3f  chain %.3f  roof %.3f  pipelined %.2f" % (d["ms_per_step"], k["sa1_mlp"], k["sa2_mlp"], k["fps"], k["ball_query"], k["sa3_chain"], d["roofline"]["frac"], d.get("pipelined_two_streams", {}).get("ms_per_step", 0)))
print(json.dumps(d["result_check"])[:300])
PY
