#!/bin/bash
# Headline step with the default library and with each A/B build named on the command line (tools/ab_build.sh <name> ...):
# usage (on the GPU box): bash tools/ab_bench.sh <name> [<name> ...]   -> one line per build: step and per-kernel ms
mkdir -p gpurun_out/ab
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/ab
run() {
  MPX_LIB_PATH=$2 timeout 300 python bench.py --steps 5 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --all-slots-steps 0 > $O/bench_$1.log 2>&1
  python - "$1" "$O/bench_$1.log" <<'PY'
import json, sys
l = [x for x in open(sys.argv[2]) if x.startswith("{")]
if not l:
    print(sys.argv[1], "NO LINE"); sys.exit(0)
d = json.loads(l[-1]); k = d["kernels_ms"]
print("%-8s step %.2f ms  sa1 %.3f  sa2 %.3f  fps %.3f  bq %.3f  chain %.3f" % (sys.argv[1], d["ms_per_step"], k["sa1_mlp"], k["sa2_mlp"], k["fps"], k["ball_query"], k["sa3_chain"]))
PY
}
run base ""
for v in "$@"; do run $v $REPO/build_ab/libmpinets_hip_$v.so; done
run base2 ""
