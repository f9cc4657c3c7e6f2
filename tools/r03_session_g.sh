#!/bin/bash
# A/B builds (tools/ab_build.sh): SA1 queries per wave, wave-per-query ball-query walk
mkdir -p gpurun_out/r03g
export PYTHONUNBUFFERED=1
REPO=$(pwd); O=$REPO/gpurun_out/r03g
run() {  # name -> one bench line reduced to the numbers that matter
  MPX_LIB_PATH=$2 timeout 300 python bench.py --steps 5 --warmup 1 --cpu-envs 0 --extra 0 --fast-steps 0 --pipeline-steps 0 --all-slots-steps 0 > $O/bench_$1.log 2>&1
  python - "$1" "$O/bench_$1.log" <<'PY'
import json, sys
l = [x for x in open(sys.argv[2]) if x.startswith("{")]
if not l:
    print(sys.argv[1], "NO LINE"); sys.exit(0)
d = json.loads(l[-1]); k = d["kernels_ms"]
print("%-8s step %.2f ms  sa1 %.3f  sa2 %.3f  fps %.3f  bq %.3f  chain %.3f  tiles1 %d" % (sys.argv[1], d["ms_per_step"], k["sa1_mlp"], k["sa2_mlp"], k["fps"], k["ball_query"], k["sa3_chain"], k["sa1_tiles_walked"]))
PY
}
run base ""
run sa1q32 $REPO/build_ab/libmpinets_hip_sa1q32.so
run sa1q8 $REPO/build_ab/libmpinets_hip_sa1q8.so
run bqcoop $REPO/build_ab/libmpinets_hip_bqcoop.so
MPX_LIB_PATH=$REPO/build_ab/libmpinets_hip_bqcoop.so timeout 600 python -m pytest tests/test_gpu_pointnet.py tests/test_gpu_soak.py -m gpu -q --tb=short -p no:cacheprovider -k "ball or soak or hits" > $O/pytest_bqcoop.log 2>&1; tail -3 $O/pytest_bqcoop.log | cut -c1-200
