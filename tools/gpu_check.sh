#!/bin/bash
# GPU box: the gpu test suite + one bench line per BASELINE config share.  usage: bash tools/gpu_check.sh <tag>
set -o pipefail
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc $? $(python -c "import json,sys; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('%.4g samples/s  kernel %.3f ms cold %.3f  valu frac %.3f' % (d['value'] or 0, r['kernel_ms'], r['kernel_ms_cold_first_launch'] or 0, r['frac']))" 2>&1)"; }
run bench_c2 --steps 20 --warmup 5
run bench_indirect --scene Scene_indirect --steps 10 --warmup 3
run bench_c3_r0 --config 3 --rank 0/8 --steps 10 --warmup 3
run bench_c3_r4 --config 3 --rank 4/8 --steps 10 --warmup 3
run bench_c3_r7 --config 3 --rank 7/8 --steps 10 --warmup 3
run bench_c4 --config 4 --steps 10 --warmup 3
run bench_c5_r4 --config 5 --rank 4/8 --steps 3 --warmup 1
run bench_c5_r0 --config 5 --rank 0/8 --steps 3 --warmup 1
