#!/bin/bash
# GPU box: development statistics (dev libraries with STATS counters).  usage: bash tools/gpu_stats.sh <tag>
TAG=${1:-stats}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{
echo "== mesh stats (STATS=1), config 4 scene, 8 spp"; timeout -k 10 200 python tests/mesh_stats.py --spp 8
echo "== mesh histogram (STATS=2)"; SRT_STATS_MODE=2 timeout -k 10 200 python tests/mesh_stats.py --spp 8
for sc in "--scene Scene1" "--scene Scene_indirect" "--scene Scene1 --mesh 224"; do echo "== section profile $sc"; timeout -k 10 200 python tests/section_profile.py $sc --spp 32; done
} > $OUT/stats.log 2>&1
cat $OUT/stats.log
