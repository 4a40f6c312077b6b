#!/bin/bash
# Every profile of a round in one GPU call: bench (stats + PMC passes), the 64-episode streaming launch, the evaluation loop,
# the value network and the render / observation stage.  bash scripts/profile_all.sh r04 ; summaries land under gpurun_out/.
TAG=${1:-r04}
bash scripts/profile_bench.sh $TAG > gpurun_out/profile_bench_$TAG.log 2>&1
bash scripts/profile_stream64.sh $TAG > gpurun_out/profile_stream64_$TAG.log 2>&1
bash scripts/profile_eval.sh $TAG > gpurun_out/profile_eval_$TAG.log 2>&1
bash scripts/profile_cnn.sh $TAG > gpurun_out/profile_cnn_$TAG.log 2>&1
bash scripts/profile_render.sh $TAG > gpurun_out/profile_render_$TAG.log 2>&1
du -sh gpurun_out/prof_$TAG
tail -3 gpurun_out/profile_*_$TAG.log
