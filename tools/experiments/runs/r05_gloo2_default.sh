#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# the multi-rank flow with the DEFAULT flags the driver uses (compare run included), two and eight gloo ranks on the one GPU (test-only backend), small config
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for n in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --backend gloo --arch small --batch 4 --steps 6 --warmup 2 2> gpurun_out/r5_gloo${n}_default.err | tail -1 > gpurun_out/r5_gloo${n}_default.json
echo "rc $?"; cut -c1-1500 gpurun_out/r5_gloo${n}_default.json; tail -3 gpurun_out/r5_gloo${n}_default.err
done
