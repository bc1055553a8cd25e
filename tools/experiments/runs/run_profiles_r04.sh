#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# Round-4 profile set in one gpurun call (outputs under gpurun_out/, summaries copied to profiles/ afterwards).  Profiled passes run the ONE-stream
# schedule (--encoder-streams 1): with two sub-batch streams kernels overlap and a traced duration is not the kernel's own.
#   1. HBM / fabric traffic per op, B/16 batch 32 and L/14 batch 16 (tools/run_traffic_r04.sh)   -> r4_traffic.json (+ profiles/r04_traffic.json for step 2)
#   2. bench lines: default (with cpu_baseline), L/14 batch 16, trained_like                        -> r4_bench_default.json, r4_bench_l14.json, r4_bench_trained_like.json
#   3. rocprofv3 --kernel-trace --stats over the bench: B/16 batch 32, L/14 batch 16                -> r4_prof_*_summary.md
#   4. matrix-pipe / VALU busy + effective clock per kernel (tools/run_pmc_r03.sh r4)               -> r4_pmc_b16.md, r4_pmc_l14.md
R=$GRAFT_REPO_ROOT
bash $R/tools/run_traffic_r04.sh > $R/gpurun_out/r4_traffic.log 2>&1
cp $R/gpurun_out/r4_traffic.json $R/profiles/r04_traffic.json
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2> $R/gpurun_out/r4_bench_default.err > $R/gpurun_out/r4_bench_default.json
python $R/bench.py --arch owlvit-large-patch14 --batch 16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null > $R/gpurun_out/r4_bench_l14.json
python $R/bench.py --weights trained_like --no-cpu-baseline 2>/dev/null > $R/gpurun_out/r4_bench_trained_like.json
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4_prof_b16 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 > $R/gpurun_out/r4_prof_b16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4_prof_l14 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --arch owlvit-large-patch14 --batch 16 --steps 4 --warmup 2 > $R/gpurun_out/r4_prof_l14.log 2>&1
cd $R
python tools/prof_summary.py $(ls gpurun_out/r4_prof_b16/*.db | head -1) 60 > gpurun_out/r4_prof_b16_summary.md
python tools/prof_summary.py $(ls gpurun_out/r4_prof_l14/*.db | head -1) 40 > gpurun_out/r4_prof_l14_summary.md
rm -rf gpurun_out/r4_prof_b16/*.db gpurun_out/r4_prof_l14/*.db 2>/dev/null
bash tools/run_pmc_r03.sh r4 > /dev/null 2>&1
cut -c1-500 gpurun_out/r4_bench_default.json; cut -c1-300 gpurun_out/r4_bench_l14.json; cut -c1-300 gpurun_out/r4_bench_trained_like.json; cat gpurun_out/r4_traffic.json; head -20 gpurun_out/r4_prof_b16_summary.md
