#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# Round-3 profile set in one gpurun call (outputs under gpurun_out/, summaries copied to profiles/ afterwards).  Every profiled pass runs the
# ONE-stream schedule (--encoder-streams 1): with two sub-batch streams kernels overlap and a traced duration is not the kernel's own.
#   0. python bench.py (default schedule, with cpu_baseline)                              -> r3_bench_default.log
#   1. rocprofv3 --kernel-trace --stats over the bench: B/16 batch 32, L/14 batch 16       -> r3_prof_b16/, r3_prof_l14/
#   2. HBM traffic: FETCH_SIZE and WRITE_SIZE, one TCC counter per pass                    -> r3_pmc_fetch/, r3_pmc_write/ -> r3_traffic.json
#   3. steady-state launches per step (two traces differenced)                            -> r3_steady_state.md
#   4. matrix-pipe / VALU busy + effective clock per kernel (tools/run_pmc_r03.sh)         -> r3_pmc_b16.md, r3_pmc_l14.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/r3_bench_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_prof_b16 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 > $R/gpurun_out/r3_prof_b16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_prof_l14 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --arch owlvit-large-patch14 --batch 16 --steps 4 --warmup 2 > $R/gpurun_out/r3_prof_l14.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r3_pmc_fetch -o p -f csv -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r3_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r3_pmc_write -o p -f csv -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r3_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r3_ss4 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 4 > $R/gpurun_out/r3_ss4.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r3_ss14 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 14 > $R/gpurun_out/r3_ss14.log 2>&1
cd $R
python tools/steady_state_counts.py $(ls gpurun_out/r3_ss4/*.db | head -1) 4 $(ls gpurun_out/r3_ss14/*.db | head -1) 14 > gpurun_out/r3_steady_state.md
rm -rf gpurun_out/r3_ss4 gpurun_out/r3_ss14
python tools/prof_summary.py $(ls gpurun_out/r3_prof_b16/*.db | head -1) 60 > gpurun_out/r3_prof_b16_summary.md
python tools/prof_summary.py $(ls gpurun_out/r3_prof_l14/*.db | head -1) 40 > gpurun_out/r3_prof_l14_summary.md
python tools/pmc_traffic.py gpurun_out/r3_pmc_fetch gpurun_out/r3_pmc_write --json gpurun_out/r3_traffic.json > gpurun_out/r3_hbm_traffic.md
rm -f gpurun_out/r3_pmc_fetch/*kernel_trace.csv gpurun_out/r3_pmc_write/*kernel_trace.csv; gzip -f gpurun_out/r3_pmc_fetch/*counter_collection.csv gpurun_out/r3_pmc_write/*counter_collection.csv 2>/dev/null
rm -rf gpurun_out/r3_prof_b16/*.db gpurun_out/r3_prof_l14/*.db 2>/dev/null
bash tools/run_pmc_r03.sh r3 > /dev/null 2>&1
tail -1 gpurun_out/r3_bench_default.log | cut -c1-600; grep -h '"metric"' gpurun_out/r3_prof_b16.log | cut -c1-200; grep -h '"metric"' gpurun_out/r3_prof_l14.log | cut -c1-200; head -8 gpurun_out/r3_hbm_traffic.md; cat gpurun_out/r3_traffic.json; head -12 gpurun_out/r3_steady_state.md | tail -6
