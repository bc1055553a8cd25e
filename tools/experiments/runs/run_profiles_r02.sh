#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# Round-2 profile set, run on the GPU box through gpurun (outputs under gpurun_out/, summaries copied to profiles/ afterwards):
# Every pass runs the ONE-stream schedule (--encoder-streams 1): with two sub-batch streams (the default) kernels overlap and a kernel's
# traced duration is not its own.  The default-schedule bench line is taken separately (pass 0).
#   0. python bench.py (default schedule)                                                   -> r2_bench_default.log
#   1. rocprofv3 --kernel-trace --stats over the bench (B/16, batch 32)          -> r2_prof_b16/
#   2. the same for L/14 840x840 batch 16                                                -> r2_prof_l14/
#   3. HBM traffic: two --pmc passes (FETCH_SIZE, WRITE_SIZE -- each alone: together they exceed the TCC counter slots) over a
#      short bench run, --kernel-trace only, every pass under its own timeout             -> r2_pmc_fetch/, r2_pmc_write/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --no-compare > $R/gpurun_out/r2_bench_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof_b16 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 > $R/gpurun_out/r2_prof_b16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof_l14 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --arch owlvit-large-patch14 --batch 16 --steps 4 --warmup 2 > $R/gpurun_out/r2_prof_l14.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2_pmc_fetch -o p -f csv -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r2_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2_pmc_write -o p -f csv -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 > $R/gpurun_out/r2_pmc_write.log 2>&1
#   4. steady-state launches per step: two kernel traces (--steps 4 / --steps 14), differenced by tools/steady_state_counts.py -> r2_steady_state.md
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r2_ss4 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 4 > $R/gpurun_out/r2_ss4.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r2_ss14 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 14 > $R/gpurun_out/r2_ss14.log 2>&1
cd $R
python tools/steady_state_counts.py $(ls gpurun_out/r2_ss4/*.db | head -1) 4 $(ls gpurun_out/r2_ss14/*.db | head -1) 14 > gpurun_out/r2_steady_state.md
rm -rf gpurun_out/r2_ss4 gpurun_out/r2_ss14
python tools/prof_summary.py $(ls gpurun_out/r2_prof_b16/*.db | head -1) 60 > gpurun_out/r2_prof_b16_summary.md
python tools/prof_summary.py $(ls gpurun_out/r2_prof_l14/*.db | head -1) 40 > gpurun_out/r2_prof_l14_summary.md
python tools/pmc_traffic.py gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write --json gpurun_out/r2_traffic.json > gpurun_out/r2_hbm_traffic.md
tail -1 gpurun_out/r2_bench_default.log | cut -c1-300; grep -h '"metric"' gpurun_out/r2_prof_b16.log | cut -c1-300; grep -h '"metric"' gpurun_out/r2_prof_l14.log | cut -c1-300; head -12 gpurun_out/r2_hbm_traffic.md; cat gpurun_out/r2_traffic.json
