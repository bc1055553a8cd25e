#!/bin/bash
# ARCHIVED (round 6): the record of a gpurun call of an earlier round, kept as it was run.  Paths (tools/..., ab_libs/...) are those of that round;
# some copy untracked library builds over the shipped libowlhip.so.  It refuses to run unless OWL_RUN_ARCHIVED=1.
if [ "${OWL_RUN_ARCHIVED:-0}" != "1" ]; then echo "$0: archived record of a past gpurun call (see tools/experiments/README.md); set OWL_RUN_ARCHIVED=1 to run it anyway" >&2; exit 1; fi
# Round-3 counter set for the kernels that SHIP (VERDICT r02 "Missing #2"): matrix-pipe busy, VALU busy, wait / stall shares and the effective clock
# (GRBM_GUI_ACTIVE / wall of the same profiled launch) per kernel of the train step, one-stream schedule, B/16 batch 32 and L/14 batch 16.
# Two SQ passes per model, GRBM_GUI_ACTIVE in both; --kernel-trace only (no other trace domain next to --pmc); every pass under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
COMMON="--no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1"
timeout 400 rocprofv3 --kernel-trace --pmc $A -d $R/gpurun_out/${TAG}_pmc_b16_a -o p -f csv -- python $R/bench.py $COMMON > $R/gpurun_out/${TAG}_pmc_b16_a.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $B -d $R/gpurun_out/${TAG}_pmc_b16_b -o p -f csv -- python $R/bench.py $COMMON > $R/gpurun_out/${TAG}_pmc_b16_b.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc $A -d $R/gpurun_out/${TAG}_pmc_l14_a -o p -f csv -- python $R/bench.py $COMMON --arch owlvit-large-patch14 --batch 16 > $R/gpurun_out/${TAG}_pmc_l14_a.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc $B -d $R/gpurun_out/${TAG}_pmc_l14_b -o p -f csv -- python $R/bench.py $COMMON --arch owlvit-large-patch14 --batch 16 > $R/gpurun_out/${TAG}_pmc_l14_b.log 2>&1
cd $R
python tools/pmc_pipes.py gpurun_out/${TAG}_pmc_b16_a gpurun_out/${TAG}_pmc_b16_b > gpurun_out/${TAG}_pmc_b16.md
python tools/pmc_pipes.py gpurun_out/${TAG}_pmc_l14_a gpurun_out/${TAG}_pmc_l14_b > gpurun_out/${TAG}_pmc_l14.md
# keep the merged-back payload small: the per-dispatch csv of a whole bench is tens of MB
for d in gpurun_out/${TAG}_pmc_*_?; do rm -f $d/*kernel_trace.csv; gzip -f $d/*counter_collection.csv 2>/dev/null; done
head -30 gpurun_out/${TAG}_pmc_b16.md; head -30 gpurun_out/${TAG}_pmc_l14.md
