#!/bin/bash
# Round-6 profile set in ONE gpurun call (outputs under gpurun_out/, summaries are copied to profiles/ afterwards; profiles/r06_traffic.json is written here so
# that the bench lines of step 2 quote it).  Profiled passes run the ONE-stream schedule (--encoder-streams 1): with two sub-batch streams kernels overlap and a
# traced duration is not the kernel's own.  PMC passes: one TCC counter per pass, --kernel-trace only beside --pmc, every pass under its own timeout.
#   1. HBM / fabric traffic per op, B/16 batch 32 and L/14 batch 16                       -> r6_traffic.json, r6_hbm_traffic_{b16,l14}.md
#   2. bench lines: default (5 windows, product input stage, cpu_baseline), L/14 batch 16, trained_like, batch 1, forward batch 8, 8 gloo ranks (soak)
#   3. rocprofv3 --kernel-trace --stats over the bench: B/16 batch 32, L/14 batch 16      -> r6_prof_*_summary.md (+ launch-by-launch listing of one step)
#   4. matrix-pipe / VALU busy + effective clock per kernel                                -> r6_pmc_b16.md, r6_pmc_l14.md
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/r6_traffic.json
for wl in "owlvit-base-patch16 32 b16" "owlvit-large-patch14 16 l14"; do
  set -- $wl
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r6_pmc_${3}_$c -o p -f csv -- python $R/bench.py --arch $1 --batch $2 --no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 --windows 1 > $R/gpurun_out/r6_pmc_${3}_$c.log 2>&1
  done
  (cd $R && python tools/pmc_traffic.py gpurun_out/r6_pmc_${3}_FETCH_SIZE gpurun_out/r6_pmc_${3}_WRITE_SIZE --json gpurun_out/r6_traffic.json --workload $1/$2 > gpurun_out/r6_hbm_traffic_$3.md)
  rm -rf $R/gpurun_out/r6_pmc_${3}_FETCH_SIZE $R/gpurun_out/r6_pmc_${3}_WRITE_SIZE
done
cp $R/gpurun_out/r6_traffic.json $R/profiles/r06_traffic.json
echo "== traffic"; cat $R/gpurun_out/r6_traffic.json
python $R/bench.py --steps 20 --warmup 3 2> $R/gpurun_out/r6_bench_default.err > $R/gpurun_out/r6_bench_default.json
python $R/bench.py --arch owlvit-large-patch14 --batch 16 --steps 6 --warmup 2 --windows 3 --no-cpu-baseline 2>/dev/null > $R/gpurun_out/r6_bench_l14.json
python $R/bench.py --weights trained_like --no-cpu-baseline --no-compare --steps 20 2>/dev/null > $R/gpurun_out/r6_bench_trained_like.json
python $R/bench.py --batch 1 --no-cpu-baseline --no-compare --steps 50 --warmup 5 2>/dev/null > $R/gpurun_out/r6_bench_batch1.json
python $R/bench.py --forward-only --batch 8 --no-cpu-baseline --no-compare --steps 50 --warmup 5 2>/dev/null > $R/gpurun_out/r6_bench_forward_batch8.json
(cd $R && timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --backend gloo --batch 4 --steps 4 --warmup 1 --windows 2 --no-cpu-baseline --no-compare 2> gpurun_out/r6_soak_gloo8.err | tail -1 > gpurun_out/r6_soak_gloo8.json)
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6_prof_b16 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --windows 1 > $R/gpurun_out/r6_prof_b16.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6_prof_l14 -o rf -- python $R/bench.py --no-cpu-baseline --no-compare --encoder-streams 1 --windows 1 --arch owlvit-large-patch14 --batch 16 --steps 4 --warmup 2 > $R/gpurun_out/r6_prof_l14.log 2>&1
cd $R
python tools/prof_summary.py $(ls gpurun_out/r6_prof_b16/*.db | head -1) 60 > gpurun_out/r6_prof_b16_summary.md
python tools/prof_summary.py $(ls gpurun_out/r6_prof_l14/*.db | head -1) 40 > gpurun_out/r6_prof_l14_summary.md
python tools/step_listing.py $(ls gpurun_out/r6_prof_b16/*.db | head -1) 5 --list > gpurun_out/r6_step_listing_b16.txt 2>&1
python tools/step_listing.py $(ls gpurun_out/r6_prof_l14/*.db | head -1) 3 > gpurun_out/r6_step_listing_l14.txt 2>&1
rm -rf gpurun_out/r6_prof_b16/*.db gpurun_out/r6_prof_l14/*.db 2>/dev/null
cd /tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
B="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
COMMON="--no-cpu-baseline --no-compare --no-kernel-events --encoder-streams 1 --steps 2 --warmup 1 --windows 1"
timeout 400 rocprofv3 --kernel-trace --pmc $A -d $R/gpurun_out/r6_pmc_b16_a -o p -f csv -- python $R/bench.py $COMMON > $R/gpurun_out/r6_pmc_b16_a.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $B -d $R/gpurun_out/r6_pmc_b16_b -o p -f csv -- python $R/bench.py $COMMON > $R/gpurun_out/r6_pmc_b16_b.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc $A -d $R/gpurun_out/r6_pmc_l14_a -o p -f csv -- python $R/bench.py $COMMON --arch owlvit-large-patch14 --batch 16 > $R/gpurun_out/r6_pmc_l14_a.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc $B -d $R/gpurun_out/r6_pmc_l14_b -o p -f csv -- python $R/bench.py $COMMON --arch owlvit-large-patch14 --batch 16 > $R/gpurun_out/r6_pmc_l14_b.log 2>&1
cd $R
python tools/pmc_pipes.py gpurun_out/r6_pmc_b16_a gpurun_out/r6_pmc_b16_b > gpurun_out/r6_pmc_b16.md
python tools/pmc_pipes.py gpurun_out/r6_pmc_l14_a gpurun_out/r6_pmc_l14_b > gpurun_out/r6_pmc_l14.md
rm -rf gpurun_out/r6_pmc_b16_a gpurun_out/r6_pmc_b16_b gpurun_out/r6_pmc_l14_a gpurun_out/r6_pmc_l14_b
echo "== default bench"; cut -c1-600 gpurun_out/r6_bench_default.json; echo; echo "== l14"; cut -c1-300 gpurun_out/r6_bench_l14.json; echo; head -20 gpurun_out/r6_prof_b16_summary.md; head -14 gpurun_out/r6_pmc_b16.md
