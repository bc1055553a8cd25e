#!/bin/bash
# End-of-round check in ONE gpurun call: (1) the two TCC traffic passes per workload on the final kernel sources -> gpurun_out/r6_traffic.json (copy it over
# profiles/r06_traffic.json: bench.py quotes it only while its digest matches csrc/), (2) the whole GPU suite, (3) smoke(), (4) the default bench line.
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
cd $R
echo "== traffic"; cat gpurun_out/r6_traffic.json
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6_final_tests_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/r6_final_tests_full.log | tail -3 | tee gpurun_out/r6_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r6_final_smoke.log
python bench.py --steps 20 --warmup 3 2> gpurun_out/r6_final_bench.err | tee gpurun_out/r6_final_bench.json | cut -c1-700
