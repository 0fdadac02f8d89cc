#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: full GPU test suite, smoke, bench line, rocprof kernel
# stats of the bench, PMC passes of the roofline kernel.  Outputs under gpurun_out/evidence/.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/evidence
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?
echo "pytest exit $rc" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-op-table > $OUT/bench_prof.log 2>&1; echo "rocprof bench exit $?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -- python3 $GRAFT_REPO_ROOT/scripts/roofline_kernel.py > $OUT/roofline_trace.log 2>&1; echo "rocprof roofline exit $?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/roofline_kernel.py > $OUT/pmc_fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/roofline_kernel.py > $OUT/pmc_write.log 2>&1; echo "pmc exit $?"
cd $GRAFT_REPO_ROOT
python scripts/parse_pmc.py $OUT/pmc_fetch qbp_cell; python scripts/parse_pmc.py $OUT/pmc_write qbp_cell
python scripts/make_traffic_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/roofline_traffic.json $OUT/roofline
python scripts/parse_trace.py $OUT/prof_roofline qbp_cell > $OUT/roofline_kernel_durations.txt; cat $OUT/roofline_kernel_durations.txt
