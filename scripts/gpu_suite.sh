# Full GPU pass: parity tests, smoke, headline bench, rocprof kernel stats + HBM counters.
# Usage (from the repo root on the GPU box): bash scripts/gpu_suite.sh <tag>
TAG=${1:-r01}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then   # SKIP_TESTS=1: the parity tests were run in a call of their own
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_gpu_$TAG.log
if [ $rc -ge 124 ]; then exit $rc; fi
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
rc=$?; tail -2 gpurun_out/smoke_$TAG.log
if [ $rc -ge 124 ]; then exit $rc; fi
MXH_VERBOSE=1 timeout -k 10 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
if [ $rc -ge 124 ]; then exit $rc; fi
ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$TAG -- python $ROOT/bench.py --steps 50 --warmup 5 --eig-n 0 --cpu-seconds 0 --no-compressed > $OUT/prof_stats_$TAG.log 2>&1
rc=$?; echo "stats rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${ctr}_$TAG -- python $ROOT/bench.py --steps 10 --warmup 2 --eig-n 0 --cpu-seconds 0 --spmm-k 0 > $OUT/pmc_${ctr}_$TAG.log 2>&1
  rc=$?; echo "pmc $ctr rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
done
