mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_mv_gpu.py -m gpu -x -q > gpurun_out/pytest_mv.log 2>&1
rc=$?; tail -4 gpurun_out/pytest_mv.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python scripts/bench_mv.py > gpurun_out/bench_mv.txt 2>&1
cat gpurun_out/bench_mv.txt | tail -14
