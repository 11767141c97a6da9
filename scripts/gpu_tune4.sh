mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_spmv_gpu.py -m gpu -x -q > gpurun_out/pytest_spmv.log 2>&1
rc=$?; tail -5 gpurun_out/pytest_spmv.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python scripts/tune_spmv.py 256 0x4000,0x8000,0x0,0x20 > gpurun_out/tune256d.txt 2>&1
cat gpurun_out/tune256d.txt | tail -6
