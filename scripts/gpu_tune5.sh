mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_spmv_gpu.py -m gpu -q > gpurun_out/pytest_spmv.log 2>&1
rc=$?; tail -8 gpurun_out/pytest_spmv.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python scripts/tune_spmv.py 256 0x0,0x20,0x440,0x540,0x480,0x380 > gpurun_out/tune256e.txt 2>&1
cat gpurun_out/tune256e.txt | tail -8
