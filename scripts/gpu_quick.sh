mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_quick.log 2>&1
rc=$?; tail -15 gpurun_out/pytest_gpu_quick.log
