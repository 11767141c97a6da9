mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_eigen_gpu.py -m gpu -x -q > gpurun_out/pytest_eig.log 2>&1
rc=$?; tail -30 gpurun_out/pytest_eig.log
