mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_solve_gpu.py tests/test_eigen_gpu.py -m gpu -x -q > gpurun_out/pytest_cg.log 2>&1
rc=$?; tail -6 gpurun_out/pytest_cg.log
if [ $rc -ne 0 ]; then exit $rc; fi
for n in 128 256; do
  timeout -k 10 900 python bench.py --cells 128 --steps 50 --cpu-seconds 0 --eig-n $n > gpurun_out/eigbench_$n.json 2> gpurun_out/eigbench_$n.err
  rc=$?; python -c "import json,sys; d=json.load(open('gpurun_out/eigbench_$n.json')); print($n, d['eigen'])"
  if [ $rc -ge 124 ]; then exit $rc; fi
done
