mkdir -p gpurun_out
for n in 64 96 128; do
  timeout -k 10 900 python bench.py --cells 128 --steps 50 --cpu-seconds 0 --eig-n $n > gpurun_out/eigbench_$n.json 2> gpurun_out/eigbench_$n.err
  rc=$?; python -c "import json,sys; d=json.load(open('gpurun_out/eigbench_$n.json')); print($n, d['eigen'])"
  if [ $rc -ge 124 ]; then exit $rc; fi
done
