# A/B of the eigen-solve on ONE box: VAR=value pairs alternate, REPS runs each (default 2), CELLS^3 (default 256).
# usage: [CELLS=384] [REPS=1] bash scripts/ab_eigen.sh VAR A B
VAR=$1; A=$2; B=$3
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 800 python bench.py --cells ${CELLS:-256} --eig-n ${CELLS:-256} --steps 20 --warmup 5 --cpu-seconds 0 --no-compressed > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
    python -c "import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$VAR=$v', d['eigen']['eigensolve_seconds'], d['eigen']['vec_lin_iters'], d['eigen']['sca_lin_iters'])"
  done
done
