# rocprofv3 kernel statistics of the eigen-solve alone (no SpMV timing loop, no CPU baseline), 256^3 unless CELLS is set.
# Usage (on the GPU box, from the repo root): [CELLS=384] bash scripts/gpu_prof_eigen.sh <tag>
TAG=${1:-r02}
N=${CELLS:-256}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MXH_VERBOSE=1 MXG_PAT_STATS=1 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eigen_$TAG -- python $ROOT/bench.py --cells $N --eig-n $N --steps 2 --warmup 1 --cpu-seconds 0 --no-compressed > $OUT/prof_eigen_$TAG.log 2>&1
rc=$?; echo "rc=$rc"; tail -3 $OUT/prof_eigen_$TAG.log
f=$(find $OUT/prof_eigen_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/eigen_kernel_stats_$TAG.csv && head -30 $f
# busy / idle time of the GPU inside the inner solves, then drop the trace (large): keep only the statistics
t=$(find $OUT/prof_eigen_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python $ROOT/scripts/trace_busy.py $t > $OUT/eigen_busy_$TAG.json && cat $OUT/eigen_busy_$TAG.json
find $OUT/prof_eigen_$TAG -name "*kernel_trace.csv" -delete
