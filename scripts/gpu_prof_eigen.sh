# rocprofv3 kernel breakdown of the 256^3 eigen-solve (multigrid-CG inner solves)
TAG=${1:-r01c}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eigen_$TAG -- python $ROOT/bench.py --steps 5 --warmup 2 --eig-n ${EIGN:-256} --cpu-seconds 0 > $OUT/prof_eigen_$TAG.log 2>&1
rc=$?; tail -3 $OUT/prof_eigen_$TAG.log | cut -c1-600
f=$(find $OUT/prof_eigen_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/eigen_kernel_stats_$TAG.csv && cut -d, -f2-5 $f | head -25
rm -rf $OUT/prof_eigen_$TAG      # the per-dispatch trace is hundreds of MiB; only the summary is kept
exit $rc
