# Multivector block kernels at the pillbox-256^3 vector length: host wall clock per call (scripts/bench_mv.py) and the
# device time of each kernel from rocprofv3 (the roofline figure: the host figure adds launch + stream wait, which
# is a fifth of a 50 us dot).  Usage (GPU box, repo root): bash scripts/gpu_prof_mv.sh <tag>
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
timeout -k 10 300 python scripts/bench_mv.py > $OUT/bench_mv_$TAG.log 2>&1 || exit $?
cp $OUT/bench_mv.json $OUT/bench_mv_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mv_$TAG -- python $ROOT/scripts/bench_mv.py > /dev/null 2>&1
rc=$?; echo "rocprof rc=$rc"
f=$(find $OUT/prof_mv_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/mv_kernel_stats_$TAG.csv && head -20 $f
find $OUT/prof_mv_$TAG -name "*kernel_trace.csv" -delete
