# device time of the pattern product's two kernels (pattern rows / individual rows), rocprofv3 kernel statistics
# usage: MXG_PAT=2 bash scripts/prof_pat.sh <tag>
TAG=${1:-pat}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $ROOT/scripts/tune_pat.py worker 256 vecLapl > $OUT/prof_$TAG.log 2>&1
echo "rc=$?"; grep RESULT $OUT/prof_$TAG.log
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-200
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete
