set -x
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max > gpurun_out/cpumax.txt 2>&1
timeout -k 10 600 python scripts/tune_spmv.py 256 > gpurun_out/tune256.txt 2>&1
rc=$?; cat gpurun_out/tune256.txt | tail -12
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --cells 256 --steps 50 --warmup 5 --eig-n 0 --cpu-seconds 0 > $OUT/prof_stats.log 2>&1
rc=$?; echo "stats rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout -k 10 600 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$tag -- python $GRAFT_REPO_ROOT/bench.py --cells 256 --steps 10 --warmup 2 --eig-n 0 --cpu-seconds 0 > $OUT/pmc_$tag.log 2>&1
  rc=$?; echo "pmc $tag rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
done
ls -R $OUT | head -50
