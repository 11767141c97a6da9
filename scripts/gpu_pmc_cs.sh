# wave-state and LDS counters of the compressed-stream kernel (where does a latency-bound kernel wait?)
TAG=${1:-r01e}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"; do
  name=$(echo $set | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $OUT/pmccs_${name} -- python $ROOT/scripts/tune_cs.py 256 > $OUT/pmccs_${name}.log 2>&1
  rc=$?; echo "pmc [$set] rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmccs_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "cs" if "k_spmv_cs" in r["Kernel_Name"] else ("rowgather" if "k_spmv_rowgather" in r["Kernel_Name"] else None)
        if k and int(r["Grid_Size"]) > 5000000:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())})
PY
rm -rf $OUT/pmccs_*/
