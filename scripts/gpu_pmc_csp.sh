# where does the persistent compressed-stream kernel spend its time?  SQ wave states, LDS, vector-memory path (TA / TCP)
TAG=${1:-r02b}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/pmc_list_$TAG.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LDS_MEM_VIOLATIONS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmccsp_$i -- python $ROOT/scripts/tune_cs.py 256 vecLapl > $OUT/pmccsp_$i.log 2>&1
  rc=$?; echo "pmc [$set] rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmccsp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "csp" if "k_spmv_csp" in n else ("cs" if "k_spmv_cs" in n else ("rgp" if "k_spmv_rgp" in n else None))
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())} for k, d in agg.items()}
json.dump(out, open("$OUT/pmc_csp_$TAG.json", "w"), indent=1)
for k, d in out.items():
    print(k, d)
PY
rm -rf $OUT/pmccsp_*/
