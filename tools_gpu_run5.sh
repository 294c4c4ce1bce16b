#!/bin/bash
mkdir -p gpurun_out/pmc
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|LDSBankConflict|OccupancyPercent|MemUnitStalled)\b" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters.txt
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/tools/conv_bench.py "G.b5.conv2" "D.b3.conv2" > $R/gpurun_out/pmc/$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $R/gpurun_out/pmc/$n.txt <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
for k,v in agg.items():
    if 'gemm' in k: print(k, dict(v))
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
run b SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS
run c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES
cd $R; cat gpurun_out/pmc/counters.txt | head -c 3000; echo; cat gpurun_out/pmc/a.txt gpurun_out/pmc/b.txt gpurun_out/pmc/c.txt; tail -3 gpurun_out/pmc/a.log
