#!/bin/bash
# rocprofv3 PMC counters of the bench workload, one pass per counter group (gpurun refuses --pmc combined with
# sys/hip traces; kernel-trace is fine; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Usage on the GPU box:  bash profiles/run_pmc.sh <tag> [group ...]      groups: sq sq2 vmem tcp tcc ta fetch write
# Writes gpurun_out/pmc_<tag>.md (per-kernel averages per launch) and gpurun_out/pmc_<tag>.json.
set -u
TAG=${1:-rX}; shift
GROUPS_="${*:-sq sq2 fetch write}"
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
CMD="${PMC_CMD:-python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-iteration-window}"     # PMC_CMD: another workload (e.g. profiles/bench_mlp.py)
declare -A G
G[sq]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
G[sq2]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
G[vmem]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
G[tcp]="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
G[tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
G[tcc2]="TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_REQ_sum"
G[ta]="TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"
G[ic]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
DBS=""
for g in $GROUPS_; do
  timeout ${PMC_TIMEOUT:-120} rocprofv3 --kernel-trace --pmc ${G[$g]} -d $R/gpurun_out/pmc_$TAG -o $g -- $CMD > /dev/null 2> $R/gpurun_out/pmc_$TAG/$g.err
  [ -f $R/gpurun_out/pmc_$TAG/${g}_results.db ] && DBS="$DBS gpurun_out/pmc_$TAG/${g}_results.db" || tail -3 $R/gpurun_out/pmc_$TAG/$g.err
done
cd $R
python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}.md $DBS > /dev/null
rm -rf gpurun_out/pmc_$TAG
