#!/bin/bash
# Round 6: the footprint knee of the lone C4 launch in counters.  For K = 1, 6, 8, 12 rotating sets, rocprofv3 --pmc passes
# (each with --kernel-trace only, one block's counters per pass) over tools/exp/r06_pmc_probe.py; tools/exp/r06_pmc_footprint_summary.py makes the table.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r06_pmc_footprint; mkdir -p $O
PASSES=(
"TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
"TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
"TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
"TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"
"TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum"
"GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_TC_BUSY GRBM_GUI_ACTIVE"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum"
)
for K in ${KS:-1 6 8 12}; do
  p=0
  for C in "${PASSES[@]}"; do
    D=$O/k${K}_p${p}
    (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/exp/r06_pmc_probe.py $K 48 $PROBE_OPTS > $D.log 2>&1; echo "K=$K pass=$p rc=$?" >> $O/rc.txt)
    p=$((p+1))
  done
done
python tools/exp/r06_pmc_footprint_summary.py $O > $O/summary.json 2> $O/summary.err
# keep what is small: the per-pass counter files are a few hundred KB each
find $O -name "*agent_info.csv" -delete
cat $O/rc.txt | tail -5; head -c 3000 $O/summary.json
