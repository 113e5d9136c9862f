#!/bin/bash
# PMC collection, one counter group per run (rocprofv3 --pmc, no trace domains besides kernel-trace).
# usage: tests/tools/pmc.sh <outdir> -- <program and args>
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
while read -r grp; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o pmc -- "$@" > $out.g$i.log 2>&1
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS
FETCH_SIZE
WRITE_SIZE
TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_HIT TCC_MISS
TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_STALL TCC_REQ
GROUPS
ls $out
