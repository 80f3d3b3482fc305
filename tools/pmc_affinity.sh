#!/bin/bash
# PMC counters of the affinity kernels (separate passes, --kernel-trace only): LDS pipe, bank conflicts, occupancy, traffic
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmca
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmca -o p -- python3 /root/repo/tools/probe_affinity.py 2 > /tmp/pmca.log 2>&1 || { echo "pass failed: $set"; tail -3 /tmp/pmca.log; continue; }
  python3 /root/repo/tools/pmc_kernel.py /tmp/pmca k_weights_lanes k_neighbours
done
