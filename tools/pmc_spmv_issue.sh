#!/bin/bash
# Instruction mix and issue utilisation of fk_spmv on full-size launches (tools/solo_fullsize_steps.py 40); run from /tmp on the GPU box:
#   bash $GRAFT_REPO_ROOT/tools/pmc_spmv_issue.sh   -> gpurun_out/r3/pmc_issue_*.csv (per-kernel averages of each counter pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; O=$R/gpurun_out/r3; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmci$i -o p -- python3 $R/tools/solo_fullsize_steps.py 40 > $O/pmci$i.log 2>&1 || { tail -3 $O/pmci$i.log; continue; }
  python3 - "$O/pmci$i" "$set" <<'PY'
import csv, glob, sys, collections
d, names = sys.argv[1], sys.argv[2].split()
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
vals = collections.defaultdict(list)
for fn in f:
    for row in csv.DictReader(open(fn)):
        if "fk_spmv" in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for n in names:   # full-size launches only: those within 30 % of the largest value of the counter
    v = vals.get(n, [])
    big = [x for x in v if x >= 0.7 * max(v)] if v else []
    out[n] = (round(sum(big) / max(len(big), 1), 1), len(big), len(v))
print(out)
PY
  rm -rf $O/pmci$i
done
