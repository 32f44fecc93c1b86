#!/bin/bash
# round 4: instruction mix and per-unit busy cycles of conv1 forward (conv_fwd_k16_kernel<18,5,2,2>) and conv1 dW -- three counter
# passes over the quick bench (counters only: never combined with a sys / hip trace).  Runs on the GPU box from the repo root.
set -u
OUT=$PWD/gpurun_out/k16pmc
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PB="python $REPO/bench.py --quick --steps 10 --warmup 5 --profile-steps 5"
pass() {  # name, counters...
  local n=$1; shift
  rm -rf $OUT/$n
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o p -- $PB > /dev/null 2> $OUT/$n.err
}
pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pass stalls SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_THREAD_CYCLES_VALU
pass base SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16
# (the databases are ~45 MB per pass: parsed below, then removed -- gpurun merges back at most 64 MiB)
python - <<PY
import sqlite3, glob, json
res = {}
for db in sorted(glob.glob("$OUT/*/p_results.db")):
    con = sqlite3.connect(db)
    for name, ctr, n, mean in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if name.startswith("void conv_fwd_k16_kernel<18") or "conv1_dw_pair_gather" in name or name.startswith("void conv_fwd_k16_kernel<10") or "conv2_bwd_pair" in name:
            res.setdefault(name[:48], {})[ctr] = round(mean, 1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/insts $OUT/active $OUT/stalls $OUT/base
