#!/bin/bash
# PMC passes over a probe binary: usage pmc.sh <binary> ; prints per-kernel counter means
set -u
BIN=$PWD/$1
OUT=$PWD/gpurun_out/rs16pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { local n=$1; shift; rm -rf $OUT/$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o p -- $BIN 5 > /dev/null 2> $OUT/$n.err; }
pass insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pass stalls SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_THREAD_CYCLES_VALU
pass base SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_LDS_BANK_CONFLICT
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
python3 - <<PY
import sqlite3, glob, json
res = {}
for db in sorted(glob.glob("$OUT/*/p_results.db")):
    con = sqlite3.connect(db)
    try:
        rows = list(con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for name, ctr, n, mean in rows:
        if "rs16" in name or "k16_kernel" in name:
            res.setdefault(name[:40], {})[ctr] = round(mean, 1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/insts $OUT/active $OUT/stalls $OUT/base $OUT/lds
