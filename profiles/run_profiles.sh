#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: the default bench line, the rocprofv3 kernel trace of the
# same command, and three separate PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE -- FETCH_SIZE and WRITE_SIZE do not
# fit the TCC slots together; counters are never combined with sys/hip traces).  Outputs under gpurun_out/;
# profiles/make_profiles.py turns them into the committed summaries.
set -u
R=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/bench_$R.json 2> $OUT/bench_$R.err
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --profile-steps 5"
rm -rf $OUT/prof_$R $OUT/pmc_${R}_sq $OUT/pmc_${R}_fetch $OUT/pmc_${R}_write
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$R -o k -- $BENCH > $OUT/prof_$R.json 2> $OUT/prof_$R.err
PB="python $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --profile-steps 5"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_${R}_sq -o p -- $PB > /dev/null 2> $OUT/pmc_${R}_sq.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_${R}_fetch -o p -- $PB > /dev/null 2> $OUT/pmc_${R}_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_${R}_write -o p -- $PB > /dev/null 2> $OUT/pmc_${R}_write.err
# keep the merge-back small: only the databases
find $OUT/prof_$R $OUT/pmc_${R}_* -type f ! -name '*.db' -delete 2>/dev/null
ls -la $OUT/prof_$R $OUT/pmc_${R}_* | head -30
cat $OUT/bench_$R.json
