#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: the default bench line, the rocprofv3 kernel trace of the
# same command, and three separate PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE -- FETCH_SIZE and WRITE_SIZE do not
# fit the TCC slots together; counters are never combined with sys/hip traces), then the bench lines + kernel traces of the
# other BASELINE configs (cfg2, cfg4 = NAF, cfg5 with 6000 rows and with one GPU's 125 000-row u8 shard, r50, batch norm).
# Outputs under gpurun_out/; profiles/make_profiles.py turns them into the committed summaries.
set -u
R=${1:-r06}
OUT=$PWD/gpurun_out
mkdir -p $OUT
REPO=$PWD
# ONLY=<config name> (environment): re-run one of the run_cfg configurations below and nothing else
if [ -z "${ONLY:-}" ]; then
python bench.py > $OUT/bench_$R.json 2> $OUT/bench_$R.err
fi
cd /tmp && export TMPDIR=/tmp
if [ -z "${ONLY:-}" ]; then
BENCH="python $REPO/bench.py --quick --steps 50 --warmup 10 --profile-steps 5"
rm -rf $OUT/prof_$R $OUT/pmc_${R}_sq $OUT/pmc_${R}_fetch $OUT/pmc_${R}_write
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$R -o k -- $BENCH > $OUT/prof_$R.json 2> $OUT/prof_$R.err
PB="python $REPO/bench.py --quick --steps 10 --warmup 5 --profile-steps 5"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_${R}_sq -o p -- $PB > /dev/null 2> $OUT/pmc_${R}_sq.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_${R}_fetch -o p -- $PB > /dev/null 2> $OUT/pmc_${R}_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_${R}_write -o p -- $PB > /dev/null 2> $OUT/pmc_${R}_write.err
python $REPO/profiles/shrink_pmc.py $OUT/pmc_${R}_sq $OUT/pmc_${R}_fetch $OUT/pmc_${R}_write      # (45 MB databases -> agg.json)
fi
# the other configurations: one bench line and one kernel trace each
run_cfg() {   # name, bench flags
  local name=$1; shift
  if [ -n "${ONLY:-}" ] && [ "$ONLY" != "$name" ]; then return; fi
  python $REPO/bench.py --quick "$@" > $OUT/bench_${R}_$name.json 2> $OUT/bench_${R}_$name.err
  rm -rf $OUT/prof_${R}_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${R}_$name -o k -- python $REPO/bench.py --quick --profile-steps 5 "$@" > /dev/null 2> $OUT/prof_${R}_$name.err
}
run_cfg cfg2 --workload cfg2 --steps 100 --warmup 10
run_cfg cfg4 --workload cfg4 --steps 100 --warmup 10
run_cfg cfg5 --workload cfg5 --steps 30 --warmup 10
run_cfg cfg5_shard --workload cfg5 --steps 30 --warmup 10 --replay-rows 125000 --replay-store u8
# one GPU's share of configs[4]'s 10^6-row replay as the reference's own f16 store: 187 500 states x 983 040 B = 184 GB of the 288 GB
run_cfg cfg5_shard_f16 --workload cfg5 --steps 30 --warmup 10 --replay-rows 125000 --replay-store f16
run_cfg r50 --workload r50 --steps 100 --warmup 10
run_cfg cfg3_bn --workload cfg3 --steps 50 --warmup 10 --use-batch-norm
run_cfg cfg3_dp1 --workload cfg3 --steps 100 --warmup 10 --force-dp
if [ -z "${ONLY:-}" ]; then
# the reference's literal loop against the fused step (profiles/bench_host_path.py), and N = 2 as a plain command (gloo diagnostic)
(cd $REPO && bash profiles/run_host_path.sh > /dev/null 2>&1)
(cd $REPO && timeout 600 python bench.py --gpus 2 --diag-backend gloo --quick --steps 20 --warmup 5 > $OUT/bench_${R}_gpus2_gloo_diag.json 2> $OUT/bench_${R}_gpus2_gloo_diag.err)
fi
# keep the merge-back small: only the databases
find $OUT/prof_${R}* $OUT/pmc_${R}_* -type f ! -name '*.db' ! -name 'agg.json' -delete 2>/dev/null
ls $OUT | grep $R | head -40
cat $OUT/bench_$R.json | head -c 3000
