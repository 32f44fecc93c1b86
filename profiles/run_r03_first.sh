#!/bin/bash
# round 3, first GPU call: the new literal-loop / replay tests first, then the whole gpu suite, the host-path bench, and the N > 1 launchers
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_literal_loop.py tests/test_gpu_replay.py -x -q -m gpu > $OUT/r03_t1.log 2>&1; echo "t1 rc=$?" | tee -a $OUT/r03_t1.log
tail -25 $OUT/r03_t1.log
timeout 1500 python -m pytest tests -q -m gpu > $OUT/r03_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/r03_tests.log
tail -15 $OUT/r03_tests.log
bash profiles/run_host_path.sh
# plain-command N = 2 on this 1-GPU box: gloo diagnostic backend (both ranks on GPU 0, torch all_reduce of the library's gradient buffer)
timeout 600 python bench.py --gpus 2 --diag-backend gloo --quick --steps 20 --warmup 5 > $OUT/r03_gpus2_gloo.json 2> $OUT/r03_gpus2_gloo.err; echo "gloo2 rc=$?"
head -c 1500 $OUT/r03_gpus2_gloo.json; echo
# can RCCL form a 2-rank communicator with both ranks on one GPU?  (expected: no -- "Duplicate GPU detected")
timeout 300 python bench.py --gpus 2 --quick --steps 20 --warmup 5 > $OUT/r03_gpus2_rccl.json 2> $OUT/r03_gpus2_rccl.err; echo "rccl2 rc=$?"
tail -5 $OUT/r03_gpus2_rccl.err; head -c 600 $OUT/r03_gpus2_rccl.json; echo
python bench.py --quick --steps 100 > $OUT/r03_quick.json 2> $OUT/r03_quick.err; head -c 800 $OUT/r03_quick.json
