#!/bin/bash
# conv1 forward / dW on the f16 pipes with TWO f16 pieces of the f32 operand (weights resp. dY; 22 significand bits, the raw pixel
# operand stays exact) against three (exact) and against the f32-input MFMA kernels: error against the float64 oracle and steps/s.
# build first: UNITS="conv_fwd_k16 conv_fwd_k16_pair conv_dw16 conv_dw16_pair conv1_dw_gather" bash profiles/experiments/build_dw16_variants.sh two "-DF16_PIECES=2"
cd "$(dirname "$0")/../.."
python - <<'PY'
import re
src = open("tests/test_gpu_fullsize.py").read()
for n in ("_CONV1_ERR_SNIPPET", "_CONV1_DW_ERR_SNIPPET", "_CONV2_ERR_SNIPPET"):
    open("/tmp/%s.py" % n, "w").write(re.search(n + r' = r"""(.*?)"""', src, re.S).group(1))
PY
for v in "three 1 1" "f32 1 0" "two two 1"; do
  set -- $v
  echo "== $1"
  for n in _CONV1_ERR_SNIPPET _CONV1_DW_ERR_SNIPPET; do
    PYTHONPATH=. CARTPOLEPP_ABLATION=$2 CPP_CONV_K16=$3 python /tmp/$n.py 2>&1 | grep -E "CONV1ERR|C1DW|weights|Error|error" | head -8
  done
done
for i in 1 2; do
  for v in 1 two; do
    echo "== bench ABLATION=$v"; CARTPOLEPP_ABLATION=$v python bench.py --quick --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], [(l['layer'], l['avg_launch_us']) for l in d['layers']])"
  done
done
echo "== random geometries, two pieces"; CARTPOLEPP_ABLATION=two python profiles/diag/random_geometry_parity.py 7 30 2>&1 | grep -E "FAIL|GEODONE"
