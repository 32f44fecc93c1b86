#!/bin/bash
# conv2 forward / dW on the bf16 pipes with six of the nine piece products (B16_MAX_ORDER=2: h*h, h*m, m*h, h*l, m*m, l*h; the dropped
# m*l, l*m, l*l are <= 2^-24 of the product with round-to-nearest pieces) against all nine and against the f32-input MFMA kernels:
# error against the float64 oracle (pooled conv2 output, conv2 weight gradient) and steps/s on the same box.
# build first: UNITS="conv conv2_bwd_pair conv_dwb16 conv_fwd_k16" bash profiles/experiments/build_dw16_variants.sh six "-DB16_MAX_ORDER=2"
cd "$(dirname "$0")/../.."
python - <<'PY' > /tmp/c2snip.py
import re
src = open("tests/test_gpu_fullsize.py").read()
print(re.search(r'_CONV2_ERR_SNIPPET = r"""(.*?)"""', src, re.S).group(1))
PY
for v in "nine 1 1" "f32 1 0" "six six 1"; do
  set -- $v
  echo "== $1"; PYTHONPATH=. CARTPOLEPP_ABLATION=$2 CPP_CONV_B16=$3 python /tmp/c2snip.py 2>&1 | grep -E "C2FWD|C2DW|Error|error" 
done
for i in 1 2; do
  for v in 1 six; do
    echo "== bench ABLATION=$v"; CARTPOLEPP_ABLATION=$v python bench.py --quick --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], [(l['layer'], l['avg_launch_us']) for l in d['layers']])"
  done
done
