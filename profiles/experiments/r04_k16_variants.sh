#!/bin/bash
# round 4: conv1-forward variants (conv_fwd_k16.hip rebuilt with extra -D flags; build_dw16_variants.sh recipe), timed with bench.py --quick
set -e
cd "$(dirname "$0")/../.."
UNITS="conv_fwd_k16" bash profiles/experiments/build_dw16_variants.sh "$@"
