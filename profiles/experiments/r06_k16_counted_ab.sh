# conv_k16.h's operand loads as builtins the compiler counts (-DK16_COUNTED_A, lib/libcartpolepp_hip_counted.so) against the inline-asm
# loads with hand-counted vmcnt waits: conv2 forward at cfg3 (B16 mode), conv1 + conv2 forward at cfg5 (30 channels / 64-wide rows), the
# 50x50 render (conv1 on the ring kernel)
for w in cfg3 cfg5 r50; do
  for i in 1 2; do
    for v in "" counted; do
      CARTPOLEPP_ABLATION=$v python bench.py --quick --workload $w --steps 60 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w variant=%-8s' % '$v', d['value'], [(l['layer'], l['avg_launch_us']) for l in d['layers'][:3]])"
    done
  done
done
