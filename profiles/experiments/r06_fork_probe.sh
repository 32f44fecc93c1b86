# does a throughput-bound launch on a second stream hide under the step's latency-bound launches?  (ablation build, CPP_EXP_FORK=n: n
# duplicate conv1 forwards of the two target networks -- ~42 us each alone -- forked off behind the trunks' forward pass, joined in
# front of the conv backward; rt_ddpg.cpp)
for i in 1 2; do
  for f in 0 1 2; do
    CARTPOLEPP_ABLATION=1 CPP_EXP_FORK=$f python bench.py --quick --steps 200 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fork=$f', d['value'], 'steps/s', d['ms_per_step'] * 1000, 'us/step')"
  done
done
