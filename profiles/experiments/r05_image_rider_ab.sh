# A/B of conv1's operand images riding in the optimiser's launch (default) against the launch of their own (CPP_RIDE_IMAGE=0, ablation build)
cd /root/repo
cartpoleplusplus_amd/lib/conv1_rs16_probe 20 2>&1 | grep -E "f32 pool1|image|rs16_kernel"
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fused_fullsize.py -x -q 2>&1 | tail -4
for i in 1 2; do
CARTPOLEPP_ABLATION=1 CPP_RIDE_IMAGE=0 python bench.py --quick --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no-ride', d['value'], d['kernels'].get('clip_sgd'), d['kernels'].get('conv1_image'), d['kernels'].get('dw_reduce'))"
CARTPOLEPP_ABLATION=1 python bench.py --quick --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ride   ', d['value'], d['kernels'].get('clip_sgd'), d['kernels'].get('conv1_image'), d['kernels'].get('dw_reduce'))"
done
