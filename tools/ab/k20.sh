p() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'value %.3f M'%(d['value']/1e6), 'ms_per_step %.4f'%d['ms_per_step'], 'kernel_ms_mean %.4f'%r['kernel_ms_mean'], 'settle', d['config'].get('settle_steps'))"; }
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | p direct_k20_w5
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-settle 2>/dev/null | p direct_k20_w5_nosettle
python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | p direct_k50_w5
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | p torchrun_k20_w5
python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | p direct_k20_w5_with_cpu_baseline
