set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
python3 -c "
import json;d=json.load(open('gpurun_out/bench_q.json'));print('VALUE %.4g  ms/step %.3f  kernel_ms %.3f  valu_frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline_valu']['frac']))"
