python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest_gpu_4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_pytest_gpu_4.log
tail -5 gpurun_out/r04_pytest_gpu_4.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_fwd3_c.json 2> gpurun_out/r04_bench_fwd3_c.err
python - <<'PY'
import json
for line in open('gpurun_out/r04_bench_fwd3_c.json'):
    if line.startswith('{'):
        j=json.loads(line); r=j['roofline']; print(j['value'], j['ms_per_step'], 'bwd', r['frac'], 'fwd_in_step', r['fwd_in_step']['frac'], r['fwd_in_step']['measured'][-110:])
PY
