timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/pytest_r2f.txt; cat gpurun_out/pytest_r2f.txt
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r2f.json 2> gpurun_out/bench_r2f.err; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r2f.json') if l.startswith('{')][-1])
print(d['value'], d['verified'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
s=d['sprot_like']; print(s['value'], s['verified'], s['ms_per_step'], s['cpu_baseline']['value'])
PY
tail -3 gpurun_out/bench_r2f.err
