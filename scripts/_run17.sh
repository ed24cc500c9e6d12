cd /root/repo
timeout 900 python bench.py --ops-json gpurun_out/ops_r03g.json > gpurun_out/bench_r03g.json 2> gpurun_out/bench_r03g.err; tail -c 300 gpurun_out/bench_r03g.err
python - <<PY
import json
b=json.load(open('gpurun_out/bench_r03g.json'))
print(b['value'], b['ms_per_nfe_batch'], b['roofline']['frac'], b['roofline']['traffic'])
h=b['roofline_hbm']; print(h['kernel'], h['achieved'], h['frac'], h['traffic'], h['family_ms_per_nfe'])
print(list(h['by_kernel']))
PY
timeout 600 python -m pytest tests/test_model.py -x -q -m gpu 2>&1 | tail -2
