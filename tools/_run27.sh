cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_d.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_pytest_d.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/refresh_profiles.sh r03 2>&1 | tail -2
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_b32.json'))
print(d['value'], d['ms_per_step'], d['batch1']['p50_ms'], d['batch1']['queries_per_s'], d['pipeline']['queries_per_s'], d['startup']['first_step_s'], d['cpu_baseline']['value'], d['alt_precision']['value'])
print({k:round(v['queries_per_s'],1) for k,v in d['configs'].items()}, d['configs']['config5_oxford_stream_b1']['p50_ms'])
r=d['roofline']; print(r['kernel'], round(r['frac'],3), round(r['avg_launch_ms'],4), r['all_mfma_kernels'], r['hbm_bound']['kernel'], round(r['hbm_bound']['achieved']), r['serial_step_ms'])
PY
