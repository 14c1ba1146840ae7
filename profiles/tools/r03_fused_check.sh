# RS(255,223) decode chain: algebraic tests, rs_bench.py, per-kernel times under rocprofv3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_algebraic.py tests/test_gpu_bitslice.py tests/test_gpu_encode.py -x -q > gpurun_out/fused_tests.log 2>&1 || { tail -20 gpurun_out/fused_tests.log; exit 1; }
tail -1 gpurun_out/fused_tests.log
python profiles/tools/rs_bench.py 20 > gpurun_out/fused_rs_bench.txt 2>&1
cut -c1-80 gpurun_out/fused_rs_bench.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fused_prof -o rs -- python3 profiles/tools/rs_bench.py 20 only > gpurun_out/fused_prof.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fused_prof/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'ccamd' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
