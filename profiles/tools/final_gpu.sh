set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err
python profiles/benchline.py gpurun_out/bench_h.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_h -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_h.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_h | head
