# Round-end check on the GPU box: full GPU suite, the bench line, and the rocprofv3 kernel stats of the headline
# command (no secondary workloads, so the dominant kernel's average is the one bench.py reports).
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python profiles/benchline.py gpurun_out/bench_final.json
bash profiles/tools/profile_headline.sh
