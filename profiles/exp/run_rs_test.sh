#!/bin/bash
# profiles/exp/run_rs_test.sh LIB: algebraic GPU tests + rs_bench + per-kernel times with an experiment library
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export CHANNELCODING_AMD_LIB=$GRAFT_REPO_ROOT/profiles/exp/lib_$1.so
timeout -k 10 600 python -m pytest tests/test_gpu_algebraic.py tests/test_gpu_bitslice.py tests/test_gpu_encode.py -x -q > gpurun_out/rsx_tests.log 2>&1 || { tail -20 gpurun_out/rsx_tests.log; exit 1; }
tail -1 gpurun_out/rsx_tests.log
python profiles/tools/rs_bench.py 20 > gpurun_out/rsx_rs_bench.txt 2>&1
cut -c1-80 gpurun_out/rsx_rs_bench.txt
bash profiles/exp/run_rs_exp.sh $1 CC_X=0
