# rocprofv3 kernel trace + stats of the headline bench command (no secondary workloads, so that the average
# duration of the dominant kernel is the one bench.py reports); run on the GPU box through gpurun.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_headline
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- \
    python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT.log 2>&1
tail -1 $OUT.log > $OUT.json || true
ls $OUT
