# Round-3 profile refresh on the GPU box (through gpurun): everything DESIGN.md / bench.py cite from profiles/ is
# regenerated here on the shipped library and copied under profiles/ with an r03_ prefix by the caller.
#   bash profiles/tools/r03_refresh.sh        (writes gpurun_out/r03_refresh/)
set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_refresh; rm -rf $OUT; mkdir -p $OUT
echo "== bench"; timeout -k 10 600 python3 bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err; python3 profiles/benchline.py $OUT/final_bench.json
echo "== side benches"
python3 profiles/tools/hard_bench.py > $OUT/hard_bench.txt 2>&1
python3 profiles/tools/codes_bench.py > $OUT/codes_bench.txt 2>&1
python3 profiles/tools/variants_bench.py > $OUT/variants_bench.txt 2>&1
python3 profiles/tools/mc_bench.py > $OUT/mc_bench.txt 2>&1
python3 profiles/tools/host_path_bench.py > $OUT/host_path.txt 2>&1
python3 profiles/tools/rs_bench.py 20 > $OUT/rs_bench.txt 2>&1
python3 profiles/tools/rs_bench.py 20 erasures > $OUT/rs_erasures.txt 2>&1
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 stats of the headline command"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_headline -o bench -- \
    python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/headline_under_rocprof.json 2> $OUT/prof_headline.err
echo "== PMC headline"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_hl/p$i -o hl -- \
      python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $OUT/pmc_hl_p$i.log 2>&1
done
echo "== PMC O0"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_o0/p$i -o o0 -- \
      python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --stop-rule 0 > $OUT/pmc_o0_p$i.log 2>&1
done
echo "== PMC RS(255,223) decode, 2^20 frames"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_rs/p$i -o rs -- \
      python3 $GRAFT_REPO_ROOT/profiles/tools/rs_bench.py 20 only > $OUT/pmc_rs_p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_summary.py --frame-iters 18246271 $(find $OUT/pmc_hl -name "*counter_collection.csv") > $OUT/pmc_minsum_diag.txt 2>&1 || true
python3 profiles/pmc_summary.py --frame-iters 1048576 $(find $OUT/pmc_o0 -name "*counter_collection.csv") > $OUT/pmc_minsum_single_o0.txt 2>&1 || true
python3 profiles/pmc_summary.py --frame-iters 1048576 $(find $OUT/pmc_rs -name "*counter_collection.csv") > $OUT/pmc_rs_decode.txt 2>&1 || true
python3 profiles/trim_stats.py $(find $OUT/prof_headline -name "*kernel_stats.csv" | head -1) > $OUT/final_headline_kernel_stats.csv 2>/dev/null || true
find $OUT -name "*.csv" -size +2M -delete
ls $OUT
