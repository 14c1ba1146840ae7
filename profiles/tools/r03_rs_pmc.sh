set -e
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_refresh; mkdir -p $OUT; rm -rf $OUT/pmc_rs
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_rs/p$i -o rs -- \
      python3 $GRAFT_REPO_ROOT/profiles/tools/rs_bench.py 20 only > $OUT/pmc_rs_p$i.log 2>&1
done
python3 $GRAFT_REPO_ROOT/profiles/tools/rs_bench.py 20 > $OUT/rs_bench.txt 2>&1; cat $OUT/rs_bench.txt
