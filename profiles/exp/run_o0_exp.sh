#!/bin/bash
# profiles/exp/run_o0_exp.sh LIB ...: the message-free kernel (stop rule O0) and the 8 dB two-pass route with experiment libraries
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export CHANNELCODING_AMD_LIB=$GRAFT_REPO_ROOT/profiles/exp/lib_$lib.so
  for args in "--stop-rule 0" "--ebno 8.0"; do
    python bench.py $args --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > gpurun_out/o0_$lib.json 2> gpurun_out/o0_$lib.err
    python - gpurun_out/o0_$lib.json "$lib $args" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("%-28s %8.1f M frames/s  kernel_ms %.3f  verified %s" % (sys.argv[2], d["value"]/1e6 if d["value"] else -1, r["kernel_ms"], d["verified"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
