#!/bin/bash
# Timing experiments on variants of the headline kernel (profiles/exp/build_variant.sh).  Variants other than `base`
# compute WRONG results by construction; only kernel_ms per frame-iteration is read.  2 dB: every frame runs all 20
# iterations whatever the (wrong) arithmetic does.   usage: run_exp.sh OUTDIR name[:ENV=V] ...
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
  lib=${name%%+*}
  env CHANNELCODING_AMD_LIB=$PWD/profiles/exp/lib_$lib.so $envs python bench.py --ebno 2.0 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; c=d["config"]
    print("%-22s kernel_ms %.3f mean_iters %.2f  ns/frame-iter %.4f verified %s" % (sys.argv[2], r["kernel_ms"], c["mean_iterations_run"], r["kernel_ms"]*1e6/(c["frames_per_gpu"]*c["mean_iterations_run"]), d["verified"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
