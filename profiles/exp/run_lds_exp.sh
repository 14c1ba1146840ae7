#!/bin/bash
# LDS-boundness experiments (results are WRONG by construction; only kernel_ms per frame-iteration is read)
out=gpurun_out/r02b; mkdir -p $out
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --ebno 2.0 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; c=d["config"]
print("%-10s kernel_ms %.3f mean_iters %.2f  ns/frame-iter %.4f verified %s" % (sys.argv[2], r["kernel_ms"], c["mean_iterations_run"], r["kernel_ms"]*1e6/(c["frames_per_gpu"]*c["mean_iterations_run"]), d["verified"]))
PY
}
for lib in cn0_cy0 cn2_cy0 cn0_cy2 cn2_cy2 cn4_cy4; do
  run $lib CHANNELCODING_AMD_LIB=$PWD/profiles/exp/lib_$lib.so
done
run fake_deal CC_AMD_EXP_FAKE_DEAL=1 CHANNELCODING_AMD_LIB=$PWD/profiles/exp/lib_cn0_cy0.so
