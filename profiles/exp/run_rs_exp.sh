#!/bin/bash
# profiles/exp/run_rs_exp.sh LIB "ENV=V" ...: RS(255,223) BM decode (rs_bench.py 20 only) per kernel under rocprofv3, one run per setting
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
lib=$1; shift
export CHANNELCODING_AMD_LIB=$GRAFT_REPO_ROOT/profiles/exp/lib_$lib.so
i=0
for setting in "$@"; do
  i=$((i+1))
  export $setting
  echo "== $setting"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rsexp/$i -o rs -- python3 profiles/tools/rs_bench.py 20 only > gpurun_out/rsexp_$i.log 2>&1 || { tail -5 gpurun_out/rsexp_$i.log; exit 1; }
  python3 - $i <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/rsexp/%s/**/*kernel_stats.csv' % sys.argv[1],recursive=True)
for r in list(csv.DictReader(open(f[0]))):
    if 'ccamd' in r['Name'] and int(r['Calls']) >= 5: print('   %-60s %s %8.1f us' % (r['Name'].replace('ccamd::(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
