#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 32 8 16 3 9 34 35 43 59; do
  E2E_CONV_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/abl_$d -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/conv_prof_one.py layer1fwd > /dev/null 2>&1
  f=$(find /tmp/abl_$d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $d <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_conv_gemm' in r['Name'] and 'false' in r['Name']:
        print(f"dbg={int(sys.argv[2]):3d}  {float(r['AverageNs'])/1e3:7.1f} us  calls {r['Calls']}")
PY
done
