#!/bin/bash
# rocprofv3 counter passes over a short command, one pass per counter group:
#   tools/pmc_kernel.sh <outdir-name> "<command>" "<counters of pass 1>" ["<counters of pass 2>" ...]
# (--pmc is only ever combined with --kernel-trace; results land in gpurun_out/<outdir-name>/passN)
set -u
NAME=$1; CMD=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $REPO/gpurun_out/$NAME/pass$i -- $CMD > $REPO/gpurun_out/$NAME.pass$i.log 2>&1
  echo "pass $i ($grp): rc=$?"
done
python3 - "$REPO/gpurun_out/$NAME" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print('   %-32s %16.0f per launch (%d launches)' % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
PY
