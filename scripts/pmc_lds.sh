#!/bin/bash
# one PMC pass (LDS counters) of bench.py --workload $1 -> prints per-kernel averages of SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for w in "$@"; do
  rm -rf /tmp/pmc_lds_$w
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_lds_$w -o pmc -- python $R/bench.py --workload $w --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /tmp/pmc_lds_$w.log 2>&1
  python3 - $w <<'P'
import csv, glob, collections, sys
w = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(f'/tmp/pmc_lds_{w}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'lp::' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(w, k, {c: round(sum(x) / len(x)) for c, x in v.items()})
P
done
