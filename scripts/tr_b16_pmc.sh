#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of scripts/tr_b16_timing.hip's eight (instruction, address pattern) kernels: does the
# row layout of the limb images change the bank conflicts of ds_read_b64_tr_b16?  -> gpurun_out/tr_b16_pmc.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/tr_b16_timing.hip -o /tmp/trt || exit 1
cd /tmp
rm -rf /tmp/trt_pmc
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/trt_pmc -o pmc -- /tmp/trt > /tmp/trt.log 2>&1
python3 - <<'P' > $R/gpurun_out/tr_b16_pmc.txt
import csv, glob, collections
rows = []
for p in glob.glob('/tmp/trt_pmc/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
by = collections.defaultdict(dict)
for r in rows:
    by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    by[int(r['Dispatch_Id'])]['k'] = r['Kernel_Name'][:40]
ids = [d for d in sorted(by) if by[d]['k'].startswith('void k<')]
print("dispatch order: instruction (ds_read_b64, ds_read_b64_tr_b16) x pattern (0 broadcast, 1 tr-style 72-byte rows, 2 tr-style skewed 64-byte rows, 3 dense, 4 row reads 72-byte rows, 5 row reads skewed rows, 6 row reads 80-byte rows, 7 row reads 64 B + 16 B skew) x 5 repeats; 256 reads per dispatch")
for i, d in enumerate(ids):
    v = by[d]
    if i % 5 == 4:
        print(f"{'tr_b16' if i >= 40 else 'b64   '} pattern {(i // 5) % 8}: SQ_INSTS_LDS {v.get('SQ_INSTS_LDS')}, SQ_LDS_IDX_ACTIVE {v.get('SQ_LDS_IDX_ACTIVE')}, SQ_LDS_BANK_CONFLICT {v.get('SQ_LDS_BANK_CONFLICT')}   [{v['k']}]")
P
cat $R/gpurun_out/tr_b16_pmc.txt; tail -3 /tmp/trt.log
