#!/bin/bash
# round 6: memory-pipeline counters of single conv kernels (tools/r6_one_conv.py), separate --pmc passes, the program directly after "--"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/pmc
RAW=/tmp/r6pmc
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  i=0
  for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_READ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $RAW/${what}_$i -- python3 $GRAFT_REPO_ROOT/tools/r6_one_conv.py $what 12 > $OUT/${what}_$i.log 2>&1 || { tail -3 $OUT/${what}_$i.log; }
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY' > $OUT/summary.txt
import csv, glob, collections
for d in sorted(glob.glob('/tmp/r6pmc/*')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f: print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'conv' not in k: continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, cs in acc.items():
        print(d.split('/')[-1], k[:90])
        for c, (n, v) in cs.items(): print('     %-44s %16.0f per launch (%d launches)' % (c, v / n, n))
PY
cat $OUT/summary.txt
