#!/bin/bash
# chunk schedule under two lanes (the tail of a launch overlaps the next launch): C3 and C4-shard shapes
set -o pipefail
O=gpurun_out
rm -f $O/r03_lanes_sched.txt
for opts in "tail_split=1" "tail_split=0" "tail_split=0,chunk_len=480" "tail_split=0,chunk_len=544" "tail_split=0,chunk_len=608" "tail_split=1,chunk_len=480" "tail_split=1,chunk_len=544" "tail_split=2" "tail_split=0,chunk_len=352"; do
  for w in c3 c3const; do
    MCD_BENCH_OPTIONS="$opts" timeout -k 10 200 python bench.py --workload $w --no-mcmc --no-cpu-baseline --no-c4-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('$w', '$opts', '%.4e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'one-lane kern %.1f'%(r.get('kernel_us_one_lane') or 0), d['launch']['chunks'])" >> $O/r03_lanes_sched.txt || exit 1
  done
done
cat $O/r03_lanes_sched.txt
