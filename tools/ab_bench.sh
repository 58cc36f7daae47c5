#!/bin/bash
# A/B the default bench line against env-var variants, alternating, N rounds:  tools/ab_bench.sh "VAR1=1 VAR2=1" [rounds]
VARS="$1"; N="${2:-3}"
for i in $(seq $N); do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A default', d['value'], d['ms_per_step'], d['end_to_end']['one_batch_in_flight']['ms_per_step'])"
  env $VARS python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B $VARS', d['value'], d['ms_per_step'], d['end_to_end']['one_batch_in_flight']['ms_per_step'])"
done
