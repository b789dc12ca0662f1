#!/bin/bash
# config A's step at 128 / 256 / 512 / 1024 rows per GPU: the row-blocked 4-launch form against the 7-launch form it replaces
for rows in 128 256 512 1024; do
  for rb in 1 0; do
    [ $rows = 128 ] && [ $rb = 0 ] && continue
    TNN_HEAD_ROW_BLOCKS=$rb python bench.py --rows $rows --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows %4d row_blocks=%s  %.2f us/step  %.0f samples/s  launches %s' % ($rows, '$rb', d['ms_per_step']*1e3, d['value'], d['config'].get('launches_per_step')))"
  done
done
