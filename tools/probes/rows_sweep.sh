#!/bin/bash
# config A's step at 128 / 256 / 512 / 1024 rows per GPU: the 4-launch row-block form (forward in its row-panel form / with the
# arrival-counter tail the data-parallel step uses) against the 7-launch form it replaces
run() { python bench.py --rows $1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows %4d %-28s %.2f us/step  %.0f samples/s' % ($1, '$2', d['ms_per_step']*1e3, d['value']))"; }
export TNN_HEAD_ROW_BLOCKS_MAX=${TNN_HEAD_ROW_BLOCKS_MAX:-1024}
run 128 "4 launches"
for rows in 256 512 1024; do
  TNN_HEAD_ROW_BLOCKS=1 TNN_HEAD_ROW_PANELS=1 run $rows "row blocks, row-panel forward"
  TNN_HEAD_ROW_BLOCKS=1 TNN_HEAD_ROW_PANELS=0 run $rows "row blocks, counter tail"
  TNN_HEAD_ROW_BLOCKS=0 run $rows "7 launches"
done
