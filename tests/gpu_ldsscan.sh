#!/bin/bash
# development aid: the overlapped step against the LDS bytes that cap the Newton walk inside a pair
for v in "$@"; do
  GHIP_PAIR_NEWTON_LDS=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null |
    python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k: round(v, 2) for k, v in d['phases_ms_rank0'].items()}, round(d['roofline']['kernel_ms_alone'], 2))
" || exit 1
done
