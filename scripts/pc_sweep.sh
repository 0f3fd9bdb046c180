#!/bin/bash
# pc_apply_ms as a function of the Schur Chebyshev degree (latency per sweep step)
for k in 1 2 4 8 16; do
  timeout -k 10 400 python bench.py --schur-its $k --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null > /tmp/o.json
  python3 -c "
import json; d=json.load(open('/tmp/o.json')); print('schur_its', $k, 'pc ms', d['config']['pc_apply_ms'])"
done
