#!/bin/bash
# ARCHIVED (end of round 6): record of a gpurun call (encoder sub-batch stream count 1..4, L/14 batch 16 and B/16 batch 32, one box): profiles/r06_streams_sweep.log
cd /root/repo
for s in 1 2 3 4; do
  python bench.py --arch owlvit-large-patch14 --batch 16 --steps 6 --warmup 2 --windows 3 --no-cpu-baseline --no-compare --encoder-streams $s 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('L/14 streams $s:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', d['config']['window_values'])"
done
for s in 1 2 3 4; do
  python bench.py --no-cpu-baseline --no-compare --steps 20 --warmup 3 --windows 5 --encoder-streams $s 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('B/16 streams $s:', d['value'], 'img/s', d['ms_per_step'], 'ms; windows', d['config']['window_values'])"
done
