#!/bin/bash
# Dev: one bench run, compact print.  usage: [ENV=..] tools/mppi_one.sh
timeout 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(' small: %.1f Mroll/s tick %.1f us kernels %s' % (d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']))
l=d['roofline_large']; print(' large: %.1f Mroll/s tick %.1f us frac(tick) %.3f kernels %s' % (l['rollouts_per_s']/1e6, l['whole_tick']['ms']*1e3, l['whole_tick']['frac'], l['kernel_ms']))"
