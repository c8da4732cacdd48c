#!/bin/bash
cd /root/repo || exit 1
mkdir -p gpurun_out/r04_prio; rm -f gpurun_out/r04_prio/*
for lab in d8 d10; do
  bd=8; [ $lab = d10 ] && bd=10
  timeout 400 python bench.py --cpu-seconds 4 --bit-depth $bd 2>gpurun_out/r04_prio/err_$lab.txt | grep '^{' > gpurun_out/r04_prio/$lab.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_prio/$lab.json').read()); p=d.get('pipelined') or {}
print('$lab', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('kernel_ms'), p.get('value'), d.get('parity_ok'), d['cpu_baseline']['value'])"
done
for ch in full pixel; do
  timeout 400 python bench.py --cpu-seconds 0 --chain $ch 2>/dev/null | grep '^{' > gpurun_out/r04_prio/$ch.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_prio/$ch.json').read()); p=d.get('pipelined') or {}
print('$ch', d['value'], d['ms_per_step'], p.get('value'), (d.get('rdo_only') or {}).get('value'))"
done
