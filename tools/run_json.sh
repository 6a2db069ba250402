#!/bin/bash
# tools/run_json.sh <bench args...>: run bench.py and print the interesting numbers of its line
python3 bench.py --full-line "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
def show(name, r):
    rf = r.get('roofline', {})
    print('%-10s value %s %s  ms/step %s  spread %s  roofline %s %s %s frac %s kernel_ms %s' % (name, r.get('value'), r.get('unit'), r.get('ms_per_step'), (r.get('step_ms_spread') or {}).get('median'), rf.get('bound'), rf.get('achieved'), rf.get('unit'), rf.get('frac'), rf.get('kernel_ms')))
    for k in ('x_realtime_all_receivers', 'waterfall_frames_per_s', 'audio_blocks_per_s'):
        if k in r: print('           %s %s' % (k, r[k]))
if 'workloads' in d:
    for k, v in d['workloads'].items(): show(k, v)
else: show(d.get('config', {}).get('workload', '?')[:10], d)
"
