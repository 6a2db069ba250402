#!/bin/bash
# GPU box: copy dispatches PER STEP of the receivers workload = the difference of two kernel traces that differ only in --steps
# (a trace's total includes the one-time configuration uploads of two banks of 128 receivers: thousands of small hipMemcpy).
export TMPDIR=/tmp
out=gpurun_out/prof_copies; mkdir -p $out
for st in 40 240; do
  KIWIGPU_BENCH_PREROLL_S=0.12 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$st -- python3 bench.py --full-line --no-cpu --no-live-traffic --workload receivers --steps $st --warmup 4 > $out/t$st.log 2>&1
done
python3 - <<'PY'
import csv,glob
def counts(d):
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)[0]
    return {r['Name']:int(r['Calls']) for r in csv.DictReader(open(f))}
a,b=counts('gpurun_out/prof_copies/t40'),counts('gpurun_out/prof_copies/t240')
fr=[k for k in a if k.startswith('void wf_frame_kernel')][0]
dsteps=b[fr]-a[fr]
print("receivers (SURVEY's mix, 128 receivers), two traces that differ only in --steps: %d more steps (wf_frame_kernel launches %d -> %d)" % (dsteps,a[fr],b[fr]))
for k in sorted(set(a)|set(b), key=lambda k:-(b.get(k,0)-a.get(k,0))):
    d=b.get(k,0)-a.get(k,0)
    if d or 'copy' in k.lower() or 'fill' in k.lower():
        print("  %-70s %6d -> %6d   per step %.3f" % (k[:70],a.get(k,0),b.get(k,0),d/dsteps))
PY
find $out -name "*.csv" -delete
