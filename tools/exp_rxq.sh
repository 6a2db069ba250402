cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for cond in cpu nocpu; do
  if [ $cond = cpu ]; then extra=""; else extra="--no-cpu"; fi
  rm -rf $R/gpurun_out/rxq_$cond
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rxq_$cond -- python3 bench.py --full-line --no-live-traffic $extra --steps 40 --warmup 4 > $R/gpurun_out/rxq_$cond.log 2>&1
  echo "== $cond"; grep SUMMARY $R/gpurun_out/rxq_$cond.log | cut -c1-300
  python3 - $cond <<'PY'
import csv,glob,collections,sys
fs=glob.glob('gpurun_out/rxq_%s/**/*kernel_trace.csv'%sys.argv[1],recursive=True)
best=max(fs,key=lambda f:sum(1 for _ in open(f)))
rows=list(csv.DictReader(open(best)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
pk=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("wf_packet_kernel")]
byp=[i for i,r in enumerate(rows) if i>pk[0] and r["Kernel_Name"].startswith("ddc_wf_bypass")]
lo=byp[0]
acq=[i for i,r in enumerate(rows) if i>lo and r["Kernel_Name"].startswith("void acq_correlate")]
hi=acq[0] if acq else len(rows)
# steady region: bypass launches 80..160 of the light bank (the timed loop)
b2=[i for i in byp if i<hi]
win=rows[b2[70]:b2[150]] if len(b2)>150 else rows[lo:hi]
t0=int(win[0]["Start_Timestamp"]); t1=int(win[-1]["Start_Timestamp"])
print("  window %.3f ms per step over %d steps"%((t1-t0)/1e6/80,80))
d=collections.defaultdict(list)
for r in win: d[(r["Kernel_Name"][:30],r["Queue_Id"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(),key=lambda kv:-sum(kv[1]))[:12]: print("   %-32s q%s n %3d avg %7.1f us"%(k[0],k[1],len(v),sum(v)/len(v)))
PY
done
find gpurun_out/rxq_cpu gpurun_out/rxq_nocpu -name "*.csv" -delete
