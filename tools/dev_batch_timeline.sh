cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 12 --warmup 4 --in-flight 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/prof_tl/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
# find the last full batch: locate k1_roi_count occurrences
idx=[i for i,r in enumerate(rows) if 'k1_roi_count' in r['Kernel_Name']]
a,b=idx[9],idx[10]
t0=int(rows[a]['Start_Timestamp']); prev_end=t0
tot_k=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%-44s start %8.1f us  dur %7.1f  gap %6.1f'%(r['Kernel_Name'][:44],(s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3))
    prev_end=e; tot_k+=e-s
print('batch span %.1f us, kernels %.1f us, next batch starts at %.1f us'%((prev_end-t0)/1e3,tot_k/1e3,(int(rows[b]['Start_Timestamp'])-t0)/1e3))
PY
