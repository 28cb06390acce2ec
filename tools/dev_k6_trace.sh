cd /tmp && export TMPDIR=/tmp
ILCC_SEED_STRIDE_T=5 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_k6 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 5 --in-flight ${INFLIGHT:-3} > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/prof_k6/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    name=r['Kernel_Name'][:40]
    key=(name, r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'))
    agg[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:14]:
    v=sorted(v); print(k, 'n',len(v),'avg us %.1f'%(sum(v)/len(v)/1e3),'med %.1f'%(v[len(v)//2]/1e3),'min %.1f'%(v[0]/1e3))
PY
