#!/bin/bash
# Kernel timeline of one training step of the bench (rocprofv3 --kernel-trace): how much of the step no kernel covers (launch gaps,
# stream hops) and the longest gaps with their neighbours.  usage: bash tools/step_timeline.sh [tag]  ->  gpurun_out/<tag>/step_timeline.txt
TAG=${1:-tl}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o x -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err || { echo "rocprofv3 exited $?"; tail -3 $OUT/prof.err; exit 1; }
python3 - <<PY > $OUT/step_timeline.txt
import csv, glob
rows=list(csv.DictReader(open(glob.glob('$OUT/prof/**/x_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
S=lambda r:int(r['Start_Timestamp']); E=lambda r:int(r['End_Timestamp'])
name=lambda r:r['Kernel_Name'].replace('(anonymous namespace)::','')[:56]
ad=[i for i,r in enumerate(rows) if 'adadelta_kernel' in r['Kernel_Name']]
a,b=ad[-3],ad[-2]                          # one whole step: from the end of an optimizer kernel to the end of the next
seg=rows[a+1:b+1]
t0,t1=E(rows[a]),E(rows[b])
iv=sorted((max(S(r),t0),min(E(r),t1)) for r in seg)
busy=0; cur_s,cur_e=iv[0]; gaps=[]
last_name={}
for (s,e),r in zip(iv, sorted(seg,key=S)):
    if s>cur_e:
        busy+=cur_e-cur_s; gaps.append((s-cur_e,cur_e,s)); cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print('step period %.1f us, %d kernels, covered by at least one kernel %.1f us, uncovered %.1f us (%.1f %%) in %d gaps'%((t1-t0)/1e3,len(seg),busy/1e3,(t1-t0-busy)/1e3,100.0*(t1-t0-busy)/(t1-t0),len(gaps)))
gaps.sort(reverse=True)
print('longest gaps (us): before-kernel -> after-kernel')
for g,ge,gs in gaps[:25]:
    prev=max((r for r in seg if E(r)<=ge+1),key=E,default=None); nxt=min((r for r in seg if S(r)>=gs-1),key=S,default=None)
    print('  %7.1f   %-56s -> %s'%(g/1e3,name(prev) if prev else '-',name(nxt) if nxt else '-'))
hist={}
for g,_,_ in gaps:
    k='<3' if g<3000 else '3-6' if g<6000 else '6-12' if g<12000 else '12-30' if g<30000 else '>30'
    hist[k]=hist.get(k,[0,0]); hist[k][0]+=1; hist[k][1]+=g
print('gap histogram (us): '+', '.join('%s: %d gaps / %.0f us'%(k,v[0],v[1]/1e3) for k,v in hist.items()))
PY
rm -rf $OUT/prof
cat $OUT/step_timeline.txt
