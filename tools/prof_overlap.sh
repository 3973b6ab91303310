#!/bin/bash
# kernel trace of the bench step with the CU-masked overlap off / on (gpurun_out/<tag>/): phase lengths per step.
TAG=${1:-ovl}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for mode in 0 1; do
export ASR_OVERLAP=$mode
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof$mode -o x -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/prof_bench$mode.json 2> $OUT/prof$mode.err || { echo "rocprofv3 exited $?"; exit 1; }
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/prof$mode/x_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
S=lambda r:int(r['Start_Timestamp']); E=lambda r:int(r['End_Timestamp'])
db=[i for i,r in enumerate(rows) if 'dec_bwd_persist' in r['Kernel_Name']]
ad=[i for i,r in enumerate(rows) if 'adadelta_kernel' in r['Kernel_Name']]
df=[i for i,r in enumerate(rows) if 'dec_fwd_persist' in r['Kernel_Name']]
print('mode $mode')
for k in range(3, len(db)):
    a_prev=max(i for i in ad if i<db[k]); a_next=min(i for i in ad if i>db[k]); f=max(i for i in df if i<db[k])
    print('  step %d: forward (prev adadelta end -> dec_fwd start) %.0f us, dec_fwd %.0f, dec_fwd end -> dec_bwd start %.0f, dec_bwd %.0f, dec_bwd end -> adadelta end %.0f ; period %.0f'%(
        k,(S(rows[f])-E(rows[a_prev]))/1e3,(E(rows[f])-S(rows[f]))/1e3,(S(rows[db[k]])-E(rows[f]))/1e3,(E(rows[db[k]])-S(rows[db[k]]))/1e3,(E(rows[a_next])-E(rows[db[k]]))/1e3,(E(rows[a_next])-E(rows[a_prev]))/1e3))
i0=db[-2]; t0=S(rows[i0])
with open('$OUT/timeline$mode.txt','w') as f:
    for r in rows[i0:i0+130]:
        f.write('%9.1f +%8.1f us  q%s  %s\n'%((S(r)-t0)/1e3,(E(r)-S(r))/1e3,r['Queue_Id'],r['Kernel_Name'][:60].replace('(anonymous namespace)::','')))
PY
rm -rf $OUT/prof$mode
done
echo done
