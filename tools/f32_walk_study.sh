#!/bin/bash
# FETCH_SIZE of the fp32 engine's ViT linears under different tile walks (PP_F_WALK = band rows << 8 | group columns); on the GPU box
cd /tmp && export TMPDIR=/tmp
for w in ${WALKS:-0 1032 2056 2064 1040}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/walk_$w
  PP_F_WALK=$w CFGS=${CFGS:-3} REPS=4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_f32.py > $out.log 2>&1
  python3 - $out $w <<'PY'
import csv,glob,sys,collections
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True): rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    if r["Counter_Name"]=="FETCH_SIZE" and "pp_gemm_f_kernel" in r["Kernel_Name"]:
        k=(r["Kernel_Name"][:60], r["Grid_Size"])
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])*2*1024
print("walk",sys.argv[2], {k[0][17:48]+" g"+k[1]: round(v[1]/v[0]/1e6) for k,v in agg.items()}, "MB per launch")
PY
  grep linear $out.log
done
