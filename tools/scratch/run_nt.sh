cd /tmp && export TMPDIR=/tmp
for sfx in "" _nt0; do
  PP_LIB_SUFFIX=$sfx CFGS=3,5 REPS=20 python3 $GRAFT_REPO_ROOT/tools/bench_f32.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$sfx] /"
  out=$GRAFT_REPO_ROOT/gpurun_out/nt$sfx
  PP_LIB_SUFFIX=$sfx CFGS=3 REPS=4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_f32.py > $out.log 2>&1
done
