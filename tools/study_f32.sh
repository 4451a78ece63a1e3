#!/bin/bash
# K-loop study of the fp32 engine (csrc/pp_gemm_f.hip): variant builds that drop one ingredient of the loop each (results are garbage,
# timings are not), run on the shapes of tools/bench_f32.py.  Build here (no GPU needed): tools/study_f32.sh build; on the GPU box: run.
set -e
cd "$(dirname "$0")/.."
VARS="nd:-DPP_STUDY_F_NODMA nb:-DPP_STUDY_F_NOBAR nl:-DPP_STUDY_F_NOLDS ne:-DPP_STUDY_F_NOEPI ndnb:-DPP_STUDY_F_NODMA,-DPP_STUDY_F_NOBAR"
if [ "$1" = build ]; then
    python -m picopose_amd.build > /dev/null
    for v in $VARS; do
        sfx=_f${v%%:*}; flags=$(echo ${v#*:} | tr , ' ')
        rm -rf picopose_amd/lib/obj$sfx; cp -rp picopose_amd/lib/obj picopose_amd/lib/obj$sfx; rm -f picopose_amd/lib/obj$sfx/pp_gemm_f.o
        PP_LIB_SUFFIX=$sfx PP_HIPCC_FLAGS="$flags" python -m picopose_amd.build > /dev/null &
    done
    wait
    ls -la picopose_amd/lib/*.so
else
    for sfx in "" _fnd _fnb _fnl _fne _fndnb; do echo "== variant '$sfx'"; PP_LIB_SUFFIX=$sfx CFGS=${CFGS:-5} REPS=${REPS:-10} python tools/bench_f32.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done
fi
