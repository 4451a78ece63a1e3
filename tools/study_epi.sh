cd $GRAFT_REPO_ROOT
for sfx in ${VARIANTS:-"" _ns _ne _ss}; do for M in ${MS:-41120 164480}; do echo "== variant '$sfx' M=$M"; PP_LIB_SUFFIX=$sfx CFGS=${CFGS:-5} PLANES=1 python tools/bench_linear.py $M 2>&1 | grep -v "fc1-noact\|Warn\|amdgpu.ids"; done; done
