for sfx in ${VARIANTS:-"" _a _b _c _d}; do echo "== variant '$sfx'"; PP_LIB_SUFFIX=$sfx CFGS=${CFGS:-5,4} python tools/bench_linear.py 2>&1 | grep -v "fc1-noact\|Warn\|amdgpu.ids"; done
