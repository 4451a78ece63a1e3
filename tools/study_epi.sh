# usage (GPU box): VARIANTS="base _old" MS="41120" CFGS=5,4 bash tools/study_epi.sh — the ViT linears under variant builds of the library
# (PP_LIB_SUFFIX; "base" = the shipped one): full kernel vs -DPP_STUDY_NOSTORE (_ns) / -DPP_STUDY_NOEPI (_ne) / ... builds
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-base _ns _ne}; do sfx=$v; [ "$v" = base ] && sfx=""; for M in ${MS:-41120 164480}; do echo "== variant '$v' M=$M"; PP_LIB_SUFFIX=$sfx CFGS=${CFGS:-5} PLANES=1 python tools/bench_linear.py $M 2>&1 | grep -v "fc1-noact\|Warn\|amdgpu.ids"; done; done
