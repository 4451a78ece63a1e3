#!/bin/bash
# rocprofv3 kernel trace of the per-image latency leg -> profiles/r05/latency/ (kernel stats, the idle share of an image).  On the GPU box.
out=$GRAFT_REPO_ROOT/gpurun_out/latency_${MODE:-f16x3}
cd /tmp && export TMPDIR=/tmp
IMAGES=${IMAGES:-4} rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/latency_image.py > $out.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $out.log
t=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/latency_idle.py "$t" $(( (${DETS:-8} + ${BS:-4} - 1) / ${BS:-4} )) | tee $out.idle.txt
s=$(find $out -name "*kernel_stats.csv" | head -1); cp "$s" $out.kernel_stats.csv
