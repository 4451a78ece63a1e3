# The non-headline workloads on the current code -> gpurun_out/wl/bench_*.json (copy to profiles/rNN/workloads/):
# BASELINE configs[1] (stage 1, batch 8 / 42 templates / ViT-S), the stage-1 shapes of configs[2] and [4], the extended template bank
# (SURVEY 8f row 1), the small full paths, config/base.yaml's own configuration (ViT-L, test batch 4), configs[4] in fp16 mode.
mkdir -p gpurun_out/wl
for w in stage1_b32_n162_c768 stage1_b8_n42_c384 stage1_b32_n162_c1024 stage1_b64_n512_c1024_f16bank full_cached_b32_n162_vitb full_b8_n42_vits full_b8_n42_vitl full_b4_n162_vitl; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-exact-leg --no-train-leg --no-latency-leg > gpurun_out/wl/bench_$w.json 2> gpurun_out/wl/bench_$w.err || { echo "FAILED $w"; exit 1; }
  python -c "
import json,sys; d=json.loads(open('gpurun_out/wl/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', 'roofline', d['roofline']['bound'], 'frac', round(d['roofline']['frac'],3))"
done
timeout -k 10 900 python bench.py --workload full_b64_n512_vitl --mode fp16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/wl/bench_full_b64_n512_vitl_fp16.json 2> gpurun_out/wl/bench_full_b64_n512_vitl_fp16.err || { echo "FAILED vitl fp16"; exit 1; }
python -c "
import json,sys; d=json.loads(open('gpurun_out/wl/bench_full_b64_n512_vitl_fp16.json').read().strip().splitlines()[-1]); print('full_b64_n512_vitl fp16', round(d['value'],1), d['unit'], round(d['ms_per_step'],2), 'ms', 'roofline frac', round(d['roofline']['frac'],3))"
