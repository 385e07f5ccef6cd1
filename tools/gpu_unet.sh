cd /root/repo
mkdir -p gpurun_out
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --netd unet --detail > gpurun_out/r02c_variant_bench_netdunet.json.log 2> gpurun_out/unet_detail.txt
tail -1 gpurun_out/r02c_variant_bench_netdunet.json.log | cut -c1-250
grep -v "amdgpu.ids\|FeatureExtractor" gpurun_out/unet_detail.txt | head -40
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --netd unet --feed resrgan > gpurun_out/r02c_variant_bench_feedresrgannetdunet.json.log 2>/dev/null
tail -1 gpurun_out/r02c_variant_bench_feedresrgannetdunet.json.log | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --feed paired > gpurun_out/r02c_variant_bench_feedpaired.json.log 2>/dev/null
tail -1 gpurun_out/r02c_variant_bench_feedpaired.json.log | cut -c1-200
