# round 3: validation of the final tree -- suite (both arithmetic modes) + smoke, the default bench line (with cpu_baseline), rocprofv3 kernel stats and PMC passes
cd /root/repo; mkdir -p gpurun_out
bash tools/gpu_suite.sh r03fin
( time timeout 900 python bench.py ) > gpurun_out/r03fin_bench_default.json.log 2> gpurun_out/r03fin_bench_default.err
tail -1 gpurun_out/r03fin_bench_default.json.log | cut -c1-1500
bash tools/gpu_prof.sh r03fin bf16x3 > gpurun_out/r03fin_prof_summary.txt 2>&1
tail -4 gpurun_out/r03fin_prof_summary.txt | cut -c1-600
