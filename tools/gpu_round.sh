# full validation of the current tree under a profile tag: every GPU test, smoke, the default bench line (with cpu_baseline), the
# per-shape table, rocprofv3 kernel stats of the bench command, PMC passes (HBM traffic, MFMA busy), the variant benches.
# usage: bash tools/gpu_round.sh r02b   -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
TAG=${1:-rXX}
cd /root/repo
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${TAG}_smoke.log 2>&1
( time timeout 600 python bench.py ) > gpurun_out/${TAG}_bench_n1.json.log 2> gpurun_out/${TAG}_bench_n1.err
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-variant --detail > gpurun_out/${TAG}_bench_detail.json 2> gpurun_out/${TAG}_per_shape_table.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_${TAG}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_${TAG} -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof_${TAG}.log 2>&1
cd /root/repo
find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" -delete
cp $(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats_bench_steps3.csv
bash tools/gpu_pmc_bench.sh > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do cp gpurun_out/pmc_bench/pmc_$c.summary.csv gpurun_out/${TAG}_pmc_$c.summary.csv; done
cp gpurun_out/pmc_bench/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
timeout 300 python bench.py --amp --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_amp_bf16.json.log 2>/dev/null
# TNR_MMA=bf16x3 (fp32 arithmetic on the bf16 matrix core): bench line + per-shape table, and the WHOLE GPU suite in that mode
timeout 300 python bench.py --mma bf16x3 --steps 6 --warmup 2 --no-cpu-baseline --detail > gpurun_out/${TAG}_bench_mma_bf16x3.json.log 2> gpurun_out/${TAG}_per_shape_table_mma_bf16x3.txt
( time TNR_MMA=bf16x3 timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/${TAG}_pytest_gpu_mma_bf16x3.log 2>&1
timeout 300 python tools/bench_i2i.py --model pix2pix > gpurun_out/${TAG}_variant_bench_pix2pix.json.log 2>/dev/null
timeout 300 python tools/bench_i2i.py --model cyclegan --batch 8 > gpurun_out/${TAG}_variant_bench_cyclegan.json.log 2>/dev/null
tail -3 gpurun_out/${TAG}_pytest_gpu.log; tail -3 gpurun_out/${TAG}_pytest_gpu_mma_bf16x3.log; tail -4 gpurun_out/${TAG}_smoke.log | cut -c1-300; tail -1 gpurun_out/${TAG}_bench_n1.json.log | cut -c1-1200
head -4 gpurun_out/${TAG}_kernel_stats_bench_steps3.csv | cut -c1-150
python - $TAG <<'P'
import json,sys
t=json.load(open("gpurun_out/%s_pmc_traffic.json" % sys.argv[1] if len(sys.argv)>1 else "gpurun_out/pmc_bench/pmc_traffic.json"))
print({k:v for k,v in t.items() if k in ("conv_chain","conv_tile_3x3")})
P
