# rocprofv3 kernel stats + separate PMC passes (HBM traffic, MFMA busy) of the bench command in one matrix-core mode
# usage: bash tools/gpu_prof.sh <tag> <bf16x3|f32>
TAG=${1:-rXX}; MODE=${2:-bf16x3}
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --mma $MODE --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-variant > $O/${TAG}_prof_${MODE}.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_bench_steps3_${MODE}.csv
CMD="python $R/bench.py --mma $MODE --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-variant"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$N -- $CMD > $O/${TAG}_pmc_run_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_$N > $O/${TAG}_pmc_${N}_${MODE}.summary.csv
done
python $R/tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE_${MODE}.summary.csv $O/${TAG}_pmc_WRITE_SIZE_${MODE}.summary.csv $O/${TAG}_pmc_traffic_${MODE}.json
python $R/tools/pmc_busy.py $O/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_${MODE}.summary.csv $O/${TAG}_pmc_mfma_busy_${MODE}.json
head -14 $O/${TAG}_kernel_stats_bench_steps3_${MODE}.csv | cut -c1-150
python - <<P
import json
t=json.load(open("$O/${TAG}_pmc_traffic_${MODE}.json")); b=json.load(open("$O/${TAG}_pmc_mfma_busy_${MODE}.json"))
print({k:v for k,v in t.items() if k in ("conv_chain","conv_tile_3x3")}); print({k:v for k,v in b.items() if k in ("conv_chain","conv_tile_3x3","wgrad_tile")})
P
