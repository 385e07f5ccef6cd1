# separate rocprofv3 --pmc passes over one bench step (HBM traffic + MFMA busy), summaries into gpurun_out/pmc_bench/
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc_bench; mkdir -p $O
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$N -- $CMD > $O/run_$N.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pmc_$N > $O/pmc_$N.summary.csv
done
python $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE.summary.csv $O/pmc_WRITE_SIZE.summary.csv $O/pmc_traffic.json
head -5 $O/pmc_FETCH_SIZE.summary.csv
