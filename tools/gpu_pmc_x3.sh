# one rocprofv3 --pmc pass (MFMA busy) over one bench step with TNR_MMA=bf16x3
cd /tmp && export TMPDIR=/tmp
R=/root/repo; mkdir -p $R/gpurun_out
rm -rf /tmp/pmc_x3
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_x3 -- python $R/bench.py --mma bf16x3 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_x3_run.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_x3 > $R/gpurun_out/${1:-r02h}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_mma_bf16x3.summary.csv
head -16 $R/gpurun_out/${1:-r02h}_pmc_SQ_VALU_MFMA_BUSY_CYCLES_mma_bf16x3.summary.csv
