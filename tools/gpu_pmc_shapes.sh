cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/pmc_shapes
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc1 -- python $R/tools/pmc_shapes.py > $R/gpurun_out/pmc_shapes/run1.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc1 > $R/gpurun_out/pmc_shapes/sq1.summary.csv
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc2 -- python $R/tools/pmc_shapes.py > $R/gpurun_out/pmc_shapes/run2.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc2 > $R/gpurun_out/pmc_shapes/sq2.summary.csv
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc3 -- python $R/tools/pmc_shapes.py > $R/gpurun_out/pmc_shapes/run3.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc3 > $R/gpurun_out/pmc_shapes/grbm.summary.csv
cp /tmp/pmc3/*/*kernel_trace.csv $R/gpurun_out/pmc_shapes/kernel_trace.csv 2>/dev/null
tail -3 $R/gpurun_out/pmc_shapes/run1.log
