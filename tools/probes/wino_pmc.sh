#!/bin/bash
# rocprofv3 --pmc passes (SQ issue / wait / LDS / MFMA counters, three passes of 8 slots) over ten launches of conv5's shape in the Winograd kernel and in the
# direct weight-stream kernel: gpurun -- 'bash tools/probes/wino_pmc.sh'  ->  gpurun_out/r09f_pmc_{wino,d4}.csv  (profiles/r09f_pmc_wino.csv)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for W in wino d4; do
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1)); rm -rf /tmp/pw_$W$i
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pw_$W$i -- python $R/tools/probes/wino_check.py --one $W > $O/r09f_pmc_run_$W$i.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pw_$W$i | grep -E "kernel|wino_kernel|d4_kernel" >> $O/r09f_pmc_$W.csv
done
done
cat $O/r09f_pmc_wino.csv $O/r09f_pmc_d4.csv | cut -c1-200
