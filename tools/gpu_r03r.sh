cd /root/repo; mkdir -p gpurun_out
timeout 120 ./tools/probes/mfma_slots > gpurun_out/r03r_mfma_slots.txt 2>&1; cat gpurun_out/r03r_mfma_slots.txt
