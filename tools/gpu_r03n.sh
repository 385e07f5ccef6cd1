cd /root/repo; mkdir -p gpurun_out
timeout 60 ./tools/probes/mfma_chain > gpurun_out/r03n_mfma_chain.txt 2>&1; cat gpurun_out/r03n_mfma_chain.txt
