# round 3, call 1: (a) the GPU suite on the tree with the round-2 ADVICE fixes, (b) phase breakdown of the dense-block chain on the
# fp32 and on the bf16x3 matrix-core path (tools/probes/conv_timeline), (c) ds_read_b64_tr_b16 semantics, (d) chain microbench per mode
cd /root/repo; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/r03a_pytest_gpu.log 2>&1
tail -3 gpurun_out/r03a_pytest_gpu.log
( for m in 0 2; do echo "== TNR_PROBE_MMA=$m"; TNR_PROBE_MMA=$m timeout 120 ./tools/probes/conv_timeline; done ) > gpurun_out/r03a_chain_timeline.txt 2>&1
cat gpurun_out/r03a_chain_timeline.txt
timeout 60 ./tools/probes/tr_read > gpurun_out/r03a_tr_read.txt 2>&1; head -40 gpurun_out/r03a_tr_read.txt
( for e in "TNR_MMA=f32" "TNR_MMA=bf16x3" "TNR_MMA=bf16x3 TNR_CHAIN_X3W8=1"; do echo "== $e"; env $e timeout 120 python tools/microbench_chain.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r03a_microbench_chain.txt 2>&1
cat gpurun_out/r03a_microbench_chain.txt
