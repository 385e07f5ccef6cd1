# Same-box check: does any group of GPU tests change the speed of the dense-block sweep afterwards?  bench before / after each group,
# rocm-smi clocks sampled beside it (profiles/r08c_*: it does not -- the slow boxes of DESIGN.md 3.2 are slow from the start).
#   gpurun -- "bash tools/probes/statecheck.sh"
b() { python bench.py --no-cpu-baseline --no-variant --steps 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], j['roofline']['avg_launch_us'])"; }
( while true; do rocm-smi --showclocks 2>/dev/null | grep -E "fclk|mclk|sclk" | tr '\n' ' '; echo; sleep 2; done ) > gpurun_out/r08c_clocks.txt &
SMI=$!
b A
TNR_TEST_MMA=bf16x3 python -m pytest tests/test_gpu_dp.py -q -m gpu 2>&1 | tail -1
b B_after_dp
TNR_TEST_MMA=bf16x3 python -m pytest tests/test_gpu_step.py -q -m gpu -k "dp_collectives or headline" 2>&1 | tail -1
b C_after_step_dp_headline
TNR_TEST_MMA=bf16x3 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -1
b D_after_kernels
kill $SMI
