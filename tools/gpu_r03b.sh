# round 3: tnr_conv_sweep correctness (bit equality with per-layer launches), timing, variant builds (timing only)
cd /root/repo; mkdir -p gpurun_out
TAG=${1:-r03b}; shift
( TNR_MMA=bf16x3 timeout 300 python tools/probes/sweep_check.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${TAG}_sweep_check.txt 2>&1
for v in "$@"; do
  [ -f trainner_amd/lib/variants/lib$v.so ] || continue
  echo "== variant $v (timing only)" >> gpurun_out/${TAG}_sweep_check.txt
  ( TNR_HIP_LIB=/root/repo/trainner_amd/lib/variants/lib$v.so TNR_MMA=bf16x3 timeout 120 python tools/probes/sweep_check.py --time-only 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/${TAG}_sweep_check.txt 2>&1
done
grep -v "done; chain" gpurun_out/${TAG}_sweep_check.txt | tail -30
