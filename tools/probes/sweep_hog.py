"""How do the dense-block sweep launches behave when NOT every CU is available?  A spinning one-workgroup kernel (tools/probes/cu_hog.hip: 100 KB
of LDS, so no sweep workgroup fits next to it) holds `blocks` CUs on a side stream while tnr_conv_sweep runs; timing per launch with and without it,
for the four-wave form (tiles dispensed from per-XCD counters) and the eight-wave form (static deal: needs the whole grid resident).
    TNR_SWEEP_WAVES=4|8 python tools/probes/sweep_hog.py"""
import ctypes as C
import os
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import ops  # noqa: E402
from tools.probes.sweep_check import block  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(os.path.join(HERE, "libcu_hog.so")):       # (probe binaries are not tracked: built where they run)
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "cu_hog.hip"), "-o",
                    os.path.join(HERE, "libcu_hog.so")], check=True)
hog = C.CDLL(os.path.join(HERE, "libcu_hog.so"))
hog.cu_hog.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int]
dev = torch.device("cuda")
out = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
run = block(16, 128, 128, seed=5, grad_shape=False)
_, _, st = run("layers")
for _ in range(3):
    ops.conv_chain(st)
torch.cuda.synchronize()


def timed(blocks, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if blocks:
        hog.cu_hog(C.c_void_p(side.cuda_stream), 60.0, C.c_void_p(out.data_ptr()), blocks)      # ~60 ms of spinning
        torch.cuda._sleep(2_000_000)      # let it start
    e0.record()
    for _ in range(reps):
        ops.conv_chain(st)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


print("TNR_SWEEP_WAVES=%s TNR_SWEEP_FORM=%s: per launch alone %.1f us; with 1 / 4 / 8 / 16 CUs held (what RCCL's ring kernels do to a launch next to them) "
      "%.1f / %.1f / %.1f / %.1f us; error flag %d" % (os.environ.get("TNR_SWEEP_WAVES", "4"), os.environ.get("TNR_SWEEP_FORM", "direct"), timed(0), timed(1), timed(4),
                                                      timed(8), timed(16), ops.chain_error_flag()))
# and the results do not depend on who else is on the chip: bit-identical with 16 CUs held
ref_buf, ref_out, _ = run("sweep")
hog.cu_hog(C.c_void_p(side.cuda_stream), 60.0, C.c_void_p(out.data_ptr()), 16)
torch.cuda._sleep(2_000_000)
got_buf, got_out, _ = run("sweep")
torch.cuda.synchronize()
print("results with 16 CUs held bit-identical:", bool(torch.equal(ref_buf, got_buf) and torch.equal(ref_out, got_out)), "; error flag", ops.chain_error_flag())
