"""Is this box one of the slow ones for the dense-block sweep, and if so which form of the dense block suffers least?
    python tools/probes/slowbox_diag.py            # times the default form; on a slow box re-runs itself once per alternative form
    python tools/probes/slowbox_diag.py --one      # (child) time the form the environment selects and print one line"""
import os
import subprocess
import sys

os.environ.setdefault("TNR_MMA", "bf16x3")
os.environ.setdefault("TNR_SWEEP_AUTO", "0")      # time the forms themselves, not the per-box choice between them
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trainner_amd import ops  # noqa: E402
from tools.probes.sweep_check import block  # noqa: E402


def time_block(grad_shape, reps=60):
    run = block(16, 128, 128, seed=5, grad_shape=grad_shape, with_r2=False)
    _, _, st = run("sweep")
    ops.CONV_SWEEP = os.environ.get("TNR_CONV_SWEEP", "1") != "0"      # (run() leaves it True)
    for _ in range(10):
        ops.conv_chain(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.conv_chain(st)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


if __name__ == "__main__":
    us_f, us_g = time_block(False), time_block(True)
    tag = os.environ.get("DIAG_TAG", "default")
    print("DIAG %-28s forward %7.1f us   gradient mirror %7.1f us   latch %d" % (tag, us_f, us_g, ops.chain_error_flag()), flush=True)
    if "--one" in sys.argv:
        sys.exit(0)
    if us_f < 750.0 and "--force" not in sys.argv:
        print("DIAG normal box")
        sys.exit(0)
    print("DIAG SLOW BOX: trying the other forms", flush=True)
    os.system("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'fclk|mclk|sclk|socclk|Power' | tr '\\n' ' '; echo")
    for name, env in (("dma form", {"TNR_SWEEP_FORM": "dma"}), ("eight waves", {"TNR_SWEEP_WAVES": "8"}), ("dispensers per XCD", {"TNR_SWEEP_DISPENSERS": "8"}),
                      ("chain kernel (x3)", {"TNR_CONV_SWEEP": "0"}), ("chain 8-wave x3", {"TNR_CONV_SWEEP": "0", "TNR_CHAIN_X3W8": "1"}),
                      ("per layer", {"TNR_CONV_CHAIN": "0"}), ("fp32 mfma chain", {"TNR_MMA": "f32"}), ("default again", {})):
        e = dict(os.environ, DIAG_TAG=name, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=e, capture_output=True, text=True, timeout=300)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("DIAG")) or ("DIAG %s FAILED: %s" % (name, r.stderr[-300:])), flush=True)
