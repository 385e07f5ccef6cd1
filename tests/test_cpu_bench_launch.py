"""`python bench.py --gpus 2` without a launcher must start its own two ranks (torch.distributed.run on 127.0.0.1),
run the data-parallel step and print ONE JSON line from rank 0 (-m "not gpu": CPU dry run over gloo + the torch
stand-in for the C ABI; the line is marked invalid because it is a plumbing check, not a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["world_size_observed"] == 2 and j["config"]["parallelism"] == "dp2"
    assert j["config"]["global_batch"] == 4 and j["scaling"] == "weak" and j["steps"] == 1
    assert j["value"] is None and j["invalid"] == "dry run"          # never mistaken for a measurement
    assert all(v == v for v in j["losses"].values())                 # finite
    # the contract's keys are all there; the second measurement (the other fp32 arithmetic) only exists for a real 1-GPU run
    for k in ("metric", "unit", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "variant_f32_mfma"):
        assert k in j, k
    assert j["variant_f32_mfma"] is None and j["dtype"] == "f32 (bf16x3)" and j["config"]["mma"].startswith("fp32 arithmetic on the bf16 matrix core")
    for k in ("variant_amp", "variant_feed_paired", "variant_config4", "variant_config5"):      # 1-GPU-only measurements: present, null here
        assert k in j and j[k] is None, k
    check_first_multi_gpu_run_keys(j, 2)


def check_first_multi_gpu_run_keys(j, world):
    """What a first real N-GPU run must show in its ONE line: the IPC mode RCCL starts with, every rank's own step
    time (a straggler), every rank's dense-block form record and the form all ranks agreed on (a split vote)."""
    c = j["config"]
    assert c["hsa_enable_ipc_mode_legacy"] == "0"                    # defaulted in bench.main() itself, not only by self_launch
    assert len(c["per_rank_ms_per_step"]) == world and all(t > 0 for t in c["per_rank_ms_per_step"])
    assert max(c["per_rank_ms_per_step"]) == j["ms_per_step"]
    form = c["dense_block_form"]
    assert [r["rank"] for r in form["per_rank"]] == list(range(world))
    assert all(set(r) >= {"choice", "sweep_us", "layers_us"} for r in form["per_rank"])
    assert form["choice"] is None                                    # nothing to time on the CPU stand-in: the default form


def test_bench_under_the_drivers_eight_rank_launch():
    """The scaling run's own command line at N = 8 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W`), as a CPU dry run: eight ranks rendezvous, shard a global batch
    of 8 x the per-rank batch, step the data-parallel model and rank 0 alone prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    port = 29700 + os.getpid() % 200
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                          "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["world_size_observed"] == 8 and j["config"]["parallelism"] == "dp8"
    assert j["config"]["global_batch"] == 16 and j["scaling"] == "weak" and j["steps"] == 1 and j["warmup"] == 1
    assert j["value"] is None and j["invalid"] == "dry run"
    assert all(v == v for v in j["losses"].values())
    check_first_multi_gpu_run_keys(j, 8)
