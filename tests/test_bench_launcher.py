"""bench.py's N-rank contract without a GPU: `python bench.py --gpus 2 --dry-run` must start two ranks ITSELF
(fresh child process, torch.distributed.run, gloo), give every rank its own seeded inputs, bracket the timed
steps with barriers, take the MAX over ranks and print ONE JSON line from rank 0; a --gpus / WORLD_SIZE mismatch
under a launcher must fail."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e,
                          timeout=timeout)


def test_gpus_2_launches_two_ranks_itself():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "6", "--dry-run"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # ONE line on stdout, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 12 and d["config"]["meshes_per_gpu"] == 6
    assert d["rank_seeds"] == [1000, 1001]              # per-rank inputs differ ...
    assert d["rank_input_checksums"][0] != d["rank_input_checksums"][1]
    # ... the reported time is the MAX over ranks of the barrier-bracketed region, hence >= every rank's own work
    # (rank 1's stand-in step sleeps twice as long as rank 0's)
    elapsed = d["ms_per_step"] * d["steps"] * 1e-3
    assert abs(elapsed - max(d["rank_elapsed_s"])) < 1e-9
    assert elapsed >= max(d["rank_work_s"]) and d["rank_work_s"][1] > d["rank_work_s"][0]
    assert abs(d["value"] - 12 * 4 / elapsed) < 0.1     # whole-job rate: all ranks' meshes / max time
    # the train leg's protocol (a CPU stand-in under DDP / gloo here): the step as trained, the same step without the
    # gradient all-reduce, the all-reduce's volume - the keys the GPU leg prints
    t = d["train_step"]
    assert t["stand_in"] and t["ddp"] and t["backend"] == "gloo" and t["n_gpus"] == 2 and t["scaling"] == "weak"
    assert t["global_batch"] == 2 * t["images_per_gpu"] and t["allreduce_MiB"] > 0
    for k in ("ddp_step", "no_sync_step"):
        assert t[k]["ms_per_step"] > 0 and t[k]["images_per_s"] > 0
    assert abs(t["allreduce_exposed_ms"] - (t["ddp_step"]["ms_per_step"] - t["no_sync_step"]["ms_per_step"])) < 2e-3


def test_gpus_8_dry_run_is_the_eight_rank_protocol():
    """The round-end scaling run's shape (N = 8, BASELINE configs[4]: global 1024 = 128 per GPU) rehearsed over gloo:
    eight ranks started by bench.py itself, per-rank seeds, MAX reduction, the strong split 1024 -> 128, every rank
    listed in `rccl_ranks_seen` with its own time, and the train leg's keys (train.py:196,205-210: 4 x num_gpus split)."""
    r = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--scaling", "strong", "--global-batch", "1024", "--dry-run",
              "--train-batch", "8", "--train-steps", "2"], timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 1024 and d["config"]["meshes_per_gpu"] == 128
    assert d["rank_seeds"] == [1000 + r_ for r_ in range(8)] and len(set(d["rank_input_checksums"])) == 8
    elapsed = d["ms_per_step"] * d["steps"] * 1e-3
    assert abs(elapsed - max(d["rank_elapsed_s"])) < 1e-9 and elapsed >= max(d["rank_work_s"])
    assert d["rank_work_s"][7] > d["rank_work_s"][0]            # rank 7's stand-in step sleeps 8 x rank 0's
    assert abs(d["value"] - 1024 * 3 / elapsed) < 0.5
    seen = d["rccl_ranks_seen"]
    assert [e["rank"] for e in seen] == list(range(8)) and len({e["name"] for e in seen}) == 8      # 8 distinct processes
    assert all(e["backend"] == "gloo" for e in seen)
    assert d["rank_ms_per_step"] == [e["ms_per_step"] for e in seen] and max(d["rank_ms_per_step"]) <= d["ms_per_step"] + 1e-3
    t = d["train_step"]
    assert t["ddp"] and t["n_gpus"] == 8 and t["global_batch"] == 64 and t["backend"] == "gloo"
    for k in ("ddp_step", "no_sync_step"):
        assert t[k]["ms_per_step"] > 0
    assert "allreduce_exposed_ms" in t and t["allreduce_MiB"] > 0


def test_a_failing_rank_fails_the_fresh_child():
    """A rank that dies must not leave a JSON line or a zero exit code behind: bench.py started the ranks as a fresh
    child process (never an exec) and returns that child's code."""
    r = _run(["--gpus", "4", "--steps", "2", "--warmup", "0", "--batch", "4", "--dry-run", "--no-train-leg"],
             env={"BENCH_DRY_FAIL_RANK": "2"}, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert b"rank 2 fails on purpose" in r.stderr


def test_strong_scaling_splits_a_fixed_global_batch():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--scaling", "strong", "--global-batch", "10", "--dry-run",
              "--no-train-leg"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.strip()][0])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 10 and d["config"]["meshes_per_gpu"] == 5
    assert d["train_step"] is None
    r = _run(["--gpus", "2", "--scaling", "strong", "--global-batch", "9", "--dry-run"])
    assert r.returncode != 0 and b"does not split" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0
    assert b"--gpus 4" in r.stderr and b"WORLD_SIZE=2" in r.stderr
    assert r.stdout.strip() == b""


def test_single_rank_default_needs_no_launcher():
    r = _run(["--steps", "2", "--warmup", "0", "--batch", "3", "--dry-run"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip())
    assert d["n_gpus"] == 1 and d["rank_seeds"] == [1000]
    assert [e["rank"] for e in d["rccl_ranks_seen"]] == [0] and d["rccl_ranks_seen"][0]["backend"] is None


import pytest


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box():
    """The real N = 2 path on hardware: `bench.py --gpus 2` starts two ranks itself, both on cuda:0 (local rank modulo
    the device count), process group over gloo (RCCL needs a device per rank): HIP graph capture, barrier-bracketed
    timing, MAX over ranks, one line with n_gpus = 2 and the whole-job rate."""
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-breakdown", "--train-batch", "16",
              "--train-steps", "3", "--train-global-batch", "64"],
             env={"BENCH_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 256 and d["config"]["meshes_per_gpu"] == 128
    assert d["value"] > 0 and abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    assert "dry_run" not in d and d["config"]["launch"] in ("graph", "eager")
    w = d["ms_per_step_windows"]
    assert w["n"] == 10 and w["min"] <= w["median"] <= w["max"] and w["min"] > 0
    assert len(d["build_id"]) == 64
    assert [e["rank"] for e in d["rccl_ranks_seen"]] == [0, 1] and all(e["device"] == 0 for e in d["rccl_ranks_seen"])
    assert len(d["rank_ms_per_step"]) == 2 and d["warmup_actual"] >= 200
    # the data-parallel train step (BASELINE configs[4]: ENet + IEF + decoder with both heads + both losses + Adam
    # under DistributedDataParallel): the step as trained, without the all-reduce, and the all-reduce's volume
    t = d["train_step"]
    assert t["ddp"] and t["n_gpus"] == 2 and t["images_per_gpu"] == 16 and t["global_batch"] == 32
    assert 50.0 < t["allreduce_MiB"] < 60.0                          # ENet + IEF: 14.58 M parameters = 55.6 MiB
    for k in ("ddp_step", "no_sync_step"):
        assert t[k]["ms_per_step"] > 0 and t[k]["images_per_s"] > 0
    assert "allreduce_exposed_ms" in t and t["strong_global_64"]["scaling"] == "strong" and t["strong_global_64"]["images_per_gpu"] == 32
