"""The self-launch of `python bench.py --gpus N` (gficf_amd/launch.py): N ranks from a plain command, one JSON line from
rank 0, a failing or hanging rank ends the job with a non-zero code instead of a hang.  CPU only (gloo)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "launch_stub.py")
DRIVER = (
    "import sys; sys.path.insert(0, %r)\n"
    "from gficf_amd import launch\n"
    "sys.exit(launch.spawn_ranks(sys.argv[1:], int(sys.argv[sys.argv.index('--gpus') + 1]), timeout_s=%s))\n"
)


def _run(args, timeout_s="None", env=None):
    e = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(v, None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", DRIVER % (ROOT, timeout_s), STUB] + args, capture_output=True, text=True, timeout=240, env=e)


def test_three_ranks_one_json_line():
    r = _run(["--gpus", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                     # rank 0 only; the other ranks' stdout went to stderr
    assert json.loads(lines[0]) == {"n_gpus": 3, "sum": 6}
    assert "rank 1 says hello" in r.stderr and "rank 2 says hello" in r.stderr


def test_failing_rank_stops_the_job():
    t0 = time.time()
    r = _run(["--gpus", "3", "--fail-rank", "1"])
    assert r.returncode == 3
    assert "rank 1 exited with code 3" in r.stderr
    assert r.stdout.strip() == ""
    assert time.time() - t0 < 120                        # the survivors were stopped, not waited for


def test_hanging_ranks_time_out():
    r = _run(["--gpus", "2", "--hang"], timeout_s="3")
    assert r.returncode == 124 and "still running after 3 s" in r.stderr


def test_need_gpus_fails_fast_without_devices():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("this box has the GPUs")
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=240, env=e)
    assert r.returncode == 2 and "2 GPUs asked for" in r.stderr and r.stdout.strip() == ""


def test_bench_self_launch_reaches_the_ranks_on_a_cpu_box():
    """Without a GPU the ranks of a rehearsal start, find no device and say so: the launcher reports failure (no hang,
    no JSON).  On a GPU box the same command is exercised for real by tests/test_dist_gpu.py."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present: covered by the -m gpu test")
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "1", "--warmup", "0", "--no-extras"],
                       capture_output=True, text=True, timeout=240, env=e)
    assert r.returncode != 0 and "needs a GPU" in r.stderr and r.stdout.strip() == ""


def test_gpu_count_comes_from_sysfs_not_from_the_hip_runtime(tmp_path, monkeypatch):
    """The launcher counts GPUs from the KFD topology (nodes with SIMDs), narrowed by the *_VISIBLE_DEVICES lists — the HIP
    runtime is never loaded in the parent."""
    sys.path.insert(0, ROOT)
    from gficf_amd import launch

    root = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):              # two CPU nodes, three GPUs
        (root / str(i)).mkdir(parents=True)
        (root / str(i) / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    assert launch.kfd_gpu_nodes(str(root)) == 3
    assert launch.kfd_gpu_nodes(str(tmp_path / "missing")) is None
    monkeypatch.setattr(launch, "kfd_gpu_nodes", lambda root=None: 8)
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert launch.visible_gpus() == 8
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert launch.visible_gpus() == 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4")
    assert launch.visible_gpus() == 1
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
    assert launch.visible_gpus() == 0
    # and the module does not pull torch (hence no HIP runtime) into the launcher
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); import gficf_amd.launch as l; l.visible_gpus(); "
                        "print('torch' in sys.modules)"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "False", (r.stdout, r.stderr[-500:])
