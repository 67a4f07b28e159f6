"""One process per GPU, started from a plain `python script.py --gpus N`.

`spawn_ranks` is what `bench.py` (and any other driver script of the sharded path, gficf_amd/dist.py) calls when it is
asked for N > 1 GPUs but was not started by `torch.distributed.run`: it starts N fresh interpreters of the same script
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them, hands rank 0's stdout through and
returns a non-zero code if any rank failed.  The caller must not have touched the GPU yet (the children are new
processes, nothing is re-exec'ed, but a parent holding a HIP context would only be in the way); the GPUs are counted from
the KFD topology in sysfs (`visible_gpus`), so the launcher never loads the HIP runtime at all.

What is sharded by the ranks so started: the cells of the reference's `parallelFor(0, N, worker)`
(src/rcpp_parallel_jaccard_coeff.cpp:73), see gficf_amd/dist.py.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched_by_torchrun() -> bool:
    """True when the rank environment of `torch.distributed.run` (or of spawn_ranks) is present."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def _parse_visible(var: str) -> list[str] | None:
    v = os.environ.get(var)
    if v is None:
        return None
    return [t.strip() for t in v.split(",") if t.strip() != ""]


def kfd_gpu_nodes(root: str = "/sys/class/kfd/kfd/topology/nodes") -> int | None:
    """GPUs the kernel driver lists (KFD topology nodes with SIMDs: CPU nodes have ``simd_count 0``), read from sysfs —
    no HIP, no HSA, nothing initialised.  None when the topology is not readable (no amdgpu driver: a CPU box)."""
    try:
        nodes = sorted(os.listdir(root), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            props = open(os.path.join(root, d, "properties")).read()
        except OSError:
            continue
        for line in props.splitlines():
            parts = line.split()
            if len(parts) == 2 and parts[0] == "simd_count" and parts[1].isdigit() and int(parts[1]) > 0:
                n += 1
                break
    return n


def runtime_gpus_in_a_child() -> int:
    """What the HIP runtime itself counts, asked of a throw-away child process (this process stays runtime-free)."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return 0


def visible_gpus() -> int:
    """Number of GPUs a rank of this job could use, WITHOUT initialising any of them and without the HIP runtime: the KFD
    topology in sysfs, narrowed by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES as the runtime would
    (a list shorter than the node count wins; an empty list is 0).  Where sysfs says nothing (no driver) a throw-away child
    process counts with torch — the launcher itself never loads the HIP runtime either way (on ROCm builds without
    amdsmi `torch.cuda.device_count()` falls back to hipGetDeviceCount, which does initialise it)."""
    n = kfd_gpu_nodes()
    if n is None:
        return runtime_gpus_in_a_child()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = _parse_visible(var)
        if lst is not None:
            n = min(n, len(lst))
    return n


def spawn_ranks(argv: list[str], n: int, *, need_gpus: int | None = None, timeout_s: float | None = None,
                poll_s: float = 0.05, extra_env: dict | None = None) -> int:
    """Run `python argv...` as `n` ranks on this node; returns the job's exit code (0 = every rank exited 0).

    argv       the script and its arguments (the same for every rank);
    need_gpus  fail fast (exit code 2, message on stderr) when fewer GPUs are visible; None: no check
               (a rehearsal in which the ranks share one device, or a CPU test);
    timeout_s  kill the ranks that are still running after this many seconds (exit code 124).
    Of rank 0's stdout the lines that are JSON objects reach this process's stdout (the one line of bench.py; gloo and
    RCCL print their own chatter there too), everything else and the other ranks' stdout goes to stderr.
    When a rank fails the remaining ones — which would wait for it in their next collective — are terminated, by PID."""
    if n < 1:
        raise ValueError("spawn_ranks: n must be >= 1")
    if need_gpus is not None:
        have = visible_gpus()
        if have < need_gpus:                                       # before refusing the job: what the runtime itself says (a topology this
            have = max(have, runtime_gpus_in_a_child())            # parser misreads must not cost a run on a box that has the GPUs)
        if have < need_gpus:
            sys.stderr.write(f"{os.path.basename(argv[0])}: {need_gpus} GPUs asked for, {have} visible on this node\n")
            return 2
    port = free_port()
    procs: list[subprocess.Popen] = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GFICF_SPAWNED_RANK": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
        if extra_env:
            env.update({k: str(v) for k, v in extra_env.items()})
        try:
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          text=(r == 0) or None))
        except OSError as ex:                                      # could not start rank r: the ranks already running would wait for it forever
            sys.stderr.write(f"could not start rank {r}: {ex}\n")
            for q in procs:
                q.terminate()
            for q in procs:
                try:
                    q.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    q.kill()
            return 1

    def forward():                                                 # rank 0's stdout: JSON lines through, the rest to stderr
        for line in procs[0].stdout:
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()

    fw = threading.Thread(target=forward, daemon=True)
    fw.start()
    t0 = time.monotonic()
    rc = 0
    t_term = None                                                  # when the survivors were told to stop
    alive = set(range(n))

    def stop_others(why: str):
        nonlocal t_term
        sys.stderr.write(why + "\n")
        for q in alive:
            procs[q].terminate()
        t_term = time.monotonic()

    while alive:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 128 - c
                stop_others(f"rank {r} exited with code {c}; stopping the other ranks")
        if alive and rc == 0 and timeout_s is not None and time.monotonic() - t0 > timeout_s:
            rc = 124
            stop_others(f"ranks {sorted(alive)} still running after {timeout_s:.0f} s; stopping them")
        if alive:
            time.sleep(poll_s)
            if t_term is not None and time.monotonic() - t_term > 10:
                for q in alive:                                    # a rank that ignored SIGTERM
                    if procs[q].poll() is None:
                        procs[q].kill()
    fw.join(timeout=10)
    return rc
