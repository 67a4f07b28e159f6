"""GPU, two processes on one device, gloo: the complete N > 1 path with the real HIP kernels — ingest,
bit-packed transport rows, all-gather, unpack, pipelined edges on alternating streams; sharded GF-ICF
with the all-reduce of gene counts; the sharded exact kNN search chained into a sharded Jaccard build.  (RCCL itself needs one GPU per rank; its call pattern is covered
by test_rccl_single_rank_collectives_and_bench_launch_path.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gficf_amd
        from gficf_amd import synth
        from gficf_amd.dist import GficfShard, JaccardShard, shard_bounds

        ops = gficf_amd.HipOps(0)
        N, k = 30011, 30
        mats = [synth.knn_windowed(N, k, seed=s) for s in (1, 2, 3)]
        b, e = shard_bounds(N, world, rank)
        idx = [torch.from_numpy(np.ascontiguousarray(m[b:e].T)).cuda() for m in mats]
        sh = JaccardShard(ops, N, k, device="cuda", pipeline=True, packed_transport=True)
        assert sh.packed is not None and sh.pw < sh.row_words
        res = []
        for rep in range(2):
            for i in range(3):
                out = sh.step(idx[i])
                sh.wait()
                res.append((i, out.clone()))
                sh.release()
        sh.sync()
        torch.cuda.synchronize()
        for n, (i, r) in enumerate(res):
            np.save(os.path.join(outdir, f"j_{rank}_{n}.npy"), r.cpu().numpy())
        # the halo form of the exchange (only the rows the block names) on device tables, ordered ids: same bits as the all-gather form
        mo = synth.knn_windowed(N, k, seed=9, perm_seed=None)
        io = torch.from_numpy(np.ascontiguousarray(mo[b:e].T)).cuda()
        sa = JaccardShard(ops, N, k, device="cuda", exchange="allgather")
        shh = JaccardShard(ops, N, k, device="cuda", exchange="halo")
        oa, oh = sa.step(io).clone(), shh.step(io).clone()
        ops.sync()
        assert torch.equal(oa, oh) and 0 < shh.rows_received < 500 < sa.rows_received
        np.save(os.path.join(outdir, f"halo_{rank}.npy"), oh.cpu().numpy())
        # the local-id form (fixed-capacity request slots, compact rows for the sub-problem): the same bits again
        from gficf_amd.dist import JaccardHaloShard

        sl = JaccardHaloShard(ops, N, k, device="cuda")
        ol = sl.step(io).clone()
        sl.sync()
        assert torch.equal(ol, oa) and 0 < sl.rows_named_outside() < 500
        # exact kNN sharded by the same cell blocks, its index block chained into a sharded Jaccard build
        from gficf_amd.dist import KnnShard

        Nk, dk, kk = 5003, 50, 16
        X = np.random.default_rng(23).normal(size=(Nk, dk)) * np.linspace(0.5, 3.0, dk)
        kb, ke = shard_bounds(Nk, world, rank)
        ks = KnnShard(ops, Nk, dk, kk, "manhattan", device="cuda")
        kidx = ks.step(torch.from_numpy(np.ascontiguousarray(X[kb:ke].T)).cuda())
        sh2 = JaccardShard(ops, Nk, kk - 1, device="cuda")
        kj = sh2.step(kidx[1:].contiguous())
        ops.sync()
        np.save(os.path.join(outdir, f"knn_{rank}.npy"), kidx.cpu().numpy())
        np.save(os.path.join(outdir, f"kjac_{rank}.npy"), kj.cpu().numpy())
        # GF-ICF
        G, Nc = 1200, 901
        cp, ri, x = synth.counts_csc(G, Nc, seed=5)
        cb, ce = shard_bounds(Nc, world, rank)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        gs = GficfShard(ops, G, Nc, ce - cb, int(cp[ce] - cp[cb]), device="cuda")
        ws = gs.step(d((cp[cb:ce + 1] - cp[cb]).astype(np.int64)), d(ri[cp[cb]:cp[ce]]), d(x[cp[cb]:cp[ce]]), 0.05, 1.0)
        ops.sync()
        n = int(ws["out_colptr"][ce - cb])
        np.savez(os.path.join(outdir, f"g_{rank}.npz"), rowidx=ws["out_rowidx"][:n].cpu().numpy(), x=ws["out_x"][:n].cpu().numpy(),
                 keep=ws["keep"].cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_full_sharded_path(tmp_path):
    import torch.multiprocessing as mp

    import oracle
    from gficf_amd import synth

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    N, k = 30011, 30
    want = [oracle.jaccard(synth.knn_windowed(N, k, seed=s), nthreads=8)[0] for s in (1, 2, 3)]
    for n in range(6):
        got = np.concatenate([np.load(tmp_path / f"j_{r}_{n}.npy") for r in range(world)], axis=1).T
        assert np.array_equal(got, want[n % 3]), n
    hwant = oracle.jaccard(synth.knn_windowed(N, k, seed=9, perm_seed=None), nthreads=8)[0]
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"halo_{r}.npy") for r in range(world)], axis=1).T, hwant)
    Nk, dk, kk = 5003, 50, 16
    X = np.random.default_rng(23).normal(size=(Nk, dk)) * np.linspace(0.5, 3.0, dk)
    widx, _ = oracle.knn(X, kk, "manhattan", nthreads=8)
    kgot = np.concatenate([np.load(tmp_path / f"knn_{r}.npy") for r in range(world)], axis=1).T
    assert np.array_equal(kgot, widx)
    kwant, _ = oracle.jaccard(np.ascontiguousarray(widx[:, 1:]), nthreads=8)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"kjac_{r}.npy") for r in range(world)], axis=1).T, kwant)
    G, Nc = 1200, 901
    cp, ri, x = synth.counts_csc(G, Nc, seed=5)
    ref = oracle.gficf_csc(G, Nc, cp, ri, x, 0.05, 1.0)
    parts = [np.load(tmp_path / f"g_{r}.npz") for r in range(world)]
    assert all(np.array_equal(p["keep"].astype(bool), ref["keep"]) for p in parts)
    assert np.array_equal(np.concatenate([p["rowidx"] for p in parts]), ref["rowidx"])
    assert np.allclose(np.concatenate([p["x"] for p in parts]), ref["x"], rtol=1e-6, atol=1e-6)


def test_plain_bench_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run (how the driver's N = 1 command looks with another N): the
    script spawns one rank per GPU itself before anything touches a device.  Here the two ranks share the box's one GPU
    over gloo (--rehearse-one-gpu); the line must be the N = 2 line, with the exchange's byte counts."""
    import json
    import subprocess

    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--cells-per-gpu", "8000"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    # N > 1 (round 5): the line is printed as soon as `value` exists and again, cumulative, after every further leg — every line a
    # whole record (a run cut short leaves a parseable last line behind), the last one the most complete
    recs = [json.loads(l) for l in lines]
    assert len(recs) >= 1 and all(x["value"] == recs[0]["value"] and x["checked_vs_oracle"] is True and "roofline" in x and "exchange" in x for x in recs)
    assert all(len(a["legs_done"]) <= len(b["legs_done"]) for a, b in zip(recs, recs[1:])) and recs[0]["legs_done"] == ["value"]
    out = recs[-1]
    assert out["skipped_legs"] == [] and out["leg_seconds"]["value"] > 0 and out["budget_s"] == 420.0
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["metric"] == "jaccard_edges_per_sec" and out["value"] > 0
    assert out["config"]["cells_total"] == 16000
    ex = out["exchange"]
    assert ex["rows_received_per_rank_per_data_set"] == 8000 and ex["bytes_received_per_rank_per_data_set"] > 0
    assert ex["backend"] == "gloo" and "backend_note" not in ex           # (a rehearsal asks for gloo; a FALLBACK to it would carry the note)


def test_bench_line_measures_the_edge_kernels_traffic_in_the_same_run():
    """`roofline.traffic` of the N = 1 line comes from two `rocprofv3 --pmc` passes bench.py runs itself after the timed region
    (a child process launching the same kernel on the same workload), not from a committed file: the line says so, and the bytes
    lie between the algorithmic 28 B/edge and a few times that (a gathered row fills a whole 128 B line)."""
    import json
    import subprocess

    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-gficf",
                        "--no-knn", "--batch", "2"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    rf = out["roofline"]
    if not rf["traffic_source"].startswith("live"):
        why = str((rf.get("traffic_detail") or {}).get("live_failed"))
        # a box on which the profiler itself cannot run (absent, refused, hung) is not a defect of the library: the line then
        # carries the committed figure and says why; anything else (a counter missing, a kernel not found) is ours
        if any(t in why for t in ("not found", "exited", "timed out", "under a profiler")):
            import pytest

            pytest.skip("rocprofv3 passes not available here: " + why)
        raise AssertionError((rf["traffic_source"], why))
    alg = rf["algorithmic_bytes_per_launch"]
    assert alg == 28 * 100000 * 30
    assert 1.2 * alg < rf["traffic"] < 6 * alg, rf
    d = rf["traffic_detail"]
    assert d["launches_profiled"] >= 4 and d["read_requests"]["128B"] > 10 * (d["read_requests"]["64B"] + d["read_requests"]["32B"])


def _ordered_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gficf_amd
        from gficf_amd.dist import JaccardHaloShard, KnnShard, edges_to_original_ids, shard_bounds

        ops = gficf_amd.HipOps(0)
        N, d, kk = 40000, 20, 16
        X = _blobs(N, d)
        b, e = shard_bounds(N, world, rank)
        ks = KnnShard(ops, N, d, kk, "euclidean", device="cuda")
        xl = torch.from_numpy(np.ascontiguousarray(X[b:e].T)).cuda()
        # cells in the order they came (clusters scattered): the block names nearly every remote row -> the slots overflow
        idx_plain = ks.step(xl).clone()
        hs = JaccardHaloShard(ops, N, kk - 1, device="cuda")
        hs.step(idx_plain[1:].contiguous())
        over = False
        try:
            hs.sync()
        except gficf_amd.GficfError as ex:
            over = ex.status == "GFICF_ERR_CAPACITY"
        named_plain = hs.rows_named_outside()
        # cells renumbered in the search's pivot order: few rows named outside the block, the halo form fits
        idx_ord, order = ks.step_ordered(xl)
        hs2 = JaccardHaloShard(ops, N, kk - 1, device="cuda")
        out = hs2.step(idx_ord[1:].contiguous()).clone()
        hs2.sync()
        named_ord = hs2.rows_named_outside()
        edges_to_original_ids(out, order)
        np.save(os.path.join(outdir, f"ord_edges_{rank}.npy"), out.cpu().numpy())
        np.save(os.path.join(outdir, f"ord_cells_{rank}.npy"), order[b:e].cpu().numpy())
        np.save(os.path.join(outdir, f"ord_meta_{rank}.npy"), np.array([int(over), named_plain, named_ord, hs2.cap]))
    finally:
        dist.destroy_process_group()


def _blobs(N, d):
    rng = np.random.default_rng(77)
    centers = rng.normal(scale=8.0, size=(25, d))
    lab = rng.integers(0, 25, size=N)                       # clusters scattered over the cell order, as in real input
    return centers[lab] + rng.normal(size=(N, d)) * rng.uniform(0.6, 1.5, size=(25, 1))[lab]


@pytest.mark.parametrize("world", [2, 3])
def test_knn_chain_in_pivot_order_feeds_the_halo_form(tmp_path, world):
    """kNN -> Jaccard sharded over two ranks on clustered points that arrive in no particular order (R/clustCells.R:57-65).
    In the given numbering a block names nearly every remote row (request slots overflow: the all-gather form is the one to
    take); with the cells renumbered in the search's pivot order (KnnShard.step_ordered) the halo form fits, names a small
    fraction of the remote rows, and — mapped back to the original ids — gives the oracle's edges bit for bit."""
    import torch.multiprocessing as mp

    import oracle

    mp.spawn(_ordered_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    N, d, kk = 40000, 20, 16
    X = _blobs(N, d)
    widx, _ = oracle.knn(X, kk, "euclidean", nthreads=8)
    want, _ = oracle.jaccard(np.ascontiguousarray(widx[:, 1:]), nthreads=8)
    k = kk - 1
    got = np.zeros_like(want)
    for r in range(world):
        cells = np.load(tmp_path / f"ord_cells_{r}.npy").astype(np.int64)          # original 0-based ids of the rank's cells, in its order
        edges = np.load(tmp_path / f"ord_edges_{r}.npy").T.reshape(len(cells), k, 3)
        got.reshape(N, k, 3)[cells] = edges
        over, named_plain, named_ord, cap = np.load(tmp_path / f"ord_meta_{r}.npy")
        assert over == 1 and named_plain == (world - 1) * cap                    # scattered clusters: every slot of the other owners in use
        assert 0 < named_ord < (0.25 if world == 2 else 0.5) * (N // world)
    assert np.array_equal(got, want)


def test_launcher_counts_the_gpus_of_this_box_from_sysfs_and_need_gpus_runs():
    """gficf_amd.launch on hardware: the sysfs count agrees with the HIP runtime's (asked of a child process, so that this
    process' launcher code path stays runtime-free), and the branch the driver's `bench.py --gpus N` takes — spawn_ranks with
    need_gpus — has executed on a GPU box: one rank, one GPU, the real bench line."""
    import json
    import subprocess

    from gficf_amd import launch

    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
    assert launch.kfd_gpu_nodes() is not None, "no KFD topology in sysfs on a GPU box"
    assert launch.visible_gpus() == int(r.stdout.strip().splitlines()[-1]) >= 1
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    drv = ("import sys; sys.path.insert(0, %r)\nfrom gficf_amd import launch\n"
           "sys.exit(launch.spawn_ranks(sys.argv[1:], 1, need_gpus=1, timeout_s=250))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", drv, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras",
                        "--cells-per-gpu", "20000", "--pre-warm-ms", "0"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["checked_vs_oracle"] is True
    # more GPUs asked for than the box has: exit code 2 before anything is started
    r = subprocess.run([sys.executable, "-c", drv.replace("need_gpus=1", "need_gpus=64"), os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 2 and "64 GPUs asked for" in r.stderr


def test_plain_multi_gpu_bench_line_carries_the_whole_scaling_answer():
    """One `python bench.py --gpus 2` line (rehearsed on the one GPU of the box: numbers meaningless, structure and byte counts
    real): value + exchange, pipelined, the other id model in both modes, the single-GPU step timed in the same run, the
    single-process peer-copy leg, and the efficiencies computed from them."""
    import json
    import subprocess

    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "2", "--warmup", "1",
                        "--cells-per-gpu", "8000", "--no-gficf", "--no-chain"], capture_output=True, text=True, timeout=580, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["n_gpus"] == 2 and out["checked_vs_oracle"] is True and out["config"]["cells_total"] == 16000
    assert out["pipelined"]["edges_per_sec"] > 0
    sp = out["spatial_ids"]
    assert sp["exchange"] == "halo" and 0 < sp["rows_named_outside"] <= 400 and sp["checked_vs_oracle"] is True
    assert sp["in_order"]["edges_per_sec"] > 0 and sp["overlapped"]["edges_per_sec"] > 0
    assert out["single_gpu_step"]["cells"] == 8000 and out["single_gpu_step"]["edges_per_sec"] > 0
    pe = out["peer"]
    assert "error" not in pe, pe
    assert pe["checked_vs_oracle"] is True and pe["devices"] == [0, 0] and pe["edges_per_sec"] > 0
    eff = out["efficiency"]
    for key in ("in_order_permuted", "overlapped_permuted", "in_order_spatial", "overlapped_spatial", "peer_in_order_permuted"):
        assert eff[key] > 0, key
    # round 6: the model's projection beside every measured figure — ONE run on the node then says whether the projections hold; in a
    # rehearsal (the ranks share one GPU) the measured side is labelled for what it is
    forms = eff["forms"]
    for name in ("value", "pipelined", "spatial_ids", "spatial_ids_pipelined", "peer"):
        f = forms[name]
        assert 0 < f["projected"] <= 1.0 and f["measured"] > 0 and abs(f["residual"] - (f["measured"] - f["projected"])) < 1e-3, (name, f)
        assert f["measured_is"].startswith("time-sliced, meaningless"), f
    assert forms["value"]["projected"] < forms["pipelined"]["projected"] <= 1.0          # hiding the exchange can only help the model
    assert eff["model"]["link_GB_per_s_per_direction"] == 55.0 and eff["model"]["single_gpu_data_set_us"] > 0
    ra = out["exchange"]["rccl_algo"]                                                  # gloo in a rehearsal: RCCL ran nothing, and the line says so
    assert ra["reported"] is False and "gloo" in ra["why"]


def _bench_env():
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    return env


def test_bench_wall_budget_skips_and_names_the_legs_it_has_no_time_for():
    """`--budget-s`: a leg that would start with less than its reserve left is skipped on every rank alike and named in
    `skipped_legs`; `value` always runs; the run ends with code 0 and a whole line.  `--legs` selects."""
    import json
    import subprocess

    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "1", "--warmup", "0", "--cells-per-gpu", "6000", "--pre-warm-ms", "0",
            "--no-gficf"]
    r = subprocess.run(base + ["--budget-s", "5"], capture_output=True, text=True, timeout=280, env=_bench_env())
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["legs_done"] == ["value"] and out["checked_vs_oracle"] is True and out["value"] > 0
    assert out["skipped_legs"] == ["pipelined", "other_ids", "single_gpu_step", "chain", "peer"] and out["budget_s"] == 5.0
    r = subprocess.run(base + ["--legs", "single_gpu_step,pipelined"], capture_output=True, text=True, timeout=280, env=_bench_env())
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["legs_done"] == ["value", "pipelined", "single_gpu_step"] and out["skipped_legs"] == []
    assert out["efficiency"]["in_order_permuted"] > 0 and "peer" not in out and "chain" not in out


def test_bench_killed_at_any_moment_after_its_first_line_leaves_a_whole_last_line():
    """the first real `bench.py --gpus N` must be impossible to lose — the cumulative line is printed as soon as
    `value` exists and after every leg, so killing the whole job (SIGKILL to its process group: no handler runs) after the first print
    leaves stdout ending in a parseable record."""
    import json
    import signal
    import subprocess
    import time

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "1", "--warmup", "0", "--cells-per-gpu", "6000", "--pre-warm-ms", "0",
           "--no-gficf"]
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=_bench_env(), start_new_session=True)
    try:
        first = pr.stdout.readline()                            # blocks until the line behind `value` is out
        assert first.lstrip().startswith("{"), first[:200]
        time.sleep(1.7)                                         # ... somewhere inside a later leg
    finally:
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except OSError:
            pass
    rest = pr.stdout.read()
    pr.wait(timeout=30)
    lines = [l for l in (first + rest).split("\n") if l.strip()]
    whole = [l for l in lines if l.rstrip().endswith("}")]
    rec = json.loads(whole[-1])
    assert rec["value"] > 0 and rec["checked_vs_oracle"] is True and "roofline" in rec and "exchange" in rec and rec["legs_done"][0] == "value"


def test_bench_under_torch_distributed_run_with_two_ranks():
    """The driver's documented N > 1 launch — `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` — with two ranks
    (sharing the box's one GPU over gloo): no self-spawn, the rank environment comes from the launcher; rank 0's cumulative lines are the
    only JSON on stdout and the last one is complete."""
    import json
    import subprocess

    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "1", "--warmup", "0",
                        "--cells-per-gpu", "6000", "--pre-warm-ms", "0", "--no-gficf", "--legs", "single_gpu_step"], capture_output=True, text=True, timeout=400, env=_bench_env())
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(recs) >= 2 and recs[-1]["legs_done"] == ["value", "single_gpu_step"] and recs[-1]["n_gpus"] == 2
    assert recs[-1]["checked_vs_oracle"] is True and recs[-1]["efficiency"]["in_order_permuted"] > 0


def test_bench_falls_back_to_gloo_and_says_so_when_rccl_cannot_build_its_communicator():
    """First contact with RCCL happens in the driver's own run.  If the communicator cannot be built the run must still leave a line:
    here RCCL is asked for with both ranks on the box's one GPU (test hook GFICF_BENCH_RANKS_SHARE_GPU0), which it refuses — the run
    goes on over gloo and labels its line."""
    import json
    import subprocess

    env = dict(_bench_env(), GFICF_BENCH_RANKS_SHARE_GPU0="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells-per-gpu", "6000",
                        "--pre-warm-ms", "0", "--no-extras"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.lstrip().startswith("{")][-1])
    ex = out["exchange"]
    assert ex["backend"] == "gloo" and "RCCL init failed" in ex["backend_note"] and out["checked_vs_oracle"] is True and out["n_gpus"] == 2
