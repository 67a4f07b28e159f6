"""CPU: the C-ABI library loads and exports every symbol include/gficf_hip.h declares
(no compute calls — there is no GPU here), and fails loudly without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

import gficf_amd
from gficf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "gficf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gficf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    L = _lib.load()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in gficf_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_kpad():
    L = _lib.load()
    assert L.gficf_hip_abi_version() == 7
    assert [L.gficf_jaccard_kpad(k) for k in (0, 1, 15, 16, 17, 30, 32, 33, 50, 64, 65, 128, 129, 256)] == \
        [16, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 256, 256]
    # k > 256: "sorted" rows (slot-order ids + the same ids ascending, each half padded to 64): kpad = row pitch = 2 * ceil64(k)
    assert [L.gficf_jaccard_kpad(k) for k in (257, 300, 320, 321, 513, 65535)] == [640, 640, 640, 768, 1152, 131072]
    assert L.gficf_jaccard_kpad(65536) == -1 and L.gficf_jaccard_kpad(-1) == -1
    # row pitch of the table: half the slots for data sets of fewer than 2^17 cells when k leaves room for the bitmap
    rw = L.gficf_jaccard_row_words
    # (32 < k <= 55 below 131071 cells: dual rows — the compact row and a planar copy for the bit-set edge kernel: 64 words)
    assert [rw(100000, k) for k in (15, 16, 17, 30, 31, 32, 33, 50, 55, 56, 60, 61, 100, 120, 121, 240, 241, 256)] == \
        [16, 16, 16, 16, 32, 32, 64, 64, 64, 32, 32, 64, 64, 64, 128, 128, 256, 256]
    assert rw(131070, 50) == 64 and rw(131071, 50) == 32
    assert rw(131071, 30) == 16 and rw(131072, 30) == 32 and rw(1000000, 30) == 32 and rw(1000000, 50) == 64
    assert rw(-1, 30) == -1 and rw(100, 257) == 640 and rw(10**6, 513) == 1152 and rw(100, 65536) == -1
    assert L.gficf_jaccard_packed_words(5000, 300) == 640           # sorted rows travel as they are


def test_exported_surface_is_the_header_and_the_binder_table_is_current():
    """ABI 7 retired five entries (an accidental export, an internal helper of gficf_phenograph_host, two superseded calls, an unused
    query): what the library exports under the gficf_ prefix is exactly what the header declares, and the header's table of who binds
    what (tools/abi_binders.py) is regenerated whenever a binding changes."""
    import shutil
    import subprocess
    import sys

    names = header_functions()
    if shutil.which("nm"):
        out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = sorted(set(re.findall(r" T (gficf_[a-z0-9_]+)$", out, re.M)))
        assert exported == names, set(exported) ^ set(names)
    assert len(names) == 85
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_binders.py"), "--check"])
    assert r.returncode == 0, "include/gficf_hip.h: the BINDERS table is stale — run python tools/abi_binders.py --write"


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_every_jaccard_kernel_header_compiles_on_its_own():
    """Round 6: the kernel headers of jaccard.hip include what they use (jaccard_shared.h) and open their own namespace; through round 5
    they compiled only inside jaccard.hip's anonymous namespace, in one order."""
    import subprocess

    csrc = os.path.join(ROOT, "gficf_amd", "csrc")
    for h in ("jaccard_shared", "jaccard_ingest", "jaccard_edges_general", "jaccard_edges_pipe", "jaccard_edges_bits", "jaccard_sorted", "jaccard_direct"):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", "-x", "hip",
                            os.path.join(csrc, h + ".h")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, h + ".h:\n" + r.stderr[-2000:]


def test_status_enum_matches_header():
    src = open(os.path.join(ROOT, "include", "gficf_hip.h")).read()
    for code, name in _lib.STATUS_NAMES.items():
        assert re.search(rf"\b{name}\s*=\s*{code}\b", src), name


@pytest.mark.skipif(gficf_amd.device_count() > 0, reason="GPU present")
def test_no_gpu_fails_loudly_no_fallback():
    with pytest.raises(gficf_amd.GficfError) as ei:
        gficf_amd.Context(0)
    assert ei.value.status == "GFICF_ERR_NO_DEVICE"
    with pytest.raises(gficf_amd.GficfError):
        gficf_amd.rcpp_parallel_jaccard_coef(np.array([[2], [1]], dtype=np.int32), False)


def test_null_ctx_is_rejected_not_crashing():
    L = _lib.load()
    rc = L.gficf_ctx_sync(None)
    assert rc == 1 and b"ctx" in L.gficf_last_error()
    rc = L.gficf_jaccard_host(None, None, 0, 10, 3, 10, None, 0)
    assert rc == 1


def test_kept_values_host_is_the_row_subset_of_the_counts():
    """gficf_csc_kept_values_host (no device involved): the values — and, on request, the renumbered row ids — of `M[keep, ]` (normCounts,
    reference R/gficf.R:40; $rawCounts :22) given its column pointer, against scipy's row subsetting: int32 and int64 pointers, empty cells,
    explicitly stored zeros (they stay), enough entries for several host threads; a column pointer that is not M[keep, ]'s is refused."""
    import ctypes

    import scipy.sparse as sp

    L = _lib.load()
    rng = np.random.default_rng(5)
    for G, N, dens, pt in ((50, 40, 0.3, np.int32), (3000, 9000, 0.2, np.int64), (1, 1, 1.0, np.int32), (7, 0, 0.5, np.int32), (0, 5, 0.5, np.int32)):
        M = sp.random(G, N, density=dens, format="csc", random_state=rng, dtype=np.float64)
        M.data = np.ceil(M.data * 9)
        M.data[::53] = 0.0
        if N > 10:
            M = sp.hstack([M[:, :5], sp.csc_matrix((G, 3)), M[:, 5:], sp.csc_matrix((G, 2))], format="csc")      # empty cells, the last two included
        G, N = M.shape
        keep = (rng.random(G) < 0.6).astype(np.uint8)
        if N > 10:                                        # ... and cells that hold dropped genes only: the last stored ones, one in the middle
            M = M.tolil()
            for c in (N - 4, N - 3, 7):
                M[:, c] = 0
                M[np.flatnonzero(keep == 0)[:3], c] = 2.0
            M = M.tocsc()
        want = M[np.flatnonzero(keep), :]
        cp, ri, x = M.indptr.astype(pt), M.indices.astype(np.int32), np.ascontiguousarray(M.data)
        kcp = want.indptr.astype(pt)
        for with_ids in (True, False):
            oi = np.full(want.nnz, -7, dtype=np.int32)
            ox = np.full(want.nnz, np.nan)
            rc = L.gficf_csc_kept_values_host(G, N, cp.ctypes.data, int(pt is np.int64), ri.ctypes.data, x.ctypes.data, keep.ctypes.data, kcp.ctypes.data,
                                              oi.ctypes.data if with_ids else None, ox.ctypes.data)
            assert rc == 0, L.gficf_last_error()
            assert np.array_equal(ox, want.data) and (not with_ids or np.array_equal(oi, want.indices))
        none = np.zeros(G, dtype=np.uint8)                # nothing kept: nothing is written (the output vector may be empty)
        zcp = np.zeros(N + 1, dtype=pt)
        guard = np.full(4, 7.0)
        assert L.gficf_csc_kept_values_host(G, N, cp.ctypes.data, int(pt is np.int64), ri.ctypes.data, x.ctypes.data, none.ctypes.data, zcp.ctypes.data, None, guard.ctypes.data) == 0
        assert (guard == 7.0).all()
        if want.nnz:
            bad = kcp.copy()                              # still monotone, but one cell's slot is an entry short (its neighbour's one too long)
            d = np.diff(kcp)
            bad[1 + int(np.flatnonzero(d > 0)[0])] -= 1
            ox = np.zeros(want.nnz)
            rc = L.gficf_csc_kept_values_host(G, N, cp.ctypes.data, int(pt is np.int64), ri.ctypes.data, x.ctypes.data, keep.ctypes.data, bad.ctypes.data, None, ox.ctypes.data)
            assert _lib.STATUS_NAMES[rc] == "GFICF_ERR_BAD_CSC"
            assert b"kept_colptr" in L.gficf_last_error()
            # a pointer that does not start at 0 (either of the two) is refused before anything is read or written (advisor, round 5)
            for shifted_in, shifted_out in ((cp + 1, kcp), (cp, kcp + 1), (cp, kcp - 1)):
                ox = np.full(want.nnz, 7.0)
                rc = L.gficf_csc_kept_values_host(G, N, shifted_in.ctypes.data, int(pt is np.int64), ri.ctypes.data, x.ctypes.data, keep.ctypes.data,
                                                  shifted_out.ctypes.data, None, ox.ctypes.data)
                assert _lib.STATUS_NAMES[rc] == "GFICF_ERR_BAD_CSC" and (ox == 7.0).all()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "gficf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f


def test_committed_bench_line_keeps_the_contract():
    """profiles/r02_bench.json is the line bench.py printed on the GPU box: the keys the driver and the judge read are there,
    the roofline and the CPU baseline objects included, and the numbers are self-consistent."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_bench.json")
    d = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["data"] == "synthetic"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    edges = d["config"]["edges_per_step"]
    assert abs(d["value"] - edges / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["checked_vs_oracle"] is True and d["gficf"]["checked_vs_oracle"] is True and d["knn"]["checked_vs_oracle"] is True


def test_round4_records_are_consistent_and_results_md_is_generated_from_them():
    """The committed round-4 line (profiles/r04_bench.json) carries what round 4 added — `value_from_idle`, the GF-ICF pass in the
    pointerB / pointerE form equal to the canonical result — and RESULTS.md is what tools/make_results.py makes of profiles/."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
    assert d["checked_vs_oracle"] is True and 0 < d["value_from_idle"] < d["value"]
    be = d["gficf"]["begin_end_form"]
    assert be["equals_canonical_result"] is True and be["transpose_equals_canonical"] is True
    assert be["ms_per_pass"] < d["gficf"]["ms_per_pass"] and be["roofline_frac"] > d["gficf"]["roofline"]["frac"]
    for cfg, kern in (("c4", "k_jaccard_edges_bits"), ("c5", "k_jaccard_edges_pipe")):
        c = json.load(open(os.path.join(root, "profiles", f"r04_bench_{cfg}.json")))
        assert c["checked_vs_oracle"] is True and c["roofline"]["kernel"] == kern and c["scaling"] == "strong"
    for n in (2, 3):
        r = json.load(open(os.path.join(root, "profiles", f"r04_rehearsal_gpus{n}.json")))
        assert r["n_gpus"] == n and r["checked_vs_oracle"] and r["spatial_ids"]["checked_vs_oracle"] and r["peer"]["checked_vs_oracle"] and r["chain"]["checked_vs_oracle"]
        assert set(r["efficiency"]) >= {"in_order_permuted", "overlapped_permuted", "in_order_spatial", "overlapped_spatial", "peer_in_order_permuted"}
    # round 5: the generator checks a round's records against each other — and round 4's do NOT hold together (its GF-ICF figure was the
    # best of four boxes next to another call's trace; its per-launch event pairs overstated the kernels): the generator says so and fails
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "make_results.py"), "r04"], capture_output=True, text=True, timeout=120)
    assert gen.returncode == 1 and gen.stdout == "" and "gficf.ms_per_pass 0.4650 ms differs from the sum of its kernels' trace averages 0.5372 ms" in gen.stderr
    assert "does not fit in ms_per_step" in gen.stderr


def test_the_rounds_records_hold_together_and_results_md_is_generated_from_them():
    """every number of RESULTS.md comes from records that agree with each other — the bench line and the rocprofv3
    trace of the SAME gpurun call: the edge kernel's in-run time within 5 % of the trace's average, data sets x (kernel + ingest) inside
    ms_per_step, the GF-ICF pass within 5 % of the sum of its kernels' trace averages (tools/make_results.py exits 1 otherwise) — and the
    headline is the MEDIAN over the round's runs, the detailed record the kept run NEAREST the medians (every kept run
    carries its own trace).  RESULTS.md is what the generator makes of profiles/.  Round 6's asks that are figures of the line: the graph
    hand-off timed by itself, the Louvain legs as medians with their spread, ten starts run together."""
    import json
    import os
    import statistics
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    P = os.path.join(root, "profiles")
    d = json.load(open(os.path.join(P, "r06_bench.json")))
    r = d["roofline"]
    assert d["checked_vs_oracle"] is True and d["legs_done"][0] == "value" and d["skipped_legs"] == [] and set(d["leg_seconds"]) == set(d["legs_done"])
    assert d["config"]["data_sets_per_step"] * (r["kernel_ms"] + r["ingest_kernel_ms"]) <= d["ms_per_step"] * 1.02          # (c) inside the line itself
    assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 2e-4
    assert d["pipelined"]["refused_on_one_rank"] is True and d["pipelined"]["edges_per_sec"] >= d["value"]                       # item 6
    g = d["gficf"]
    assert g["ms_per_pass"] == statistics.median(g["ms_per_pass_batches"]) or abs(g["ms_per_pass"] - statistics.median(g["ms_per_pass_batches"])) < 1e-3
    assert g["host_abi"]["checked_vs_oracle"] is True and g["host_abi"]["ms_per_call"] > g["ms_per_pass"]                         # the PCIe-inclusive figure is in the line
    # the small configs: one launch / one library call per step (item 3)
    c1 = json.load(open(os.path.join(P, "r06_bench_c1.json")))
    c2 = json.load(open(os.path.join(P, "r06_bench_c2.json")))
    assert c1["roofline"]["kernel"] == "k_jaccard_direct" and c1["ms_per_step"] <= 0.010 and c1["checked_vs_oracle"] is True and c1["steps"] >= 1000
    assert c2["ms_per_step"] <= 0.014 and c2["checked_vs_oracle"] is True
    c5 = json.load(open(os.path.join(P, "r06_bench_c5.json")))
    assert c5["spatial_ids"]["checked_vs_oracle"] is True and c5["spatial_ids"]["kernel_ms"] < 0.75 * c5["roofline"]["kernel_ms"]   # item 7
    runs = [f for f in os.listdir(os.path.join(P, "r06_runs")) if f.endswith(".json") and not f.endswith("_traced.json")]
    assert len(runs) >= 3 and all(os.path.exists(os.path.join(P, "r06_runs", f[:-5] + "_kernel_stats.csv")) for f in runs)      # any of them can be the detailed record
    kn = d["knn"]
    lv = kn["louvain"]
    assert kn["checked_vs_oracle"] is True and kn["graph_build"]["ms_jaccard_filter_adjacency"] < 0.5 < kn["graph_build"]["ms_total"]      # 0.77 in round 5
    assert lv["ms_min"] <= lv["ms"] <= lv["ms_max"] and lv["ms"] < 6.5 and lv["ten_starts"]["ms"] < 5 * lv["ms"]      # 8.5 ms a start in round 5; ten starts together
    assert abs(lv["modularity"] - lv["ten_starts"]["modularity"]) < 0.01
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "make_results.py"), "r06"], capture_output=True, text=True, timeout=120)
    assert gen.returncode == 0, gen.stderr[-2000:]
    assert gen.stdout == open(os.path.join(root, "RESULTS.md")).read(), "RESULTS.md is stale: python tools/make_results.py r06 > RESULTS.md"
