"""Multi-GPU form of the hot path: one process per GPU, cells sharded by contiguous block.

New design (the reference is single-process):

* Jaccard — the edges of cell i need row i and the k rows it names, which may live on any
  rank, so there is exactly one exchange step: every rank transposes its own block of the
  kNN matrix into table rows, then ONE all-gather (RCCL over xGMI) replicates the
  int32 table, then every rank builds the edges of its own cell block.  No other
  collective touches the data path; outputs stay sharded (each rank owns rows
  [b*k, e*k) of the reference's edge matrix).
* kNN (the step in front, "next" row N2) — queries shard by the same cell blocks; every rank prepares
  its own block of point rows (f32, row-major), ONE all-gather replicates the points (N x dpad x 4 B:
  20 MB at 100 k x 50), then every rank searches its own queries against all points.  The index
  block it writes is exactly the input block of the sharded Jaccard build, already in place.
* GF-ICF — cells (columns) are independent except for the per-gene cell counts nt_g, so
  there is exactly one all-reduce(sum) of G int64 counters between the counting pass and
  the scaling pass.

``ops`` is the object that runs the stage kernels (``gficf_amd.api.HipOps`` in the product
path).  It is a parameter only so that the collective / sharding logic can be exercised on
CPU with the gloo backend and a test double; there is no fallback here.
"""
from __future__ import annotations

import collections

import torch
import torch.distributed as dist


def genes_words(G: int) -> int:
    from .api import genes_words as _gw

    return _gw(G)


def rows_per_rank(n: int, world: int) -> int:
    return (n + world - 1) // world


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, equal-pitch cell blocks: rank r owns [r*rpr, min(n, (r+1)*rpr))."""
    rpr = rows_per_rank(n, world)
    b = min(n, rank * rpr)
    return b, min(n, b + rpr)


def shard_bounds_by_nnz(colptr, world: int) -> list[tuple[int, int]]:
    """Contiguous cell (column) blocks of a CSC matrix balanced by stored entries rather than by cell count
    (the GF-ICF passes stream nnz, so that is what a rank's time follows).  ``colptr``: the
    N+1 column pointers (any integer array-like).  Returns ``world`` (begin, end) pairs covering [0, N); block r
    ends at the first cell boundary at or after r+1 equal shares of the entries."""
    import numpy as np

    cp = np.asarray(colptr, dtype=np.int64)
    n = len(cp) - 1
    nnz = int(cp[-1] - cp[0]) if n > 0 else 0
    cuts = [0]
    for r in range(1, world):
        target = cp[0] + (nnz * r + world - 1) // world
        c = int(np.searchsorted(cp, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


_FLAT_ALL_GATHER: dict = {}          # backend name -> does all_gather_into_tensor exist there (decided once, not per call)


def _all_gather_rows(table, local_view, group):
    """All-gather equal-sized row blocks into ``table`` (``local_view`` aliases this rank's block).

    The form of the collective is a property of the backend, decided once: a communication failure is never
    answered with a different collective (the other ranks would not issue it) — it propagates."""
    backend = dist.get_backend(group)
    flat = _FLAT_ALL_GATHER.get(backend)
    if flat is None:
        if backend == "nccl":                              # RCCL: in-place flat all-gather
            flat = True
        else:
            # gloo grew all_gather_into_tensor over time: probe on a throw-away tensor, every rank alike
            world = dist.get_world_size(group)
            probe_in = torch.zeros(1, dtype=torch.int32, device=table.device)
            probe_out = torch.zeros(world, dtype=torch.int32, device=table.device)
            try:
                dist.all_gather_into_tensor(probe_out, probe_in, group=group)
                flat = True
            except (NotImplementedError, RuntimeError) as ex:
                msg = str(ex).lower()
                if isinstance(ex, NotImplementedError) or "not supported" in msg or "not implemented" in msg or "no backend" in msg:
                    flat = False
                else:
                    raise
        _FLAT_ALL_GATHER[backend] = flat
    if flat:
        dist.all_gather_into_tensor(table, local_view, group=group)
    else:
        world = dist.get_world_size(group)
        chunks = list(table.view(world, -1).unbind(0))
        dist.all_gather(chunks, local_view.reshape(-1).clone(), group=group)


def _all_to_all(out, inp, out_split, in_split, group):
    """all_to_all_single; device tensors under the gloo backend (test set-ups: ranks sharing one GPU) go through the host."""
    if out.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_split, input_split_sizes=in_split, group=group)


class JaccardShard:
    """Per-rank state of the sharded Jaccard build (buffers allocated once, reused per step).

    ``pipeline=True`` (GPU only): steps are software-pipelined over two tables and two output
    buffers.  The ingest (and, for N > 1, pack + all-gather + unpack) of a step runs on a side stream,
    so it overlaps the edge kernel of the step before it, which still reads the other table; the edge
    kernels of consecutive steps run on two alternating streams, so the tail of one overlaps the ramp
    of the next.  Every step does the same work and yields the same bits; only stream placement
    changes.  Contract in this mode: the input block handed to :meth:`step` is read in stream order behind
    the caller's current stream (the side stream waits for it); the returned buffer is one of two and is
    overwritten by the step after next; call :meth:`wait` to make the caller's current stream wait for the
    latest step before reading it, and :meth:`release` once it has been read (the step that reuses the buffer
    then waits for that point instead of racing the reader).

    With ONE rank there is no exchange to hide, and what is left to overlap (a 6 us ingest under the tail of a 41 us edge kernel
    that already fills every CU) is worth less than the stream hops cost: measured 50.8 G edges/s against 63.4 in order at
    100 k x 30 (profiles/r04_bench.json; the pipelined step is ~10 runtime calls and host-bound).  So ``pipeline=True`` on one rank
    RUNS IN ORDER on the caller's stream (``pipeline_in_order`` is set); ``pipeline="force"`` keeps the two-stream machinery (tests of
    that machinery on a one-GPU box).  The CONTRACT above still holds in that mode: the result alternates between two buffers and is
    overwritten only by the step after next, every step records an event that :meth:`wait` makes another consumer stream wait for, and
    the step that reuses a buffer waits for the point :meth:`release` marked — a caller written against the pipelined contract reads the
    same data whether the steps overlap or not.
    """

    def __init__(self, ops, N_total: int, k: int, group=None, device=None, with_u: bool = False,
                 pipeline: bool = False, packed_transport: bool = True, time_edges: bool = False,
                 exchange: str = "allgather"):
        if exchange not in ("allgather", "halo"):
            raise ValueError("exchange must be 'allgather' or 'halo'")
        self.exchange = exchange
        self.bytes_received = 0                 # table-row bytes this rank received in the last step (both forms)
        self.rows_received = 0
        self.ops, self.N, self.k, self.group = ops, int(N_total), int(k), group
        self.time_edges = bool(time_edges)      # HIP events around every edge-kernel launch, on its launch stream
        self.edge_events = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.kpad = ops.kpad(k)
        self.row_words = ops.row_words(self.N, self.k)      # row pitch of the table (the library's choice for this N, k)
        _format_agreed(ops, self.N, self.k, group, device)  # N > 1, once per (N, k): every rank lays the rows out alike, or every rank raises
        self.rpr = rows_per_rank(self.N, self.world)
        self.b, self.e = shard_bounds(self.N, self.world, self.rank)
        self.n_local = self.e - self.b
        self.pipeline = bool(pipeline) and device is not None and torch.device(device).type == "cuda"
        self.pipeline_in_order = self.pipeline and self.world == 1 and pipeline != "force"
        if self.pipeline_in_order:
            self.pipeline = False
        nbuf = 2 if self.pipeline else 1
        nout = 2 if (self.pipeline or self.pipeline_in_order) else 1      # (in order: one table is enough — nothing reads it behind the step —, two results)
        # full table(s), padded to world*rpr rows so that every rank contributes an equal block
        self.tables = [torch.zeros((self.world * self.rpr, self.row_words), dtype=torch.int32, device=device) for _ in range(nbuf)]
        self.outs = [torch.zeros((3, self.n_local * self.k), dtype=torch.float64, device=device) for _ in range(nout)]
        self.us = [torch.zeros(self.n_local * self.k, dtype=torch.int32, device=device) if with_u else None for _ in range(nout)]
        self.table, self.out, self.u = self.tables[0], self.outs[0], self.us[0]
        self.t = 0
        # one rank: prepared single-call steps of the LAST FEW input blocks (see step).  A prepared call keeps its input block alive, so the
        # cache is small: a streaming caller that hands over a fresh block every step (KnnShard output) must not pin 64 of them (7.7 GB at
        # 1 M x 30), nor keep the caching allocator from reusing them.
        self._prepared = collections.OrderedDict()
        if self.pipeline_in_order:
            self.ev_step = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_consumed = [None, None]
            self.last_done = None
            self.last_p = None
        # N > 1: rows travel bit-packed (ceil(log2(N+1)) bits per id) and are unpacked after the all-gather
        self.packed = None
        if self.world > 1 and packed_transport and exchange == "allgather":
            self.pw = ops.packed_words(self.N, self.k)
            if self.pw < self.row_words:                     # (compact rows of a small data set may already be as short)
                self.packed = torch.zeros((self.world * self.rpr, self.pw), dtype=torch.int32, device=device)
        if self.pipeline:
            self.side = torch.cuda.Stream(device=device)
            self.edge_streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
            self.ev_table_ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_edges_done = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_consumed = [None, None]      # recorded by release(): the reader of outs[p] has got past it
            self.last_done = None
            self.last_p = None

    def _fill_table_halo(self, table, idx_local_cm):
        """Exchange form for inputs whose ids have locality (cells numbered in a spatial order — e.g. the pivot order of
        the device kNN search): a rank fetches only the remote rows its own block names.  Request lists (sorted unique ids
        per owner) go out in one all-to-all, the rows come back in a second; the split sizes need one host round trip.
        With scrambled ids nearly every row is named and the all-gather is the better form (same result either way)."""
        my_rows = table[self.rank * self.rpr:(self.rank + 1) * self.rpr]
        if self.n_local > 0:
            self.ops.jaccard_ingest(idx_local_cm, self.n_local, self.k, self.N, my_rows)
        if self.world == 1:
            return
        ids = idx_local_cm.reshape(-1)
        if ids.dtype != torch.int64:
            ids = ids.to(torch.int64)
        ids = ids[(ids >= 1) & (ids <= self.N)]                         # (invalid ids are reported by the ingest)
        need = torch.unique(ids[(ids <= self.b) | (ids > self.e)])      # sorted: grouped by owner, ascending
        owner = torch.div(need - 1, self.rpr, rounding_mode="floor")
        send_counts = torch.bincount(owner, minlength=self.world).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        _all_to_all(recv_counts, send_counts, None, None, self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()             # the one host round trip
        req_in = torch.empty(sum(rc), dtype=torch.int64, device=need.device)
        _all_to_all(req_in, need, rc, sc, self.group)
        rw = table.shape[1]
        rows_out = table.index_select(0, req_in - 1).contiguous()       # the requested rows of my block
        rows_in = torch.empty((len(need), rw), dtype=table.dtype, device=table.device)
        _all_to_all(rows_in.view(-1), rows_out.view(-1), [c * rw for c in sc], [c * rw for c in rc], self.group)
        table.index_copy_(0, need - 1, rows_in)
        self.rows_received = int(len(need))
        self.bytes_received = int(len(need)) * rw * 4 + sum(rc) * 8

    def _fill_table(self, table, idx_local_cm):
        if self.exchange == "halo":
            return self._fill_table_halo(table, idx_local_cm)
        my_rows = table[self.rank * self.rpr:(self.rank + 1) * self.rpr]
        if self.world > 1:
            wire = self.pw if self.packed is not None else self.row_words
            self.rows_received = self.N - self.n_local
            self.bytes_received = self.rows_received * wire * 4
        if self.n_local > 0:
            self.ops.jaccard_ingest(idx_local_cm, self.n_local, self.k, self.N, my_rows)
        if self.world > 1 and self.packed is not None:
            mine = self.packed[self.rank * self.rpr:(self.rank + 1) * self.rpr]
            if self.n_local > 0:
                self.ops.jaccard_pack_rows(my_rows, self.n_local, self.k, self.N, mine)
            _all_gather_rows(self.packed.view(-1), mine.reshape(-1), self.group)
            # every other rank's block (the own block is already in place, unpacked): the blocks are equal-pitch and contiguous in
            # both buffers, so the rows in front of the own block and the rows behind it are ONE launch each (a launch per
            # remote block cost 7 x 3 us of launch overhead at 8 ranks)
            for lo, hi in ((0, self.b), (self.e, self.N)):
                if hi > lo:
                    self.ops.jaccard_unpack_rows(self.packed[lo:hi], hi - lo, self.k, self.N, table[lo:hi])
        elif self.world > 1:
            _all_gather_rows(table.view(-1), my_rows.reshape(-1), self.group)

    def step(self, idx_local_cm):
        """idx_local_cm: (k, n_local) tensor == column-major n_local x k block of the kNN matrix
        (global 1-based ids).  Returns this rank's (3, n_local*k) slice of the edge matrix (valid in
        stream order on the caller's current stream)."""
        if not self.pipeline:
            p, cur = 0, None
            if self.pipeline_in_order:
                # the pipelined contract, in order: two result buffers in turn; a buffer is reused only behind the point its reader marked
                p = self.t & 1
                cur = torch.cuda.current_stream(self.outs[p].device)
                if self.ev_consumed[p] is not None:
                    cur.wait_event(self.ev_consumed[p])
                    self.ev_consumed[p] = None
            out, u = self.outs[p], self.us[p]
            if self.world == 1 and self.exchange == "allgather" and not self.time_edges and self.n_local > 0:
                # one rank: nothing to exchange — the library's single-device sequence in ONE call (for a small problem under
                # set_jaccard_distinct that is one launch), its arguments converted once per input block and result buffer
                key = (idx_local_cm.data_ptr(), tuple(idx_local_cm.shape), tuple(idx_local_cm.stride()), idx_local_cm.dtype, p)
                run = self._prepared.get(key)
                if run is None:
                    while len(self._prepared) >= 4:
                        self._prepared.popitem(last=False)
                    run = self._prepared[key] = self.ops.jaccard_prepared(idx_local_cm, self.N, self.k, self.table, out, u)
                else:
                    self._prepared.move_to_end(key)
                run()
            else:
                self._fill_table(self.table, idx_local_cm)
                if self.time_edges:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if self.n_local > 0:
                    self.ops.jaccard_edges(self.table, self.N, self.k, self.b, self.e, out, u)
                if self.time_edges:
                    e1.record()
                    self.edge_events.append((e0, e1))
            self.out, self.u = out, u
            if self.pipeline_in_order:
                self.ev_step[p].record(cur)
                self.last_done, self.last_p = self.ev_step[p], p
                self.t += 1
            return self.out
        p = self.t & 1
        table = self.tables[p]
        # the input block may have been produced just before this call on the caller's stream (e.g. by KnnShard.step)
        self.side.wait_stream(torch.cuda.current_stream(self.out.device))
        if self.t >= 2:
            self.side.wait_event(self.ev_edges_done[p])      # edges of step t-2 have finished reading this table
        with torch.cuda.stream(self.side):
            self._fill_table(table, idx_local_cm)
            self.ev_table_ready[p].record(self.side)
        es = self.edge_streams[p]                            # in order behind step t-2, which wrote the same buffers
        es.wait_event(self.ev_table_ready[p])
        if self.ev_consumed[p] is not None:                  # ... and behind whoever read what step t-2 wrote there
            es.wait_event(self.ev_consumed[p])
            self.ev_consumed[p] = None
        with torch.cuda.stream(es):
            if self.time_edges:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(es)
            if self.n_local > 0:
                self.ops.jaccard_edges(table, self.N, self.k, self.b, self.e, self.outs[p], self.us[p])
            if self.time_edges:
                e1.record(es)
                self.edge_events.append((e0, e1))
            self.ev_edges_done[p].record(es)
        self.table, self.out, self.u = table, self.outs[p], self.us[p]
        self.last_done = self.ev_edges_done[p]
        self.last_p = p
        self.t += 1
        return self.out

    def edge_kernel_ms(self, last: int | None = None) -> float:
        """Mean duration of the (last ``last``) timed edge-kernel launches; call after a device sync."""
        ev = self.edge_events if last is None else self.edge_events[-last:]
        return sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)

    def wait(self):
        """Make the caller's current stream wait for the latest step (pipelined mode, also when it runs in order; no-op otherwise)."""
        if (self.pipeline or self.pipeline_in_order) and self.last_done is not None:
            torch.cuda.current_stream(self.out.device).wait_event(self.last_done)

    def release(self):
        """Pipelined mode: the caller's current stream has read the latest returned buffer up to this point; the step
        that overwrites that buffer (the one after next) waits for it.  No-op otherwise."""
        if (self.pipeline or self.pipeline_in_order) and self.last_p is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.out.device))
            self.ev_consumed[self.last_p] = ev

    def sync(self):
        """Wait for all streams and surface deferred input-validation errors."""
        if self.pipeline:
            self.side.synchronize()
            for es in self.edge_streams:
                es.synchronize()
        self.ops.sync()


# deferred statuses a sharded step can end in, by severity (the collective sync raises the most severe one on every rank)
_STATUS_SEVERITY = {"GFICF_ERR_CAPACITY": 1, "GFICF_ERR_SET_OVERFLOW": 2, "GFICF_ERR_DUPLICATE_IDS": 2, "GFICF_ERR_BAD_ID": 3, "GFICF_ERR_HIP": 4}


class JaccardHaloShard:
    """The sharded Jaccard build on LOCAL ids (csrc/halo.hip): a rank's sub-problem is its own block of cells plus the remote
    rows the block names, renumbered 1..n_ext — with n_ext < 2^17 it runs on the compact 64 B-row table and the fast edge
    kernel whatever N_total is (the all-gather form takes 128 B rows from 2^17 cells on), and only the named rows travel.

    Exchange: two all-to-alls with EQUAL, host-known splits (``cap`` request slots per owner: ids out, raw index rows back) —
    no count exchange and no host round trip inside a step.  ``cap`` defaults to what keeps n_ext below 2^17 (at most 8192).
    A block that names more than ``cap`` rows of one owner (ids without locality) makes the next :meth:`sync` raise
    ``GFICF_ERR_CAPACITY``; the caller then builds a :class:`JaccardShard` (all-gather) for that input — the choice is a
    property of the data (bench.py makes it on the warm-up step).  Same interface and the same bits as JaccardShard.
    Input blocks must be int32 (what the kNN search and uwot produce).

    ``pipeline=True`` (GPU only): everything in front of the edge kernel — plan, the two all-to-alls, serve, relabel,
    ingest — runs on a side stream over two sets of buffers, so the exchange of a step overlaps the edge kernel of the step
    before it (same contract as JaccardShard's pipelined mode: :meth:`wait` before reading the returned buffer,
    :meth:`release` after; the input block is read in stream order behind the caller's current stream)."""

    def __init__(self, ops, N_total: int, k: int, group=None, device=None, with_u: bool = False, cap: int | None = None,
                 time_edges: bool = False, pipeline: bool = False):
        self.ops, self.N, self.k, self.group = ops, int(N_total), int(k), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rpr = rows_per_rank(self.N, self.world)
        self.b, self.e = shard_bounds(self.N, self.world, self.rank)
        self.n_local = self.e - self.b
        if cap is None:
            room = (1 << 17) - 1 - self.rpr                      # rows left below 2^17 next to the largest block
            cap = max(64, min(8192, room // self.world)) if room >= 64 * self.world else 1024
        self.cap = int(cap)
        self.n_ext = self.n_local + self.world * self.cap
        self.exchange = "halo"
        self.time_edges = bool(time_edges)
        self.edge_events = []
        self.pipeline = bool(pipeline) and device is not None and torch.device(device).type == "cuda"
        nbuf = 2 if self.pipeline else 1
        i32 = dict(dtype=torch.int32, device=device)
        self.row_words = ops.row_words(self.n_ext, self.k)
        # (a rank's sub-problem table is private — only raw index rows travel — but the request slots, the block pitch and the library
        # build must agree: compared on the size of the largest sub-problem, which is the same number on every rank)
        _format_agreed(ops, self.rpr + self.world * self.cap, self.k, group, device, extra=(self.cap, self.N))
        ws_bytes = ops.halo_workspace_bytes(self.N, self.world)
        self.bufs = [dict(ws=torch.zeros(ws_bytes, dtype=torch.uint8, device=device),
                          req_out=torch.zeros(self.world * self.cap, **i32), req_in=torch.zeros(self.world * self.cap, **i32),
                          rows_out=torch.zeros(self.world * self.cap * self.k, **i32), rows_in=torch.zeros(self.world * self.cap * self.k, **i32),
                          idx_ext=torch.zeros((self.k, max(self.n_ext, 1)), **i32), l2g=torch.zeros(max(self.n_ext, 1), **i32),
                          table=torch.zeros((max(self.n_ext, 1), self.row_words), **i32),
                          out=torch.zeros((3, self.n_local * self.k), dtype=torch.float64, device=device),
                          u=torch.zeros(self.n_local * self.k, **i32) if with_u else None) for _ in range(nbuf)]
        self._use(0)
        self.t = 0
        self.packed = None
        # what a step moves: the request slots and the reply slots of the other ranks (fixed), and what of it is used
        self.bytes_received = (self.world - 1) * self.cap * 4 * (1 + self.k)
        self.rows_received = 0
        if self.pipeline:
            self.side = torch.cuda.Stream(device=device)
            self.edge_stream = torch.cuda.Stream(device=device)
            self.ev_ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_done = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_consumed = [None, None]
            self.last_done = None
            self.last_p = None

    def _use(self, p):
        bf = self.bufs[p]
        self.ws, self.req_out, self.req_in, self.rows_out, self.rows_in = bf["ws"], bf["req_out"], bf["req_in"], bf["rows_out"], bf["rows_in"]
        self.idx_ext, self.l2g, self.table, self.out, self.u = bf["idx_ext"], bf["l2g"], bf["table"], bf["out"], bf["u"]

    def _front(self, bf, idx_local_cm):
        """Everything in front of the edge kernel, on the current stream."""
        o, P, k, nl = self.ops, self.world, self.k, self.n_local
        o.halo_plan(idx_local_cm, nl, k, self.N, self.b, P, self.rpr, self.cap, bf["ws"], bf["req_out"])
        if P > 1:
            _all_to_all(bf["req_in"], bf["req_out"], None, None, self.group)
        else:
            bf["req_in"].copy_(bf["req_out"])
        split = getattr(o, "halo_serve_ingest", None)     # k <= 64: serve + the own cells' table rows in ONE launch between the exchanges
        if split is not None and split(idx_local_cm, nl, k, self.N, self.b, P, self.rpr, self.cap, bf["ws"], bf["req_out"], bf["req_in"], bf["rows_out"],
                                       bf["table"], bf["l2g"]):
            if P > 1:
                _all_to_all(bf["rows_in"], bf["rows_out"], None, None, self.group)
            else:
                bf["rows_in"].copy_(bf["rows_out"])
            # ... and only the halo slots in use (a few hundred rows) behind the second one
            o.halo_ingest_slots(idx_local_cm, nl, k, self.N, self.b, P, self.rpr, self.cap, bf["ws"], bf["req_out"], bf["rows_in"], bf["table"], bf["l2g"])
            return
        o.halo_serve(idx_local_cm, nl, k, self.b, bf["req_in"], bf["rows_out"])
        if P > 1:
            _all_to_all(bf["rows_in"], bf["rows_out"], None, None, self.group)
        else:
            bf["rows_in"].copy_(bf["rows_out"])
        # k > 64 (or an ops double without the fused launches): global ids -> local ids, then the plain ingest of the sub-problem
        o.halo_relabel(idx_local_cm, nl, k, self.N, self.b, P, self.rpr, self.cap, bf["ws"], bf["req_out"], bf["rows_in"], bf["idx_ext"], bf["l2g"])
        o.jaccard_ingest_local(bf["idx_ext"], self.n_ext, k, bf["table"])

    def _edges(self, bf, stream=None):
        if self.time_edges:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream) if stream is not None else e0.record()
        if self.n_local > 0:
            self.ops.jaccard_edges_mapped(bf["table"], self.n_ext, self.k, self.n_local, self.b, bf["l2g"], bf["out"], bf["u"])
        if self.time_edges:
            e1.record(stream) if stream is not None else e1.record()
            self.edge_events.append((e0, e1))

    def step(self, idx_local_cm):
        """idx_local_cm: (k, n_local) int32 == column-major block of the kNN matrix, global 1-based ids.  Returns this rank's
        (3, n_local*k) slice of the edge matrix."""
        if not self.pipeline:
            self._front(self.bufs[0], idx_local_cm)
            self._edges(self.bufs[0])
            return self.out
        p = self.t & 1
        bf = self.bufs[p]
        self.side.wait_stream(torch.cuda.current_stream(bf["out"].device))      # the input block may have just been produced there
        if self.t >= 2:
            self.side.wait_event(self.ev_done[p])                               # the edges of step t-2 have finished with these buffers
        with torch.cuda.stream(self.side):
            self._front(bf, idx_local_cm)
            self.ev_ready[p].record(self.side)
        es = self.edge_stream                                                   # one edge stream: steps stay in order
        es.wait_event(self.ev_ready[p])
        if self.ev_consumed[p] is not None:
            es.wait_event(self.ev_consumed[p])
            self.ev_consumed[p] = None
        with torch.cuda.stream(es):
            self._edges(bf, es)
            self.ev_done[p].record(es)
        self._use(p)
        self.last_done, self.last_p = self.ev_done[p], p
        self.t += 1
        return self.out

    def rows_named_outside(self) -> int:
        """Rows this rank's block named outside itself in the last step (a device read: not for a timed loop)."""
        self.rows_received = int((self.req_out != 0).sum().item())
        return self.rows_received

    def edge_kernel_ms(self, last: int | None = None) -> float:
        ev = self.edge_events if last is None else self.edge_events[-last:]
        return sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)

    def wait(self):
        """Make the caller's current stream wait for the latest step (pipelined mode, also when it runs in order; no-op otherwise)."""
        if (self.pipeline or self.pipeline_in_order) and self.last_done is not None:
            torch.cuda.current_stream(self.out.device).wait_event(self.last_done)

    def release(self):
        """Pipelined mode: the caller's current stream has read the latest returned buffer up to this point."""
        if (self.pipeline or self.pipeline_in_order) and self.last_p is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.out.device))
            self.ev_consumed[self.last_p] = ev

    def sync(self, collective: bool = False):
        """Wait for the streams; raises GficfError(GFICF_ERR_CAPACITY) when a step overflowed the request slots (or
        GFICF_ERR_DUPLICATE_IDS in distinct mode).

        A deferred error is raised on the rank whose block overflowed (or owns the offending row) ONLY.  A caller that answers
        it by switching forms — building a :class:`JaccardShard` — must do so on EVERY rank or the ranks issue different
        collectives and the job hangs.  ``collective=True`` makes that agreement here: every rank calls it (it is a
        collective), the ranks all-reduce their status and every rank raises the SAME error — the most severe status any rank
        saw, with the number of the first rank that saw it — or none does."""
        if self.pipeline:
            self.side.synchronize()
            self.edge_stream.synchronize()
        if not collective or self.world == 1:
            self.ops.sync()
            return
        from ._lib import STATUS_NAMES, GficfError

        # Every rank joins the all-reduce whatever its own sync did (an exception that skipped it would leave the others waiting),
        # and the reduced word carries the NUMERIC status (severity, status code, rank), so a status outside the severity table
        # arrives as itself.
        err, other, sev, num = None, None, 0, 0
        try:
            self.ops.sync()
        except GficfError as ex:
            err, sev, num = ex, _STATUS_SEVERITY.get(ex.status, 5), int(ex.code)
        except BaseException as ex:                               # noqa: BLE001  (re-raised below, behind the collective)
            other, sev, num = ex, 6, 5                            # counted as a HIP-level failure on the other ranks
        dev = self.out.device if dist.get_backend(self.group) != "gloo" else torch.device("cpu")
        t = torch.tensor([(sev * 256 + num) * 4096 + (4095 - self.rank) if sev else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        if other is not None:
            raise other
        top = int(t.item())
        if top == 0:
            return
        who, word = 4095 - top % 4096, top // 4096
        top_sev, top_num = word // 256, word % 256
        if err is not None and sev == top_sev and num == top_num:
            raise err
        status = STATUS_NAMES.get(top_num, f"status {top_num}")
        raise GficfError(top_num, f"rank {who} reported {status} in the sharded Jaccard step (this rank's own block was fine); every rank fails alike")


_AGREED = set()


def _format_agreed(ops, N_total, k, group, device, extra=()):
    """assert_same_format, once per (N, k, extra) and process (every rank constructs its shards in the same order, so the
    all-gathers match up)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    key = (int(N_total), int(k), tuple(int(v) for v in extra), id(group))
    if key in _AGREED:
        return
    assert_same_format(ops, N_total, k, group=group, device=device, extra=extra)
    _AGREED.add(key)


def assert_same_format(ops, N_total: int, k: int, group=None, device=None, extra: tuple = ()):
    """Every rank of a sharded Jaccard build must lay table rows out alike: the row pitch is a function of (N, k) AND of the
    library build (ABI version) and its GFICF_JACCARD_* switches, which are read from each rank's own environment.  One
    all-gather of a few words, once, before the first step; raises on EVERY rank (they all see the same gathered words) when
    two ranks differ.  ``extra``: further integers the caller wants agreed on (e.g. the halo form's capacity)."""
    import os
    import zlib

    from . import _lib

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return
    L = _lib.load()
    env = "|".join(f"{k_}={v}" for k_, v in sorted(os.environ.items()) if k_.startswith("GFICF_JACCARD_") or k_ in ("GFICF_BITS_WAVES", "GFICF_BITS_DEPTH"))
    words = [int(L.gficf_hip_abi_version()), int(ops.row_words(N_total, k)), int(ops.kpad(k)), int(ops.packed_words(N_total, k)),
             zlib.crc32(env.encode()), int(N_total), int(k)] + [int(v) for v in extra]
    gloo = dist.get_backend(group) == "gloo"
    if not gloo and device is None:                               # (RCCL moves device tensors only)
        device = torch.device("cuda", torch.cuda.current_device())
    mine = torch.tensor(words, dtype=torch.int64, device="cpu" if gloo else device)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    rows = [tuple(int(v) for v in t.tolist()) for t in got]
    if any(r != rows[0] for r in rows):
        names = ["abi_version", "row_words", "kpad", "packed_words", "crc32(GFICF_JACCARD_* environment)", "N_total", "k"] + [f"extra[{i}]" for i in range(len(extra))]
        diff = [n for i, n in enumerate(names) if any(r[i] != rows[0][i] for r in rows)]
        raise RuntimeError(f"the ranks of this sharded Jaccard build do not agree on the table format: {diff} differ across ranks "
                           f"(per rank: {rows}); start every rank with the same library and the same GFICF_JACCARD_* environment")


class GficfShard:
    """Per-rank state of the sharded GF-ICF normalisation (local CSC block of cells)."""

    def __init__(self, ops, G: int, N_total: int, n_local: int, nnz_local: int, group=None, device=None):
        self.ops, self.G, self.N, self.n_local, self.group = ops, int(G), int(N_total), int(n_local), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = device
        self.ws = dict(
            nt=torch.zeros(max(G, 1), dtype=torch.int64, device=dev),
            keep=torch.zeros(max(G, 1), dtype=torch.uint8, device=dev),
            genes=torch.zeros(genes_words(G), dtype=torch.float64, device=dev),   # opaque per-gene tables
            w=torch.zeros(max(G, 1), dtype=torch.float64, device=dev),
            gkept=torch.zeros(1, dtype=torch.int64, device=dev),
            out_colptr=torch.zeros(n_local + 1, dtype=torch.int64, device=dev),
            out_rowidx=torch.zeros(max(nnz_local, 1), dtype=torch.int32, device=dev),
            out_x=torch.zeros(max(nnz_local, 1), dtype=torch.float64, device=dev),
        )

    def step(self, colptr, rowidx, x, prop_min=0.05, prop_max=1.0, w_in=None):
        """colptr (int64, n_local+1, starting at 0), rowidx, x: this rank's block of cells."""
        ws, ops = self.ws, self.ops
        ws["nt"].zero_()
        ops.csc_count(self.G, self.n_local, colptr, rowidx, x, ws["nt"])
        if self.world > 1:
            dist.all_reduce(ws["nt"], op=dist.ReduceOp.SUM, group=self.group)
        ops.csc_genes(self.G, self.N, ws["nt"], prop_min, prop_max, w_in, ws["keep"], ws["genes"], ws["w"], ws["gkept"])
        ops.csc_colptr(self.G, self.n_local, colptr, rowidx, ws["keep"], ws["gkept"], ws["out_colptr"])
        ops.csc_scale(self.G, self.n_local, colptr, rowidx, x, ws["genes"], ws["gkept"], ws["out_colptr"],
                      ws["out_rowidx"], ws["out_x"])
        return ws


class KnnShard:
    """Per-rank state of the sharded exact kNN search (reference call-site R/clustCells.R:57,60).

    ``step(X_local_cm)`` takes this rank's block of the cells x components matrix ((d, n_local) tensor ==
    column-major n_local x d) and returns the (k, n_local) int32 index block (1-based global ids, column 0 =
    the cell itself) — ``idx[1:]`` is what :meth:`JaccardShard.step` takes (``neigh[,-1]``, R/clustCells.R:63).
    """

    def __init__(self, ops, N_total: int, d: int, k: int, metric: str = "manhattan", group=None, device=None,
                 with_dist: bool = False):
        self.ops, self.N, self.d, self.k, self.metric, self.group = ops, int(N_total), int(d), int(k), metric, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rpr = rows_per_rank(self.N, self.world)
        self.b, self.e = shard_bounds(self.N, self.world, self.rank)
        self.n_local = self.e - self.b
        self.dpad = ops.knn_dpad(self.d)
        # all points, padded to world*rpr rows so that every rank contributes an equal block
        self.points = torch.zeros((self.world * self.rpr, self.dpad), dtype=torch.float32, device=device)
        self.idx = torch.zeros((self.k, max(self.n_local, 1)), dtype=torch.int32, device=device)
        self.dist = torch.zeros((self.k, max(self.n_local, 1)), dtype=torch.float32, device=device) if with_dist else None
        self.ws = torch.zeros(max(ops.knn_workspace_bytes(self.n_local, self.N, self.k), 16), dtype=torch.uint8, device=device)

    def step(self, X_local_cm):
        mine = self.points[self.rank * self.rpr:(self.rank + 1) * self.rpr]
        if self.n_local > 0:
            self.ops.knn_prepare(X_local_cm, self.n_local, self.d, self.metric, mine)
        if self.world > 1:
            _all_gather_rows(self.points.view(-1), mine.reshape(-1), self.group)
        if self.n_local > 0:
            # blocks are equal-pitch and only the last can be short, so rows [0, N) of the table are the N points
            self.ops.knn_search(self.points, self.N, self.d, self.k, self.metric, self.b, self.e, self.ws, self.idx, self.dist)
        return self.idx

    def step_ordered(self, X_local_cm):
        """The search with the cells RENUMBERED in the pruned search's pivot order (R/clustCells.R:57-65: find_nn -> neigh[,-1] ->
        the Jaccard build): every rank derives the same order from the all-gathered points (``gficf_knn_pivot_order_device``), lays
        the points out in it and searches the cells at positions [b, e) of the order.  Returns ``(idx, order)``: ``idx`` (k, n_local)
        int32 holds 1-based ids IN THE NEW NUMBERING (column 0 = the cell itself) — ids with locality: a block names few rows
        outside itself, which is what :class:`JaccardHaloShard` wants; ``order`` (N int32, the same on every rank): ``order[p]`` =
        original 0-based cell at position p, so this rank's cells are ``order[b:e]`` and :func:`edges_to_original_ids` maps a
        block of edges back.  The neighbour lists are those of :meth:`step` (ties between equal distances are broken by the
        position in the order instead of the original index)."""
        mine = self.points[self.rank * self.rpr:(self.rank + 1) * self.rpr]
        if self.n_local > 0:
            self.ops.knn_prepare(X_local_cm, self.n_local, self.d, self.metric, mine)
        if self.world > 1:
            _all_gather_rows(self.points.view(-1), mine.reshape(-1), self.group)
        if not hasattr(self, "order"):
            dev = self.points.device
            self.order = torch.zeros(max(self.N, 1), dtype=torch.int32, device=dev)
            self.ws_order = torch.zeros(max(self.ops.knn_workspace_bytes(self.N, self.N, 1), 16), dtype=torch.uint8, device=dev)
        self.ops.knn_pivot_order(self.points, self.N, self.d, self.metric, self.ws_order, self.order)
        self.points_sorted = self.points[:self.N].index_select(0, self.order[:self.N].long())
        if self.n_local > 0:
            self.ops.knn_search(self.points_sorted, self.N, self.d, self.k, self.metric, self.b, self.e, self.ws, self.idx, self.dist)
        return self.idx, self.order[:self.N]


def edges_to_original_ids(out3, order):
    """Columns 1 and 2 (from, to) of a block of edges built on renumbered cells (:meth:`KnnShard.step_ordered`) back in the
    original 1-based ids: id -> order[id - 1] + 1, zero rows stay zero.  out3: (3, n) float64, changed in place."""
    o1 = torch.cat([torch.zeros(1, dtype=torch.float64, device=out3.device), order.to(torch.float64) + 1.0])
    out3[0] = o1[out3[0].long()]
    out3[1] = o1[out3[1].long()]
    return out3
