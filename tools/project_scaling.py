#!/usr/bin/env python3
"""Per-GPU efficiency of the N > 1 Jaccard step, projected from what can be measured on one GPU: the per-rank stage times of
tools/halo_stage_times.py (kernels, HIP events) and the bytes each exchange form moves, plus a stated model of the collectives.
Nothing here is a measurement of a multi-GPU run; the driver's SCALE run is.  Usage: project_scaling.py stage_times.jsonl [single_gpu_ms]

Model (constants below): an all-gather moves every rank's block over each of its xGMI links once -> time = block bytes / LINK_GBS
+ LAT_US; an all-to-all with equal splits of s bytes per peer -> s / LINK_GBS + LAT_US (one link per peer); kernels of one data
set run in order on one stream with GAP_US between dependent launches (non-pipelined: what bench.py reports as `value`);
pipelined = max(sum of the kernel times, exchange) (exchange of the next data set under the edge kernel of this one; `--pipeline`,
never `value`).  Third form, round 4: "peer" — the single-process step of the C ABI (gficf_multi_jaccard_device): ingest, then
every device pulls the other P - 1 UNPACKED table slices with hipMemcpyPeerAsync (one slice per link, all pairs at once), then
edges on the full table; no collective, so the fixed cost of the exchange is a copy's start-up (PEER_LAT_US) instead of LAT_US,
but the rows travel as the table holds them (128 B from 2^17 cells on) and no pack / unpack kernels run.  Fourth form: "peer halo"
(gficf_multi_jaccard_halo_device) — the halo step's kernels with NOTHING exchanged: the launch that ingests the halo slots reads
the rows named outside where they lie, in the owners' blocks of ids, through the peer mapping (a few hundred rows of k ids: one
dependent remote read per lane, PEER_READ_US on top of the kernel); no collective, no copy, no event between devices.  Its
"overlapped" column is two such contexts taking the data sets in turn (the front end of one under the edge kernel of the other)."""
import json
import sys

LINK_GBS = 55.0      # achieved per link and direction in RCCL collectives (xGMI, 76.8 GB/s nominal per direction); VERDICT r2 item 2 uses 55
LAT_US = 20.0        # launch + protocol latency of one small RCCL collective
GAP_US = 2.0         # between dependent kernel launches on one stream
PEER_READ_US = 3.0   # a dependent read over xGMI inside a kernel (the peer halo form's slots launch)
PEER_LAT_US = 6.0    # start-up of a peer copy behind an event (SDMA engine); the host's enqueue of a step runs ahead on its own threads

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
one = next(r for r in rows if r["P"] == 1)
base_ms = float(sys.argv[2]) if len(sys.argv) > 2 else one["allgather_permuted"]["compute_ms"] + GAP_US * 1e-3
print("single GPU, per data set: %.1f us (ingest + edges + one launch gap; the bench's own figure may be passed as argv[2])" % (base_ms * 1e3))
print()
hdr = "| P | ids | exchange | rows named outside | bytes received / rank | table row | compute (kernels) | exchange (model) | step, in order | efficiency | step, overlapped | efficiency |"
print(hdr)
print("|" + "---|" * (hdr.count("|") - 1))
for r in rows:
    P = r["P"]
    if P == 1:
        continue
    for ids in ("spatial", "permuted"):
        for form in ("allgather", "peer", "halo", "peer halo"):
            d = r.get(f"{form}_{ids}") if form != "peer halo" else r.get(f"halo_{ids}")
            if form == "peer halo":
                if not d or not d.get("fits") or "peer_ingest_us" not in d:
                    continue
                d = {"plan_ms": d["plan_ms"], "table_ms": d["peer_ingest_us"] * 1e-3, "edges_ms": d["edges_ms"], "table_row_bytes": d["table_row_bytes"],
                     "rows_named_outside": d["rows_named_outside"], "bytes_received": 0, "fits": True}
                d["compute_ms"] = d["plan_ms"] + d["table_ms"] + d["edges_ms"]
            if form == "peer":
                ag = r.get(f"allgather_{ids}")
                if not ag:
                    continue
                n_rem = ag["bytes_received"] // ag["wire_bytes_per_row"]        # rows of the other ranks
                d = {"ingest_ms": ag["ingest_ms"], "edges_ms": ag["edges_ms"], "table_row_bytes": ag["table_row_bytes"],
                     "bytes_received": n_rem * ag["table_row_bytes"]}
                d["compute_ms"] = d["ingest_ms"] + d["edges_ms"]
            if not d or (form == "halo" and not d.get("fits")):
                if d is not None and form == "halo":
                    print(f"| {P} | {ids} | halo | {d['rows_named_outside']}+ | - | - | - | request slots overflow: falls back to the all-gather | | | | |")
                continue
            n_k = sum(1 for kk in d if kk.endswith("_ms") and kk not in ("compute_ms", "chain_ms"))
            comp = d["compute_ms"] * 1e3 + GAP_US * n_k
            if form == "allgather":
                ex = d["bytes_received"] / (P - 1) / (LINK_GBS * 1e3) + LAT_US           # one block per link
                named = "all"
            elif form == "peer":
                ex = d["bytes_received"] / (P - 1) / (LINK_GBS * 1e3) + PEER_LAT_US      # one unpacked slice per link, pulled side by side
                named = "all"
            elif form == "peer halo":
                ex = PEER_READ_US
                named = str(d["rows_named_outside"])
            else:
                per_peer_req, per_peer_rows = d["cap"] * 4, d["cap"] * 4 * r["k"]
                ex = per_peer_req / (LINK_GBS * 1e3) + LAT_US + per_peer_rows / (LINK_GBS * 1e3) + LAT_US
                named = str(d["rows_named_outside"])
            step = comp + ex
            over = max(d["compute_ms"] * 1e3, ex)                 # overlapped: the launch gaps are filled by the other stream
            if form == "peer halo":                               # two contexts in turn: the longer of the edge kernel and the front end (+ the remote read)
                front = d["compute_ms"] * 1e3 - d["edges_ms"] * 1e3 + ex
                over = max(d["edges_ms"] * 1e3 + GAP_US, front)
            recv = d["bytes_received"] if form != "peer halo" else d["rows_named_outside"] * 4 * r["k"]
            print(f"| {P} | {ids} | {form} | {named} | {recv / 1e6:.2f} MB | {d['table_row_bytes']} B | {comp:.0f} us | {ex:.0f} us | "
                  f"{step:.0f} us | {base_ms * 1e3 / step:.2f} | {over:.0f} us | {base_ms * 1e3 / over:.2f} |")
print()
print("(Both efficiency columns divide the single GPU's IN-ORDER step by the step of the form; a form that overlaps data sets can exceed 1 "
      "against it — the single GPU overlapping its own data sets runs a step in about %.0f us, its edge kernel plus a launch gap.)"
      % (one["allgather_permuted"]["edges_ms"] * 1e3 + GAP_US))
