// multi.hip — the host entry points over several GPUs of one node, single process (C ABI: gficf_multi_*).
//
// What an R session can bind: the reference's call sites are one `.Call` each (R/clustCells.R:65 for the Jaccard
// build, gficf() R/gficf.R:17-33 for the normalisation), so a drop-in that shards over the GPUs of a node has to do
// it underneath that one call — one context and one stream per device, and ONE HOST THREAD PER DEVICE for every stage
// that moves host data (the caller's buffers are pageable: hipMemcpyAsync from or to pageable memory holds the calling
// thread until the copy is done, so a single thread would run the devices one after another); the threads are joined
// before the call returns and never touch the R API.  The partitioning is the one of the multi-process path (gficf_amd/dist.py):
//   * Jaccard: cells in contiguous equal-pitch blocks; every device uploads and ingests ITS block of the kNN
//     matrix, the table rows are exchanged device to device (hipMemcpyPeerAsync over xGMI, pulled by the receiving
//     device on its own stream behind an event of the sending one; with no peer access every device ingests the
//     whole matrix instead), every device builds the edges of its block and copies them straight into its three
//     column slices of the caller's (N*k) x 3 matrix.
//   * GF-ICF: cells in contiguous blocks balanced by stored entries; the per-gene cell counts nt_g are the only
//     global quantity: each device counts its block, the G counters are summed on the host (G x 8 B per device —
//     for a host entry the all-reduce is two small copies) and handed back; filter, weights and the scaling pass
//     then run per block and the kept entries land at their offsets in the caller's arrays.
// Kernels and arithmetic are those of the single-device entries: same bits.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "halo_map.h"

struct gficf_multi_block {
  int64_t b = 0, e = 0;            // cells [b, e)
  int64_t p0 = 0, p1 = 0;          // stored entries [p0, p1) (GF-ICF)
  int64_t nnz_kept = 0, out_off = 0;
  // plan state (device pointers into pool slot 4 of the block's context)
  int64_t* d_colptr = nullptr;
  int32_t* d_rowidx = nullptr;
  double* d_x = nullptr;
  double* d_w_in = nullptr;
  int64_t* d_nt = nullptr;
  uint8_t* d_keep = nullptr;
  gficf_gene_entry* d_genes = nullptr;
  double* d_w = nullptr;
  int64_t* d_gkept = nullptr;
  int64_t* d_out_colptr = nullptr;
};

struct gficf_multi_worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_job, cv_idle;
  std::deque<std::shared_ptr<const std::function<int(int)>>> q;
  std::atomic<unsigned> posted{0};
  bool busy = false, quit = false;
  int rc = GFICF_OK;               // first failure since the failures were last collected (multi_drain)
  std::string msg;
};

struct gficf_multi {
  int ndev = 0;
  std::vector<int> dev;
  std::vector<gficf_ctx*> ctx;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> ev;
  bool peer = true;                // every pair of distinct devices can access each other
  // device-resident step (gficf_multi_jaccard_device): device d pulls the other devices' table slices on P - 1 copy streams of
  // its own, all pairs at once; events order the pulls against the ingests (this step) and the edge kernels (the step before)
  std::vector<std::vector<hipStream_t>> cstream;   // [d][t - 1]: the stream of device d's t-th pull
  std::vector<std::vector<hipEvent_t>> cev;        // ... and "that pull is done"
  std::vector<hipEvent_t> ev_prev;                 // device d: everything enqueued before this step (the last step's edge kernel) is done
  std::vector<hipEvent_t> ev_pulled;               // device d: has pulled every other slice of this step
  bool step_valid = false;                         // ev_pulled holds a recorded step
  bool step_resources_ok = false;                  // the copy streams and events above exist, all of them
  // ... enqueued by one PERSISTENT host thread per device (a step is ~45 runtime calls per device — launches, peer copies, event
  // records and waits: from one thread that is 8 x 45 calls in a row, several hundred microseconds per step at 8 devices against
  // ~100 us of device work; threads made per call would cost as much).  Every thread has a queue of jobs of its own: a job is POSTED
  // to all of them and either waited for (the phases of the peer-copy step, which depend on each other across devices) or not (the
  // halo step: its devices do not depend on each other, so the caller's thread is free to post the next step while this one is
  // still being enqueued; failures are kept and come out of gficf_multi_sync).  A thread spins for a few tens of microseconds
  // before it parks on its condition variable: inside a burst of steps the next job arrives within that.
  std::vector<std::unique_ptr<gficf_multi_worker>> workers;
  // GF-ICF plan
  bool has_plan = false;
  int64_t G = 0, N = 0, g_kept = 0, nnz_kept = 0;
  int colptr_is_i64 = 0;
  std::vector<gficf_multi_block> blk;
};

namespace {

// first error of a sweep over the devices wins; the others are still synchronised
struct FirstError {
  int rc = GFICF_OK;
  char msg[768] = "";
  void note(int r) {
    if (r != GFICF_OK && rc == GFICF_OK) {
      rc = r;
      snprintf(msg, sizeof(msg), "%s", gficf_last_error());
    }
  }
  int done() {
    if (rc != GFICF_OK) gficf_set_error("%s", msg);
    return rc;
  }
};

int hip_fail(const char* what, hipError_t e) {
  gficf_set_error("%s failed: %s", what, hipGetErrorString(e));
  return GFICF_ERR_HIP;
}

// One stage of a multi-device entry: body(r) for every device r, each on its own host thread (inline for one device),
// joined before returning.  body returns a gficf_status; its message is thread-local to the worker, so the worker keeps
// a copy and the first failing device (in device order) is the one reported.  A stage is skipped once `fe` holds an error.
// A stage body may allocate (std::vector, std::function copies): no C++ exception may leave a worker thread (std::terminate
// would take the R session down) nor cross the C ABI — it becomes a status with a message.
int guarded(const std::function<int(int)>& body, int r) {
  try {
    return body(r);
  } catch (const std::bad_alloc&) {
    gficf_set_error("out of host memory in a multi-device stage (device slot %d)", r);
    return GFICF_ERR_HIP;
  } catch (const std::exception& ex) {
    gficf_set_error("C++ exception in a multi-device stage (device slot %d): %s", r, ex.what());
    return GFICF_ERR_INVALID_ARG;
  } catch (...) {
    gficf_set_error("unknown C++ exception in a multi-device stage (device slot %d)", r);
    return GFICF_ERR_INVALID_ARG;
  }
}

void for_each_device(int P, FirstError& fe, const std::function<int(int)>& body) {
  if (fe.rc != GFICF_OK) return;
  if (P == 1) { fe.note(guarded(body, 0)); return; }
  std::vector<FirstError> each;
  std::vector<std::thread> th;
  try {
    each.resize((size_t)P);
    th.reserve((size_t)P);
  } catch (...) {
    gficf_set_error("out of host memory setting up the per-device threads");
    fe.note(GFICF_ERR_HIP);
    return;
  }
  for (int r = 0; r < P; ++r) {
    try {
      th.emplace_back([&, r]() { each[(size_t)r].note(guarded(body, r)); });
    } catch (...) {                                 // no thread to be had (std::system_error must not cross the C ABI): this device's stage runs here
      each[(size_t)r].note(guarded(body, r));
    }
  }
  for (auto& t : th) t.join();
  for (int r = 0; r < P && fe.rc == GFICF_OK; ++r)
    if (each[(size_t)r].rc != GFICF_OK) { fe.rc = each[(size_t)r].rc; snprintf(fe.msg, sizeof(fe.msg), "%s", each[(size_t)r].msg); }
}

}  // namespace

extern "C" {

/* Cell blocks of the Jaccard build: block r = [r*pitch, min(N, (r+1)*pitch)), pitch = ceil(N / ndev)
 * (gficf_amd/dist.py: shard_bounds).  bounds: ndev + 1 entries. */
int gficf_multi_cell_blocks(int64_t N, int ndev, int64_t* bounds) {
  if (N < 0 || ndev < 1 || !bounds) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N < 0, ndev < 1 or bounds is NULL");
  const int64_t pitch = (N + ndev - 1) / ndev;
  for (int r = 0; r <= ndev; ++r) {
    const int64_t b = (int64_t)r * pitch;
    bounds[r] = b < N ? b : N;
  }
  return GFICF_OK;
}

/* Cell blocks of the GF-ICF passes, balanced by stored entries: block r ends at the first cell boundary at or after
 * r + 1 equal shares of the entries (gficf_amd/dist.py: shard_bounds_by_nnz).  bounds: ndev + 1 entries. */
int gficf_multi_cell_blocks_by_nnz(int64_t N, const void* colptr, int colptr_is_i64, int ndev, int64_t* bounds) {
  if (N < 0 || ndev < 1 || !bounds || !colptr) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N < 0, ndev < 1 or a NULL pointer");
  auto cp = [&](int64_t c) -> int64_t { return colptr_is_i64 ? ((const int64_t*)colptr)[c] : (int64_t)((const int32_t*)colptr)[c]; };
  const int64_t base = cp(0), nnz = N > 0 ? cp(N) - base : 0;
  bounds[0] = 0;
  for (int r = 1; r < ndev; ++r) {
    const int64_t target = base + (nnz * r + ndev - 1) / ndev;
    int64_t lo = 0, hi = N + 1;                     // first c with cp(c) >= target
    while (lo < hi) {
      const int64_t mid = (lo + hi) / 2;
      if (mid <= N && cp(mid) < target) lo = mid + 1;
      else hi = mid;
    }
    int64_t c = lo < bounds[r - 1] ? bounds[r - 1] : lo;
    bounds[r] = c < N ? c : N;
  }
  bounds[ndev] = N;
  return GFICF_OK;
}

int gficf_multi_create(const int* devices, int ndev, gficf_multi** out) {
  if (!out) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (ndev < 1 || ndev > 64) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ndev = %d outside [1, 64]", ndev);
  gficf_multi* m = new gficf_multi();
  m->ndev = ndev;
  for (int r = 0; r < ndev; ++r) {
    const int d = devices ? devices[r] : r;
    hipStream_t st = nullptr;
    gficf_ctx* c = nullptr;
    hipEvent_t ev = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { gficf_multi_destroy(m); GFICF_FAIL(GFICF_ERR_NO_DEVICE, "no HIP device visible (libgficf_hip needs an AMD GPU; there is no CPU fallback)"); }
    if (d < 0 || d >= n) { gficf_multi_destroy(m); GFICF_FAIL(GFICF_ERR_NO_DEVICE, "device %d out of range (%d visible)", d, n); }
    hipError_t e = hipSetDevice(d);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      if (ev) (void)hipEventDestroy(ev);             // not yet owned by m
      if (st) (void)hipStreamDestroy(st);
      gficf_multi_destroy(m);
      return hip_fail("stream / event creation", e);
    }
    m->dev.push_back(d);
    m->stream.push_back(st);
    m->ev.push_back(ev);
    m->ctx.push_back(nullptr);
    const int rc = gficf_ctx_create(d, st, &c);
    if (rc) { gficf_multi_destroy(m); return rc; }
    m->ctx[r] = c;
  }
  // peer access between every pair of distinct devices (the same device may be named twice: one block each)
  for (int a = 0; a < ndev && m->peer; ++a) {
    for (int b = 0; b < ndev; ++b) {
      if (m->dev[a] == m->dev[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, m->dev[a], m->dev[b]) != hipSuccess || !can) { m->peer = false; break; }
      (void)hipSetDevice(m->dev[a]);
      const hipError_t e = hipDeviceEnablePeerAccess(m->dev[b], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { m->peer = false; break; }
      (void)hipGetLastError();
    }
  }
  if (getenv("GFICF_HIP_NO_PEER")) m->peer = false;        // test hook: the every-device-ingests-everything form
  *out = m;
  return GFICF_OK;
}

static void multi_workers_stop(gficf_multi* m);
static int multi_drain(gficf_multi* m);

void gficf_multi_destroy(gficf_multi* m) {
  if (!m) return;
  multi_workers_stop(m);
  for (size_t r = 0; r < m->ctx.size(); ++r) {
    if (m->ctx[r]) gficf_ctx_destroy(m->ctx[r]);
  }
  for (size_t r = 0; r < m->cstream.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    for (hipStream_t cs : m->cstream[r]) if (cs) (void)hipStreamDestroy(cs);
    for (hipEvent_t ce : m->cev[r]) if (ce) (void)hipEventDestroy(ce);
    if (r < m->ev_prev.size() && m->ev_prev[r]) (void)hipEventDestroy(m->ev_prev[r]);
    if (r < m->ev_pulled.size() && m->ev_pulled[r]) (void)hipEventDestroy(m->ev_pulled[r]);
  }
  for (size_t r = 0; r < m->stream.size(); ++r) {
    (void)hipSetDevice(m->dev[r]);
    if (m->ev[r]) (void)hipEventDestroy(m->ev[r]);
    if (m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
  }
  delete m;
}


int gficf_multi_set_print(gficf_multi* m, void (*fn)(const char*)) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  for (gficf_ctx* c : m->ctx) c->print_fn = fn;
  return GFICF_OK;
}

/* The Jaccard entry of clustcells() (R/clustCells.R:65 -> src/rcpp_parallel_jaccard_coeff.cpp:59-80) over the
 * devices of the context.  Same arguments and the same (N*k) x 3 result as gficf_jaccard_host. */
int gficf_jaccard_host_multi(gficf_multi* m, const void* idx, int idx_is_f64, int64_t N, int k, int64_t ld, double* rmat,
                             int print_output) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  { const int drc = multi_drain(m); if (drc) return drc; }                     // device-resident steps still being enqueued come first
  if (N < 0 || k < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N = %lld or k = %d is negative", (long long)N, k);
  const int roww = gficf_jaccard_row_words(N, k);
  if (roww < 0) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d exceeds GFICF_JACCARD_MAX_K_EXACT = %d or N = %lld exceeds int32 ids", k, GFICF_JACCARD_MAX_K_EXACT, (long long)N);
  if (print_output) gficf_print(m->ctx[0], "Running Parallell Jaccard Coefficient Estimation...\n");  // reference :63
  const int64_t E = N * (int64_t)k;
  if (E > 0) {
    if (!idx || !rmat) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL host pointer");
    if (ld < N) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld = %lld < N = %lld", (long long)ld, (long long)N);
    gficf_advise_hugepages(rmat, sizeof(double) * 3 * (size_t)E);      // (a fresh R matrix: first touched by the per-device downloads)
    const int P = m->ndev;
    std::vector<int64_t> bd((size_t)P + 1);
    gficf_multi_cell_blocks(N, P, bd.data());
    const size_t esz = idx_is_f64 ? sizeof(double) : sizeof(int32_t);
    std::vector<void*> d_idx(P, nullptr);
    std::vector<int32_t*> d_table(P, nullptr);
    std::vector<double*> d_out(P, nullptr);
    FirstError fe;
    // 1. upload + ingest: each device its own block (peer exchange) or the whole matrix (no peer access); one host
    //    thread per device, so the uploads of the pageable matrix run side by side
    for_each_device(P, fe, [&](int r) -> int {
      gficf_ctx* c = m->ctx[r];
      const int64_t n = bd[r + 1] - bd[r];
      hipError_t e = hipSetDevice(m->dev[r]);
      const int64_t rows_up = m->peer ? n : N, row0 = m->peer ? bd[r] : 0;
      if (e == hipSuccess) e = gficf_pool_get(c, 0, esz * (size_t)(rows_up > 0 ? rows_up : 1) * (size_t)k, &d_idx[r]);
      if (e == hipSuccess) e = gficf_pool_get(c, 1, sizeof(int32_t) * (size_t)N * (size_t)roww, (void**)&d_table[r]);
      if (e == hipSuccess) e = gficf_pool_get(c, 2, sizeof(double) * 3 * (size_t)(n > 0 ? n : 1) * (size_t)k, (void**)&d_out[r]);
      if (e == hipSuccess && rows_up > 0)     // k columns of rows_up ids out of the column-major matrix (leading dimension ld)
        e = hipMemcpy2DAsync(d_idx[r], esz * (size_t)rows_up, (const char*)idx + esz * (size_t)row0, esz * (size_t)ld, esz * (size_t)rows_up,
                             (size_t)k, hipMemcpyHostToDevice, m->stream[r]);
      if (e != hipSuccess) return hip_fail("upload of the kNN block", e);
      if (rows_up > 0) {
        const int rc = gficf_jaccard_ingest_device(c, d_idx[r], idx_is_f64, rows_up, k, rows_up, N, d_table[r] + (size_t)row0 * roww);
        if (rc) return rc;
      }
      if (m->peer) {
        e = hipEventRecord(m->ev[r], m->stream[r]);
        if (e != hipSuccess) return hip_fail("hipEventRecord", e);
      }
      return GFICF_OK;
    });
    // 2. exchange: every device pulls the other blocks' table rows behind their ingest
    if (fe.rc == GFICF_OK && m->peer && P > 1) {
      for (int d = 0; d < P && fe.rc == GFICF_OK; ++d) {
        hipError_t e = hipSetDevice(m->dev[d]);
        for (int t = 1; t < P && e == hipSuccess; ++t) {
          const int s = (d + t) % P;                           // start with the next device: the pulls of a step are spread over the links
          const int64_t n = bd[s + 1] - bd[s];
          if (n <= 0) continue;
          e = hipStreamWaitEvent(m->stream[d], m->ev[s], 0);
          const size_t off = (size_t)bd[s] * roww, bytes = sizeof(int32_t) * (size_t)n * roww;
          if (e == hipSuccess) e = hipMemcpyPeerAsync(d_table[d] + off, m->dev[d], d_table[s] + off, m->dev[s], bytes, m->stream[d]);
        }
        if (e != hipSuccess) fe.note(hip_fail("exchange of table rows", e));
      }
    }
    // 3. edges of the own block, straight into the three column slices of rmat (a host thread per device: the
    //    device-to-host copies into the pageable result hold their thread)
    for_each_device(P, fe, [&](int r) -> int {
      const int64_t n = bd[r + 1] - bd[r];
      if (n <= 0) return GFICF_OK;
      hipError_t e = hipSetDevice(m->dev[r]);
      if (e != hipSuccess) return hip_fail("hipSetDevice", e);
      const size_t ne = (size_t)n * (size_t)k;
      const int rc = gficf_jaccard_edges_device(m->ctx[r], d_table[r], N, k, bd[r], bd[r + 1], d_out[r], d_out[r] + ne, d_out[r] + 2 * ne, nullptr);
      if (rc) return rc;
      for (int col = 0; col < 3 && e == hipSuccess; ++col)
        e = hipMemcpyAsync(rmat + (size_t)col * (size_t)E + (size_t)bd[r] * (size_t)k, d_out[r] + (size_t)col * ne, sizeof(double) * ne,
                           hipMemcpyDeviceToHost, m->stream[r]);
      if (e != hipSuccess) return hip_fail("download of the edge block", e);
      return GFICF_OK;
    });
    // 4. wait for every device (also on failure: nothing may still read the caller's buffers) and collect deferred errors
    for (int r = 0; r < P; ++r) fe.note(gficf_ctx_sync(m->ctx[r]));
    const int rc = fe.done();
    if (rc) return rc;
  }
  if (print_output) gficf_print(m->ctx[0], "Done!!\n");  // reference :77
  return GFICF_OK;
}

// ------------------------------------------------------------------------------------------------ device-resident step
// body(r) for every device slot r on the slot's persistent thread; returns when all are done.  First failing slot wins.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

static int multi_workers_start(gficf_multi* m) {
  if (!m->workers.empty() || m->ndev == 1) return GFICF_OK;
  try {
    for (int r = 0; r < m->ndev; ++r) {
      m->workers.emplace_back(new gficf_multi_worker());
      gficf_multi_worker* const w = m->workers.back().get();
      w->th = std::thread([m, w, r]() {
        (void)hipSetDevice(m->dev[r]);
        unsigned taken = 0;
        for (;;) {
          for (int spin = 0; spin < 4000 && w->posted.load(std::memory_order_acquire) == taken; ++spin) cpu_relax();
          std::shared_ptr<const std::function<int(int)>> job;
          {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv_job.wait(lk, [&] { return w->quit || !w->q.empty(); });
            if (w->q.empty()) return;                 // (quit: only once every posted job has run)
            job = std::move(w->q.front());
            w->q.pop_front();
            w->busy = true;
          }
          ++taken;
          const int rc = guarded(*job, r);
          std::string msg;
          if (rc != GFICF_OK) {
            try { msg = gficf_last_error(); } catch (...) {}
          }
          job.reset();
          {
            std::lock_guard<std::mutex> lk(w->mu);
            if (rc != GFICF_OK && w->rc == GFICF_OK) { w->rc = rc; w->msg.swap(msg); }
            w->busy = false;
            if (w->q.empty()) w->cv_idle.notify_all();
          }
        }
      });
    }
  } catch (...) {                                   // no threads to be had: the caller's thread does the work (workers stays short: never used)
    multi_workers_stop(m);
  }
  return GFICF_OK;
}

static void multi_workers_stop(gficf_multi* m) {
  for (auto& w : m->workers) {
    if (!w || !w->th.joinable()) continue;
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->quit = true;
    }
    w->cv_job.notify_all();
    w->th.join();
  }
  m->workers.clear();
}

// Posts body(r) to every device slot r's thread and returns (one device, or no threads: runs it here).  The job is kept alive by the queues.
static int multi_post(gficf_multi* m, std::function<int(int)> body) {
  const int P = m->ndev;
  if ((int)m->workers.size() != P) {
    for (int r = 0; r < P; ++r) {
      const int rc = guarded(body, r);
      if (rc) return rc;
    }
    return GFICF_OK;
  }
  std::shared_ptr<const std::function<int(int)>> job;
  try {
    job = std::make_shared<const std::function<int(int)>>(std::move(body));
    for (auto& w : m->workers) {
      {
        std::lock_guard<std::mutex> lk(w->mu);
        w->q.push_back(job);
      }
      w->posted.fetch_add(1, std::memory_order_release);
      w->cv_job.notify_one();
    }
  } catch (...) {
    // part of the workers may hold the job already: it refers to the caller's stack, so they are waited for before the error goes back
    (void)multi_drain(m);
    GFICF_FAIL(GFICF_ERR_HIP, "out of host memory posting a multi-device job");
  }
  return GFICF_OK;
}

// Waits until every posted job has run; the first failure (in device order) since the last call is returned and forgotten.
static int multi_drain(gficf_multi* m) {
  int rc = GFICF_OK;
  std::string msg;
  for (auto& w : m->workers) {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv_idle.wait(lk, [&] { return w->q.empty() && !w->busy; });
    if (w->rc != GFICF_OK && rc == GFICF_OK) { rc = w->rc; msg.swap(w->msg); }
    w->rc = GFICF_OK;
    w->msg.clear();
  }
  if (rc != GFICF_OK) gficf_set_error("%s", msg.c_str());
  return rc;
}

// body(r) for every device slot r on the slot's thread; returns when all are done.  First failing slot wins.
static int multi_run(gficf_multi* m, const std::function<int(int)>& body) {
  const int rc = multi_post(m, body);
  if (rc) return rc;
  return multi_drain(m);
}

// destroys whatever copy streams / events exist (also after a creation that failed half-way) and empties the lists
static void multi_step_resources_free(gficf_multi* m) {
  for (size_t d = 0; d < m->cstream.size(); ++d) {
    if (d < m->dev.size()) (void)hipSetDevice(m->dev[d]);
    for (hipStream_t s_ : m->cstream[d]) if (s_) (void)hipStreamDestroy(s_);
    if (d < m->cev.size()) for (hipEvent_t e_ : m->cev[d]) if (e_) (void)hipEventDestroy(e_);
  }
  for (hipEvent_t e_ : m->ev_prev) if (e_) (void)hipEventDestroy(e_);
  for (hipEvent_t e_ : m->ev_pulled) if (e_) (void)hipEventDestroy(e_);
  m->cstream.clear(); m->cev.clear(); m->ev_prev.clear(); m->ev_pulled.clear();
  m->step_resources_ok = false;
}

// lazily: the copy streams and events of gficf_multi_jaccard_device
static int multi_step_resources(gficf_multi* m) {
  const int P = m->ndev;
  if (m->step_resources_ok) return GFICF_OK;
  multi_step_resources_free(m);                                // (what a failed attempt left behind)
  m->cstream.assign((size_t)P, {});
  m->cev.assign((size_t)P, {});
  m->ev_prev.assign((size_t)P, nullptr);
  m->ev_pulled.assign((size_t)P, nullptr);
  for (int d = 0; d < P; ++d) {
    hipError_t e = hipSetDevice(m->dev[d]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_prev[d], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_pulled[d], hipEventDisableTiming);
    for (int t = 1; t < P && e == hipSuccess; ++t) {
      hipStream_t cs = nullptr;
      hipEvent_t ce = nullptr;
      e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ce, hipEventDisableTiming);
      m->cstream[d].push_back(cs);
      m->cev[d].push_back(ce);
    }
    if (e != hipSuccess) {
      multi_step_resources_free(m);                            // nothing half-made stays behind: the next call starts over
      return hip_fail("copy streams of the device-resident step", e);
    }
  }
  m->step_resources_ok = true;
  return GFICF_OK;
}

int gficf_multi_set_jaccard_distinct(gficf_multi* m, int assume_distinct) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  const int drc = multi_drain(m);                                              // (posted steps read the setting when they are enqueued)
  if (drc) return drc;
  for (gficf_ctx* c : m->ctx) {
    const int rc = gficf_ctx_set_jaccard_distinct(c, assume_distinct);
    if (rc) return rc;
  }
  return GFICF_OK;
}

/* The sharded Jaccard step with everything resident in HBM, single process (what is sharded: the cells of the
 * reference's parallelFor(0, N, worker), src/rcpp_parallel_jaccard_coeff.cpp:73; the call being sharded: R/clustCells.R:64-65).
 * Enqueue only — nothing here waits for a device. */
int gficf_multi_jaccard_device(gficf_multi* m, const void* const* d_idx, int idx_is_f64, const int64_t* ld, int64_t N, int k,
                               int32_t* const* d_table, double* const* d_out) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  if (N < 0 || k < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N = %lld or k = %d is negative", (long long)N, k);
  const int roww = gficf_jaccard_row_words(N, k);
  if (roww < 0) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "k = %d exceeds GFICF_JACCARD_MAX_K_EXACT = %d or N = %lld exceeds int32 ids", k, GFICF_JACCARD_MAX_K_EXACT, (long long)N);
  if (N == 0 || k == 0) return GFICF_OK;
  if (!d_idx || !d_table || !d_out) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer array");
  const int P = m->ndev;
  std::vector<int64_t> bd;
  try { bd.resize((size_t)P + 1); } catch (...) { GFICF_FAIL(GFICF_ERR_HIP, "out of host memory"); }
  gficf_multi_cell_blocks(N, P, bd.data());
  for (int r = 0; r < P; ++r) {
    const int64_t n = bd[r + 1] - bd[r];
    if (!d_table[r] || (n > 0 && (!d_idx[r] || !d_out[r]))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer for device slot %d", r);
    if (ld && ld[r] < n) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld[%d] = %lld < %lld rows of the block", r, (long long)ld[r], (long long)n);
  }
  int rc = multi_step_resources(m);
  if (!rc) rc = multi_workers_start(m);
  if (rc) return rc;
  const bool prev_valid = m->step_valid;
  // 1. every device (on its own host thread): its block of ids -> its slice of ITS table, behind (a) its own last edge kernel
  //    (same stream) and (b) the other devices' pulls of that slice in the step before
  rc = multi_run(m, [&](int r) -> int {
    const int64_t n = bd[r + 1] - bd[r];
    hipError_t e = hipSetDevice(m->dev[r]);
    if (e == hipSuccess) e = hipEventRecord(m->ev_prev[r], m->stream[r]);
    for (int d = 0; d < P && e == hipSuccess && prev_valid; ++d)
      if (d != r) e = hipStreamWaitEvent(m->stream[r], m->ev_pulled[d], 0);
    if (e != hipSuccess) return hip_fail("ordering the ingest behind the last step", e);
    if (n > 0) {
      const int irc = gficf_jaccard_ingest_device(m->ctx[r], d_idx[r], idx_is_f64, n, k, ld ? ld[r] : n, N, d_table[r] + (size_t)bd[r] * roww);
      if (irc) return irc;
    }
    e = hipEventRecord(m->ev[r], m->stream[r]);
    if (e != hipSuccess) return hip_fail("hipEventRecord", e);
    return GFICF_OK;
  });
  if (rc) return rc;
  // (every ev[s] has been RECORDED by now: a wait enqueued on an event captures the record in force at that moment)
  // 2. every device pulls the P - 1 other slices, each on a copy stream of its own (all pairs at once: a device's pulls
  //    run side by side over its links instead of one after the other), then builds the edges of its block
  rc = multi_run(m, [&](int d) -> int {
    hipError_t e = hipSetDevice(m->dev[d]);
    for (int t = 1; t < P && e == hipSuccess; ++t) {
      const int s = (d + t) % P;                              // start with the next device: the pulls of a step are spread over the links
      const int64_t n = bd[s + 1] - bd[s];
      if (n <= 0) continue;
      hipStream_t cs = m->cstream[d][(size_t)t - 1];
      e = hipStreamWaitEvent(cs, m->ev[s], 0);                // the slice has been ingested on its owner
      if (e == hipSuccess) e = hipStreamWaitEvent(cs, m->ev_prev[d], 0);   // this device's last edge kernel no longer reads its table
      const size_t off = (size_t)bd[s] * roww, bytes = sizeof(int32_t) * (size_t)n * roww;
      if (e == hipSuccess) e = hipMemcpyPeerAsync(d_table[d] + off, m->dev[d], d_table[s] + off, m->dev[s], bytes, cs);
      if (e == hipSuccess) e = hipEventRecord(m->cev[d][(size_t)t - 1], cs);
      if (e == hipSuccess) e = hipStreamWaitEvent(m->stream[d], m->cev[d][(size_t)t - 1], 0);
    }
    if (e == hipSuccess) e = hipEventRecord(m->ev_pulled[d], m->stream[d]);
    if (e != hipSuccess) return hip_fail("exchange of table slices (peer copies)", e);
    const int64_t n = bd[d + 1] - bd[d];
    if (n > 0) {
      const size_t ne = (size_t)n * (size_t)k;
      const int erc = gficf_jaccard_edges_device(m->ctx[d], d_table[d], N, k, bd[d], bd[d + 1], d_out[d], d_out[d] + ne, d_out[d] + 2 * ne, nullptr);
      if (erc) return erc;
    }
    return GFICF_OK;
  });
  if (rc) return rc;
  m->step_valid = true;
  return GFICF_OK;
}

/* The same step for blocks whose ids have LOCALITY, with nothing exchanged at all: every device plans the rows its block names outside
 * (halo.hip), builds the table of its own sub-problem — own cells from its block, the few requested rows READ WHERE THEY LIE, in the
 * owners' blocks of ids, through the peer mapping — and its edges.  Three kernels after the plan's mark (four launches) per device and no
 * event between devices: the blocks of ids are inputs.  Same blocks, same rows of rmat, bit for bit. */
int gficf_multi_jaccard_halo_device(gficf_multi* m, const int32_t* const* d_idx, const int64_t* ld, int64_t N, int k, int cap,
                                    void* const* d_ws, int32_t* const* d_req, int32_t* const* d_table, int32_t* const* d_l2g, double* const* d_out) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  if (N < 0 || k < 0 || cap < 1) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "N = %lld, k = %d or cap = %d out of range", (long long)N, k, cap);
  if (k > 64) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "the halo step of the multi-device context covers k <= 64 (k = %d): use gficf_multi_jaccard_device", k);
  const int P = m->ndev;
  if (P > GFICF_HALO_MAX_PEERS) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "the halo step of the multi-device context covers %d devices, %d given", GFICF_HALO_MAX_PEERS, P);
  if (!m->peer && P > 1) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "the halo step reads the other devices' blocks in place: it needs peer access between every pair of devices");
  if (N == 0 || k == 0) return GFICF_OK;
  if (!d_idx || !d_ws || !d_req || !d_table || !d_l2g || !d_out) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer array");
  // (the step is posted, not waited for: everything the job reads is its own copy)
  struct Step {
    std::vector<int64_t> bd, lds;
    std::vector<const int32_t*> idx;
    std::vector<void*> ws;
    std::vector<int32_t*> req, table, l2g;
    std::vector<double*> out;
  };
  std::shared_ptr<Step> st;
  try {
    st = std::make_shared<Step>();
    st->bd.resize((size_t)P + 1); st->lds.resize((size_t)P); st->idx.resize((size_t)P); st->ws.resize((size_t)P);
    st->req.resize((size_t)P); st->table.resize((size_t)P); st->l2g.resize((size_t)P); st->out.resize((size_t)P);
  } catch (...) { GFICF_FAIL(GFICF_ERR_HIP, "out of host memory"); }
  gficf_multi_cell_blocks(N, P, st->bd.data());
  const int64_t rpr = (N + P - 1) / P;
  for (int r = 0; r < P; ++r) {
    const int64_t n = st->bd[r + 1] - st->bd[r];
    if (!d_ws[r] || !d_req[r] || !d_table[r] || !d_l2g[r] || (n > 0 && (!d_idx[r] || !d_out[r]))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL device pointer for device slot %d", r);
    if (ld && ld[r] < n) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "ld[%d] = %lld < %lld rows of the block", r, (long long)ld[r], (long long)n);
    st->lds[r] = ld ? ld[r] : n;
    st->idx[r] = d_idx[r]; st->ws[r] = d_ws[r]; st->req[r] = d_req[r]; st->table[r] = d_table[r]; st->l2g[r] = d_l2g[r]; st->out[r] = d_out[r];
  }
  if (gficf_jaccard_row_words(rpr + (int64_t)P * cap, k) < 0) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "no table format for %lld rows of k = %d", (long long)(rpr + (int64_t)P * cap), k);
  int rc = multi_workers_start(m);
  if (rc) return rc;
  // every device on its own: plan -> the table (own cells' rows, and the requested rows read in the owners' blocks) -> edges.  No event between the
  // devices: the blocks of ids are inputs, complete before the call (the contract of every device entry).
  return multi_post(m, [m, st, N, k, cap, P, rpr](int r) -> int {
    const int64_t n = st->bd[r + 1] - st->bd[r], n_ext = n + (int64_t)P * cap, b = st->bd[r];
    gficf_ctx* c = m->ctx[r];
    int q = gficf_jaccard_halo_plan_device(c, st->idx[r], n, k, st->lds[r], N, b, P, rpr, cap, st->ws[r], st->req[r]);
    if (!q) q = gficf_jaccard_halo_ingest_peer_device(c, st->idx[r], n, k, st->lds[r], N, b, P, rpr, cap, st->ws[r], st->req[r], st->idx.data(), st->lds.data(), st->table[r], st->l2g[r]);
    if (!q && n > 0) {
      const size_t ne = (size_t)n * (size_t)k;
      q = gficf_jaccard_edges_mapped_device(c, st->table[r], n_ext, k, n, b, st->l2g[r], st->out[r], st->out[r] + ne, st->out[r] + 2 * ne, nullptr);
    }
    return q;
  });
}

int gficf_multi_sync(gficf_multi* m) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  FirstError fe;
  fe.note(multi_drain(m));                                                     // steps still being enqueued; what their enqueue reported
  for (int r = 0; r < m->ndev; ++r) fe.note(gficf_ctx_sync(m->ctx[r]));      // (the copy streams were joined into the device's stream)
  return fe.done();
}

// ------------------------------------------------------------------------------------------------ GF-ICF
static void multi_plan_clear(gficf_multi* m) {
  m->has_plan = false;
  m->blk.clear();
}

/* gficf() over the devices of the context: same arguments and results as gficf_normalize_csc_host_plan / _finish. */
int gficf_normalize_csc_host_multi_plan(gficf_multi* m, int64_t G, int64_t N, const void* colptr, int colptr_is_i64,
                                        const int32_t* rowidx, const double* x, double prop_min, double prop_max,
                                        const double* w_in, int64_t* G_kept, int64_t* nnz_kept) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  { const int drc = multi_drain(m); if (drc) return drc; }                     // device-resident steps still being enqueued come first
  if (G < 0 || N < 0) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "negative dimension");
  if (G > 0x7FFFFFFFll) GFICF_FAIL(GFICF_ERR_UNSUPPORTED, "G = %lld exceeds int32 row indices", (long long)G);
  if (!colptr || !G_kept || !nnz_kept) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  multi_plan_clear(m);
  std::vector<int64_t> cp((size_t)N + 1);
  for (int64_t c = 0; c <= N; ++c)
    cp[(size_t)c] = colptr_is_i64 ? ((const int64_t*)colptr)[c] : (int64_t)((const int32_t*)colptr)[c];
  if (cp[0] != 0) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr[0] = %lld, expected 0", (long long)cp[0]);
  for (int64_t c = 0; c < N; ++c)
    if (cp[(size_t)c + 1] < cp[(size_t)c]) GFICF_FAIL(GFICF_ERR_BAD_CSC, "colptr not monotone at cell %lld", (long long)c);
  const int64_t nnz = cp[(size_t)N];
  if (nnz > 0 && (!rowidx || !x)) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL pointer");
  const int P = m->ndev;
  std::vector<int64_t> bd((size_t)P + 1);
  gficf_multi_cell_blocks_by_nnz(N, cp.data(), 1, P, bd.data());
  m->G = G; m->N = N; m->colptr_is_i64 = colptr_is_i64;
  m->blk.assign((size_t)P, gficf_multi_block());
  const size_t gsz = (size_t)(G > 0 ? G : 1);
  std::vector<std::vector<int64_t>> cpl((size_t)P), ntl((size_t)P);
  FirstError fe;
  // 1. per block: upload, count (a host thread per device: the uploads of the pageable matrix run side by side)
  for_each_device(P, fe, [&](int r) -> int {
    gficf_multi_block& B = m->blk[r];
    gficf_ctx* c = m->ctx[r];
    B.b = bd[r]; B.e = bd[r + 1]; B.p0 = cp[(size_t)B.b]; B.p1 = cp[(size_t)B.e];
    const int64_t n = B.e - B.b, nz = B.p1 - B.p0;
    cpl[r].resize((size_t)n + 1);
    for (int64_t t = 0; t <= n; ++t) cpl[r][(size_t)t] = cp[(size_t)(B.b + t)] - B.p0;      // the block's colptr starts at 0
    ntl[r].assign(gsz, 0);
    hipError_t e = hipSetDevice(m->dev[r]);
    gficf_arena ar;
    const size_t nsz = (size_t)(nz > 0 ? nz : 1);
    const size_t o_cp = ar.take(sizeof(int64_t) * ((size_t)n + 1)), o_ri = ar.take(sizeof(int32_t) * nsz), o_x = ar.take(sizeof(double) * nsz);
    const size_t o_nt = ar.take(sizeof(int64_t) * gsz), o_keep = ar.take(gsz), o_genes = ar.take(gficf_csc_genes_bytes(G));
    const size_t o_w = ar.take(sizeof(double) * gsz), o_gk = ar.take(sizeof(int64_t)), o_ocp = ar.take(sizeof(int64_t) * ((size_t)n + 1));
    const size_t o_win = ar.take(sizeof(double) * gsz);
    if (e == hipSuccess) e = ar.bind(c, 4);
    if (e != hipSuccess) return hip_fail("device buffers of the GF-ICF block", e);
    B.d_colptr = ar.at<int64_t>(o_cp); B.d_rowidx = ar.at<int32_t>(o_ri); B.d_x = ar.at<double>(o_x);
    B.d_nt = ar.at<int64_t>(o_nt); B.d_keep = ar.at<uint8_t>(o_keep); B.d_genes = ar.at<gficf_gene_entry>(o_genes);
    B.d_w = ar.at<double>(o_w); B.d_gkept = ar.at<int64_t>(o_gk); B.d_out_colptr = ar.at<int64_t>(o_ocp);
    B.d_w_in = (w_in && G > 0) ? ar.at<double>(o_win) : nullptr;
    hipStream_t st = m->stream[r];
    e = hipMemcpyAsync(B.d_colptr, cpl[r].data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nz > 0) e = hipMemcpyAsync(B.d_rowidx, rowidx + B.p0, sizeof(int32_t) * (size_t)nz, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nz > 0) e = hipMemcpyAsync(B.d_x, x + B.p0, sizeof(double) * (size_t)nz, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && B.d_w_in) e = hipMemcpyAsync(B.d_w_in, w_in, sizeof(double) * (size_t)G, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(B.d_nt, 0, sizeof(int64_t) * gsz, st);
    if (e != hipSuccess) return hip_fail("upload of the GF-ICF block", e);
    const int rc = gficf_csc_count_device(c, G, n, B.d_colptr, B.d_rowidx, B.d_x, nz, B.d_nt);
    if (rc) return rc;
    if (G > 0) {
      e = hipMemcpyAsync(ntl[r].data(), B.d_nt, sizeof(int64_t) * (size_t)G, hipMemcpyDeviceToHost, st);
      if (e != hipSuccess) return hip_fail("download of the gene counts", e);
    }
    return GFICF_OK;
  });
  for (int r = 0; r < P; ++r) fe.note(gficf_ctx_sync(m->ctx[r]));
  if (fe.rc) { multi_plan_clear(m); return fe.done(); }
  // 2. the one global quantity: nt_g summed over the blocks (the all-reduce of the sharded path)
  std::vector<int64_t> nt(gsz, 0);
  for (int r = 0; r < P; ++r)
    for (int64_t g = 0; g < G; ++g) nt[(size_t)g] += ntl[r][(size_t)g];
  // 3. per block: filter + weights from the global counts, kept entries per cell
  std::vector<int64_t> hk((size_t)P * 2, 0);
  for_each_device(P, fe, [&](int r) -> int {
    gficf_multi_block& B = m->blk[r];
    gficf_ctx* c = m->ctx[r];
    const int64_t n = B.e - B.b;
    hipStream_t st = m->stream[r];
    hipError_t e = hipSetDevice(m->dev[r]);
    if (e == hipSuccess && G > 0) e = hipMemcpyAsync(B.d_nt, nt.data(), sizeof(int64_t) * (size_t)G, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return hip_fail("upload of the summed gene counts", e);
    int rc = gficf_csc_genes_device(c, G, N, B.d_nt, prop_min, prop_max, B.d_w_in, B.d_keep, B.d_genes, B.d_w, B.d_gkept);
    if (!rc) rc = gficf_csc_colptr_device(c, G, n, B.d_colptr, B.d_rowidx, B.d_keep, B.d_gkept, B.d_out_colptr);
    if (rc) return rc;
    e = hipMemcpyAsync(&hk[(size_t)r * 2], B.d_gkept, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&hk[(size_t)r * 2 + 1], B.d_out_colptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return hip_fail("download of the kept counts", e);
    return GFICF_OK;
  });
  for (int r = 0; r < P; ++r) fe.note(gficf_ctx_sync(m->ctx[r]));
  if (fe.rc) { multi_plan_clear(m); return fe.done(); }
  int64_t total = 0;
  for (int r = 0; r < P; ++r) {
    m->blk[r].nnz_kept = hk[(size_t)r * 2 + 1];
    m->blk[r].out_off = total;
    total += hk[(size_t)r * 2 + 1];
  }
  m->g_kept = hk[0];
  m->nnz_kept = total;
  m->has_plan = true;
  *G_kept = m->g_kept;
  *nnz_kept = total;
  return GFICF_OK;
}

int gficf_normalize_csc_host_multi_finish(gficf_multi* m, uint8_t* keep, int64_t* nt, double* w, void* out_colptr,
                                          int32_t* out_rowidx, double* out_x) {
  if (!m) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "multi context is NULL");
  { const int drc = multi_drain(m); if (drc) return drc; }                     // device-resident steps still being enqueued come first
  if (!m->has_plan) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "gficf_normalize_csc_host_multi_finish without a plan");
  if (!out_colptr || (m->nnz_kept > 0 && (!out_rowidx || !out_x))) GFICF_FAIL(GFICF_ERR_INVALID_ARG, "NULL output pointer");
  const int P = m->ndev;
  const int64_t G = m->G, N = m->N;
  std::vector<std::vector<int64_t>> ocp((size_t)P);
  FirstError fe;
  // (the caller's result vectors are fresh as a rule: huge pages asked for before the per-device copies first touch them)
  if (m->nnz_kept > 0) {
    gficf_advise_hugepages(out_rowidx, sizeof(int32_t) * (size_t)m->nnz_kept);
    gficf_advise_hugepages(out_x, sizeof(double) * (size_t)m->nnz_kept);
  }
  for_each_device(P, fe, [&](int r) -> int {          // a host thread per device: the downloads into the pageable results run side by side
    gficf_multi_block& B = m->blk[r];
    gficf_ctx* c = m->ctx[r];
    const int64_t n = B.e - B.b, nz = B.p1 - B.p0;
    hipStream_t st = m->stream[r];
    hipError_t e = hipSetDevice(m->dev[r]);
    const size_t ksz = (size_t)(B.nnz_kept > 0 ? B.nnz_kept : 1);
    gficf_arena ar;
    const size_t o_ri = ar.take(sizeof(int32_t) * ksz), o_x = ar.take(sizeof(double) * ksz);
    if (e == hipSuccess) e = ar.bind(c, 7);
    if (e != hipSuccess) return hip_fail("output buffers of the GF-ICF block", e);
    int32_t* const d_ori = ar.at<int32_t>(o_ri);
    double* const d_ox = ar.at<double>(o_x);
    const int rc = gficf_csc_scale_device(c, G, n, B.d_colptr, B.d_rowidx, B.d_x, nz, B.d_genes, B.d_gkept, B.d_out_colptr, d_ori, d_ox);
    if (rc) return rc;
    ocp[r].resize((size_t)n + 1);
    e = hipMemcpyAsync(ocp[r].data(), B.d_out_colptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && B.nnz_kept > 0) e = hipMemcpyAsync(out_rowidx + B.out_off, d_ori, sizeof(int32_t) * (size_t)B.nnz_kept, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && B.nnz_kept > 0) e = hipMemcpyAsync(out_x + B.out_off, d_ox, sizeof(double) * (size_t)B.nnz_kept, hipMemcpyDeviceToHost, st);
    if (r == 0 && G > 0) {                   // the per-gene results are the same on every device
      if (e == hipSuccess && keep) e = hipMemcpyAsync(keep, B.d_keep, (size_t)G, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess && nt) e = hipMemcpyAsync(nt, B.d_nt, sizeof(int64_t) * (size_t)G, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess && w) e = hipMemcpyAsync(w, B.d_w, sizeof(double) * (size_t)G, hipMemcpyDeviceToHost, st);
    }
    if (e != hipSuccess) return hip_fail("download of the GF-ICF block", e);
    return GFICF_OK;
  });
  for (int r = 0; r < P; ++r) fe.note(gficf_ctx_sync(m->ctx[r]));
  int rc = fe.done();
  if (!rc) {
    for (int r = 0; r < P; ++r) {
      const gficf_multi_block& B = m->blk[r];
      for (int64_t t = (r == 0 ? 0 : 1); t <= B.e - B.b; ++t) {
        const int64_t v = ocp[r][(size_t)t] + B.out_off;
        if (m->colptr_is_i64) ((int64_t*)out_colptr)[B.b + t] = v;
        else ((int32_t*)out_colptr)[B.b + t] = (int32_t)v;
      }
    }
    if (N == 0) {
      if (m->colptr_is_i64) ((int64_t*)out_colptr)[0] = 0;
      else ((int32_t*)out_colptr)[0] = 0;
    }
  }
  multi_plan_clear(m);
  return rc;
}

}  // extern "C"
