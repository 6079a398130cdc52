// Single-controller multi-GPU group behind the C-ABI (include/tgp.h, "tgp_group_*").
//
// One host process drives n model replicas, one per device.  Replicated state (hyper-parameters, data,
// factorisation, trajectory weights) is brought up by the same deterministic single-device entry points on
// every member, concurrently (one host worker thread per member: `update` is a chain of dependent kernels
// that ends in a host-side check of the Cholesky info flag).  Sharded sweeps are only ENQUEUED, member by
// member, from the calling thread (tgp_acq_argmax_async / tgp_traj_argmin_async on the members' private
// streams); the (value, index) pairs they leave on their devices meet in ONE in-process RCCL all-gather
// (ncclCommInitAll communicator: xGMI, 16 B per member and vectorised function) or in 16-byte peer copies,
// a merge kernel on member 0 applies (max value, min index) and one small copy reaches the host.
//
// RCCL is resolved at run time (dlopen) so that libtgp.so has no link-time dependency on it: inside a
// Python process that imported torch the already loaded librccl.so.1 (torch's own copy, bound to the same
// HIP runtime as this library) is reused; from plain C the ROCm one is loaded.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only; every call goes through dlsym

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "tgp_host.hpp"

using namespace tgp;

namespace {

thread_local std::string g_group_create_error;
constexpr int MAXV = 16;  // pairs per member and step: 1 (acquisition arg-max) or B <= 16 trajectories

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string& err) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      err = std::string("RCCL is not loadable (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?");
      return false;
    }
#define TGP_SYM(field, sym)                                      \
  field = (decltype(field))dlsym(lib, sym);                      \
  if (!field) {                                                  \
    err = std::string("RCCL symbol missing: ") + sym;            \
    return false;                                                \
  }
    TGP_SYM(CommInitAll, "ncclCommInitAll");
    TGP_SYM(CommDestroy, "ncclCommDestroy");
    TGP_SYM(CommCount, "ncclCommCount");
    TGP_SYM(AllGather, "ncclAllGather");
    TGP_SYM(GroupStart, "ncclGroupStart");
    TGP_SYM(GroupEnd, "ncclGroupEnd");
    TGP_SYM(GetErrorString, "ncclGetErrorString");
#undef TGP_SYM
    return true;
  }
};

// One host thread per member: runs the synchronous single-device entry points of that member.
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, stop = false;
  int rc = 0;
  void start() {
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || stop; });
        if (stop) return;
        std::function<int()> j = std::move(job);
        has_job = false;
        lk.unlock();
        const int r = j();
        lk.lock();
        rc = r;
        done = true;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<int()> j) {
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j);
    has_job = true;
    done = false;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
    return rc;
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};

}  // namespace

struct tgp_group_s {
  int n = 0, d = 0, kind = 0, merge = TGP_MERGE_RCCL, rccl_ranks = 0;
  std::vector<int> devs;
  std::vector<tgp_handle> h;
  std::vector<Worker*> workers;
  std::string err;
  // resident candidate table, sharded: member r owns rows [lo[r], hi[r])
  int64_t M = 0;
  std::vector<int64_t> lo, hi;
  std::vector<DevBuf> cand, pair, gather;  // [hi-lo][d]; [2][MAXV]; [n][2][MAXV] (+ the merged [2][MAXV])
  std::vector<hipEvent_t> ev;
  Rccl rccl;
  std::vector<ncclComm_t> comms;
};

struct tgp_group_traj_s {
  tgp_group g = nullptr;
  int B = 0;
  std::vector<tgp_traj> t;
};

namespace {

int gfail(tgp_group g, int code, const char* fmt, ...) {
  char buf[640];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (g) g->err = buf;
  else g_group_create_error = buf;
  return code;
}

#define GHIP(g, expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return gfail(g, e_ == hipErrorOutOfMemory ? TGP_ERR_ALLOC : TGP_ERR_HIP, "%s: %s", #expr,    \
                   hipGetErrorString(e_));                                                         \
  } while (0)
#define GNCCL(g, expr)                                                                             \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) return gfail(g, TGP_ERR_HIP, "%s: %s", #expr, g->rccl.GetErrorString(r_)); \
  } while (0)

// status of a member call -> the group's status and message
int member_rc(tgp_group g, int i, int rc) {
  if (rc == TGP_OK) return TGP_OK;
  return gfail(g, rc, "member %d (device %d): %s", i, g->devs[i], tgp_last_error(g->h[i]));
}

// fn(i) on every member's worker thread, concurrently; the first failing member's status is returned
int run_all(tgp_group g, const std::function<int(int)>& fn) {
  for (int i = 0; i < g->n; ++i) g->workers[i]->submit([&fn, i] { return fn(i); });
  int first = TGP_OK, who = -1;
  for (int i = 0; i < g->n; ++i) {
    const int rc = g->workers[i]->wait();
    if (rc != TGP_OK && first == TGP_OK) {
      first = rc;
      who = i;
    }
  }
  return who >= 0 ? member_rc(g, who, first) : TGP_OK;
}

void shard(int64_t M, int n, std::vector<int64_t>& lo, std::vector<int64_t>& hi) {  // == distributed.shard_range
  const int64_t per = (M + n - 1) / n;
  lo.resize(n);
  hi.resize(n);
  for (int r = 0; r < n; ++r) {
    lo[r] = std::min<int64_t>((int64_t)r * per, M);
    hi[r] = std::min<int64_t>(lo[r] + per, M);
  }
}

// Every member's stream holds the producer of its [2][V] pairs in g->pair[i].  Gather them on member 0 (RCCL
// all-gather or peer copies), merge there, and hand the V winners to the host: ONE host synchronisation.
int merge_and_fetch(tgp_group g, int V, int minimize, double* out_val, int64_t* out_idx) {
  const int n = g->n;
  const size_t cnt = 2 * (size_t)V;
  if (g->merge == TGP_MERGE_RCCL) {
    GNCCL(g, g->rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
      if (hipSetDevice(g->devs[i]) != hipSuccess) {
        (void)g->rccl.GroupEnd();
        return gfail(g, TGP_ERR_HIP, "hipSetDevice(%d) failed", g->devs[i]);
      }
      const ncclResult_t r = g->rccl.AllGather(g->pair[i].p, g->gather[i].p, cnt, ncclDouble, g->comms[i],
                                               g->h[i]->stream);
      if (r != ncclSuccess) {
        (void)g->rccl.GroupEnd();
        return gfail(g, TGP_ERR_HIP, "ncclAllGather (member %d): %s", i, g->rccl.GetErrorString(r));
      }
    }
    GNCCL(g, g->rccl.GroupEnd());
  } else {
    double* g0 = g->gather[0].as<double>();
    for (int i = 0; i < n; ++i) {
      GHIP(g, hipSetDevice(g->devs[i]));
      if (i == 0) {
        GHIP(g, hipMemcpyAsync(g0, g->pair[0].p, cnt * sizeof(double), hipMemcpyDeviceToDevice, g->h[0]->stream));
      } else {
        GHIP(g, hipMemcpyPeerAsync(g0 + (size_t)i * cnt, g->devs[0], g->pair[i].p, g->devs[i], cnt * sizeof(double),
                                   g->h[i]->stream));
        GHIP(g, hipEventRecord(g->ev[i], g->h[i]->stream));
      }
    }
    GHIP(g, hipSetDevice(g->devs[0]));
    for (int i = 1; i < n; ++i) GHIP(g, hipStreamWaitEvent(g->h[0]->stream, g->ev[i], 0));
  }
  GHIP(g, hipSetDevice(g->devs[0]));
  double* merged = g->gather[0].as<double>() + (size_t)n * 2 * MAXV;
  launch_merge_winners(g->h[0]->stream, g->gather[0].as<double>(), n, V, minimize, merged);
  double host[2 * MAXV];
  GHIP(g, hipMemcpyAsync(host, merged, cnt * sizeof(double), hipMemcpyDeviceToHost, g->h[0]->stream));
  GHIP(g, hipStreamSynchronize(g->h[0]->stream));
  GHIP(g, hipGetLastError());
  for (int v = 0; v < V; ++v) {
    if (out_val) out_val[v] = host[v];
    if (out_idx) memcpy(&out_idx[v], &host[V + v], sizeof(int64_t));
  }
  return TGP_OK;
}

// Group calls walk over the members' devices; the caller's current device (torch code in the same thread relies on
// it) is put back when the entry point returns.
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceGuard() {
    if (dev >= 0) (void)hipSetDevice(dev);
    (void)hipGetLastError();
  }
};

// First use of a fresh communicator: all-gather (member ordinal, device id) and check every member's copy before
// any winner travels through it -- a mis-wired communicator (ranks permuted against the members' buffers) would
// otherwise merge the right values under the wrong global indices without any error.
int rccl_self_check(tgp_group g) {
  const int n = g->n;
  for (int i = 0; i < n; ++i) {
    GHIP(g, hipSetDevice(g->devs[i]));
    const double tag[2] = {(double)i, (double)g->devs[i]};
    GHIP(g, hipMemcpyAsync(g->pair[i].p, tag, sizeof tag, hipMemcpyHostToDevice, g->h[i]->stream));
    GHIP(g, hipStreamSynchronize(g->h[i]->stream));  // `tag` is a stack buffer
  }
  GNCCL(g, g->rccl.GroupStart());
  for (int i = 0; i < n; ++i) {
    if (hipSetDevice(g->devs[i]) != hipSuccess) {
      (void)g->rccl.GroupEnd();
      return gfail(g, TGP_ERR_HIP, "hipSetDevice(%d) failed", g->devs[i]);
    }
    const ncclResult_t r = g->rccl.AllGather(g->pair[i].p, g->gather[i].p, 2, ncclDouble, g->comms[i], g->h[i]->stream);
    if (r != ncclSuccess) {
      (void)g->rccl.GroupEnd();
      return gfail(g, TGP_ERR_HIP, "ncclAllGather self-check (member %d): %s", i, g->rccl.GetErrorString(r));
    }
  }
  GNCCL(g, g->rccl.GroupEnd());
  std::vector<double> got(2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    GHIP(g, hipSetDevice(g->devs[i]));
    GHIP(g, hipMemcpyAsync(got.data(), g->gather[i].p, got.size() * sizeof(double), hipMemcpyDeviceToHost, g->h[i]->stream));
    GHIP(g, hipStreamSynchronize(g->h[i]->stream));
    for (int j = 0; j < n; ++j)
      if (got[2 * j] != (double)j || got[2 * j + 1] != (double)g->devs[j])
        return gfail(g, TGP_ERR_HIP, "RCCL self-check: member %d received (%g, %g) in slot %d, expected (%d, %d)", i,
                     got[2 * j], got[2 * j + 1], j, j, g->devs[j]);
  }
  return TGP_OK;
}

int need_candidates(tgp_group g) {
  if (g->M < 1) return gfail(g, TGP_ERR_STATE, "no resident candidates: call tgp_group_set_candidates / _sample_candidates");
  return TGP_OK;
}

}  // namespace

extern "C" {

const char* tgp_group_last_error(tgp_group g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int tgp_group_create(const int* device_ids, int n_dev, int d, int kernel_kind, int merge, tgp_group* out) {
  if (!out) return gfail(nullptr, TGP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (!device_ids || n_dev < 1 || n_dev > 64) return gfail(nullptr, TGP_ERR_ARG, "need 1..64 device ids");
  if (merge != TGP_MERGE_RCCL && merge != TGP_MERGE_PEER) return gfail(nullptr, TGP_ERR_ARG, "unknown merge %d", merge);
  DeviceGuard guard;
  // A device may be listed more than once only as a TEST AID (TGP_GROUP_ALLOW_DUPLICATES=1, peer merge): several
  // members then share one GPU, which exercises the whole multi-member path -- worker threads, shards, per-member
  // streams, peer copies + events, the P-way merge -- on a single-GPU box.  RCCL needs distinct devices.
  const bool allow_dup = getenv("TGP_GROUP_ALLOW_DUPLICATES") != nullptr && merge == TGP_MERGE_PEER;
  for (int i = 0; i < n_dev; ++i)
    for (int j = 0; j < i; ++j)
      if (device_ids[i] == device_ids[j] && !allow_dup)
        return gfail(nullptr, TGP_ERR_ARG, "device %d listed twice", device_ids[i]);
  tgp_group g = new (std::nothrow) tgp_group_s();
  if (!g) return gfail(nullptr, TGP_ERR_ALLOC, "host allocation failed");
  g->n = n_dev;
  g->d = d;
  g->kind = kernel_kind;
  g->merge = merge;
  g->devs.assign(device_ids, device_ids + n_dev);
  g->cand.resize(n_dev);
  g->pair.resize(n_dev);
  g->gather.resize(n_dev);
  g->ev.assign(n_dev, nullptr);
  auto bail = [&](int code, const std::string& msg) {
    g_group_create_error = msg;
    tgp_group_destroy(g);
    return code;
  };
  for (int i = 0; i < n_dev; ++i) {
    tgp_handle h = nullptr;
    int rc = tgp_create(device_ids[i], d, kernel_kind, &h);
    if (rc != TGP_OK) return bail(rc, std::string("member create: ") + tgp_last_error(nullptr));
    g->h.push_back(h);
    if ((rc = tgp_use_private_stream(h)) != TGP_OK) return bail(rc, std::string("private stream: ") + tgp_last_error(h));
    hipError_t e = hipSetDevice(device_ids[i]);
    if (e == hipSuccess) e = g->pair[i].reserve(2 * MAXV * sizeof(double));
    if (e == hipSuccess) e = g->gather[i].reserve(((size_t)n_dev + 1) * 2 * MAXV * sizeof(double));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming);
    if (e != hipSuccess) return bail(TGP_ERR_HIP, std::string("member buffers: ") + hipGetErrorString(e));
    Worker* w = new (std::nothrow) Worker();
    if (!w) return bail(TGP_ERR_ALLOC, "host allocation failed");
    w->start();
    g->workers.push_back(w);
  }
  if (merge == TGP_MERGE_RCCL) {
    std::string err;
    if (!g->rccl.load(err)) return bail(TGP_ERR_STATE, err + " (use TGP_MERGE_PEER for a group without RCCL)");
    g->comms.assign(n_dev, nullptr);
    ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), n_dev, g->devs.data());
    if (r != ncclSuccess) {
      g->comms.clear();
      return bail(TGP_ERR_HIP, std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r));
    }
    int cnt = 0;
    if (g->rccl.CommCount(g->comms[0], &cnt) == ncclSuccess) g->rccl_ranks = cnt;
    if (int rc = rccl_self_check(g)) return bail(rc, g->err);
  }
  (void)hipGetLastError();
  *out = g;
  return TGP_OK;
}

int tgp_group_destroy(tgp_group g) {
  if (!g) return TGP_OK;
  DeviceGuard guard;
  for (size_t i = 0; i < g->h.size(); ++i) {
    (void)hipSetDevice(g->devs[i]);
    if (g->h[i] && g->h[i]->stream) (void)hipStreamSynchronize(g->h[i]->stream);
  }
  for (size_t i = 0; i < g->comms.size(); ++i)
    if (g->comms[i]) (void)g->rccl.CommDestroy(g->comms[i]);
  for (Worker* w : g->workers) {
    w->shutdown();
    delete w;
  }
  for (size_t i = 0; i < g->devs.size(); ++i) {
    (void)hipSetDevice(g->devs[i]);
    if (i < g->cand.size()) g->cand[i].release();
    if (i < g->pair.size()) g->pair[i].release();
    if (i < g->gather.size()) g->gather[i].release();
    if (i < g->ev.size() && g->ev[i]) (void)hipEventDestroy(g->ev[i]);
  }
  for (tgp_handle h : g->h) (void)tgp_destroy(h);
  delete g;
  (void)hipGetLastError();
  return TGP_OK;
}

int tgp_group_info(tgp_group g, int* n_dev, int* merge, int* rccl_ranks) {
  if (!g) return TGP_ERR_ARG;
  if (n_dev) *n_dev = g->n;
  if (merge) *merge = g->merge;
  if (rccl_ranks) *rccl_ranks = g->rccl_ranks;
  return TGP_OK;
}

int tgp_group_member(tgp_group g, int i, tgp_handle* out) {
  if (!g || !out) return TGP_ERR_ARG;
  if (i < 0 || i >= g->n) return gfail(g, TGP_ERR_ARG, "member %d out of range (have %d)", i, g->n);
  *out = g->h[i];
  return TGP_OK;
}

int tgp_group_set_hyper(tgp_group g, double variance, const double* lengthscales, double noise_variance,
                        double mean_const) {
  if (!g) return TGP_ERR_ARG;
  for (int i = 0; i < g->n; ++i)
    if (int rc = tgp_set_hyper(g->h[i], variance, lengthscales, noise_variance, mean_const)) return member_rc(g, i, rc);
  return TGP_OK;
}

int tgp_group_set_data(tgp_group g, const double* X, const double* Y, int64_t N) {
  if (!g) return TGP_ERR_ARG;
  return run_all(g, [&](int i) { return tgp_set_data(g->h[i], X, Y, N, TGP_HOST); });
}

int tgp_group_append_data(tgp_group g, const double* Xnew, const double* Ynew, int64_t k) {
  if (!g) return TGP_ERR_ARG;
  return run_all(g, [&](int i) { return tgp_append_data(g->h[i], Xnew, Ynew, k, TGP_HOST); });
}

int tgp_group_set_candidates(tgp_group g, const double* Xq, int64_t M) {
  if (!g) return TGP_ERR_ARG;
  if (M < 1 || !Xq) return gfail(g, TGP_ERR_SHAPE, "need M >= 1 candidates");
  g->M = 0;
  shard(M, g->n, g->lo, g->hi);
  const int d = g->d;
  const int rc = run_all(g, [&](int i) -> int {
    const int64_t Mi = g->hi[i] - g->lo[i];
    if (Mi == 0) return TGP_OK;
    tgp_handle h = g->h[i];
    if (int r = host_set_device(h)) return r;
    hipError_t e = g->cand[i].reserve((size_t)Mi * d * sizeof(double));
    if (e == hipSuccess)
      e = hipMemcpyAsync(g->cand[i].p, Xq + g->lo[i] * d, (size_t)Mi * d * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    return e == hipSuccess ? TGP_OK : host_fail(h, TGP_ERR_HIP, "candidate shard upload: %s", hipGetErrorString(e));
  });
  if (rc == TGP_OK) g->M = M;
  return rc;
}

int tgp_group_sample_candidates(tgp_group g, uint64_t seed, int64_t M, const double* lower, const double* upper) {
  if (!g) return TGP_ERR_ARG;
  if (M < 1 || !lower || !upper) return gfail(g, TGP_ERR_ARG, "need M >= 1 and the box bounds");
  g->M = 0;
  shard(M, g->n, g->lo, g->hi);
  const int d = g->d;
  const int rc = run_all(g, [&](int i) -> int {
    const int64_t Mi = g->hi[i] - g->lo[i];
    if (Mi == 0) return TGP_OK;
    tgp_handle h = g->h[i];
    if (int r = host_set_device(h)) return r;
    const hipError_t e = g->cand[i].reserve((size_t)Mi * d * sizeof(double));
    if (e != hipSuccess) return host_fail(h, TGP_ERR_ALLOC, "candidate shard: %s", hipGetErrorString(e));
    return tgp_sample_box(h, seed, g->lo[i], Mi, lower, upper, g->cand[i].as<double>());
  });
  if (rc == TGP_OK) g->M = M;
  return rc;
}

int tgp_group_acq_argmax(tgp_group g, int acq_kind, double param, double* best_val, int64_t* best_idx, double* best_x) {
  if (!g) return TGP_ERR_ARG;
  DeviceGuard guard;
  if (int rc = need_candidates(g)) return rc;
  for (int i = 0; i < g->n; ++i) {  // enqueue only: the members' sweeps run concurrently on their devices
    const int64_t Mi = g->hi[i] - g->lo[i];
    if (Mi == 0) {  // more members than candidates: an empty shard contributes (NaN, -1), which never wins
      GHIP(g, hipSetDevice(g->devs[i]));
      GHIP(g, hipMemsetAsync(g->pair[i].p, 0xFF, 2 * sizeof(double), g->h[i]->stream));
      continue;
    }
    if (int rc = tgp_acq_argmax_async(g->h[i], acq_kind, param, g->cand[i].as<double>(), Mi, g->lo[i],
                                      g->pair[i].as<double>()))
      return member_rc(g, i, rc);
  }
  double v;
  int64_t idx;
  if (int rc = merge_and_fetch(g, 1, 0, &v, &idx)) return rc;
  if (best_val) *best_val = v;
  if (best_idx) *best_idx = idx;
  if (best_x) {
    if (idx < 0 || idx >= g->M) return gfail(g, TGP_ERR_HIP, "arg-max produced no valid index (all NaN?)");
    int owner = 0;
    while (owner + 1 < g->n && idx >= g->hi[owner]) ++owner;
    GHIP(g, hipSetDevice(g->devs[owner]));
    GHIP(g, hipMemcpy(best_x, g->cand[owner].as<double>() + (idx - g->lo[owner]) * g->d, g->d * sizeof(double),
                      hipMemcpyDeviceToHost));
  }
  return TGP_OK;
}

int tgp_group_acq_topk(tgp_group g, int acq_kind, double param, int k, double* vals, int64_t* idx) {
  if (!g) return TGP_ERR_ARG;
  if (int rc = need_candidates(g)) return rc;
  if (k < 1 || k > 1024 || !vals || !idx) return gfail(g, TGP_ERR_ARG, "k must be in 1..1024, outputs non-NULL");
  if (g->M < k) return gfail(g, TGP_ERR_SHAPE, "top-k needs M >= k (M=%lld, k=%d)", (long long)g->M, k);
  const int n = g->n;
  std::vector<double> pv((size_t)n * k);
  std::vector<int64_t> pi((size_t)n * k);
  std::vector<int> cnt(n, 0);
  const int rc = run_all(g, [&](int i) -> int {
    const int64_t Mi = g->hi[i] - g->lo[i];
    const int ki = (int)std::min<int64_t>(k, Mi);
    cnt[i] = ki;
    if (ki == 0) return TGP_OK;
    return tgp_acq_topk(g->h[i], acq_kind, param, g->cand[i].as<double>(), Mi, g->lo[i], ki, &pv[(size_t)i * k],
                        &pi[(size_t)i * k], TGP_DEVICE);
  });
  if (rc != TGP_OK) return rc;
  std::vector<std::pair<double, int64_t>> all;
  for (int i = 0; i < n; ++i)
    for (int t = 0; t < cnt[i]; ++t) all.emplace_back(pv[(size_t)i * k + t], pi[(size_t)i * k + t]);
  // value descending, ties by lower global index (tf.math.top_k over the unsharded values); NaNs last
  std::stable_sort(all.begin(), all.end(), [](const std::pair<double, int64_t>& a, const std::pair<double, int64_t>& b) {
    const bool an = a.first != a.first, bn = b.first != b.first;
    if (an != bn) return bn;
    if (a.first != b.first) return a.first > b.first;
    return a.second < b.second;
  });
  for (int t = 0; t < k; ++t) {
    vals[t] = all[t].first;
    idx[t] = all[t].second;
  }
  return TGP_OK;
}

int tgp_group_qei(tgp_group g, const double* Xq, int64_t G, int q, const double* eps, int S, double eta, double jitter,
                  double* out) {
  if (!g) return TGP_ERR_ARG;
  if (G < 0 || (G > 0 && (!Xq || !out))) return gfail(g, TGP_ERR_ARG, "bad arguments");
  if (G == 0) return TGP_OK;
  std::vector<int64_t> lo, hi;
  shard(G, g->n, lo, hi);
  const int d = g->d;
  return run_all(g, [&](int i) -> int {
    const int64_t Gi = hi[i] - lo[i];
    if (Gi == 0) return TGP_OK;
    return tgp_qei(g->h[i], Xq + lo[i] * q * d, Gi, q, eps, S, eta, jitter, out + lo[i], TGP_HOST);
  });
}

int tgp_group_traj_create(tgp_group g, const double* rff_W, const double* rff_b, int F, const double* w,
                          const double* xi, int B, tgp_group_traj* out) {
  if (!g || !out) return TGP_ERR_ARG;
  *out = nullptr;
  if (B < 1 || B > MAXV) return gfail(g, TGP_ERR_SHAPE, "a group trajectory set holds 1..%d trajectories, got %d", MAXV, B);
  tgp_group_traj t = new (std::nothrow) tgp_group_traj_s();
  if (!t) return gfail(g, TGP_ERR_ALLOC, "host allocation failed");
  t->g = g;
  t->B = B;
  t->t.assign(g->n, nullptr);
  const int rc = run_all(g, [&](int i) { return tgp_traj_create(g->h[i], rff_W, rff_b, F, w, xi, B, &t->t[i]); });
  if (rc != TGP_OK) {
    tgp_group_traj_destroy(t);
    return rc;
  }
  *out = t;
  return TGP_OK;
}

int tgp_group_traj_destroy(tgp_group_traj t) {
  if (!t) return TGP_OK;
  DeviceGuard guard;
  for (tgp_traj m : t->t)
    if (m) (void)tgp_traj_destroy(m);
  delete t;
  return TGP_OK;
}

int tgp_group_traj_argmin(tgp_group_traj t, double* best_val, int64_t* best_idx) {
  if (!t) return TGP_ERR_ARG;
  DeviceGuard guard;
  tgp_group g = t->g;
  if (int rc = need_candidates(g)) return rc;
  const int B = t->B;
  for (int i = 0; i < g->n; ++i) {
    const int64_t Mi = g->hi[i] - g->lo[i];
    if (Mi == 0) {
      GHIP(g, hipSetDevice(g->devs[i]));
      GHIP(g, hipMemsetAsync(g->pair[i].p, 0xFF, 2 * B * sizeof(double), g->h[i]->stream));
      continue;
    }
    if (int rc = tgp_traj_argmin_async(t->t[i], g->cand[i].as<double>(), Mi, g->lo[i], g->pair[i].as<double>()))
      return member_rc(g, i, rc);
  }
  return merge_and_fetch(g, B, 1, best_val, best_idx);
}

int tgp_group_last_kernel_ms(tgp_group g, double* ms) {
  if (!g || !ms) return TGP_ERR_ARG;
  DeviceGuard guard;
  double worst = 0.0;
  for (int i = 0; i < g->n; ++i) {
    double m = 0.0;
    int nl = 0;
    if (tgp_last_kernel_ms(g->h[i], &m, &nl) == TGP_OK) worst = std::max(worst, m);
  }
  *ms = worst;
  return TGP_OK;
}

}  // extern "C"
