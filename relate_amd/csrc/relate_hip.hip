// librelate_hip.so -- C ABI (include/relate_hip.h) over the gfx950 kernels of the all-pairs IBD HMM path.
// Host side mirrors the reference's per-pair operator relateHMM::init_globals / sgl
// (src/relateHMM.cpp:3371-3591) as batched calls.  The product path has no CPU fallback: every numeric
// result comes from the HIP kernels in kernels_*.hpp; host code only sequences launches and runs the
// optimiser's control logic (fit.hpp / lbfgsb.hpp).
#include "../../include/relate_hip.h"
#include "../../include/relate_hip_bench.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is looked up at run time (relate_hip_gather)

#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fit.hpp"
#include "kernels_emis.hpp"
#include "kernels_hmm.hpp"
#include "kernels_linkage.hpp"
#include "kernels_precalc.hpp"
#include "post_text.hpp"

using namespace relate;

namespace {

std::string g_create_error;

template <class T>
struct DBuf {  // grow-only device buffer
  T *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned host buffers of one pair population (grow-only) and the events of its launch in flight.
struct LaneBufs {
  int32_t *lane = nullptr;
  double *par = nullptr, *val = nullptr;
  size_t cap = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr, done = nullptr;
  int nl = 0, chunks = 1;
  int nv = 1;  // points per lane the slot layout of this launch allows (pick_nv)
  long npoints = 0, wanted = 0;  // points the optimisers asked for in this launch (without padding) / in this round (with those held back)
  hipError_t ensure(size_t n) {
    if (!e0) {
      hipError_t e = hipEventCreate(&e0);
      if (e == hipSuccess) e = hipEventCreate(&e1);
      if (e == hipSuccess) e = hipEventCreate(&done);
      if (e != hipSuccess) return e;
    }
    if (n <= cap) return hipSuccess;
    release_mem();
    size_t c = std::max<size_t>(n + n / 4, 4096);
    hipError_t e = hipHostMalloc((void **)&lane, c * sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&par, 3 * c * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void **)&val, c * sizeof(double));
    if (e == hipSuccess) cap = c;
    return e;
  }
  void release_mem() {
    if (lane) (void)hipHostFree(lane);
    if (par) (void)hipHostFree(par);
    if (val) (void)hipHostFree(val);
    lane = nullptr; par = val = nullptr; cap = 0;
  }
  void release() {
    release_mem();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (done) (void)hipEventDestroy(done);
    e0 = e1 = done = nullptr;
  }
};

// Storage of the optimiser runs of a fit: one slab per run (the run object, then the work matrices of its L-BFGS-B
// state), handed out from 168 MB chunks and recycled run to run and block to block.  A block of 8 192 pairs with 8
// restarts each has 65 536 runs alive at once = 5.6 GB; allocated object by object that was 1.4 million page faults
// per block (0.5 s of every fit's host time).  The chunks ask for huge pages (the box grants them on madvise).
constexpr size_t kFitRunBytes = 5376;
constexpr size_t kFitSlabBytes = kFitRunBytes + sizeof(double) * Lbfgsb::kSlab;
struct RunArena {
  static constexpr size_t kPerChunk = 2048;
  std::vector<void *> chunks;
  std::vector<unsigned char *> free_;
  unsigned char *take() {
    if (free_.empty()) {
      const size_t bytes = kPerChunk * kFitSlabBytes;
      void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (m == MAP_FAILED) return nullptr;
      (void)madvise(m, bytes, MADV_HUGEPAGE);
      chunks.push_back(m);
      for (size_t i = kPerChunk; i-- > 0;) free_.push_back((unsigned char *)m + i * kFitSlabBytes);
    }
    unsigned char *s = free_.back();
    free_.pop_back();
    return s;
  }
  void give(unsigned char *s) { free_.push_back(s); }
  void release() {
    for (void *m : chunks) munmap(m, kPerChunk * kFitSlabBytes);
    chunks.clear();
    free_.clear();
  }
};

}  // namespace

struct relate_hip_ctx {
  int dev = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  std::string err;
  relate_opts o{};
  int back2 = 0;
  bool loaded = false, ready = false;
  int N = 0, Mraw = 0, M = 0, W = 0;
  int useLD = 0;
  std::vector<uint8_t> h_raw;
  std::vector<uint8_t> h_bed;    // relate_hip_load_bed: SNP-major .bed payload (h_raw stays empty)
  std::vector<int> h_bedcols;    // its rows in analysis order
  int bedBps = 0;
  std::vector<double> h_pos_raw;
  std::vector<int> h_chr_raw;
  std::vector<int> keep;  // Mraw: pre_calc_used_list
  std::vector<double> h_pos, h_maf;
  // device: data set
  DBuf<uint64_t> dG, dGt;  // SNP-major / individual-major 2-bit genotypes
  int Wt = 0;
  bool winMode = false;      // stage_pairs_win_kernel + depth table (back2 <= 32)
  DBuf<double> dPos, dMaf, dMea, dPBA, dPbA, dPBa, dPba, dTab1, dTabL, dTabD;
  DBuf<double> dTsh;       // shared distances tsh[s] = pos[s] - pos[s-1] (negative: 1e7; tsh[0] = 0), M + 2 entries
  DBuf<int> dBnd;          // chromosome breaks: the SNPs s with pos[s] < pos[s-1]
  int nBnd = 0;
  DBuf<int> dChoose;
  DBuf<uint8_t> dOrder;  // per-SNP preference order of the conditioning depths (windowed staging)
  // device: staged pair block
  DBuf<double> dRec;            // staged records (S1/S0, S2/S0; layout: kernels_emis.hpp), padded pair count recPitch
  DBuf<double> dRecV;           // the full format (t, S0, S1, S2) of a block: Viterbi and inspection only
  DBuf<int32_t> dPairs;
  std::vector<int32_t> staged;  // pairs of the block currently in dRec (2 ints per pair)
  std::vector<int32_t> stagedV; // ... in dRecV
  int recPitch = 0, recPitchV = 0;  // plane pitch in doubles (>= pairs of the staged block)
  DBuf<double> dScM, dPairM;    // the factor taken out of the records: per-run partial products, per-pair result
  DBuf<int> dScE, dScN, dPairE; // ... exponents, kept-SNP counts
  DBuf<int> dNsnp;              // SNPs kept per pair of the staged block
  DBuf<int> dPairOwn;           // per column: 1 if the pair has records that carry their own distance (boundary_fixup_kernel)
  std::vector<int> h_nsnp;      // the same on the host
  // device: work
  DBuf<int32_t> dLanePair;
  DBuf<double> dParams, dOut, dPost;
  DBuf<unsigned char> dCk;  // tile checkpoints of the decode (kernels_hmm.hpp, TileCkpt)
  DBuf<uint8_t> dBp;
  DBuf<int8_t> dPath;
  DBuf<double> dCm, dAstart, dBend;  // chunk products and boundary vectors of the chunked sweeps
  DBuf<int> dCe;
  LaneBufs lanebufs[2];  // pinned lane buffers of the two pair populations of a fit
  void *bounce[2] = {nullptr, nullptr};  // pinned staging of results on their way to pageable caller memory
  hipEvent_t bounceEv[2] = {nullptr, nullptr};
  double timings[4] = {0, 0, 0, 0};
  double like_stats[6] = {0, 0, 0, 0, 0, 0};  // relate_hip_last_like_stats
  int last_chunks[2] = {1, 1};  // SNP-axis chunks of the last like / decode launch
  long slots_like = 2048;                             // resident waves of the chunked sweeps (set at create)
  long slots_decode[4] = {6144, 6144, 6144, 6144};    // by the rows pass B writes (0 = the reference's three): its LDS tile sets its occupancy
  ncclComm_t comm = nullptr;  // relate_hip_gather: one communicator per handle, made at the first gather
  int comm_rank = -1, comm_n = 0;
  long gather_seq = 0;        // gathers issued on this handle (names the rendezvous files of each)
  int gather_ranks_seen = 0;  // ranks the last gather's communicator (or file session) held: relate_hip_gather_ranks
  DBuf<unsigned char> dGatherS, dGatherR;
  unsigned char *scratch_dev = nullptr;  // device address of `scratch` on this handle's GPU (relate_hip_like_noalloc)
  long single_sweep_lanes = 45056;              // launches below this many lanes are cut along the SNP axis (pick_chunks)
  RunArena run_arena;
  // rolling refills of a long fit (FitPop::roll): pinned pair lists and kept-SNP counts of the segments being staged, and
  // the event behind their staging work
  int32_t *refill_pairs = nullptr;
  int *refill_nsnp = nullptr;
  size_t refill_cap = 0;
  hipEvent_t refill_ev = nullptr;
  int fit_eager = -1;  // what the previous fit of this data set saw: 1 = first waves of restarts rarely settle a pair (FitPop::eager)
};

namespace {

int fail(relate_hip_t h, int code, const std::string &msg) {
  if (h) h->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIPCHK(h, expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(h, e_ == hipErrorOutOfMemory ? RELATE_E_NOMEM : RELATE_E_HIP,                 \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)

inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// Host-side control work of the batched optimiser (one L-BFGS-B step and the objective wrappers of every active run
// per round) spread over a few threads: runs are independent objects.  RELATE_HOST_THREADS overrides (default: up to 8, or OMP_NUM_THREADS if that is larger).
int host_threads() {
  static int n = [] {
    int v = 0;
    if (const char *e = getenv("RELATE_HOST_THREADS")) v = atoi(e);
    if (v < 1) {
      v = (int)std::thread::hardware_concurrency();
      if (v > 8) v = 8;
      // a larger share granted through OMP_NUM_THREADS is taken (up to 32); a smaller value is not a limit meant for
      // this library (torchrun sets it to 1 for every rank)
      if (const char *e = getenv("OMP_NUM_THREADS")) {
        int w = atoi(e);
        if (w > v) v = w > 32 ? 32 : w;
      }
    }
    return v < 1 ? 1 : v;
  }();
  return n;
}
// The workers are started once and kept (a round of a fit makes several passes of a few hundred microseconds each; a
// thread start costs a good part of that).  One pass at a time: a second caller (another handle's host thread) or a
// forked child, which has no workers, starts threads of its own as before.  Never destroyed: the workers sleep on a
// condition variable until the process ends.
class HostPool {
 public:
  explicit HostPool(int nt) : pid_(getpid()) {
    for (int t = 1; t < nt; t++) th_.emplace_back([this, t] { worker(t); });
  }
  bool try_run(size_t n, const std::function<void(size_t, size_t)> &fn) {
    if (getpid() != pid_) return false;
    std::unique_lock<std::mutex> use(use_, std::try_to_lock);
    if (!use.owns_lock()) return false;
    const int nt = (int)th_.size() + 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &fn; n_ = n; per_ = (n + nt - 1) / nt; pending_ = nt - 1; gen_++;
    }
    cv_work_.notify_all();
    fn((size_t)0, std::min(n, per_));
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    return true;
  }

 private:
  void worker(int t) {
    long seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_work_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      const std::function<void(size_t, size_t)> *job = job_;
      const size_t lo = (size_t)t * per_, hi = std::min(n_, lo + per_);
      lk.unlock();
      if (lo < hi) (*job)(lo, hi);
      lk.lock();
      if (--pending_ == 0) cv_done_.notify_one();
    }
  }
  pid_t pid_;
  std::vector<std::thread> th_;
  std::mutex mu_, use_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(size_t, size_t)> *job_ = nullptr;
  size_t n_ = 0, per_ = 0;
  long gen_ = 0;
  int pending_ = 0;
};
HostPool &host_pool() {  // one pool for the process (not one per call site of the template below)
  static HostPool *pool = new HostPool(host_threads());
  return *pool;
}
template <class F>
void parallel_for(size_t n, F &&fn, size_t serial_below = 4096) {  // fn(begin, end)
  int nt = host_threads();
  if (n < serial_below || nt == 1) { fn((size_t)0, n); return; }
  {
    const std::function<void(size_t, size_t)> job = [&fn](size_t lo, size_t hi) { fn(lo, hi); };
    if (host_pool().try_run(n, job)) return;
  }
  std::vector<std::thread> th;
  size_t per = (n + nt - 1) / nt;
  for (int t = 0; t < nt; t++) {
    size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
  }
  for (auto &x : th) x.join();
}

// Padded pair count of a staged block (a SNP row holds kRecPlanes x this many doubles): P rounded up to whole groups of
// 64 pairs, and from 256 pairs on to an odd number of groups, so that consecutive SNP rows never lie a power of two
// apart (HBM channel camping of the strided sweeps).
inline int rec_pitch(int P) {
  int u = (P + 63) / 64;
  if (P >= 256 && (u & 1) == 0) u++;
  return u * 64;
}

int set_device(relate_hip_t h) {
  HIPCHK(h, hipSetDevice(h->dev));
  return 0;
}

// Largest pair block whose buffers fit.  Every MemNeed is one group of grow-only buffers: the bytes it takes per
// (pair, SNP) and what it holds already; growing a group costs only what it lacks.  The sum of what all groups lack
// must fit in 70 % of the memory that is free now.
struct MemNeed {
  double per_unit;
  size_t held;
};
size_t pair_block_limit(relate_hip_t h, std::initializer_list<MemNeed> needs) {
  size_t freeb = 0, total = 0;
  if (hipMemGetInfo(&freeb, &total) != hipSuccess) freeb = (size_t)8 << 30;
  const double budget = 0.70 * (double)freeb;
  auto lacks = [&](size_t L) {
    double tot = 0;
    for (const MemNeed &n : needs) {
      double want = (n.per_unit * (double)h->M + 256.0) * (double)L;
      if (want > (double)n.held) tot += want - (double)n.held;
    }
    return tot;
  };
  size_t lo = 1, hi = (size_t)1 << 22;
  if (lacks(hi) <= budget) return hi;
  while (hi - lo > 1) {
    size_t mid = lo + (hi - lo) / 2;
    if (lacks(mid) <= budget) lo = mid;
    else hi = mid;
  }
  return lo;
}
// bytes per (pair, SNP) of the buffer groups
constexpr double kRecBytes = 8.0 * kRecPlanes + 0.5;          // records + the per-run partials of the staging kernels
constexpr double kRecFullBytes = 8.0 * kRecPlanesFull + 2.0;  // full-format records + Viterbi back pointers and path
constexpr double kCkptBytes = 84.0 / 16 + 0.25;               // tile checkpoints (kernels_hmm.hpp, TileCkpt)
inline MemNeed need_rec(relate_hip_t h) { return {kRecBytes, h->dRec.cap * sizeof(double)}; }
inline MemNeed need_ckpt(relate_hip_t h) { return {kCkptBytes, h->dCk.cap}; }
inline MemNeed need_post(relate_hip_t h, double bytes) { return {bytes, h->dPost.cap * sizeof(double)}; }
inline MemNeed need_path(relate_hip_t h, bool want) {
  return {want ? kRecFullBytes : 0.0, h->dRecV.cap * sizeof(double) + h->dBp.cap + h->dPath.cap};
}

// K4: n pairs into columns [col0, col0 + n) of a record buffer whose padded pair count is RP (full: dRecV, else dRec
// together with its per-pair factors and kept-SNP counts).  Returns with the work done and the counts on the host.
int check_pairs(relate_hip_t h, const int32_t *pairs, long n) {
  for (long p = 0; p < n; p++) {
    int a = pairs[2 * p], b = pairs[2 * p + 1];
    if (a < 0 || a >= h->N || b < 0 || b >= h->N)
      return fail(h, RELATE_E_ARG, "pair index out of range (the data set has " + std::to_string(h->N) + " individuals)");
  }
  return 0;
}
// the staging work queued on the library's stream, nothing waited for; d_pairs = where the pair list goes on the device
// (a list still being read by an earlier launch of the queue must not be overwritten: callers that queue several
// stagings give each its own part of dPairs)
int stage_cols_enqueue(relate_hip_t h, const int32_t *pairs, int n, int col0, int RP, bool full, int32_t *d_pairs) {
  DBuf<double> &R = full ? h->dRecV : h->dRec;
  const int words = 2, chunk = 64;  // SNPs per staging thread: 64 either way
  const int nruns = h->winMode ? (h->Wt + words - 1) / words : (h->M + chunk - 1) / chunk;
  HIPCHK(h, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice, h->stream));
  if (h->winMode) {
    unsigned grid = (unsigned)((size_t)nruns * ((n + 255) / 256));
#define STAGE_WIN(FULL)                                                                                              \
  hipLaunchKernelGGL(stage_pairs_win_kernel<FULL>, dim3(grid), dim3(256), 0, h->stream, h->dGt.p, h->M, h->Wt,         \
                     d_pairs, n, words, h->dPos.p, h->dTab1.p, h->dTabD.p, h->dMea.p, h->dChoose.p, h->dOrder.p, \
                     h->useLD, h->o.back, h->back2, R.p, RP, col0, h->dScM.p, h->dScE.p, h->dScN.p)
    if (full) STAGE_WIN(true);
    else STAGE_WIN(false);
#undef STAGE_WIN
  } else {
    static_assert(kStageRun == 64, "one bit per SNP of a run");
    const unsigned ggrid = (unsigned)((size_t)nruns * ((n + 63) / 64));
#define STAGE_GEN(FULL)                                                                                               \
  hipLaunchKernelGGL(stage_pairs_kernel<FULL>, dim3(ggrid), dim3(64), 0, h->stream, h->dG.p, h->dGt.p, h->Wt, h->M,   \
                     h->W, d_pairs, n, h->dPos.p, h->dTab1.p, h->dTabL.p, h->dChoose.p, h->dMea.p,                    \
                     h->dPBA.p, h->dPbA.p, h->dPBa.p, h->dPba.p, h->useLD, h->o.back, h->back2, h->o.epsilon, R.p,    \
                     RP, col0, h->dScM.p, h->dScE.p, h->dScN.p)
    if (full) STAGE_GEN(true);
    else STAGE_GEN(false);
#undef STAGE_GEN
  }
  HIPCHK(h, hipGetLastError());
  if (!full) {
    hipLaunchKernelGGL(scale_reduce_kernel, dim3(blocks_for(n, 64)), dim3(64 * kScaleSeg), 0, h->stream, h->dScM.p,
                       h->dScE.p, h->dScN.p, nruns, RP, col0, n, h->dPairM.p, h->dPairE.p, h->dNsnp.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemsetAsync(h->dPairOwn.p + col0, 0, sizeof(int) * n, h->stream));
    if (h->nBnd > 0) {  // runs of dropped SNPs across a chromosome break (kernels_emis.hpp); after the counts: a pair
                        // without a kept SNP would have every thread of it scan the whole row for one
      hipLaunchKernelGGL(boundary_fixup_kernel, dim3(blocks_for((long)n * h->nBnd, 256)), dim3(256), 0, h->stream,
                         h->dG.p, h->M, h->W, d_pairs, n, h->dBnd.p, h->nBnd, h->dPos.p, R.p, RP, col0,
                         h->dPairOwn.p, h->dNsnp.p);
      HIPCHK(h, hipGetLastError());
    }
  }
  return 0;
}
int stage_cols(relate_hip_t h, const int32_t *pairs, int n, int col0, int RP, bool full) {
  int rc = check_pairs(h, pairs, n);
  if (rc) return rc;
  HIPCHK(h, h->dPairs.ensure((size_t)2 * n));
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  rc = stage_cols_enqueue(h, pairs, n, col0, RP, full, h->dPairs.p);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  if (!full)
    HIPCHK(h, hipMemcpyAsync(h->h_nsnp.data() + col0, h->dNsnp.p + col0, sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->timings[0] = ms;
  return 0;
}

// K4 for a block of pairs (skipped when the same block is already staged).  full: the four-plane format into dRecV
// (Viterbi, inspection); the block in dRec and its per-pair factors stay as they are.
int stage_block(relate_hip_t h, const int32_t *pairs, int P, bool full = false) {
  std::vector<int32_t> &memo = full ? h->stagedV : h->staged;
  if ((int)memo.size() == 2 * P && std::memcmp(memo.data(), pairs, sizeof(int32_t) * 2 * P) == 0) {
    h->timings[0] = 0;
    return 0;
  }
  memo.clear();
  const int RP = rec_pitch(P);
  DBuf<double> &R = full ? h->dRecV : h->dRec;
  (full ? h->recPitchV : h->recPitch) = RP;
  const int nruns = h->winMode ? (h->Wt + 1) / 2 : (h->M + 63) / 64;
  HIPCHK(h, R.ensure((size_t)((h->M + 1) / 2) * 2 * (full ? kRecPlanesFull : kRecPlanes) * RP));
  HIPCHK(h, h->dScM.ensure((size_t)nruns * RP));
  HIPCHK(h, h->dScE.ensure((size_t)nruns * RP));
  HIPCHK(h, h->dScN.ensure((size_t)nruns * RP));
  if (!full) {
    HIPCHK(h, h->dPairM.ensure(RP));
    HIPCHK(h, h->dPairE.ensure(RP));
    HIPCHK(h, h->dNsnp.ensure(RP));
    HIPCHK(h, h->dPairOwn.ensure(RP));
    h->h_nsnp.resize(P);
  }
  int rc = stage_cols(h, pairs, P, 0, RP, full);
  if (rc) return rc;
  memo.assign(pairs, pairs + 2 * P);
  return 0;
}

// SNP-axis chunking.  A sweep lane carries a serial chain of M steps; below about two waves per SIMD the
// machine idles, so the SNP axis is cut until there are enough threads -- but a chunked step costs ~2x a plain
// one, so a launch that already has >= ~0.7 waves per SIMD stays a single sweep (relate_hip_ctx::single_sweep_lanes).  When cutting, the number of
// chunks is the largest one whose waves still fit `rounds` full turns of the `slots` waves the kernels of the launch
// can hold at once (hipOccupancy at create time; for decode the least common multiple over its three kernels): one
// chunk more would add a turn that runs nearly empty (a seventh of the run time at 6 turns).  The likelihood uses
// 12 turns; decode one turn of the common multiple (its combine kernel walks the chunks serially per pair and its
// sweeps are HBM-bound, not latency-bound).  Chunks are multiples of 16 SNPs (the posterior tiles of
// decode_bwd_kernel) and at least 512 SNPs long.
// Launches with at least this many lanes run as one sweep per lane; fewer are cut along the SNP axis.  Measured
// crossover (M = 1e5): a lone wave per SIMD walks its chain in 10.4 ms whatever the lane count, the chunked path
// sustains ~420 G lane-steps/s, equal at ~44 000 lanes.  RELATE_SINGLE_SWEEP_LANES overrides (tuning experiments).

void pick_chunks(int M, long lanes, long single_sweep_lanes, int wg_threads, long slots, int rounds, int *nchunk,
                 int *chunk_len) {
  long c = 1;
  if (lanes < single_sweep_lanes) {
    long wg = (lanes + wg_threads - 1) / wg_threads;
    long waves_per_chunk = wg * ((wg_threads + 63) / 64);
    c = (slots * rounds) / std::max(1L, waves_per_chunk);
  }
  long cmax = std::max(1, M / 512);
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  int len = (int)((M + c - 1) / c);
  len = (len + 15) / 16 * 16;
  *chunk_len = len;
  *nchunk = (M + len - 1) / len;
}

long lcm(long a, long b) {
  long x = a, y = b;
  while (y) { long t = x % y; x = y; y = t; }
  return a / x * b;
}

// waves a kernel can hold on the whole device at once
template <class K>
long wave_slots(K kernel, int block, int ncu) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess || nb < 1) nb = 1;
  return (long)nb * ((block + 63) / 64) * ncu;
}

inline int pad256(int n) { return (n + 255) / 256 * 256; }

int ensure_chunk_buffers(relate_hip_t h, size_t n, ChunkOut *O) {
  HIPCHK(h, h->dCm.ensure((size_t)9 * n));
  HIPCHK(h, h->dCe.ensure((size_t)3 * n));
  O->m = h->dCm.p;
  O->e = h->dCe.p;
  return 0;
}

// Scratch of relate_hip_like_noalloc: chunk products (84 B per lane and chunk) or the half vectors of the two-sided
// sweep (96 B per lane).  One copy per GPU, held until the process ends; its device address is looked up once per
// handle (relate_hip_create) -- in host code the array's name is not a device pointer.
constexpr size_t kNoAllocScratchBytes = (size_t)64 << 20;
constexpr long kNoAllocMaxLanes = 1L << 20;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

// Where a likelihood launch finds its lanes and leaves its values, and where its chunk products / half vectors go.
struct LikeIO {
  LaneMap map;
  const double *params;  // device, 3 per lane
  double *out;           // device, 1 per lane
  unsigned char *fixed;  // nullptr: the handle's grow-only buffers (may allocate); else a scratch of fixed_bytes
  size_t fixed_bytes;
  int nv = 1;            // every run of nv consecutive slots (from a multiple of nv) belongs to one pair: a lane may carry them
};

// Points per lane for a launch of `slots` points whose layout allows up to nv_max (kernels_hmm.hpp, LaneSet).  Two
// chains per lane pay once the lanes still fill the SIMDs: measured at M = 5e5 (10 000 pairs), 60 000 points run at 701 G
// point-steps/s one per lane (two-sided, 1.8 waves per SIMD) and 706 G two per lane; 120 000 points at 799 G against
// 874 G; 240 000 points at 870 G against 986 G (92 % of what a pure FMA stream sustains on this part).  Three per lane
// need more than 256 VGPRs.  RELATE_LIKE_NV overrides (tuning, tests).
constexpr long kTwoPerLaneFrom = 98304;
int pick_nv(long slots, int nv_max) {
  static const int forced = [] { const char *e = getenv("RELATE_LIKE_NV"); return e ? atoi(e) : 0; }();
  int nv = forced > 0 ? forced : (slots > kTwoPerLaneFrom ? 2 : 1);
  if (nv > nv_max) nv = nv_max;
  if (nv > 2) nv = 2;
  if (nv < 1) nv = 1;
  return nv;
}

// K5 launches for nl lanes of the staged block
int launch_like(relate_hip_t h, int nl, const LikeIO &io) {
  int nchunk, clen;
  pick_chunks(h->M, nl, h->single_sweep_lanes, 256, h->slots_like, 12, &nchunk, &clen);
  const int nlp = pad256(nl);
  if (io.fixed && nchunk > 1) {  // as many chunks as the fixed scratch holds products for
    long cmax = (long)(io.fixed_bytes / 84 / (size_t)nlp);
    if (cmax < nchunk) {
      if (cmax < 2) { nchunk = 1; clen = h->M; }
      else {
        int len = (int)((h->M + cmax - 1) / cmax);
        clen = (len + 15) / 16 * 16;
        nchunk = (h->M + clen - 1) / clen;
      }
    }
  }
  h->last_chunks[0] = nchunk;
  const bool dr = h->o.double_recom != 0;
  const PairScale scale{h->dPairM.p, h->dPairE.p};
  if (nchunk == 1) {
    const int NV = io.nv;
    const long lanes = (nl + NV - 1) / NV;
    const int nsp = (nl + 256 * NV - 1) / (256 * NV) * (256 * NV);  // slots padded to whole workgroups
    // ring depth by the waves per SIMD this launch can fill (kernels_hmm.hpp, like_kernel)
    const long simd_lanes = 256L * 4 * 64;
    // Up to one wave per SIMD: the two-sided sweep instead (twice the lanes, half the length, like_halves_kernel).
    // RELATE_LIKE_TWO_SIDED=0 keeps the single sweep (tests, tuning).
    static const bool two_sided = [] { const char *e = getenv("RELATE_LIKE_TWO_SIDED"); return !(e && atoi(e) == 0); }();
    // One point per lane: up to 1.5 turns of lanes (three waves per SIMD of half lanes: the shallow-ring build holds
    // three, 148 VGPRs); a plain sweep of 1.0-1.5 waves per SIMD would cost what two full waves cost.
    const long two_sided_max = NV == 1 ? simd_lanes * 3 / 2 : simd_lanes;
    if (two_sided && lanes <= two_sided_max && h->M >= 64 && (!io.fixed || (size_t)12 * nsp * sizeof(double) <= io.fixed_bytes)) {
      double *part = reinterpret_cast<double *>(io.fixed);
      if (!io.fixed) {
        HIPCHK(h, h->dCm.ensure((size_t)12 * nsp));
        part = h->dCm.p;
      }
      dim3 grid((unsigned)(2 * (nsp / (256 * NV))));
      const bool shallow = lanes > simd_lanes;
#define LAUNCH_HALVES(DR, NB, NVV)                                                                                     \
  hipLaunchKernelGGL((like_halves_kernel<DR, NB, NVV>), grid, dim3(256), 0, h->stream, h->dRec.p, h->dTsh.p, h->dPairOwn.p, h->M, h->recPitch,   \
                     io.map, io.params, nl, nsp, h->o.alim[1], part)
      if (dr) {
        if (NV == 2) LAUNCH_HALVES(true, 4, 2);
        else if (shallow) LAUNCH_HALVES(true, 2, 1);
        else LAUNCH_HALVES(true, 4, 1);
      } else {
        if (NV == 2) LAUNCH_HALVES(false, 4, 2);
        else if (shallow) LAUNCH_HALVES(false, 2, 1);
        else LAUNCH_HALVES(false, 4, 1);
      }
#undef LAUNCH_HALVES
      HIPCHK(h, hipGetLastError());
      hipLaunchKernelGGL(like_join_kernel, dim3(blocks_for(nl, 256)), dim3(256), 0, h->stream, part, io.map, scale,
                         io.params, nl, nsp, h->o.alim[1], io.out);
      HIPCHK(h, hipGetLastError());
      return 0;
    }
    const int nbuf = lanes <= simd_lanes ? 6 : (lanes <= 2 * simd_lanes ? 4 : 2);
    dim3 grid(blocks_for(lanes, 256)), block(256);
#define LAUNCH_LIKE(DR, NB, NVV)                                                                                   \
  hipLaunchKernelGGL((like_kernel<DR, NB, NVV>), grid, block, 0, h->stream, h->dRec.p, h->dTsh.p, h->dPairOwn.p, h->M, h->recPitch, io.map, \
                     scale, io.params, nl, h->o.alim[1], io.out)
#define LAUNCH_LIKE_NV(DR)                                      \
  do {                                                          \
    if (NV == 2) {                                              \
      if (nbuf == 2) LAUNCH_LIKE(DR, 2, 2);                     \
      else LAUNCH_LIKE(DR, 4, 2);                               \
    } else {                                                    \
      if (nbuf == 6) LAUNCH_LIKE(DR, 6, 1);                     \
      else if (nbuf == 4) LAUNCH_LIKE(DR, 4, 1);                \
      else LAUNCH_LIKE(DR, 2, 1);                               \
    }                                                           \
  } while (0)
    if (dr) LAUNCH_LIKE_NV(true);
    else LAUNCH_LIKE_NV(false);
#undef LAUNCH_LIKE_NV
#undef LAUNCH_LIKE
    HIPCHK(h, hipGetLastError());
    return 0;
  }
  size_t n = (size_t)nlp * nchunk;
  ChunkOut O;
  if (io.fixed) {
    O.m = reinterpret_cast<double *>(io.fixed);
    O.e = reinterpret_cast<int *>(io.fixed + (size_t)72 * n);
  } else {
    int rc = ensure_chunk_buffers(h, n, &O);
    if (rc) return rc;
  }
  dim3 grid((unsigned)((size_t)(nlp / 256) * nchunk));
  if (dr)
    hipLaunchKernelGGL((chunk_matrix_kernel<true, false>), grid, dim3(256), 0, h->stream, h->dRec.p, h->dTsh.p, h->dPairOwn.p, h->M, h->recPitch,
                       io.map, io.params, nl, nlp, nchunk, clen, h->o.alim[1], 1, O, TileCkpt{nullptr, 0}, 0);
  else
    hipLaunchKernelGGL((chunk_matrix_kernel<false, false>), grid, dim3(256), 0, h->stream, h->dRec.p, h->dTsh.p, h->dPairOwn.p, h->M, h->recPitch,
                       io.map, io.params, nl, nlp, nchunk, clen, h->o.alim[1], 1, O, TileCkpt{nullptr, 0}, 0);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(like_combine_kernel, dim3(blocks_for(nl, 256)), dim3(256), 0, h->stream, O, io.map, scale,
                     io.params, nl, nlp, nchunk, h->o.alim[1], dr ? 1 : 0, io.out);
  HIPCHK(h, hipGetLastError());
  return 0;
}
// lanes already uploaded to dLanePair/dParams; result in dOut (device)
int launch_like(relate_hip_t h, int P, int nl, int nv_max = 1) {
  (void)P;
  return launch_like(h, nl, LikeIO{LaneMap{h->dLanePair.p, 1}, h->dParams.p, h->dOut.p, nullptr, 0, pick_nv(nl, nv_max)});
}
// E points per pair in order: the largest lane width that never straddles two pairs
inline int nv_for_points(int E) { return E % 2 == 0 ? 2 : 1; }

// K5 on the staged block: nl lanes, host arrays in, host array out
int like_lanes(relate_hip_t h, int P, const int32_t *lane_pair, const double *params, int nl, double *out,
               int nv_max = 1) {
  if (nl == 0) return 0;
  HIPCHK(h, h->dLanePair.ensure(nl));
  HIPCHK(h, h->dParams.ensure((size_t)3 * nl));
  HIPCHK(h, h->dOut.ensure(nl));
  HIPCHK(h, hipMemcpyAsync(h->dLanePair.p, lane_pair, sizeof(int32_t) * nl, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dParams.p, params, sizeof(double) * 3 * nl, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  int rc = launch_like(h, P, nl, nv_max);
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  HIPCHK(h, hipMemcpyAsync(out, h->dOut.p, sizeof(double) * nl, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->timings[1] += ms;
  if (getenv("RELATE_FIT_TRACE"))
    std::fprintf(stderr, "like launch: %d lanes, %d chunks, %.3f ms\n", nl, h->last_chunks[0], ms);
  return 0;
}

int require_ready(relate_hip_t h) {
  if (!h) return fail(nullptr, RELATE_E_ARG, "null handle");
  if (!h->ready) return fail(h, RELATE_E_ARG, "call relate_hip_load and relate_hip_precalc first");
  return set_device(h);
}

// ---- batched run_optimization for the staged block ---------------------------------------------------------------
// K5 for the lanes in B (pinned), queued on the library's stream without waiting: upload, kernels, download, event.
// The device lane buffers are shared by the populations; stream order keeps one launch's use of them apart from the
// next one's upload.
int like_enqueue(relate_hip_t h, int P, LaneBufs &B) {
  const int nl = B.nl;
  HIPCHK(h, h->dLanePair.ensure(nl));
  HIPCHK(h, h->dParams.ensure((size_t)3 * nl));
  HIPCHK(h, h->dOut.ensure(nl));
  HIPCHK(h, hipMemcpyAsync(h->dLanePair.p, B.lane, sizeof(int32_t) * nl, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dParams.p, B.par, sizeof(double) * 3 * nl, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(B.e0, h->stream));
  int rc = launch_like(h, nl, LikeIO{LaneMap{h->dLanePair.p, 1}, h->dParams.p, h->dOut.p, nullptr, 0, B.nv});
  if (rc) return rc;
  B.chunks = h->last_chunks[0];
  HIPCHK(h, hipEventRecord(B.e1, h->stream));
  HIPCHK(h, hipMemcpyAsync(B.val, h->dOut.p, sizeof(double) * nl, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(B.done, h->stream));
  return 0;
}
int like_wait(relate_hip_t h, LaneBufs &B) {
  HIPCHK(h, hipEventSynchronize(B.done));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, B.e0, B.e1));
  h->timings[1] += ms;
  {
    const int k = B.chunks > 1 ? 1 : 0;  // 0: one or two sweeps per point, 1: cut along the SNP axis (three-row products)
    h->like_stats[k] += (double)B.nl * h->M;
    h->like_stats[2 + k] += ms;
    h->like_stats[4 + k] += 1;
  }
  if (getenv("RELATE_FIT_TRACE"))
    std::fprintf(stderr, "like launch: %d lanes, %d chunks, %.3f ms, %ld points of %ld wanted, nv %d, t %.6f\n", B.nl, B.chunks, ms,
                 B.npoints, B.wanted, B.nv,
                 std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
  return 0;
}

// One population of pairs of the staged block with its optimiser state: prepare() advances the multi-start rule,
// starts and retires runs and lays out the lanes of the next likelihood launch; consume() feeds the values back.
constexpr long kCompactBelowPoints = 45056;  // launches below this size are cut along the SNP axis (pick_chunks)

struct FitPop {
  struct Run {
    FindMax fm;
    int slot[7];            // for each requested point: lane index, or -1 if the wrapper short-circuits
    double akk[7][3];       // the request mapped to (a, k2, k1)
    bool ok[7];
    bool bad = false;
    double *matrices() { return reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(this) + kFitRunBytes); }
  };
  RunArena *arena = nullptr;
  void drop(Run *r) {  // the run's slab goes back to the arena
    r->~Run();
    arena->give(reinterpret_cast<unsigned char *>(r));
  }
  // What the scheduling passes of a round read of a run, side by side (the runs themselves are heap objects of tens of
  // KB each: with 65 536 of them a serial pass over their fields is a cache miss per run).
  struct Act {
    Run *r;
    int pair, idx;          // pair = index inside the population
    int nlanes;             // points of its request that need the GPU
    size_t lane0;           // first lane of this run in the launch
    bool waited;            // its request was held back in the previous round
    bool taken;             // its request is part of the launch being laid out
    bool done;              // fm.done, as of the last consume()
    signed char flight = -1;  // lane buffer of the launch its request is in (not answered yet), or -1
  };
  int p0 = 0, P = 0;  // pairs [p0, p0 + P) of the staged block
  const relate_opts *o = nullptr;
  FitSpec S{};
  int n = 0, R = 1;
  double lo[3], hi[3];
  const double *starts = nullptr;
  long pair_base = 0;
  std::vector<MultiStart> ms;
  std::vector<GlibcRand> rng;
  std::vector<std::vector<double>> results;
  std::vector<std::vector<char>> have;
  std::vector<Act> active;
  bool first_round = true, inflight = false, finished = false;
  // Two launches may be in the air at once (long rows, one population): while one runs, the host feeds the answers of
  // the previous one to the optimisers and lays out the next from the requests that are pending -- in the steady state
  // of a block two thirds of the runs wait for the launch budget anyway, so the next launch is known before the current
  // one ends, and the GPU goes from one to the next without the host in between.  B = the buffer being laid out /
  // consumed; Bs[k] = lane buffer of flight k; in_air[k] = its launch has been enqueued and not consumed yet.
  LaneBufs *B = nullptr;
  LaneBufs *Bs[2] = {nullptr, nullptr};
  bool in_air[2] = {false, false};
  bool head_in[2] = {false, false};              // the uLike / poLike points ride in that flight
  bool nothing_idle = false;                     // prepare(): every live run is in the air, nothing to lay out now
  int nflights() const { return (in_air[0] ? 1 : 0) + (in_air[1] ? 1 : 0); }
  std::vector<int> col;                          // column of every pair in the record block
  int ncols = 0;                                 // columns [p0, p0 + ncols) hold this population's live pairs
  std::function<int(FitPop &, const std::vector<int> &)> restage;  // stages the given pairs into columns p0, p0 + 1, ...
  int ul = 2;                                    // slots per pair of the uLike / poLike points in the first launch
  // Restarts ahead of need.  The multi-start rule starts the minimum of runs that could settle a pair and waits for
  // them; where they rarely do (the 2-parameter calcA mode: no two of 8 runs agree) that serialises a pair
  // into two waves and keeps the launches at a quarter of the runs the block has to make anyway.  Once the first
  // waves seen so far (here, or in the previous fit of the handle) have left at least 3 of 4 pairs unsettled, every
  // pair starts all its remaining runs at once.  Outcomes do not change: starts are drawn and results consumed in
  // index order, runs beyond the stopping point are dropped unseen.  RELATE_FIT_EAGER = 0 | 1 overrides.
  bool eager = false;
  int eager_forced = -1;
  long seen_first = 0, seen_unsettled = 0;       // pairs whose first wave has been consumed / that needed more runs
  // ~84 KB of L-BFGS-B state each (12 GB for a block of 16 384 pairs with all 8 restarts alive: what keeps two full
  // launches' worth of requests pending, so that a second launch can be in the air)
  static constexpr size_t kMaxActiveRuns = 147456;
  std::vector<std::pair<size_t, size_t>> pads;   // slot gaps [first, second) of the launch being laid out
  std::vector<int> pair_pts;                     // points of each pair taken into the launch being laid out
  bool long_rows = true;                         // rows long enough that the kernels, not the rounds, set the time

  ~FitPop() {
    for (Act &a : active)
      if (a.r) drop(a.r);
  }
  void init(int p0_, int P_, const relate_opts *o_, const FitSpec &S_, const double *starts_, long pair_base_,
            LaneBufs *B_, const int *nsnp) {
    p0 = p0_; P = P_; o = o_; S = S_; starts = starts_; pair_base = pair_base_; B = B_;
    Bs[0] = B_;
    n = S.nvar();
    R = std::max(1, o->timesToRun);
    mode_bounds(S, lo, hi);
    ms.assign(P, MultiStart());
    for (int p = 0; p < P; p++) {
      ms[p].timesToConverge = o->timesToConverge;
      ms[p].timesToRun = R;
      ms[p].convTol = o->convTol;
      if (nsnp[p0 + p] == 0) ms[p].finished = true;  // nothing to fit: the pair shares no SNP (status RELATE_E_NOSNP)
    }
    if (!starts) {
      rng.reserve(P);
      for (int p = 0; p < P; p++)
        rng.emplace_back(o->seed_mode == 1 ? o->seed + (unsigned)(pair_base + p0 + p) : o->seed);
    }
    results.assign(P, std::vector<double>((size_t)R * 4, 0.0));
    have.assign(P, std::vector<char>(R, 0));
    col.resize(P);
    for (int p = 0; p < P; p++) col[p] = p0 + p;
    ncols = P;
  }

  // returns the number of lanes laid out in B, or -1 on a pinned-memory error; sets `finished` when no run is left.
  // Zero lanes with runs left is possible: every requested point short-circuits in its wrapper (infeasible k).
  // k = the lane buffer to lay the launch out in; second_only_if_full: another launch is in the air, so this one is
  // laid out only if it fills the launch budget (else -3: wait for the one in the air -- two under-filled launches cost
  // more than one full one)
  long prepare(int k = 0, bool second_only_if_full = false) {
    B = Bs[k];
    nothing_idle = false;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq = tnow();
    auto lap = [&](int k) { double t = tnow(); tprep[k] += t - tq; tq = t; };
    auto by_pair = [](const Act &x, const Act &y) { return x.pair != y.pair ? x.pair < y.pair : x.idx < y.idx; };
    std::vector<std::array<double, 3>> x0s;
    for (;;) {
      // `active` stays sorted by (pair, restart): lanes of one pair side by side read the same records, so a wave
      // touches ~13 distinct records per step instead of 64 (measured 250 -> 330 G lane-steps/s on the chunked
      // path).  New runs arrive in that order at the tail and are merged in; retired runs are removed in place.
      const size_t n_before = active.size();
      x0s.clear();
      for (int p = 0; p < P; p++) {
        int w = ms[p].want_to_launch(eager && ms[p].launched > 0);
        if (w > 0 && ms[p].launched > ms[p].consumed && active.size() + w > kMaxActiveRuns) w = 0;  // ahead of need: bounded
        for (int q = 0; q < w; q++) {
          Act a{nullptr, p, ms[p].launched++, 0, 0, false, true, false};
          double st[4];
          std::array<double, 3> x0;
          if (starts) std::memcpy(st, starts + ((size_t)(pair_base + p0 + p) * R + a.idx) * 4, sizeof(st));
          else draw_start(rng[p], S, o->fixK, st);  // the pair's rand() stream: in index order
          start_to_x(S, st, x0.data());
          a.r = reinterpret_cast<Run *>(arena->take());
          if (!a.r) {  // out of host memory: the slabs taken in this pass hold no run yet
            for (size_t i = n_before; i < active.size(); i++) arena->give(reinterpret_cast<unsigned char *>(active[i].r));
            active.resize(n_before);
            return -1;
          }
          x0s.push_back(x0);
          active.push_back(a);
        }
      }
      parallel_for(x0s.size(), [&](size_t lo_, size_t hi_) {  // the optimiser states (their matrices are cleared here)
        for (size_t i = lo_; i < hi_; i++) {
          Run *r = new (active[n_before + i].r) Run();
          r->fm.begin(n, x0s[i].data(), lo, hi, r->matrices());
          active[n_before + i].r = r;
          active[n_before + i].done = r->fm.done;
        }
      }, 256);
      if (n_before > 0 && active.size() > n_before)
        std::inplace_merge(active.begin(), active.begin() + n_before, active.end(), by_pair);
      lap(0);
      // retire finished runs, feed the multi-start rule in index order
      bool relaunch = false;
      for (bool progressed = true; progressed && !relaunch;) {
        progressed = false;
        size_t keep_n = 0;
        for (size_t i = 0; i < active.size(); i++) {
          const Act &a = active[i];
          if (a.done) {
            x_to_result(S, a.r->fm.opt.x, a.r->fm.like, &results[a.pair][(size_t)a.idx * 4]);
            have[a.pair][a.idx] = 1;
            drop(a.r);
            progressed = true;
          } else
            active[keep_n++] = a;
        }
        active.resize(keep_n);
        for (int p = 0; p < P; p++)
          while (!ms[p].finished && ms[p].consumed < ms[p].launched && have[p][ms[p].consumed]) {
            ms[p].consume(&results[p][(size_t)ms[p].consumed * 4]);
            if (ms[p].consumed == ms[p].first_wave()) {
              seen_first++;
              if (!ms[p].finished) seen_unsettled++;
            }
          }
        if (!eager && eager_forced != 0 && seen_first >= std::max(32L, (long)P / 64) && seen_unsettled * 4 >= seen_first * 3)
          eager = true;
        // runs started ahead of need for a pair whose rule has stopped meanwhile are dropped unseen
        keep_n = 0;
        for (size_t i = 0; i < active.size(); i++) {
          if (ms[active[i].pair].finished && active[i].flight < 0) drop(active[i].r);  // (one in the air goes once it is back)
          else active[keep_n++] = active[i];
        }
        active.resize(keep_n);
        for (int p = 0; p < P && !relaunch; p++) {
          int w = ms[p].want_to_launch(eager && ms[p].launched > 0);
          if (w > 0 && !(ms[p].launched > ms[p].consumed && active.size() + w > kMaxActiveRuns)) relaunch = true;
        }
      }
      lap(1);
      if (relaunch) continue;
      break;
    }
    if (active.empty() && !first_round) {
      finished = true;
      return 0;
    }
    {
      bool idle = first_round;
      for (const Act &a : active) idle = idle || a.flight < 0;
      if (!idle) { nothing_idle = true; return -3; }
    }
    // Whole turns of the machine.  A single sweep costs the time of its fullest SIMD, so a launch of 1.1 waves per
    // SIMD costs what one of 2.0 does.  Up to one turn of lanes at two points per lane (131 072 points) a round
    // launches what it has (two-sided sweep: the cost follows the count); beyond, whole turns only, and the rest is
    // held back one round (runs are independent state machines; a held request is simply answered later, with the
    // same values).  Held runs go first in the next round, so none waits twice.
    const long kTurn = 2 * 256L * 4 * 64;
    parallel_for(active.size(), [&](size_t lo_, size_t hi_) {
      for (size_t i = lo_; i < hi_; i++) {  // lanes a run needs: the points its wrapper does not short-circuit
        if (active[i].flight >= 0) { active[i].nlanes = 0; continue; }  // its request is in the air
        Run *r = active[i].r;
        int nl = 0;
        for (int q = 0; q < r->fm.npts; q++) {
          r->ok[q] = map_point(S, r->fm.pts[q], r->akk[q]);
          nl += r->ok[q] ? 1 : 0;
        }
        active[i].nlanes = nl;
      }
    });
    long want = 0;
    for (const Act &a : active) want += a.nlanes;
    lap(2);
    // Few pairs left, scattered over the block's columns: every lane then pulls whole 128-byte lines for the 16 bytes
    // of its pair (a launch of 1 300 points over 260 scattered pairs ran HBM-bound at 136 G point-steps/s against
    // ~460 G dense).  Once the live pairs are a quarter of the columns they occupy, their records are staged again
    // side by side (K4 from the genotypes: ~3 us per pair at M = 5e5); finished pairs no longer need theirs.
    if (!first_round && restage && want > 0 && want < kCompactBelowPoints && nflights() == 0) {
      std::vector<int> live;
      for (int p = 0; p < P; p++)
        if (!ms[p].finished) live.push_back(p);
      if (!live.empty() && (long)live.size() * 4 <= ncols) {
        if (restage(*this, live)) return -2;
        for (size_t i = 0; i < live.size(); i++) col[live[i]] = p0 + (int)i;
        ncols = (int)live.size();
      }
    }
    lap(3);
    // The brackets of launch_like at long rows: <= 65 536 slots 43 ms, <= 98 304 60 ms, <= 131 072 68 ms,
    // <= 262 144 122 ms at M = 5e5.  A round that falls just above a bracket launches the bracket and holds the rest.
    long budget = want;
    if (long_rows) {
      const long head = first_round ? 2L * P : 0;  // the uLike / poLike points of the first launch count towards its bracket
      const long all = want + head;
      if (all > 2 * kTurn) budget = 2 * kTurn - head;
      else if (all > kTurn && all < kTurn * 9 / 5) budget = kTurn - head;
      else if (all > kTurn / 2 && all < kTurn * 11 / 16) budget = kTurn / 2 - head;
      if (budget < 0) budget = 0;
      if (second_only_if_full && all < 2 * kTurn) return -3;
    } else if (second_only_if_full) {
      return -3;
    }
    // Slots.  A lane of the likelihood kernels may carry nv points of one pair (pick_nv): every pair's points start at
    // a multiple of nv and are padded to one with copies of its last point (values never read); the budget counts
    // slots, padding included (a launch one slot above a bracket pays the next bracket).
    // uLike / poLike (:3478-3481) ride along with the first launch: slots p * ul, p * ul + 1
    const int nv = pick_nv(std::min(want, budget) + (first_round ? 2L * P : 0), 2);
    auto up = [nv](size_t x) { return (x + nv - 1) / nv * nv; };
    long used = 0;
    pair_pts.assign(P, 0);
    // requests held back before go first (in pair order, so the list is worked through and none starves), then the
    // others; both within the budget -- a launch above two turns of the machine falls into a bracket that costs half
    // as much again (285 000 slots: 180 ms against 127 ms for 262 144)
    for (Act &a : active) a.taken = false;
    for (int pass = 0; pass < 2; pass++)
      for (Act &a : active) {
        if (a.flight >= 0 || a.taken || a.waited != (pass == 0)) continue;
        const long delta = (long)up(pair_pts[a.pair] + a.nlanes) - (long)up(pair_pts[a.pair]);
        if (used + delta <= budget) {
          a.taken = true;
          used += delta;
          pair_pts[a.pair] += a.nlanes;
        }
      }
    B->nv = nv;
    if (first_round) ul = (int)up(2);
    size_t nl_total = first_round ? (size_t)ul * P : 0;
    pads.clear();
    int prev_pair = -1;
    for (Act &a : active) {
      if (a.flight >= 0) continue;
      a.waited = !a.taken;
      if (a.taken && a.nlanes > 0 && a.pair != prev_pair) {
        if (nl_total % nv) pads.push_back({nl_total, up(nl_total)});
        nl_total = up(nl_total);
        prev_pair = a.pair;
      }
      a.lane0 = nl_total;
      if (a.taken) nl_total += a.nlanes;
    }
    if (nl_total % nv) pads.push_back({nl_total, up(nl_total)});
    nl_total = up(nl_total);
    lap(4);
    if (B->ensure(nl_total) != hipSuccess) return -1;
    lap(5);
    B->nl = (int)nl_total;
    B->npoints = first_round ? 2L * P : 0;
    for (Act &a : active)
      if (a.taken) {
        B->npoints += a.nlanes;
        a.flight = (signed char)k;
      }
    B->wanted = want + (first_round ? 2L * P : 0);
    head_in[k] = first_round;
    if (first_round)
      for (int p = 0; p < P; p++)
        for (int j = 0; j < ul; j++) {
          size_t at = (size_t)ul * p + j;
          B->lane[at] = p0 + p;
          B->par[3 * at] = o->alim[0]; B->par[3 * at + 1] = 0; B->par[3 * at + 2] = j < 2 ? j : 1;
        }
    parallel_for(active.size(), [&](size_t lo_, size_t hi_) {
      for (size_t i = lo_; i < hi_; i++) {
        if (!active[i].taken) continue;
        Run *r = active[i].r;
        size_t at = active[i].lane0;
        const int column = col[active[i].pair];
        for (int q = 0; q < r->fm.npts; q++) {
          r->slot[q] = -1;
          if (r->ok[q]) {
            r->slot[q] = (int)at;
            B->lane[at] = column;
            B->par[3 * at] = r->akk[q][0]; B->par[3 * at + 1] = r->akk[q][1]; B->par[3 * at + 2] = r->akk[q][2];
            at++;
          }
        }
      }
    });
    for (const auto &g : pads)  // a gap follows the last point of a pair: copy that point
      for (size_t at = g.first; at < g.second; at++) {
        B->lane[at] = B->lane[g.first - 1];
        for (int c = 0; c < 3; c++) B->par[3 * at + c] = B->par[3 * (g.first - 1) + c];
      }
    lap(6);
    first_round = false;  // (the head rides in this launch: head_in[k])
    return (long)nl_total;
  }
  double tprep[7] = {0, 0, 0, 0, 0, 0, 0};  // seconds by phase of prepare(): start runs, retire, map points, restage, select, pinned buffers, lay out

  void consume(relate_fit *out, int k = 0) {  // out: records of the staged block; k: the flight whose values are in
    B = Bs[k];
    const double *val = B->val;
    if (head_in[k]) {
      for (int p = 0; p < P; p++) {
        out[p0 + p].uLike = val[(size_t)ul * p];
        out[p0 + p].poLike = val[(size_t)ul * p + 1];
      }
      head_in[k] = false;
    }
    std::atomic<bool> bad_seen{false};
    parallel_for(active.size(), [&](size_t lo_, size_t hi_) {
      for (size_t i = lo_; i < hi_; i++) {
        if (active[i].flight != k) continue;
        active[i].flight = -1;
        Run *r = active[i].r;
        double v[7];
        bool bad = false;
        for (int q = 0; q < r->fm.npts; q++) {
          v[q] = r->slot[q] < 0 ? kSysMax : val[r->slot[q]];
          if (!std::isfinite(v[q])) bad = true;
        }
        if (bad) {  // the reference prints and exit(0)s on NaN (:1513-1520); flag the pair and stop this run
          r->bad = true;
          r->fm.done = true;
          r->fm.like = NAN;
          active[i].done = true;
          bad_seen.store(true, std::memory_order_relaxed);
          continue;
        }
        r->fm.supply(v);
        active[i].done = r->fm.done;
      }
    });
    if (bad_seen.load())
      for (const Act &a : active)
        if (a.r->bad) out[p0 + a.pair].status = RELATE_E_NUMERIC;
  }

  void finish(relate_fit *out, double *conv_out) {
    for (int p = 0; p < P; p++) {
      relate_fit &f = out[p0 + p];
      f.a = ms[p].best[0];
      f.k2 = ms[p].best[1];
      f.k1 = ms[p].best[2];
      f.k0 = 1 - f.k2 - f.k1;
      f.negloglik = ms[p].best[3];
      f.timesRun = ms[p].consumed;
      f.timesConverged = ms[p].conv;
      if (conv_out)  // convInfo: the runs the rule consumed; rows beyond stay zero (speculative runs are not reported)
        std::memcpy(conv_out + (size_t)(pair_base + p0 + p) * R * 4, results[p].data(),
                    sizeof(double) * ms[p].consumed * 4);
    }
  }
};

static_assert(sizeof(FitPop::Run) <= kFitRunBytes && kFitSlabBytes % 64 == 0, "a run and its matrices share one slab");

// fills out[0..P) (and conv_out rows pair_base..) for the P pairs of the staged block
int fit_block(relate_hip_t h, int P, long pair_base, const double *starts, relate_fit *out, double *conv_out) {
  const relate_opts &o = h->o;
  for (int p = 0; p < P; p++) {
    out[p] = relate_fit{};
    out[p].status = 0;
    out[p].nsnp = h->h_nsnp[p];
    // the reference stops here: "pair doesn't share 1 non-missing SNP" (src/relateHMM.cpp:3415-3421)
    if (h->h_nsnp[p] == 0) out[p].status = RELATE_E_NOSNP;
  }
  if (o.doPar) {  // likelihood at the user's parameters, no optimisation (:3484-3510)
    std::vector<int32_t> lp;
    std::vector<double> par, val;
    auto push = [&](int p, double a, double k2, double k1) {
      lp.push_back(p);
      par.push_back(a); par.push_back(k2); par.push_back(k1);
    };
    for (int p = 0; p < P; p++) {
      push(p, o.alim[0], 0, 0);
      push(p, o.alim[0], 0, 1);
    }
    for (int p = 0; p < P; p++) push(p, o.par[0], o.par[1], o.par[2]);
    val.resize(lp.size());
    int rc = like_lanes(h, P, lp.data(), par.data(), (int)lp.size(), val.data());
    if (rc) return rc;
    for (int p = 0; p < P; p++) {
      out[p].uLike = val[2 * p];
      out[p].poLike = val[2 * p + 1];
      out[p].a = o.par[0]; out[p].k2 = o.par[1]; out[p].k1 = o.par[2];
      out[p].k0 = 1 - out[p].k2 - out[p].k1;
      out[p].negloglik = val[2 * P + p];
      out[p].timesRun = 0; out[p].timesConverged = 0;
      if (!std::isfinite(val[2 * P + p])) out[p].status = RELATE_E_NUMERIC;
    }
    return 0;
  }
  int mode = select_mode(o.fixK, o.fixA, o.calcA);
  if (mode < 0)
    return fail(h, RELATE_E_ARG,
                "optimization not implemented for this fixA/fixK/calcA combination (the reference exits, "
                "src/relateHMM.cpp:1898-1901; fixA without fixK is undefined there)");
  FitSpec S{mode, o.fixA_val, o.fixK_val, o.phi, o.alim[0], o.alim[1]};
  // Two populations of pairs take turns on the GPU: while the likelihood launch of one runs, the host retires runs,
  // steps the optimisers and lays out the next launch of the other (pinned buffers, events; nothing waits on the
  // stream as a whole).  Pairs are independent, so the split changes no value.  Small blocks stay one population.
  // Each population should still fill whole turns of the machine on its own (a few runs of 3-5 lanes per pair), or the
  // split costs more likelihood time than the overlap saves: 8 192 pairs x 5e5 SNPs ran 13.0 s unsplit, 15.6 s split;
  // 44 850 pairs x 5e4 SNPs 15.6 s unsplit, 9.7 s split.  RELATE_FIT_POPULATIONS = 1 | 2 overrides (tests, tuning).
  // With rows of 5e5 SNPs the host's share of a round is ~6 % (runs in recycled slabs, persistent workers) and one
  // population's larger launches win: 16 384 pairs 15.8 s unsplit, 16.7 s split; rows below ~1.5e5 SNPs still split.
  int npop = (P >= 16384 && h->M < 150000) ? 2 : 1;
  if (const char *e = getenv("RELATE_FIT_POPULATIONS")) npop = atoi(e) >= 2 && P >= 2 ? 2 : 1;
  FitPop pop[2];
  const std::vector<int32_t> blk(h->staged);  // the block's pairs by column (compaction clears the memo)
  static const bool compact = [] { const char *e = getenv("RELATE_FIT_COMPACT"); return !(e && atoi(e) == 0); }();
  for (int k = 0; k < npop; k++) {
    int b = (int)((long)P * k / npop), e = (int)((long)P * (k + 1) / npop);
    pop[k].init(b, e - b, &o, S, starts, pair_base, &h->lanebufs[k], h->h_nsnp.data());
    if (const char *ee = getenv("RELATE_FIT_EAGER")) pop[k].eager_forced = atoi(ee) ? 1 : 0;
    pop[k].long_rows = h->M >= 16384;
    pop[k].arena = &h->run_arena;
    pop[k].eager = pop[k].eager_forced >= 0 ? pop[k].eager_forced == 1 : h->fit_eager == 1;
    if (compact && (int)blk.size() == 2 * P)
      pop[k].restage = [h, &blk](FitPop &F, const std::vector<int> &live) -> int {
        std::vector<int32_t> sub(2 * live.size());
        for (size_t i = 0; i < live.size(); i++) {
          sub[2 * i] = blk[2 * (size_t)(F.p0 + live[i])];
          sub[2 * i + 1] = blk[2 * (size_t)(F.p0 + live[i]) + 1];
        }
        h->staged.clear();  // the block in dRec is no longer the one the memo describes
        const double t_stage = h->timings[0];
        int rc = stage_cols(h, sub.data(), (int)live.size(), F.p0, h->recPitch, false);
        h->timings[0] += t_stage;
        return rc;
      };
  }
  double t_prep = 0, t_cons = 0, t_wait = 0;  // host seconds by phase (RELATE_FIT_TRACE)
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  int rc = 0;
  // One population, long rows: two launches in the air (FitPop::Bs).  RELATE_FIT_FLIGHTS=1 keeps one (tests, tuning).
  static const int max_flights = [] { const char *e = getenv("RELATE_FIT_FLIGHTS"); return e && atoi(e) == 1 ? 1 : 2; }();
  if (npop == 1 && max_flights == 2 && pop[0].long_rows) {
    FitPop &F = pop[0];
    F.Bs[1] = &h->lanebufs[1];
    int fifo[2] = {-1, -1}, nq = 0;  // flights in the air, oldest first
    // lays a launch out in buffer k and enqueues it; 1 = enqueued, 0 = nothing to launch now (or the block is finished)
    auto launch = [&](int k, bool only_if_full) -> int {
      for (;;) {
        const double t0 = now();
        long nl = F.prepare(k, only_if_full);
        t_prep += now() - t0;
        if (nl == -2) { rc = RELATE_E_HIP; return 0; }
        if (nl == -1) { rc = fail(h, RELATE_E_NOMEM, "host memory for the optimiser runs or the likelihood lanes"); return 0; }
        if (nl == -3 || F.finished) return 0;
        if (nl > 0) {
          rc = like_enqueue(h, P, *F.Bs[k]);
          if (rc) return 0;
          F.in_air[k] = true;
          fifo[nq++] = k;
          return 1;
        }
        F.consume(out, k);  // nothing to evaluate on the GPU: the wrappers answered every point of these runs themselves
      }
    };
    // A second launch beside one in the air is only made when it fills the budget; prepare() finds that out, but a
    // prepare() that ends in "not full" is 3-5 ms of host work for nothing, so it is only tried when the requests known
    // to be pending could fill one: what the last layout left behind plus (after a flight has come back) about what
    // that flight carried.
    const long kFull = 2L * 2 * 256 * 4 * 64;
    auto left_behind = [&](int k) { return F.Bs[k]->wanted - F.Bs[k]->npoints; };
    launch(0, false);
    if (!rc && nq == 1 && left_behind(0) >= kFull) launch(1, true);
    while (nq > 0 && !rc) {
      const int k = fifo[0];
      double t0 = now();
      rc = like_wait(h, *F.Bs[k]);
      t_wait += now() - t0;
      if (rc) break;
      fifo[0] = fifo[1];
      nq--;
      F.in_air[k] = false;
      const long came_back = F.Bs[k]->npoints;
      t0 = now();
      F.consume(out, k);
      t_cons += now() - t0;
      if (nq > 0) {  // the other flight is in the air: this one goes up again only as a full launch
        const int other = fifo[0];
        if (left_behind(other) + came_back >= kFull) launch(k, true);
      } else {
        launch(k, false);
        if (!rc && nq == 1 && !F.finished && left_behind(k) >= kFull) launch(k == 0 ? 1 : 0, true);
      }
    }
  } else {
  auto start = [&](FitPop &F) -> int {
    F.inflight = false;
    for (;;) {
      const double t0 = now();
      long nl = F.prepare();
      t_prep += now() - t0;
      if (nl == -2) return RELATE_E_HIP;  // staging the live pairs again failed: the message is set
      if (nl < 0) return fail(h, RELATE_E_NOMEM, "host memory for the optimiser runs or the likelihood lanes");
      if (F.finished) return 0;
      if (nl > 0) {
        F.inflight = true;
        return like_enqueue(h, P, *F.B);
      }
      F.consume(out);  // nothing to evaluate on the GPU this round: the wrappers answered every point themselves
    }
  };
  for (int k = 0; k < npop && !rc; k++) rc = start(pop[k]);
  for (bool any = true; any && !rc;) {
    any = false;
    for (int k = 0; k < npop && !rc; k++) {
      FitPop &F = pop[k];
      if (!F.inflight) continue;
      any = true;
      double t0 = now();
      rc = like_wait(h, *F.B);
      t_wait += now() - t0;
      if (rc) break;
      t0 = now();
      F.consume(out);
      t_cons += now() - t0;
      rc = start(F);
    }
  }
  }
  if (rc) {
    (void)hipStreamSynchronize(h->stream);  // nothing of a failed block stays queued on its buffers
    return rc;
  }
  for (int k = 0; k < npop; k++) pop[k].finish(out, conv_out);
  {
    long first = 0, unsettled = 0;
    for (int k = 0; k < npop; k++) { first += pop[k].seen_first; unsettled += pop[k].seen_unsettled; }
    if (first >= 32) h->fit_eager = unsettled * 4 >= first * 3 ? 1 : 0;
  }
  if (getenv("RELATE_FIT_TRACE"))
    std::fprintf(stderr, "fit block: %d pairs, host prepare %.3f s (start runs %.3f, retire %.3f, map points %.3f, restage %.3f, "
                 "select %.3f, pinned buffers %.3f, lay out %.3f), consume %.3f s, waiting on the GPU %.3f s\n", P, t_prep,
                 pop[0].tprep[0], pop[0].tprep[1], pop[0].tprep[2], pop[0].tprep[3], pop[0].tprep[4], pop[0].tprep[5],
                 pop[0].tprep[6], t_cons, t_wait);
  return 0;
}

// Device results -> caller memory, queued behind the work on the handle's stream; returns with the data in place.
// Memory from relate_hip_host_alloc takes the DMA directly.  Any other memory is taken to be pageable, which would
// make the runtime stage through its own small buffers (~10 GB/s measured); instead the copy runs in 32 MiB pieces
// through two pinned buffers, the DMA of one piece overlapping the host threads' copy of the previous.
std::mutex g_pinned_mu;
std::map<const char *, size_t> g_pinned;  // blocks handed out by relate_hip_host_alloc: start -> bytes
bool is_library_pinned(const void *p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  auto it = g_pinned.upper_bound((const char *)p);
  if (it == g_pinned.begin()) return false;
  --it;
  return (const char *)p + bytes <= it->first + it->second;
}
constexpr size_t kBounceBytes = (size_t)32 << 20;
int copy_to_host(relate_hip_t h, void *dst, const void *src, size_t bytes) {
  if (!bytes) return 0;
  if (is_library_pinned(dst, bytes) || bytes <= ((size_t)1 << 20)) {
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
  }
  for (int i = 0; i < 2; i++) {
    if (!h->bounce[i]) HIPCHK(h, hipHostMalloc(&h->bounce[i], kBounceBytes, hipHostMallocDefault));
    if (!h->bounceEv[i]) HIPCHK(h, hipEventCreateWithFlags(&h->bounceEv[i], hipEventDisableTiming));
  }
  const size_t np = (bytes + kBounceBytes - 1) / kBounceBytes;
  auto drain = [&](size_t i) -> int {  // piece i: wait for its DMA, then copy it out with the host threads
    const int b = (int)(i & 1);
    const size_t off = i * kBounceBytes, n = std::min(kBounceBytes, bytes - off);
    HIPCHK(h, hipEventSynchronize(h->bounceEv[b]));
    const char *from = (const char *)h->bounce[b];
    char *to = (char *)dst + off;
    parallel_for(n, [&](size_t lo, size_t hi) { std::memcpy(to + lo, from + lo, hi - lo); });
    return 0;
  };
  for (size_t i = 0; i < np; i++) {
    const int b = (int)(i & 1);
    const size_t off = i * kBounceBytes, n = std::min(kBounceBytes, bytes - off);
    // bounce[b] last held piece i-2, which the previous iteration drained
    HIPCHK(h, hipMemcpyAsync(h->bounce[b], (const char *)src + off, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipEventRecord(h->bounceEv[b], h->stream));
    if (i > 0) {
      int rc = drain(i - 1);
      if (rc) return rc;
    }
  }
  return drain(np - 1);
}

// K6 launches for pairs [p0, p0 + P) of the staged block with their params already in dParams; posteriors left in dPost
// (f32_rows: 0 = doubles; 1..3 = floats, P x f32_rows x M in writer order, in the same buffer.  f64_rows: 0 = the
// reference's P x 3 x M (IBD2, IBD1, IBD0); 1..3 = doubles, P x f64_rows x M in writer order)
int launch_decode(relate_hip_t h, int p0, int P, int f32_rows = 0, int sig4 = 0, int f64_rows = 0) {
  const int M = h->M;
  int nchunk, clen;
  pick_chunks(M, P, h->single_sweep_lanes, 256, h->slots_decode[f32_rows ? f32_rows : f64_rows], 1, &nchunk, &clen);
  if (const char *e = getenv("RELATE_DECODE_CHUNKS")) {  // tuning experiments
    const int want = atoi(e);
    if (want >= 1 && want <= std::max(1, M / 32)) {  // (chunks are whole tiles of 16 SNPs; short ones are only slow)
      clen = ((M + want - 1) / want + 15) / 16 * 16;
      nchunk = (M + clen - 1) / clen;
    }
  }
  h->last_chunks[1] = nchunk;
  const bool dr = h->o.double_recom != 0;
  const int nlp = pad256(P);
  const double *rec = h->dRec.p;
  const int RP = h->recPitch;
  const size_t ntile = (size_t)(M + 15) / 16;
  const size_t tile_stride = (size_t)((P + 63) / 64) * kCkptGroupBytes;
  HIPCHK(h, h->dCk.ensure(ntile * tile_stride));
  const TileCkpt ck{h->dCk.p, tile_stride};
  size_t n = (size_t)nlp * nchunk;
  ChunkOut O;
  int rc = ensure_chunk_buffers(h, n, &O);
  if (rc) return rc;
  const double *astart = nullptr, *bend = nullptr;
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  dim3 grid((unsigned)((size_t)(nlp / 256) * nchunk));
  if (dr)
    hipLaunchKernelGGL((chunk_matrix_kernel<true, true>), grid, dim3(256), 0, h->stream, rec, h->dTsh.p, h->dPairOwn.p, M, RP, LaneMap{nullptr, 1},
                       h->dParams.p, P, nlp, nchunk, clen, h->o.alim[1], 0, O, ck, p0);
  else
    hipLaunchKernelGGL((chunk_matrix_kernel<false, true>), grid, dim3(256), 0, h->stream, rec, h->dTsh.p, h->dPairOwn.p, M, RP, LaneMap{nullptr, 1},
                       h->dParams.p, P, nlp, nchunk, clen, h->o.alim[1], 0, O, ck, p0);
  HIPCHK(h, hipGetLastError());
  if (nchunk > 1) {
    HIPCHK(h, h->dAstart.ensure((size_t)3 * n));
    HIPCHK(h, h->dBend.ensure((size_t)3 * n));
    hipLaunchKernelGGL(decode_combine_kernel, dim3(blocks_for(P, 256)), dim3(256), 0, h->stream, O, h->dParams.p, P,
                       nlp, nchunk, h->dAstart.p, h->dBend.p);
    HIPCHK(h, hipGetLastError());
    astart = h->dAstart.p;
    bend = h->dBend.p;
  }
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  dim3 bgrid((unsigned)((size_t)((P + kBwdThreads - 1) / kBwdThreads) * nchunk));
#define LAUNCH_BWD(DRV, T, OUT, WROWS)                                                                              \
  hipLaunchKernelGGL((decode_bwd_kernel<DRV, T, WROWS>), bgrid, dim3(kBwdThreads), 0, h->stream, rec, h->dTsh.p, h->dPairOwn.p, M, P, RP, p0, \
                     h->dParams.p, nlp, clen, astart, bend, ck, OUT, sig4)
#define LAUNCH_BWD_ROWS(DRV, T, OUT, N)       \
  do {                                        \
    if ((N) == 1) LAUNCH_BWD(DRV, T, OUT, 1); \
    else if ((N) == 2) LAUNCH_BWD(DRV, T, OUT, 2); \
    else LAUNCH_BWD(DRV, T, OUT, 3);          \
  } while (0)
  if (f32_rows) {
    float *pf = reinterpret_cast<float *>(h->dPost.p);
    if (dr) LAUNCH_BWD_ROWS(true, float, pf, f32_rows);
    else LAUNCH_BWD_ROWS(false, float, pf, f32_rows);
  } else if (f64_rows) {
    if (dr) LAUNCH_BWD_ROWS(true, double, h->dPost.p, f64_rows);
    else LAUNCH_BWD_ROWS(false, double, h->dPost.p, f64_rows);
  } else {
    if (dr) LAUNCH_BWD(true, double, h->dPost.p, 0);
    else LAUNCH_BWD(false, double, h->dPost.p, 0);
  }
#undef LAUNCH_BWD_ROWS
#undef LAUNCH_BWD
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  return 0;
}

// Decode (and Viterbi) of pairs [p0, p0 + P) of the staged block; `pairs` = those P pairs (needed to stage the full
// format when a path is wanted).
int decode_block(relate_hip_t h, const int32_t *pairs, int p0, int P, const double *params, double *post, int8_t *path,
                 int *nsnp_out, bool copy_back, float *post32 = nullptr, int f32_rows = 0, int sig4 = 0, int f64_rows = 0) {
  const int M = h->M;
  HIPCHK(h, h->dParams.ensure((size_t)3 * P));
  HIPCHK(h, hipMemcpyAsync(h->dParams.p, params, sizeof(double) * 3 * P, hipMemcpyHostToDevice, h->stream));
  if (nsnp_out) std::memcpy(nsnp_out, h->h_nsnp.data() + p0, sizeof(int) * P);  // counted by the staging kernel
  if (post || post32 || !copy_back) {  // (f32_rows > 0 without post32: float rows left on the device)
    const size_t rows64 = f64_rows ? (size_t)f64_rows : 3;
    HIPCHK(h, h->dPost.ensure(f32_rows ? ((size_t)f32_rows * M * P + 1) / 2 : rows64 * M * P));
    int rc = launch_decode(h, p0, P, f32_rows, sig4, f64_rows);
    if (rc) return rc;
    if (copy_back && post) {
      rc = copy_to_host(h, post, h->dPost.p, sizeof(double) * rows64 * M * P);
      if (rc) return rc;
    }
    if (copy_back && post32) {
      rc = copy_to_host(h, post32, h->dPost.p, sizeof(float) * (size_t)f32_rows * M * P);
      if (rc) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float a = 0, b = 0;
    HIPCHK(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    HIPCHK(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
    h->timings[2] += a;
    h->timings[3] += b;
  }
  if (path) {
    int rc = stage_block(h, pairs, P, true);  // the Viterbi works on log S of every state: full format, own buffer
    if (rc) return rc;
    HIPCHK(h, h->dBp.ensure(vit_bp_bytes(M, P)));
    HIPCHK(h, h->dPath.ensure((size_t)M * P));
    // a workgroup's time is its walk along the row whatever the number of pairs it holds: a short pair list goes out four
    // or eight pairs to a workgroup (tiles of 64 / 32 SNPs) so that every CU has workgroups, a long one sixteen (tiles of 16)
    static const int vit_np = [] { const char *e = getenv("RELATE_VIT_NP"); return e ? atoi(e) : 0; }();
    const int np = vit_np == 4 || vit_np == 8 || vit_np == 16 ? vit_np : P >= 8192 ? 16 : P >= 3072 ? 8 : 4;
    auto vit = np == 16 ? viterbi_fwd_kernel<16> : np == 8 ? viterbi_fwd_kernel<8> : viterbi_fwd_kernel<4>;
    hipLaunchKernelGGL(vit, dim3(blocks_for(P, np)), dim3(kVitThreads), 0, h->stream, h->dRecV.p, M, P, h->recPitchV,
                       h->dParams.p, h->o.double_recom, h->dBp.p);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(viterbi_back_kernel, dim3(blocks_for(P, kVitPairs)), dim3(kVitBack), 0, h->stream, h->dBp.p, M, P,
                       h->dPath.p);
    HIPCHK(h, hipGetLastError());
    rc = copy_to_host(h, path, h->dPath.p, (size_t)M * P);
    if (rc) return rc;
  }
  return 0;
}

void gather_release(relate_hip_t h);

}  // namespace

extern "C" {

int relate_hip_create(relate_hip_t *out, int device_id) {
  if (!out) return fail(nullptr, RELATE_E_ARG, "null output pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, RELATE_E_NODEV,
                std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                    " (librelate_hip has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, RELATE_E_ARG, "device_id out of range");
  relate_hip_ctx *h = new relate_hip_ctx();
  h->dev = device_id;
  if (const char *e = getenv("RELATE_SINGLE_SWEEP_LANES")) h->single_sweep_lanes = atol(e);
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
    delete h;
    return fail(nullptr, RELATE_E_HIP, "hipSetDevice/hipStreamCreate failed");
  }
  for (int i = 0; i < 4; i++)
    if (hipEventCreate(&h->ev[i]) != hipSuccess) {
      delete h;
      return fail(nullptr, RELATE_E_HIP, "hipEventCreate failed");
    }
  {
    hipDeviceProp_t prop;
    int ncu = 256;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    long sc = wave_slots(chunk_matrix_kernel<false, false>, 256, ncu);
    long sa = wave_slots(chunk_matrix_kernel<false, true>, 256, ncu);
    h->slots_like = sc;
    // decode launches (P / 64) waves per chunk in each of its two sweeps
    h->slots_decode[0] = lcm(sa, wave_slots(decode_bwd_kernel<false, double, 0>, kBwdThreads, ncu));
    h->slots_decode[1] = lcm(sa, wave_slots(decode_bwd_kernel<false, double, 1>, kBwdThreads, ncu));
    h->slots_decode[2] = lcm(sa, wave_slots(decode_bwd_kernel<false, double, 2>, kBwdThreads, ncu));
    h->slots_decode[3] = lcm(sa, wave_slots(decode_bwd_kernel<false, double, 3>, kBwdThreads, ncu));
  }
  {
    // the no-alloc entry point's scratch: every GPU holds its own copy of the array at its own address
    hipError_t se = hipGetSymbolAddress((void **)&h->scratch_dev, HIP_SYMBOL(scratch));
    if (se != hipSuccess || !h->scratch_dev) {
      relate_hip_destroy(h);
      return fail(nullptr, RELATE_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(se));
    }
  }
  *out = h;
  return RELATE_OK;
}

int relate_hip_destroy(relate_hip_t h) {
  if (!h) return RELATE_OK;
  (void)hipSetDevice(h->dev);
  h->dTsh.release(); h->dBnd.release(); h->dPairOwn.release();
  h->dG.release(); h->dPos.release(); h->dMaf.release(); h->dMea.release(); h->dPBA.release(); h->dPbA.release();
  h->dPBa.release(); h->dPba.release(); h->dTab1.release(); h->dTabL.release(); h->dChoose.release();
  h->dGt.release(); h->dTabD.release(); h->dOrder.release();
  h->dRec.release(); h->dPairs.release(); h->dLanePair.release(); h->dParams.release(); h->dOut.release();
  h->dPost.release(); h->dCk.release(); h->dBp.release(); h->dPath.release(); h->dNsnp.release();
  h->dRecV.release(); h->dScM.release(); h->dScE.release(); h->dScN.release(); h->dPairM.release(); h->dPairE.release();
  h->dCm.release(); h->dAstart.release(); h->dBend.release(); h->dCe.release();
  h->lanebufs[0].release(); h->lanebufs[1].release();
  for (int i = 0; i < 2; i++) {
    if (h->bounce[i]) (void)hipHostFree(h->bounce[i]);
    if (h->bounceEv[i]) (void)hipEventDestroy(h->bounceEv[i]);
  }
  for (int i = 0; i < 4; i++)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  gather_release(h);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  h->run_arena.release();
  delete h;
  return RELATE_OK;
}

int relate_hip_host_alloc(void **out, size_t bytes) {
  if (!out) return fail(nullptr, RELATE_E_ARG, "null output pointer");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(nullptr, RELATE_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  g_pinned[(const char *)*out] = bytes ? bytes : 1;
  return RELATE_OK;
}

int relate_hip_host_free(void *p) {
  if (!p) return RELATE_OK;
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    if (!g_pinned.erase((const char *)p)) return fail(nullptr, RELATE_E_ARG, "not a block of relate_hip_host_alloc");
  }
  hipError_t e = hipHostFree(p);
  if (e != hipSuccess) return fail(nullptr, RELATE_E_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
  return RELATE_OK;
}

size_t relate_post_text_stride(int M) { return (size_t)(M > 0 ? M : 0) * relate_text::kMaxValueChars + 2; }

int relate_format_post_rows(const double *post, int P, int M, int k1_only, char *out, size_t *row_len) {
  if (!post || !out || !row_len || P < 0 || M < 1) return fail(nullptr, RELATE_E_ARG, "bad arguments");
  const int rpp = k1_only ? 1 : 2;
  const long nrows = (long)P * rpp;
  const size_t stride = relate_post_text_stride(M);
  (void)relate_text::pow10_table();  // built before the threads start
  std::atomic<long> next{0};
  auto work = [&] {
    for (long r = next.fetch_add(1); r < nrows; r = next.fetch_add(1)) {
      const long p = r / rpp;
      const int which = (r % rpp) ? 1 : 2;  // IBD0 row (2) first, then the IBD1 row (1), :3771-3782
      row_len[r] = relate_text::format_row(post + ((size_t)p * 3 + which) * M, M, out + (size_t)r * stride);
    }
  };
  int nt = (int)std::min<long>(host_threads(), nrows);
  if ((long)nrows * M < 65536) nt = 1;
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto &x : th) x.join();
  return RELATE_OK;
}

int relate_format_rows(const double *rows, long nrows, int M, char *out, size_t *row_len) {
  if (!rows || !out || !row_len || nrows < 0 || M < 1) return fail(nullptr, RELATE_E_ARG, "bad arguments");
  const size_t stride = relate_post_text_stride(M);
  (void)relate_text::pow10_table();  // built before the threads start
  std::atomic<long> next{0};
  auto work = [&] {
    for (long r = next.fetch_add(1); r < nrows; r = next.fetch_add(1))
      row_len[r] = relate_text::format_row(rows + (size_t)r * M, M, out + (size_t)r * stride);
  };
  int nt = (int)std::min<long>(host_threads(), nrows);
  if (nrows * M < 65536) nt = 1;
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work);
  work();
  for (auto &x : th) x.join();
  return RELATE_OK;
}

const char *relate_hip_last_error(relate_hip_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int relate_hip_load(relate_hip_t h, const uint8_t *geno, int N, int M_raw, const double *pos, const int *chr,
                    const relate_opts *opts) {
  if (!h) return fail(nullptr, RELATE_E_ARG, "null handle");
  if (!geno || !pos || !opts || N < 2 || M_raw < 2) return fail(h, RELATE_E_ARG, "bad data set arguments");
  if (opts->doPrune && opts->back == 0)
    return fail(h, RELATE_E_ARG, "Must supply a back > 0, when you want to prune (src/runRelate.cpp:89-92)");
  if (opts->back < 0) return fail(h, RELATE_E_ARG, "back must be >= 0");
  {  // one threaded pass: copy and validate (the host keeps the matrix for relate_hip_precalc)
    const size_t total = (size_t)N * M_raw;
    h->h_raw.resize(total);
    std::atomic<bool> bad{false};
    uint8_t *dst = h->h_raw.data();
    parallel_for(total, [&](size_t lo, size_t hi) {
      uint8_t worst = 0;
      for (size_t i = lo; i < hi; i++) {
        const uint8_t v = geno[i];
        dst[i] = v;
        worst |= v;
      }
      if (worst > 3) bad.store(true);
    }, (size_t)1 << 20);
    if (bad.load()) {
      h->h_raw.clear();
      h->loaded = false;
      return fail(h, RELATE_E_ARG, "Error in genotype data: value should be between 0 and 3");
    }
  }
  h->o = *opts;
  h->back2 = 2 * opts->back;  // post_processing_of_option_pars, src/runRelate.cpp:88
  h->N = N;
  h->Mraw = M_raw;
  h->W = (N + 31) / 32;
  h->h_bed.clear();
  h->h_bed.shrink_to_fit();
  h->h_bedcols.clear();
  h->h_pos_raw.assign(pos, pos + M_raw);
  if (chr) h->h_chr_raw.assign(chr, chr + M_raw);
  else h->h_chr_raw.assign(M_raw, -1);
  h->loaded = true;
  h->ready = false;
  h->fit_eager = -1;
  h->staged.clear();
  h->stagedV.clear();
  return RELATE_OK;
}

int relate_hip_load_bed(relate_hip_t h, const uint8_t *bed, int N, int n_snp_bed, const int *cols, int M_raw,
                        const double *pos, const int *chr, const relate_opts *opts) {
  if (!h) return fail(nullptr, RELATE_E_ARG, "null handle");
  if (!bed || !cols || !pos || !opts || N < 2 || M_raw < 2 || n_snp_bed < M_raw)
    return fail(h, RELATE_E_ARG, "bad data set arguments");
  if (opts->doPrune && opts->back == 0)
    return fail(h, RELATE_E_ARG, "Must supply a back > 0, when you want to prune (src/runRelate.cpp:89-92)");
  if (opts->back < 0) return fail(h, RELATE_E_ARG, "back must be >= 0");
  for (int s = 0; s < M_raw; s++)
    if (cols[s] < 0 || cols[s] >= n_snp_bed) return fail(h, RELATE_E_ARG, "a .bed row index is out of range");
  h->o = *opts;
  h->back2 = 2 * opts->back;
  h->N = N;
  h->Mraw = M_raw;
  h->W = (N + 31) / 32;
  h->bedBps = (N + 3) / 4;
  h->h_bed.assign(bed, bed + (size_t)n_snp_bed * h->bedBps);
  h->h_bedcols.assign(cols, cols + M_raw);
  h->h_raw.clear();
  h->h_raw.shrink_to_fit();
  h->h_pos_raw.assign(pos, pos + M_raw);
  if (chr) h->h_chr_raw.assign(chr, chr + M_raw);
  else h->h_chr_raw.assign(M_raw, -1);
  h->loaded = true;
  h->ready = false;
  h->fit_eager = -1;
  h->staged.clear();
  h->stagedV.clear();
  return RELATE_OK;
}

int relate_hip_precalc(relate_hip_t h, int *keep_out, int *M_out) {
  if (!h) return fail(nullptr, RELATE_E_ARG, "null handle");
  if (!h->loaded) return fail(h, RELATE_E_ARG, "relate_hip_load has not been called");
  if (set_device(h)) return RELATE_E_HIP;
  const relate_opts &o = h->o;
  const int N = h->N, Mraw = h->Mraw, W = h->W;
  hipStream_t st = h->stream;
  // raw genotypes -> device, packed SNP-major at 2 bits
  DBuf<uint8_t> dRaw;
  DBuf<uint64_t> dG0, dG1;
  DBuf<double> dMaf0;
  DBuf<int> dCols;
  auto cleanup = [&]() { dRaw.release(); dG0.release(); dG1.release(); dMaf0.release(); dCols.release(); };
#define PCHK(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      cleanup();                                                                                       \
      return fail(h, e_ == hipErrorOutOfMemory ? RELATE_E_NOMEM : RELATE_E_HIP,                        \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                  \
    }                                                                                                  \
  } while (0)
  PCHK(dG0.ensure((size_t)Mraw * W));
  if (!h->h_bed.empty()) {  // PLINK payload: recode the 2-bit fields in place of the byte matrix
    PCHK(dRaw.ensure(h->h_bed.size()));
    PCHK(hipMemcpyAsync(dRaw.p, h->h_bed.data(), h->h_bed.size(), hipMemcpyHostToDevice, st));
    PCHK(dCols.ensure(Mraw));
    PCHK(hipMemcpyAsync(dCols.p, h->h_bedcols.data(), sizeof(int) * Mraw, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bed_recode_kernel, dim3(blocks_for((long)Mraw * W, 256)), dim3(256), 0, st, dRaw.p, h->bedBps,
                       dCols.p, Mraw, N, W, dG0.p);
    PCHK(hipGetLastError());
    PCHK(hipStreamSynchronize(st));  // dCols is reused below
  } else {
    PCHK(dRaw.ensure((size_t)N * Mraw));
    PCHK(hipMemcpyAsync(dRaw.p, h->h_raw.data(), (size_t)N * Mraw, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pack_kernel, dim3(blocks_for((long)Mraw * W, 256)), dim3(256), 0, st, dRaw.p, N, Mraw,
                       (const int *)nullptr, Mraw, W, dG0.p);
    PCHK(hipGetLastError());
  }
  // K1 on all raw SNPs, maf_stripper on the host copy of M_raw doubles (:3302-3305)
  PCHK(dMaf0.ensure(Mraw));
  hipLaunchKernelGGL(maf_kernel, dim3(blocks_for(Mraw, 256)), dim3(256), 0, st, dG0.p, Mraw, W, dMaf0.p);
  PCHK(hipGetLastError());
  std::vector<double> maf0(Mraw);
  PCHK(hipMemcpyAsync(maf0.data(), dMaf0.p, sizeof(double) * Mraw, hipMemcpyDeviceToHost, st));
  PCHK(hipStreamSynchronize(st));
  h->keep.assign(Mraw, 0);
  std::vector<int> cols;
  for (int s = 0; s < Mraw; s++)
    if (maf0[s] != 0 && maf0[s] > o.min && maf0[s] < (1 - o.min)) {
      h->keep[s] = 1;
      cols.push_back(s);
    }
  if ((int)cols.size() < 2) {
    cleanup();
    return fail(h, RELATE_E_ARG, "fewer than 2 SNPs pass the MAF filter");
  }
  int M1 = (int)cols.size();
  PCHK(dCols.ensure(Mraw));
  PCHK(hipMemcpyAsync(dCols.p, cols.data(), sizeof(int) * M1, hipMemcpyHostToDevice, st));
  PCHK(dG1.ensure((size_t)M1 * W));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for((long)M1 * W, 256)), dim3(256), 0, st, dG0.p, dCols.p, M1, W,
                     dG1.p);
  PCHK(hipGetLastError());
  const uint64_t *Gfinal = dG1.p;
  int M = M1;
  if (o.doPrune) {
    // getMea (K2 at depth back) on the device, pruning3's sequential sweep on the host (:3318-3325, 669-756)
    int B = o.back, Wd = M1 - 1;
    DBuf<double> dMeaP;
    PCHK(dMeaP.ensure((size_t)B * Wd));
    PCHK(hipMemsetAsync(dMeaP.p, 0, sizeof(double) * (size_t)B * Wd, st));
    hipLaunchKernelGGL(ld_window_kernel, dim3(blocks_for((long)M1 * B, 256)), dim3(256), 0, st, dG1.p, M1, W, B, o.LD,
                       0, dMeaP.p, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr);
    hipError_t le = hipGetLastError();
    std::vector<double> mea((size_t)B * Wd);
    if (le == hipSuccess) le = hipMemcpyAsync(mea.data(), dMeaP.p, sizeof(double) * mea.size(), hipMemcpyDeviceToHost, st);
    if (le == hipSuccess) le = hipStreamSynchronize(st);
    dMeaP.release();
    PCHK(le);
    auto MEA = [&](int r, int c) -> double & { return mea[(size_t)r * Wd + c]; };
    std::vector<int> pk(M1, 0);
    pk[0] = 1;
    if (B > 1) {
      for (int nsnp = 1; nsnp < M1; nsnp++) {
        bool strip = false;
        for (int r = 0; r < B; r++)
          if (MEA(r, nsnp - 1) > o.prune_val) { strip = true; break; }
        if (strip) {
          pk[nsnp] = 0;
          if (B > 2) {
            for (int p = 0; p < B - 1; p++) {
              if (p + nsnp >= Wd) break;
              for (int q = p + 1; q < B - 1; q++) MEA(q, nsnp + p) = MEA(q + 1, p + nsnp);
              MEA(B - 1, nsnp + p) = 0;
            }
          } else if (nsnp < Wd) {
            MEA(1, nsnp) = 0;
          }
        } else
          pk[nsnp] = 1;
      }
    } else {
      for (int i = 1; i < M1; i++) pk[i] = MEA(0, i - 1) <= o.prune_val ? 1 : 0;
    }
    std::vector<int> rows;
    std::vector<int> cols2;
    for (int i = 0; i < M1; i++)
      if (pk[i]) { rows.push_back(i); cols2.push_back(cols[i]); }
      else h->keep[cols[i]] = 0;  // merge_bArrays, :3030-3044
    if ((int)rows.size() < 2) {
      cleanup();
      return fail(h, RELATE_E_ARG, "fewer than 2 SNPs left after pruning");
    }
    cols.swap(cols2);
    M = (int)rows.size();
    PCHK(hipMemcpyAsync(dCols.p, rows.data(), sizeof(int) * M, hipMemcpyHostToDevice, st));
    PCHK(h->dG.ensure((size_t)M * W));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for((long)M * W, 256)), dim3(256), 0, st, dG1.p, dCols.p, M, W,
                       h->dG.p);
    PCHK(hipGetLastError());
    Gfinal = h->dG.p;
  } else {
    PCHK(h->dG.ensure((size_t)M * W));
    PCHK(hipMemcpyAsync(h->dG.p, dG1.p, sizeof(uint64_t) * (size_t)M * W, hipMemcpyDeviceToDevice, st));
    Gfinal = h->dG.p;
  }
  h->M = M;
  // global MAF (calcMaf(genos_global), :3337) and positions
  h->h_pos.resize(M);
  for (int s = 0; s < M; s++) h->h_pos[s] = h->h_pos_raw[cols[s]];
  PCHK(h->dPos.ensure(M));
  PCHK(hipMemcpyAsync(h->dPos.p, h->h_pos.data(), sizeof(double) * M, hipMemcpyHostToDevice, st));
  std::vector<double> tsh((size_t)M + 2, 0.0);  // diff (:391-404) on consecutive SNPs
  std::vector<int> bnd;
  for (int s = 1; s < M; s++) {
    double d = h->h_pos[s] - h->h_pos[s - 1];
    if (d < 0) { d = 10000000.0; bnd.push_back(s); }
    tsh[s] = d;
  }
  PCHK(h->dTsh.ensure(tsh.size()));
  PCHK(hipMemcpyAsync(h->dTsh.p, tsh.data(), sizeof(double) * tsh.size(), hipMemcpyHostToDevice, st));
  h->nBnd = (int)bnd.size();
  PCHK(h->dBnd.ensure(std::max<size_t>(1, bnd.size())));
  if (!bnd.empty()) PCHK(hipMemcpyAsync(h->dBnd.p, bnd.data(), sizeof(int) * bnd.size(), hipMemcpyHostToDevice, st));
  PCHK(hipStreamSynchronize(st));  // the host vectors go out of scope
  PCHK(h->dMaf.ensure(M));
  hipLaunchKernelGGL(maf_kernel, dim3(blocks_for(M, 256)), dim3(256), 0, st, Gfinal, M, W, h->dMaf.p);
  PCHK(hipGetLastError());
  h->h_maf.resize(M);
  PCHK(hipMemcpyAsync(h->h_maf.data(), h->dMaf.p, sizeof(double) * M, hipMemcpyDeviceToHost, st));
  // unconditional emission table
  PCHK(h->dTab1.ensure((size_t)M * 27));
  hipLaunchKernelGGL(table_single_kernel, dim3(blocks_for((long)M * 9, 256)), dim3(256), 0, st, h->dMaf.p, M,
                     o.epsilon, h->dTab1.p);
  PCHK(hipGetLastError());
  h->useLD = (o.ld_adj && o.back > 0) ? 1 : 0;
  if (h->useLD) {
    int B2 = h->back2;
    PCHK(h->dMea.ensure((size_t)B2 * (M - 1)));
    PCHK(h->dPBA.ensure((size_t)B2 * M));
    PCHK(h->dPbA.ensure((size_t)B2 * M));
    PCHK(h->dPBa.ensure((size_t)B2 * M));
    PCHK(h->dPba.ensure((size_t)B2 * M));
    PCHK(hipMemsetAsync(h->dMea.p, 0, sizeof(double) * (size_t)B2 * (M - 1), st));
    PCHK(hipMemsetAsync(h->dPBA.p, 0, sizeof(double) * (size_t)B2 * M, st));
    PCHK(hipMemsetAsync(h->dPbA.p, 0, sizeof(double) * (size_t)B2 * M, st));
    PCHK(hipMemsetAsync(h->dPBa.p, 0, sizeof(double) * (size_t)B2 * M, st));
    PCHK(hipMemsetAsync(h->dPba.p, 0, sizeof(double) * (size_t)B2 * M, st));
    hipLaunchKernelGGL(ld_window_kernel, dim3(blocks_for((long)M * B2, 256)), dim3(256), 0, st, Gfinal, M, W, B2, o.LD,
                       1, h->dMea.p, h->dPBA.p, h->dPBa.p, h->dPbA.p, h->dPba.p);
    PCHK(hipGetLastError());
    PCHK(h->dChoose.ensure(M));
    hipLaunchKernelGGL(choose_global_kernel, dim3(blocks_for(M, 256)), dim3(256), 0, st, h->dMea.p, M, o.back,
                       h->dChoose.p);
    PCHK(hipGetLastError());
  }
  // staging mode: the windowed kernel needs back2 <= 32 and room for the depth table (M x back2 x 1944 B)
  {
    size_t freeb = 0, total = 0;
    if (hipMemGetInfo(&freeb, &total) != hipSuccess) freeb = 0;
    size_t need = h->useLD ? (size_t)M * h->back2 * 243 * sizeof(double) : 0;
    h->winMode = h->back2 <= 32 && need <= freeb / 4;
  }
  // individual-major copy of the genotypes: the windowed kernel's input, and the kept-SNP scans of the generic one
  h->Wt = (M + 31) / 32;
  PCHK(h->dGt.ensure((size_t)N * h->Wt));
  hipLaunchKernelGGL(transpose_geno_kernel, dim3(blocks_for((long)N * h->Wt, 256)), dim3(256), 0, st, Gfinal, N, M, W,
                     h->Wt, h->dGt.p);
  PCHK(hipGetLastError());
  if (h->winMode) {
    if (h->useLD) {
      PCHK(h->dTabD.ensure((size_t)M * h->back2 * 243));
      hipLaunchKernelGGL(table_ld_depth_kernel, dim3(blocks_for((long)M * h->back2 * 81, 256)), dim3(256), 0, st,
                         h->dPBA.p, h->dPbA.p, h->dPBa.p, h->dPba.p, M, h->back2, o.epsilon, h->dTabD.p);
      PCHK(hipGetLastError());
      PCHK(h->dOrder.ensure((size_t)M * h->back2));
      hipLaunchKernelGGL(ld_order_kernel, dim3(blocks_for(M, 256)), dim3(256), 0, st, h->dMea.p, M, h->back2,
                         h->dOrder.p);
      PCHK(hipGetLastError());
    }
    h->dTabL.release();
  } else {
    h->dTabD.release();
    if (h->useLD) {
      PCHK(h->dTabL.ensure((size_t)M * 243));
      hipLaunchKernelGGL(table_ld_kernel, dim3(blocks_for((long)M * 81, 256)), dim3(256), 0, st, h->dChoose.p,
                         h->dPBA.p, h->dPbA.p, h->dPBa.p, h->dPba.p, M, o.epsilon, h->dTabL.p);
      PCHK(hipGetLastError());
    }
  }
  PCHK(hipStreamSynchronize(st));
  cleanup();
#undef PCHK
  h->ready = true;
  h->staged.clear();
  h->stagedV.clear();
  if (keep_out) std::memcpy(keep_out, h->keep.data(), sizeof(int) * Mraw);
  if (M_out) *M_out = M;
  return RELATE_OK;
}

int relate_hip_get_stripped(relate_hip_t h, uint8_t *geno, double *pos, double *maf) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (pos) std::memcpy(pos, h->h_pos.data(), sizeof(double) * h->M);
  if (maf) std::memcpy(maf, h->h_maf.data(), sizeof(double) * h->M);
  if (geno) {
    DBuf<uint8_t> d;
    HIPCHK(h, d.ensure((size_t)h->N * h->M));
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for((long)h->N * h->M, 256)), dim3(256), 0, h->stream, h->dG.p, h->N,
                       h->M, h->W, d.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(geno, d.p, (size_t)h->N * h->M, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    d.release();
    HIPCHK(h, e);
  }
  return RELATE_OK;
}

int relate_hip_get_hap(relate_hip_t h, double *mea, double *pBA, double *pBa, double *pbA, double *pba,
                       int *choose_global) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!h->useLD) return fail(h, RELATE_E_ARG, "LD tables are not built (ld_adj = 0 or back = 0)");
  size_t nm = (size_t)h->back2 * (h->M - 1), nh = (size_t)h->back2 * h->M;
  if (mea) HIPCHK(h, hipMemcpy(mea, h->dMea.p, sizeof(double) * nm, hipMemcpyDeviceToHost));
  if (pBA) HIPCHK(h, hipMemcpy(pBA, h->dPBA.p, sizeof(double) * nh, hipMemcpyDeviceToHost));
  if (pBa) HIPCHK(h, hipMemcpy(pBa, h->dPBa.p, sizeof(double) * nh, hipMemcpyDeviceToHost));
  if (pbA) HIPCHK(h, hipMemcpy(pbA, h->dPbA.p, sizeof(double) * nh, hipMemcpyDeviceToHost));
  if (pba) HIPCHK(h, hipMemcpy(pba, h->dPba.p, sizeof(double) * nh, hipMemcpyDeviceToHost));
  if (choose_global) HIPCHK(h, hipMemcpy(choose_global, h->dChoose.p, sizeof(int) * h->M, hipMemcpyDeviceToHost));
  return RELATE_OK;
}

int relate_hip_pair_emissions(relate_hip_t h, int ind1, int ind2, int *keep, double *t, double *S) {
  int rc = require_ready(h);
  if (rc) return rc;
  int32_t pr[2] = {ind1, ind2};
  rc = stage_block(h, pr, 1, true);  // full format: t, S0, S1, S2 planes of one 64-pair group
  if (rc) return rc;
  std::vector<double> r((size_t)((h->M + 1) / 2) * 2 * kRecPlanesFull * h->recPitchV);
  HIPCHK(h, hipMemcpy(r.data(), h->dRecV.p, sizeof(double) * r.size(), hipMemcpyDeviceToHost));
  for (int s = 0; s < h->M; s++) {
    const double *q = &r[rec_off(s, 0, kRecPlanesFull, h->recPitchV)];  // planes 128 doubles apart
    bool k = !std::signbit(q[0]);  // dropped SNPs carry t = -0.0
    if (keep) keep[s] = k ? 1 : 0;
    if (t) t[s] = k ? q[0] : -1.0;
    if (S) {
      S[s] = k ? q[128] : 0;
      S[(size_t)h->M + s] = k ? q[256] : 0;
      S[(size_t)2 * h->M + s] = k ? q[384] : 0;
    }
  }
  return RELATE_OK;
}

int relate_hip_like(relate_hip_t h, const int32_t *pairs, int P, const double *params, int E, double *negloglik) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || !negloglik || P < 0 || E < 1) return fail(h, RELATE_E_ARG, "bad arguments");
  h->timings[1] = 0;
  size_t lim = pair_block_limit(h, {need_rec(h)});
  std::vector<int32_t> lp;
  for (long b0 = 0; b0 < P; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, P - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (rc) return rc;
    lp.resize((size_t)pb * E);
    for (int p = 0; p < pb; p++)
      for (int e = 0; e < E; e++) lp[(size_t)p * E + e] = p;
    rc = like_lanes(h, pb, lp.data(), params + (size_t)b0 * E * 3, pb * E, negloglik + (size_t)b0 * E, nv_for_points(E));
    if (rc) return rc;
  }
  return RELATE_OK;
}

int relate_hip_like_noalloc(relate_hip_t h, int P, const double *d_params, int E, double *d_negloglik) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!d_params || !d_negloglik || P < 1 || E < 1) return fail(h, RELATE_E_ARG, "bad arguments");
  if ((int)h->staged.size() != 2 * P)
    return fail(h, RELATE_E_ARG, "relate_hip_like_noalloc works on the pair block of the previous relate_hip_like / fit / decode call (" +
                                     std::to_string(h->staged.size() / 2) + " pairs are staged, P = " + std::to_string(P) + ")");
  if ((long)P * E > kNoAllocMaxLanes)
    return fail(h, RELATE_E_ARG, "relate_hip_like_noalloc takes at most " + std::to_string(kNoAllocMaxLanes) +
                                     " lanes (P x E) per call; got " + std::to_string((long)P * E));
  rc = launch_like(h, P * E, LikeIO{LaneMap{nullptr, E}, d_params, d_negloglik, h->scratch_dev, kNoAllocScratchBytes});
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RELATE_OK;
}

namespace {
// Pairs per fitted block: what the records allow, evened out over the blocks of the call (a last block of a fifth of
// the size pays a whole block's tail of small launches for a fifth of the pairs).  RELATE_FIT_BLOCK_MAX caps it (tuning).
size_t fit_block_pairs(relate_hip_t h, long P) {
  size_t lim = pair_block_limit(h, {need_rec(h)});
  // Long rows: blocks beyond ~16 000 pairs fit no faster per pair (measured at M = 5e5 on 61 440 pairs,
  // tools/fitbench.py: blocks of 12 288, 15 360 and 20 480 pairs all at 1 064 - 1 070 pairs/s), and every live run holds
  // 84 KB of optimiser state on the host (all 8 restarts of 16 384 pairs: 11 GB), so the block is capped there.
  if (h->M >= 150000 && lim > 16384) lim = 16384;
  if (const char *e = getenv("RELATE_FIT_BLOCK_MAX")) {
    long v = atol(e);
    if (v >= 64 && (size_t)v < pair_block_limit(h, {need_rec(h)})) lim = (size_t)v;
  }
  if (P > 0 && (size_t)P > lim) {
    const size_t nb = ((size_t)P + lim - 1) / lim;
    lim = ((size_t)P + nb - 1) / nb;
  }
  return lim;
}
}  // namespace

int relate_hip_fit(relate_hip_t h, const int32_t *pairs, int P, const double *starts, relate_fit *out,
                   double *conv_out) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !out || P < 0) return fail(h, RELATE_E_ARG, "bad arguments");
  h->timings[1] = 0;
  for (double &v : h->like_stats) v = 0;
  if (conv_out) std::memset(conv_out, 0, sizeof(double) * (size_t)P * std::max(1, h->o.timesToRun) * 4);
  size_t lim = fit_block_pairs(h, P);
  for (long b0 = 0; b0 < P; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, P - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (rc) return rc;
    rc = fit_block(h, pb, b0, starts, out + b0, conv_out);
    if (rc) return rc;
  }
  return RELATE_OK;
}

int relate_hip_decode(relate_hip_t h, const int32_t *pairs, int P, const double *params, double *post, int8_t *path,
                      int *nsnp_out) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || P < 0) return fail(h, RELATE_E_ARG, "bad arguments");
  h->timings[2] = h->timings[3] = 0;
  // every buffer group counts for what it lacks: records, posteriors, tile checkpoints, and the Viterbi's own buffers
  size_t lim = pair_block_limit(h, {need_rec(h), need_post(h, post ? 24.0 : 0.0), need_ckpt(h), need_path(h, path != nullptr)});
  for (long b0 = 0; b0 < P; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, P - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (rc) return rc;
    rc = decode_block(h, pairs + 2 * b0, 0, pb, params + 3 * b0, post ? post + (size_t)b0 * 3 * h->M : nullptr,
                      path ? path + (size_t)b0 * h->M : nullptr, nsnp_out ? nsnp_out + b0 : nullptr, true);
    if (rc) return rc;
  }
  return RELATE_OK;
}

int relate_hip_decode_rows(relate_hip_t h, const int32_t *pairs, int P, const double *params, int rows, double *post,
                           int *nsnp_out) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || !post || P < 0 || rows < 1 || rows > 3) return fail(h, RELATE_E_ARG, "bad arguments");
  h->timings[2] = h->timings[3] = 0;
  size_t lim = pair_block_limit(h, {need_rec(h), need_post(h, 8.0 * rows), need_ckpt(h)});
  for (long b0 = 0; b0 < P; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, P - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (rc) return rc;
    rc = decode_block(h, pairs + 2 * b0, 0, pb, params + 3 * b0, post + (size_t)b0 * rows * h->M, nullptr,
                      nsnp_out ? nsnp_out + b0 : nullptr, true, nullptr, 0, 0, rows);
    if (rc) return rc;
  }
  return RELATE_OK;
}

int relate_hip_decode_f32(relate_hip_t h, const int32_t *pairs, int P, const double *params, int rows, float *post,
                          int *nsnp_out) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || !post || P < 0 || rows < 1 || rows > 3) return fail(h, RELATE_E_ARG, "bad arguments");
  h->timings[2] = h->timings[3] = 0;
  size_t lim = pair_block_limit(h, {need_rec(h), need_post(h, 4.0 * rows), need_ckpt(h)});
  for (long b0 = 0; b0 < P; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, P - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (rc) return rc;
    rc = decode_block(h, pairs + 2 * b0, 0, pb, params + 3 * b0, nullptr, nullptr, nsnp_out ? nsnp_out + b0 : nullptr,
                      true, post + (size_t)b0 * rows * h->M, rows);
    if (rc) return rc;
  }
  return RELATE_OK;
}

int relate_hip_run_pairs(relate_hip_t h, const int32_t *pairs, int P, const double *starts, relate_fit *out,
                         double *post, int8_t *path) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !out || P < 0) return fail(h, RELATE_E_ARG, "bad arguments");
  h->timings[1] = h->timings[2] = h->timings[3] = 0;
  // The fit wants as many pairs in flight as the records allow (its launches are single sweeps only with tens of
  // thousands of lanes); the posteriors of a fitted block are then decoded in sub-blocks sized to what is left for
  // the posterior and checkpoint buffers, each a range of columns of the records already staged.
  size_t lim = fit_block_pairs(h, P);
  std::vector<double> par;
  for (long b0 = 0; b0 < P; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, P - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (rc) return rc;
    rc = fit_block(h, pb, b0, starts, out + b0, nullptr);
    if (rc) return rc;
    if (!post && !path) continue;
    size_t dlim = pair_block_limit(h, {need_post(h, post ? 24.0 : 0.0), need_ckpt(h), need_path(h, path != nullptr)});
    if (dlim > (size_t)pb) dlim = pb;
    for (long c0 = 0; c0 < pb; c0 += (long)dlim) {
      int pc = (int)std::min<long>((long)dlim, pb - c0);
      par.resize((size_t)3 * pc);
      for (int p = 0; p < pc; p++) {
        const relate_fit &f = out[b0 + c0 + p];
        par[3 * p] = f.a; par[3 * p + 1] = f.k2; par[3 * p + 2] = f.k1;
      }
      // the fit stages the last live pairs of a block again, side by side (FitPop::prepare): then the block's records
      // are gone and the sub-block is staged afresh
      int col0 = (int)c0;
      if ((int)h->staged.size() != 2 * pb) {
        rc = stage_block(h, pairs + 2 * (b0 + c0), pc);
        if (rc) return rc;
        col0 = 0;
      }
      rc = decode_block(h, pairs + 2 * (b0 + c0), col0, pc, par.data(),
                        post ? post + (size_t)(b0 + c0) * 3 * h->M : nullptr,
                        path ? path + (size_t)(b0 + c0) * h->M : nullptr, nullptr, true);
      if (rc) return rc;
    }
  }
  return RELATE_OK;
}

int relate_findmax_bfgs(int numpars, double *invec, double (*fun)(double *x, void *user), void *user,
                        const double *lowbound, const double *upbound, double *neg_f) {
  if (numpars < 1 || numpars > Lbfgsb::kMaxN || !invec || !fun || !lowbound || !upbound) return RELATE_E_ARG;
  FindMax fm;
  fm.begin(numpars, invec, lowbound, upbound);
  while (!fm.done) {
    double v[7];
    for (int q = 0; q < fm.npts; q++) {
      double x[3];
      for (int j = 0; j < numpars; j++) x[j] = fm.pts[q][j];
      v[q] = fun(x, user);
    }
    fm.supply(v);
  }
  for (int j = 0; j < numpars; j++) invec[j] = fm.opt.x[j];
  if (neg_f) *neg_f = -fm.like;
  return RELATE_OK;
}

void relate_rand_stream(unsigned int seed, int n, int *out) {
  GlibcRand r(seed);
  for (int i = 0; i < n; i++) out[i] = r.next();
}

void relate_hip_shard(int64_t P, int rank, int nranks, int64_t *begin, int64_t *end) {
  if (nranks < 1) nranks = 1;
  int64_t q = P / nranks, r = P % nranks;
  int64_t b = rank * q + std::min<int64_t>(rank, r);
  int64_t e = b + q + (rank < r ? 1 : 0);
  if (begin) *begin = b;
  if (end) *end = e;
}

int relate_hip_last_timings(relate_hip_t h, double *out4) {
  if (!h || !out4) return RELATE_E_ARG;
  for (int i = 0; i < 4; i++) out4[i] = h->timings[i];
  return RELATE_OK;
}

int relate_hip_last_like_stats(relate_hip_t h, double *out6) {
  if (!h || !out6) return RELATE_E_ARG;
  for (int i = 0; i < 6; i++) out6[i] = h->like_stats[i];
  return RELATE_OK;
}

int relate_hip_last_chunks(relate_hip_t h, int *like_chunks, int *decode_chunks) {
  if (!h) return RELATE_E_ARG;
  if (like_chunks) *like_chunks = h->last_chunks[0];
  if (decode_chunks) *decode_chunks = h->last_chunks[1];
  return RELATE_OK;
}

int relate_hip_last_gather_ranks(relate_hip_t h) { return h ? h->gather_ranks_seen : 0; }

int relate_hip_bench_like(relate_hip_t h, const int32_t *pairs, int P, const double *params, int E, int reps,
                          double *ms_out) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || P < 1 || E < 1 || reps < 1) return fail(h, RELATE_E_ARG, "bad arguments");
  rc = stage_block(h, pairs, P);
  if (rc) return rc;
  int nl = P * E;
  std::vector<int32_t> lp((size_t)nl);
  for (int p = 0; p < P; p++)
    for (int e = 0; e < E; e++) lp[(size_t)p * E + e] = p;
  HIPCHK(h, h->dLanePair.ensure(nl));
  HIPCHK(h, h->dParams.ensure((size_t)3 * nl));
  HIPCHK(h, h->dOut.ensure(nl));
  HIPCHK(h, hipMemcpyAsync(h->dLanePair.p, lp.data(), sizeof(int32_t) * nl, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dParams.p, params, sizeof(double) * 3 * nl, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  for (int r = 0; r < reps; r++) {
    rc = launch_like(h, P, nl, nv_for_points(E));
    if (rc) return rc;
  }
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  if (ms_out) *ms_out = ms / reps;
  return RELATE_OK;
}

int relate_hip_bench_decode(relate_hip_t h, const int32_t *pairs, int P, const double *params, int reps,
                            double *ms_fwd_out, double *ms_bwd_out) {
  return relate_hip_bench_decode_rows(h, pairs, P, params, 0, reps, ms_fwd_out, ms_bwd_out);
}

int relate_hip_bench_decode_rows(relate_hip_t h, const int32_t *pairs, int P, const double *params, int rows, int reps,
                                 double *ms_fwd_out, double *ms_bwd_out) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || P < 1 || reps < 1 || rows < 0 || rows > 3) return fail(h, RELATE_E_ARG, "bad arguments");
  rc = stage_block(h, pairs, P);
  if (rc) return rc;
  h->timings[2] = h->timings[3] = 0;
  for (int r = 0; r < reps; r++) {
    rc = decode_block(h, pairs, 0, P, params, nullptr, nullptr, nullptr, false, nullptr, 0, 0, rows);
    if (rc) return rc;
  }
  if (ms_fwd_out) *ms_fwd_out = h->timings[2] / reps;
  if (ms_bwd_out) *ms_bwd_out = h->timings[3] / reps;
  return RELATE_OK;
}

}  // extern "C"

// ---- linkage (HMMtest) ---------------------------------------------------------------------------------------
namespace {

double lk_lcg01(int *seed) {  // d_uniform_01, src/HMMtest.txt:1063-1072
  int k = *seed / 127773;
  *seed = 16807 * (*seed - k * 127773) - k * 2836;
  if (*seed < 0) *seed = *seed + 2147483647;
  return (double)(*seed) * 4.656612875E-10;
}

// getCaseList (:693-829): code per pair row: 1 case pair, 2 control pair, 0 unused
void lk_case_codes(const int *cc, int n, int ccAll, uint8_t *code, int *n1, int *n0) {
  std::vector<char> aff(n);
  int na = 0;
  for (int i = 0; i < n; i++) { aff[i] = cc[i] ? 0 : 1; na += aff[i]; }
  *n1 = na;
  *n0 = n - na;
  size_t row = 0;
  for (int i = 0; i < n - 1; i++)
    for (int j = i + 1; j < n; j++, row++) {
      int v;
      if (aff[i] && aff[j]) v = 1;
      else if (aff[j]) v = -1;  // the reference's second branch (:785): excluded whatever ccAll says
      else if (!aff[i] && !aff[j]) v = 2;
      else v = 0;
      if (v == 0 && ccAll == 2) v = 2;
      else if (v == 0 && ccAll == 1) v = 1;
      code[row] = v > 0 ? (uint8_t)v : 0;
    }
}

// permutate_dFileList (:1113-1143) with perm_uniform (:1094-1109)
void lk_perm_codes(int n, int n1, int seed, uint8_t *code) {
  std::vector<int> p(n);
  std::vector<char> in(n, 0);
  for (int i = 1; i <= n; i++) p[i - 1] = i;
  for (int i = 1; i <= n; i++) {
    int j = i + (int)(lk_lcg01(&seed) * (double)(n + 1 - i));
    if (j < i) j = i;
    if (n < j) j = n;
    std::swap(p[i - 1], p[j - 1]);
  }
  for (int i = 0; i < n1; i++) in[p[i] - 1] = 1;
  size_t row = 0;
  for (int i = 0; i < n - 1; i++)
    for (int j = i + 1; j < n; j++, row++) code[row] = (in[i] && in[j]) ? 1 : ((!in[i] && !in[j]) ? 2 : 0);
}

}  // namespace

namespace {
// HMMtest on a posterior matrix that is already in device memory (dPp: R x M floats, overwritten by the inference of
// missing values when infer = 1)
int linkage_on_device(relate_hip_t h, float *dPp, int R, int M, const float *pos, const float *K, const int *cc, int n_ind,
                      const relate_linkage_opts *o, float *stat, float *xmax, float *covar, float *post_out,
                      relate_linkage_result *res) {
  if ((long)n_ind * (n_ind - 1) / 2 != R)
    return fail(h, RELATE_E_ARG, "the posterior file must hold one row per pair of the individuals in the -D file (" +
                                     std::to_string((long)n_ind * (n_ind - 1) / 2) + " rows, got " + std::to_string(R) + ")");
  if (o->infer == 1 && !pos) return fail(h, RELATE_E_ARG, "Position file required for removing missing");
  hipStream_t st = h->stream;
  const int nsim = o->nsim;
  DBuf<float> dPos, dK, dCov, dU1, dU0, dScal, dX0, dXmax;
  DBuf<int> dPsim;
  auto cleanup = [&]() {
    dPos.release(); dK.release(); dCov.release(); dU1.release(); dU0.release(); dScal.release();
    dX0.release(); dXmax.release(); dPsim.release();
  };
#define LCHK(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      cleanup();                                                                                \
      return fail(h, e_ == hipErrorOutOfMemory ? RELATE_E_NOMEM : RELATE_E_HIP,                 \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
    }                                                                                           \
  } while (0)
  int missing = 1;
  if (o->infer == 1) {
    LCHK(dPos.ensure(M));
    LCHK(hipMemcpyAsync(dPos.p, pos, sizeof(float) * M, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lk_fix_missing_kernel, dim3(blocks_for(R, 64)), dim3(64), 0, st, dPp, R, M, dPos.p);
    LCHK(hipGetLastError());
    missing = 0;
  }
  LCHK(dK.ensure(R));
  if (K) {
    std::vector<float> k(R);
    for (int r = 0; r < R; r++) k[r] = o->isK0 ? 1 - K[r] : K[r];
    LCHK(hipMemcpyAsync(dK.p, k.data(), sizeof(float) * R, hipMemcpyHostToDevice, st));
    LCHK(hipStreamSynchronize(st));
  } else {
    hipLaunchKernelGGL(lk_row_mean_kernel, dim3(blocks_for(R, 64)), dim3(64), 0, st, dPp, R, M, missing, dK.p);
    LCHK(hipGetLastError());
  }
  LCHK(dCov.ensure((size_t)R * R));
  {
    unsigned g = (unsigned)((R + kCovT - 1) / kCovT);
    hipLaunchKernelGGL(lk_covar_kernel, dim3(g, g), dim3(256), 0, st, dPp, dK.p, R, M, missing, dCov.p);
    LCHK(hipGetLastError());
  }
  // Case lists and permutations (host: integer bookkeeping on the glibc rand() / LCG streams).  The reference keeps
  // two list arrays whose LENGTHS are fixed by getCaseList; every permutation overwrites their front
  // (generate_affectLists, :455-470).  With ccAll != 0 the observed lists also hold the mixed pairs, a permutation
  // yields only C(n1,2) / C(n0,2) pairs, and the tail keeps entries of the previous lists.  Kept, list by list.
  std::vector<uint8_t> code(R);
  int n1 = 0, n0 = 0;
  lk_case_codes(cc, n_ind, o->ccAll, code.data(), &n1, &n0);
  int LA = 0, LN = 0;
  for (int r = 0; r < R; r++) { LA += code[r] == 1; LN += code[r] == 2; }
  if (LA < 1 || LN < 1) { cleanup(); return fail(h, RELATE_E_ARG, "the case/control file leaves no case pairs or no control pairs"); }
  std::vector<int> la_all((size_t)nsim * LA), ln_all((size_t)nsim * LN);
  {
    std::vector<int> la(LA), ln(LN);
    auto fill = [&]() {
      int a = 0, n = 0;
      for (int r = 0; r < R; r++) {
        if (code[r] == 1) la[a++] = r;
        else if (code[r] == 2) ln[n++] = r;
      }
    };
    GlibcRand rng(o->seed);  // srand(time(NULL)) (:1296); one rand() per permutation (:1116)
    for (int i = 0; i < nsim; i++) {
      if (i > 0) lk_perm_codes(n0 + n1, n1, rng.next(), code.data());
      fill();
      std::copy(la.begin(), la.end(), la_all.begin() + (size_t)i * LA);
      std::copy(ln.begin(), ln.end(), ln_all.begin() + (size_t)i * LN);
    }
  }
  DBuf<int> dLa, dLn;
  auto cleanup2 = [&]() { dLa.release(); dLn.release(); cleanup(); };
#define L2CHK(expr)                                                                             \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      cleanup2();                                                                               \
      return fail(h, e_ == hipErrorOutOfMemory ? RELATE_E_NOMEM : RELATE_E_HIP,                 \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
    }                                                                                           \
  } while (0)
  L2CHK(dLa.ensure(la_all.size()));
  L2CHK(dLn.ensure(ln_all.size()));
  L2CHK(hipMemcpyAsync(dLa.p, la_all.data(), sizeof(int) * la_all.size(), hipMemcpyHostToDevice, st));
  L2CHK(hipMemcpyAsync(dLn.p, ln_all.data(), sizeof(int) * ln_all.size(), hipMemcpyHostToDevice, st));
  L2CHK(dX0.ensure(M));
  L2CHK(dXmax.ensure(nsim));
  L2CHK(dPsim.ensure(M));
  L2CHK(hipMemsetAsync(dPsim.p, 0, sizeof(int) * M, st));
  const int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsim, ((size_t)256 << 20) / ((size_t)M * 8)));
  L2CHK(dU1.ensure((size_t)batch * M));
  L2CHK(dU0.ensure((size_t)batch * M));
  L2CHK(dScal.ensure((size_t)3 * batch));
  for (int b0 = 0; b0 < nsim; b0 += batch) {
    int nb = std::min(batch, nsim - b0);
    const int *a = dLa.p + (size_t)b0 * LA, *n = dLn.p + (size_t)b0 * LN;
    hipLaunchKernelGGL(lk_col_means_kernel, dim3(blocks_for(M, 256), nb), dim3(256), 0, st, dPp, M, a, LA, n, LN, missing,
                       dU1.p, dU0.p);
    L2CHK(hipGetLastError());
    const size_t list_bytes = sizeof(int) * ((size_t)LA + LN);
    const int lds = list_bytes <= 48 * 1024 ? 1 : 0;
    hipLaunchKernelGGL(lk_perm_scalars_kernel, dim3(nb), dim3(64), lds ? list_bytes : 0, st, dK.p, dCov.p, R, a, LA, n, LN,
                       nb, lds, dScal.p);
    L2CHK(hipGetLastError());
    hipLaunchKernelGGL(lk_statistic_kernel, dim3(blocks_for(M, 256), nb), dim3(256), 0, st, dU1.p, dU0.p, dScal.p, M);
    L2CHK(hipGetLastError());
    if (b0 == 0) L2CHK(hipMemcpyAsync(dX0.p, dU1.p, sizeof(float) * M, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(lk_perm_reduce_kernel, dim3(nb), dim3(256), 0, st, dU1.p, dX0.p, M, nb, b0 == 0 ? 1 : 0,
                       dXmax.p + b0, dPsim.p);
    L2CHK(hipGetLastError());
  }
  const int n_case = LA, n_ctrl = LN;
  std::vector<float> hx(nsim);
  std::vector<int> hps(M);
  L2CHK(hipMemcpyAsync(hx.data(), dXmax.p, sizeof(float) * nsim, hipMemcpyDeviceToHost, st));
  L2CHK(hipMemcpyAsync(hps.data(), dPsim.p, sizeof(int) * M, hipMemcpyDeviceToHost, st));
  if (stat) L2CHK(hipMemcpyAsync(stat, dX0.p, sizeof(float) * M, hipMemcpyDeviceToHost, st));
  if (covar) L2CHK(hipMemcpyAsync(covar, dCov.p, sizeof(float) * (size_t)R * R, hipMemcpyDeviceToHost, st));
  if (post_out) L2CHK(hipMemcpyAsync(post_out, dPp, sizeof(float) * (size_t)R * M, hipMemcpyDeviceToHost, st));
  L2CHK(hipStreamSynchronize(st));
  cleanup2();
#undef LCHK
#undef L2CHK
  if (xmax) std::memcpy(xmax, hx.data(), sizeof(float) * nsim);
  if (res) {  // the printed summaries (:1349-1362): counting, the reference's float accumulations, its sort
    float pmax = 1;
    for (int i = 1; i < nsim; i++)
      if (hx[0] <= hx[i]) pmax++;
    res->pmaxsim = pmax / nsim;
    float ps = 0;
    for (int s = 0; s < M; s++) ps += hps[s];
    res->psim_over_nsim = ps / (float)nsim;
    std::vector<float> sorted(hx);
    std::sort(sorted.begin(), sorted.end());
    int at = (int)std::round((double)((1 - o->sig) * nsim)) - 1;
    res->psigsim = (at >= 0 && at < nsim) ? sorted[at] : NAN;
    res->n_case_pairs = n_case;
    res->n_control_pairs = n_ctrl;
  }
  return RELATE_OK;
}
}  // namespace

extern "C" int relate_hip_linkage(relate_hip_t h, const float *post, int R, int M, const float *pos, const float *K,
                                  const int *cc, int n_ind, const relate_linkage_opts *o, float *stat, float *xmax,
                                  float *covar, float *post_out, relate_linkage_result *res) {
  if (!h) return fail(nullptr, RELATE_E_ARG, "null handle");
  if (!post || !cc || !o || R < 1 || M < 2 || n_ind < 2 || o->nsim < 1) return fail(h, RELATE_E_ARG, "bad arguments");
  if (set_device(h)) return RELATE_E_HIP;
  DBuf<float> dP;
  hipError_t e = dP.ensure((size_t)R * M);
  if (e == hipSuccess) e = hipMemcpyAsync(dP.p, post, sizeof(float) * (size_t)R * M, hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) {
    dP.release();
    return fail(h, e == hipErrorOutOfMemory ? RELATE_E_NOMEM : RELATE_E_HIP, std::string("posterior matrix: ") + hipGetErrorString(e));
  }
  int rc = linkage_on_device(h, dP.p, R, M, pos, K, cc, n_ind, o, stat, xmax, covar, post_out, res);
  dP.release();
  return rc;
}

// Decode -> linkage without leaving HBM: the IBD0 posterior row of every pair (the row the .post file of the linkage
// pipeline holds, src/relateHMM.cpp:3771-3782 with fixK2 = 1, value 0) is decoded block by block into one R x M float
// matrix on the device and handed to the linkage kernels.  digits = 4 rounds every value to the four significant
// digits the reference's .post text carries (POSTPRECISION, src/conf.h:8), so that the statistic is the one HMMtest
// computes from the file; digits = 0 keeps the float posteriors.
extern "C" int relate_hip_linkage_pairs(relate_hip_t h, const int32_t *pairs, int R, const double *params, int digits,
                                        const float *pos, const float *K, const int *cc, int n_ind,
                                        const relate_linkage_opts *o, float *stat, float *xmax, float *covar,
                                        float *post_out, relate_linkage_result *res) {
  int rc = require_ready(h);
  if (rc) return rc;
  if (!pairs || !params || !cc || !o || R < 1 || n_ind < 2 || o->nsim < 1 || (digits != 0 && digits != 4))
    return fail(h, RELATE_E_ARG, "bad arguments");
  const int M = h->M;
  DBuf<float> dAll;
  HIPCHK(h, dAll.ensure((size_t)R * M));
  h->timings[2] = h->timings[3] = 0;
  size_t lim = pair_block_limit(h, {need_rec(h), need_post(h, 4.0), need_ckpt(h)});
  for (long b0 = 0; b0 < R && !rc; b0 += (long)lim) {
    int pb = (int)std::min<long>((long)lim, R - b0);
    rc = stage_block(h, pairs + 2 * b0, pb);
    if (!rc)
      rc = decode_block(h, pairs + 2 * b0, 0, pb, params + 3 * b0, nullptr, nullptr, nullptr, false, nullptr, 1,
                        digits == 4 ? 1 : 0);
    if (!rc) {
      hipError_t e = hipMemcpyAsync(dAll.p + (size_t)b0 * M, h->dPost.p, sizeof(float) * (size_t)pb * M,
                                    hipMemcpyDeviceToDevice, h->stream);
      if (e != hipSuccess) rc = fail(h, RELATE_E_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
  }
  if (!rc) {
    std::vector<float> fpos;
    if (!pos) {  // the stripped positions, as a float reader of stripped.pos sees them
      fpos.resize(M);
      for (int s = 0; s < M; s++) fpos[s] = (float)h->h_pos[s];
      pos = fpos.data();
    }
    rc = linkage_on_device(h, dAll.p, R, M, pos, K, cc, n_ind, o, stat, xmax, covar, post_out, res);
  }
  dAll.release();
  return rc;
}

// ---- multi-GPU: the path's one exchange -------------------------------------------------------------------------
// Pairs are independent, so ranks work on contiguous shards of the pair list (relate_hip_shard) and meet once, to
// gather the per-pair result records.  Over RCCL (xGMI inside a node): one all-gather of the padded shards on the
// handle's stream; the communicator is made at the first gather of a handle from a ncclUniqueId that rank 0 leaves
// in a file next to `rendezvous`.  RCCL is looked up in the process at run time (a host program of the library need
// not link it; in a torch process it is torch's copy).  transport 1 moves the records through files on a filesystem
// every rank sees: no GPU involved, for ranks that share one GPU and for tests.
namespace {

struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional: how many ranks the communicator holds
  bool ok() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather && GetErrorString; }
};
RcclApi &rccl() {
  static RcclApi api = [] {
    RcclApi a;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // a copy the process already has (torch's) comes first
      if (a.lib) break;
    }
    if (!a.lib)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
      }
    if (a.lib) {
      a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
      a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
      a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
      a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
      a.CommCount = (decltype(a.CommCount))dlsym(a.lib, "ncclCommCount");
    }
    return a;
  }();
  return api;
}

void gather_release(relate_hip_t h) {
  if (h->comm && rccl().ok()) (void)rccl().CommDestroy(h->comm);
  h->comm = nullptr;
  h->dGatherS.release();
  h->dGatherR.release();
}

bool write_file_atomically(const std::string &path, const void *data, size_t bytes) {
  std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  bool ok = std::fwrite(data, 1, bytes, f) == bytes;
  ok = (std::fclose(f) == 0) && ok;
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}
// true when the file is there with exactly this size and was read in full
bool read_file_exact(const std::string &path, void *data, size_t bytes) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0 || (size_t)st.st_size != bytes) return false;
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = std::fread(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok;
}
double gather_timeout() {
  const char *e = getenv("RELATE_GATHER_TIMEOUT");
  return e ? atof(e) : 1800.0;
}

// ---- rendezvous through files -----------------------------------------------------------------------------------
// The ranks of a gather are separate processes that share nothing but a path.  Files of an earlier run under the same
// path (a run that died, or a second run in the same directory) must never be taken for this run's, so every file
// carries a nonce and the ranks first agree on a session:
//   rank r > 0 writes  <base>.hello.<r> = {its fresh nonce};
//   rank 0 writes      <base>.session   = {its fresh nonce0, state PROPOSE, the nonces it read, [ncclUniqueId]};
//   rank r finds its own nonce echoed there (a session that lists another nonce for it is stale, or was made from a
//   stale hello: it keeps waiting, rank 0 re-reads the hellos and proposes again) and writes <base>.ack.<r> =
//   {nonce0, nonce_r};
//   rank 0, holding every ack for (nonce0, nonce_r), rewrites the session with state GO; the ranks remove their hello
//   and ack files and go on.  Every later file of the exchange carries nonce0; rank 0 removes them all at the end.
constexpr uint64_t kRdvMagic = 0x333045544c4552ull;  // "RELTE03"
struct RdvHello { uint64_t magic, nonce; };
struct RdvAck { uint64_t magic, nonce0, nonce; };
struct RdvMark { uint64_t magic, nonce0; int64_t count; };
struct RdvSession {
  uint64_t magic, nonce0;
  int32_t state, nranks;  // state 0 = PROPOSE, 1 = GO
  int64_t P_total;
  unsigned char id[128];  // ncclUniqueId (transport 0)
};
static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId does not fit the session record");

uint64_t fresh_nonce() {
  uint64_t v = 0;
  if (FILE *f = std::fopen("/dev/urandom", "rb")) {
    if (std::fread(&v, 1, sizeof v, f) != sizeof v) v = 0;
    std::fclose(f);
  }
  v ^= (uint64_t)getpid() << 32;
  v ^= (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  v ^= (uint64_t)std::chrono::system_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ull;
  return v ? v : 1;
}

struct Deadline {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double seconds;
  explicit Deadline(double s) : seconds(s) {}
  bool over() const {
    return seconds >= 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds;
  }
};

// returns "" when the session stands (nonce0 and, if id is given, the ncclUniqueId: rank 0 supplies it, the others
// receive it), or what went wrong
std::string rendezvous_session(const std::string &base, int rank, int nranks, int64_t P_total, unsigned char *id,
                               uint64_t *nonce0, const Deadline &dl) {
  const size_t sbytes = sizeof(RdvSession) + sizeof(uint64_t) * (size_t)nranks;
  std::vector<unsigned char> buf(sbytes);
  RdvSession *S = (RdvSession *)buf.data();
  uint64_t *nonces = (uint64_t *)(buf.data() + sizeof(RdvSession));
  const std::string sess = base + ".session";
  if (rank > 0) {
    const std::string hello = base + ".hello." + std::to_string(rank), ack = base + ".ack." + std::to_string(rank);
    RdvHello me{kRdvMagic, fresh_nonce()};
    if (!write_file_atomically(hello, &me, sizeof me)) return "cannot write " + hello;
    uint64_t acked = 0;
    for (;;) {
      if (read_file_exact(sess, buf.data(), sbytes) && S->magic == kRdvMagic && S->nranks == nranks &&
          nonces[rank] == me.nonce) {
        if (S->P_total != P_total)
          return "the ranks disagree on the number of pairs (" + std::to_string((long long)S->P_total) + " on rank 0, " +
                 std::to_string((long long)P_total) + " here)";
        if (S->state == 1 && acked == S->nonce0) {
          *nonce0 = S->nonce0;
          if (id) std::memcpy(id, S->id, 128);
          std::remove(hello.c_str());
          std::remove(ack.c_str());
          return "";
        }
        if (S->state == 0 && acked != S->nonce0) {
          RdvAck a{kRdvMagic, S->nonce0, me.nonce};
          if (!write_file_atomically(ack, &a, sizeof a)) return "cannot write " + ack;
          acked = S->nonce0;
        }
      }
      if (dl.over()) return "rank 0 did not open a session at " + sess;
      usleep(2000);
    }
  }
  std::vector<uint64_t> seen((size_t)nranks, 0);
  std::memset(buf.data(), 0, sbytes);
  S->magic = kRdvMagic;
  S->nonce0 = *nonce0 = fresh_nonce();
  S->nranks = nranks;
  S->P_total = P_total;
  if (id) std::memcpy(S->id, id, 128);
  std::remove(sess.c_str());  // whatever an earlier run left
  bool published = false;
  for (;;) {
    bool changed = false, all = true;
    for (int r = 1; r < nranks; r++) {
      RdvHello hl;
      if (read_file_exact(base + ".hello." + std::to_string(r), &hl, sizeof hl) && hl.magic == kRdvMagic && hl.nonce &&
          hl.nonce != seen[(size_t)r]) {
        seen[(size_t)r] = hl.nonce;
        changed = true;
      }
      all = all && seen[(size_t)r] != 0;
    }
    if (all && (changed || !published)) {
      for (int r = 0; r < nranks; r++) nonces[r] = seen[(size_t)r];
      S->state = 0;
      if (!write_file_atomically(sess, buf.data(), sbytes)) return "cannot write " + sess;
      published = true;
    } else if (published) {
      bool acked = true;
      for (int r = 1; r < nranks && acked; r++) {
        RdvAck a;
        acked = read_file_exact(base + ".ack." + std::to_string(r), &a, sizeof a) && a.magic == kRdvMagic &&
                a.nonce0 == S->nonce0 && a.nonce == seen[(size_t)r];
      }
      if (acked) {
        S->state = 1;
        if (!write_file_atomically(sess, buf.data(), sbytes)) return "cannot write " + sess;
        return "";
      }
    }
    if (dl.over()) {
      for (int r = 1; r < nranks; r++)
        if (!seen[(size_t)r]) return "rank " + std::to_string(r) + " did not arrive at " + base;
      return "the ranks did not acknowledge the session at " + sess;
    }
    usleep(2000);
  }
}

}  // namespace

extern "C" int relate_hip_gather(relate_hip_t h, const relate_fit *mine, int64_t P_total, int rank, int nranks,
                                 const char *rendezvous, int transport, relate_fit *all) {
  if (P_total < 0 || nranks < 1 || rank < 0 || rank >= nranks || !all || (!mine && P_total > 0))
    return fail(h, RELATE_E_ARG, "bad arguments");
  if (transport != 0 && transport != 1) return fail(h, RELATE_E_ARG, "transport: 0 = RCCL, 1 = files");
  if (transport == 0 && !h) return fail(nullptr, RELATE_E_ARG, "the RCCL gather needs a handle (its GPU and stream)");
  if (nranks > 1 && (!rendezvous || !*rendezvous))
    return fail(h, RELATE_E_ARG, "a rendezvous path every rank sees is needed for more than one rank");
  int64_t b = 0, e = 0;
  relate_hip_shard(P_total, rank, nranks, &b, &e);
  const long seq = h ? h->gather_seq++ : 0;
  const std::string base = std::string(rendezvous ? rendezvous : "") + ".g" + std::to_string(seq);
  const Deadline dl(gather_timeout());
  if (h) h->gather_ranks_seen = 0;
  if (transport == 1) {
    if (nranks == 1) {
      std::memcpy(all, mine, sizeof(relate_fit) * (size_t)(e - b));
      if (h) h->gather_ranks_seen = 1;
      return RELATE_OK;
    }
    uint64_t nonce0 = 0;
    std::string err = rendezvous_session(base, rank, nranks, P_total, nullptr, &nonce0, dl);
    if (!err.empty()) return fail(h, RELATE_E_ARG, err);
    auto shard_file = [&](int r) { return base + ".fit." + std::to_string(r); };
    auto done_file = [&](int r) { return base + ".done." + std::to_string(r); };
    {
      std::vector<unsigned char> out(sizeof(RdvMark) + sizeof(relate_fit) * (size_t)(e - b));
      RdvMark mk{kRdvMagic, nonce0, e - b};
      std::memcpy(out.data(), &mk, sizeof mk);
      if (e > b) std::memcpy(out.data() + sizeof mk, mine, sizeof(relate_fit) * (size_t)(e - b));
      if (!write_file_atomically(shard_file(rank), out.data(), out.size()))
        return fail(h, RELATE_E_ARG, "cannot write " + shard_file(rank));
    }
    std::vector<unsigned char> in;
    for (int r = 0; r < nranks; r++) {
      int64_t rb = 0, re = 0;
      relate_hip_shard(P_total, r, nranks, &rb, &re);
      in.resize(sizeof(RdvMark) + sizeof(relate_fit) * (size_t)(re - rb));
      for (;;) {  // a file of another session (an earlier run's) is not this rank's shard: wait for the real one
        const RdvMark *mk = (const RdvMark *)in.data();
        if (read_file_exact(shard_file(r), in.data(), in.size()) && mk->magic == kRdvMagic && mk->nonce0 == nonce0 &&
            mk->count == re - rb)
          break;
        if (dl.over()) return fail(h, RELATE_E_ARG, "rank " + std::to_string(r) + " did not deliver " + shard_file(r));
        usleep(2000);
      }
      if (re > rb) std::memcpy(all + rb, in.data() + sizeof(RdvMark), sizeof(relate_fit) * (size_t)(re - rb));
    }
    RdvMark dn{kRdvMagic, nonce0, 0};
    if (!write_file_atomically(done_file(rank), &dn, sizeof dn)) return fail(h, RELATE_E_ARG, "cannot write " + done_file(rank));
    if (rank == 0) {  // every rank has read every shard: the files of this exchange go
      for (int r = 1; r < nranks; r++)
        for (;;) {
          RdvMark m;
          if (read_file_exact(done_file(r), &m, sizeof m) && m.magic == kRdvMagic && m.nonce0 == nonce0) break;
          if (dl.over()) return fail(h, RELATE_E_ARG, "rank " + std::to_string(r) + " did not finish reading (" + done_file(r) + ")");
          usleep(2000);
        }
      for (int r = 0; r < nranks; r++) {
        std::remove(shard_file(r).c_str());
        std::remove(done_file(r).c_str());
      }
      std::remove((base + ".session").c_str());
    }
    if (h) h->gather_ranks_seen = nranks;
    return RELATE_OK;
  }
  if (set_device(h)) return RELATE_E_HIP;
  RcclApi &N = rccl();
  if (!N.ok()) return fail(h, RELATE_E_HIP, std::string("librccl.so not found in the process or on the library path: ") + (dlerror() ? dlerror() : ""));
#define NCHK(expr)                                                                                    \
  do {                                                                                                \
    ncclResult_t r_ = (expr);                                                                         \
    if (r_ != ncclSuccess) return fail(h, RELATE_E_HIP, std::string(#expr) + ": " + N.GetErrorString(r_)); \
  } while (0)
  bool made_session = false;
  if (!h->comm || h->comm_rank != rank || h->comm_n != nranks) {
    if (h->comm) { (void)N.CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    unsigned char idbuf[128] = {0};
    if (rank == 0) {
      NCHK(N.GetUniqueId(&id));
      std::memcpy(idbuf, &id, sizeof id);
    }
    if (nranks > 1) {
      uint64_t nonce0 = 0;
      std::string err = rendezvous_session(base, rank, nranks, P_total, idbuf, &nonce0, dl);
      if (!err.empty()) return fail(h, RELATE_E_ARG, err);
      std::memcpy(&id, idbuf, sizeof id);
      made_session = true;
    }
    NCHK(N.CommInitRank(&h->comm, nranks, id, rank));
    h->comm_rank = rank;
    h->comm_n = nranks;
  }
  {
    int seen = 0;
    if (N.CommCount && N.CommCount(h->comm, &seen) == ncclSuccess) h->gather_ranks_seen = seen;
  }
  if (P_total == 0) return RELATE_OK;  // nothing to exchange (every rank knows it: P_total is the same on all of them)
  // equal counts per rank: shards are padded to the longest one (they differ by at most one record)
  const size_t per = (size_t)((P_total + nranks - 1) / nranks) * sizeof(relate_fit);
  HIPCHK(h, h->dGatherS.ensure(per));
  HIPCHK(h, h->dGatherR.ensure(per * nranks));
  HIPCHK(h, hipMemsetAsync(h->dGatherS.p, 0, std::max<size_t>(per, 1), h->stream));
  HIPCHK(h, hipMemcpyAsync(h->dGatherS.p, mine, sizeof(relate_fit) * (size_t)(e - b), hipMemcpyHostToDevice, h->stream));
  NCHK(N.AllGather(h->dGatherS.p, h->dGatherR.p, per, ncclInt8, h->comm, h->stream));
  std::vector<unsigned char> host(per * nranks);
  HIPCHK(h, hipMemcpyAsync(host.data(), h->dGatherR.p, host.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
#undef NCHK
  // the all-gather has completed here, so every rank has joined the communicator, i.e. has read the session
  if (made_session && rank == 0) std::remove((base + ".session").c_str());
  for (int r = 0; r < nranks; r++) {
    int64_t rb = 0, re = 0;
    relate_hip_shard(P_total, r, nranks, &rb, &re);
    std::memcpy(all + rb, host.data() + per * r, sizeof(relate_fit) * (size_t)(re - rb));
  }
  return RELATE_OK;
}
