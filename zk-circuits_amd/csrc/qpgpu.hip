// C-ABI entry points of the MI355X prover backend (include/qpgpu.h): host orchestration of
// prove_with_partition_witness (SURVEY.md §3.3 stages 1-10; reference call sites wormhole/prover/src/lib.rs:233-237 and
// wormhole/aggregator/src/circuits/tree.rs:136).
//
// Everything heavy (iNTT/LDE, Poseidon Merkle trees, Z/partial products, quotient, openings, FRI combine/fold, PoW grind,
// query gather) runs as HIP kernels on a per-call stream with a per-call workspace. The Fiat-Shamir transcript
// (row a6, ~120 sequential Poseidon permutations per proof) is inherently serial, so it runs on the host between
// stages, exactly where upstream's Challenger runs; each challenge costs one small D2H copy of a cap / opening set.
#include "../../include/qpgpu_internal.h"
#include "kernels.h"
#include "options.h"
#include "hostcpus.h"
#include "salt.h"
#include "poseidon.h"
#include "proof_format.h"
#include <string>
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <memory>
#include <chrono>
#include <time.h>
#include <sched.h>

namespace qp {
thread_local std::string g_last_error;
int set_err(int code, const std::string& m) { g_last_error = m; return code; }

// ---- options (options.h): the one place that reads the environment
// A set of options is immutable once published: a refresh that finds the environment unchanged keeps the current set, one that finds a
// change publishes a NEW set and leaves the old one alive (a few hundred bytes per deliberate change of the environment), so a proof
// that is running while another thread creates a context never sees a buffer being rewritten under it, and — unless the host changes
// the variables on purpose — never sees a knob change mid-proof either.
static std::atomic<const Options*> g_opts_cur{nullptr}; static std::mutex g_opts_mu;
static const Options g_opts_default{};
const Options& opts() { const Options* p = g_opts_cur.load(std::memory_order_acquire); return p ? *p : g_opts_default; }
void options_refresh_from_env() {
  std::lock_guard<std::mutex> lk(g_opts_mu);
  Options o;
  auto geti = [](const char* k, int dflt) { const char* v = getenv(k); return v && *v ? atoi(v) : dflt; };
  // Worker streams only overlap if each gets a hardware queue of its own: ROCm maps a process's streams onto GPU_MAX_HW_QUEUES queues
  // (default 4), and a worker uses two streams. With >= 8 queues eight workers run 383 proofs/s on config B against 356 for four (r02 notes).
  o.streams = geti("GPU_MAX_HW_QUEUES", 4) >= 8 ? 8 : 4;
  o.streams = geti("QPGPU_STREAMS", o.streams); o.handoff_spin_us = geti("QPGPU_HANDOFF_SPIN_US", o.handoff_spin_us);
  o.no_publish = geti("QPGPU_NO_PUBLISH", o.no_publish); o.timing = geti("QPGPU_TIMING", o.timing);
  o.split_quotient = geti("QPGPU_SPLIT_QUOTIENT", o.split_quotient); o.split_fork_late = geti("QPGPU_SPLIT_FORK_LATE", o.split_fork_late);
  o.split_serial = geti("QPGPU_SPLIT_SERIAL", o.split_serial); o.quotient_halves = geti("QPGPU_QUOTIENT_HALVES", o.quotient_halves);
  o.shared_mode = geti("QPGPU_SHARED_MODE", o.shared_mode); o.l0_table = geti("QPGPU_L0_TABLE", o.l0_table);
  o.pair_leaves = geti("QPGPU_PAIR_LEAVES", o.pair_leaves); o.pair_max = geti("QPGPU_PAIR_MAX", o.pair_max); o.coop_max = geti("QPGPU_COOP_MAX", o.coop_max);
  o.fuse_levels = geti("QPGPU_FUSE_LEVELS", o.fuse_levels); o.ntt16 = geti("QPGPU_NTT16", o.ntt16); o.ntt_strided = geti("QPGPU_NTT_STRIDED", o.ntt_strided);
  o.ntt_m8 = geti("QPGPU_NTT_M8", o.ntt_m8);
  o.quotient_shared_loads = geti("QPGPU_QUOTIENT_SHARED_LOADS", o.quotient_shared_loads);
  o.pow_first_shift = geti("QPGPU_POW_FIRST_SHIFT", o.pow_first_shift); o.poseidon_fast = geti("QPGPU_POSEIDON_FAST", o.poseidon_fast);
  o.verify_coop_max = geti("QPGPU_VERIFY_COOP_MAX", o.verify_coop_max);
  const Options* cur = g_opts_cur.load(std::memory_order_relaxed);
  if (cur && memcmp(cur, &o, sizeof o) == 0) return;                 // all-int struct, no padding: bytewise equality is value equality
  g_opts_cur.store(new Options(o), std::memory_order_release);       // never freed: readers may still hold the previous set
}
static_assert(sizeof(Options) % sizeof(int) == 0, "Options is plain ints");
// The library never touches the host's environment. Eight worker streams need eight hardware queues: the HOST exports GPU_MAX_HW_QUEUES=8
// before HIP initialises if it wants them (zk-circuits_amd/ffi.py and bench.py do), otherwise four workers run.
namespace { struct OptionsInit { OptionsInit() { options_refresh_from_env(); } } g_options_init; }
}
using namespace qp;

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(e_ == hipErrorOutOfMemory ? QPGPU_ERR_OOM : QPGPU_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace {

struct DevArena {
  char* base = nullptr; size_t size = 0, used = 0;
  u64* take(size_t elems) { size_t b = (elems * 8 + 255) & ~size_t(255); char* p = base + used; used += b; return (u64*)p; }
};

// ---- optional per-kernel-group timing (HIP events on the proving stream), used by bench.py for the roofline object
enum ProfId { PF_INTT = 0, PF_LDE, PF_LEAVES, PF_LEVELS, PF_ZPP, PF_QUOTIENT, PF_QUOTIENT_INTT, PF_OPENINGS, PF_FRI_COMBINE,
              PF_FRI_LAYERS, PF_POW, PF_GATHER, PF_COUNT };
const char* const PROF_NAMES[PF_COUNT] = {"intt_values_to_coeffs", "lde_batch", "merkle_leaves_colmajor", "merkle_levels", "z_partial_products",
                                          "quotient_eval", "quotient_coset_intt", "openings", "fri_compose_divide", "fri_layers", "pow_grind", "query_gather"};
struct ProfAcc { uint64_t launches = 0; double ms = 0, bytes = 0; };
std::atomic<int> g_prof_on(0); std::mutex g_prof_mu; ProfAcc g_prof[PF_COUNT];
static constexpr int PROF_EVENTS = 128;

struct Workspace {
  hipEvent_t pev[PROF_EVENTS]; int pev_made = 0, pev_used = 0; int pev_id[PROF_EVENTS / 2]; double pev_bytes[PROF_EVENTS / 2];
  DevArena arena; hipStream_t stream = nullptr; u64* pinned = nullptr; size_t pinned_elems = 0;
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_gates = nullptr, ev_perm = nullptr;   // off-critical-path quotient parts
  hipEvent_t ev_in = nullptr, ev_canon = nullptr;                                                      // off-critical-path input check
  u64 *q_gate_terms = nullptr, *q_perm_terms = nullptr;
  u64* pinned_dev = nullptr;   // device-side alias of the mapped pinned buffer; last 8 words: [flag, ...]
  u64* pub_counter = nullptr;  // device counter of multi-block publishes: the last block to arrive resets it (k_publish)
  u64 pub_seq = 0;
  bool shared_gpu = false;    // opt-in (QPGPU_SHARED_MODE=1): kernel choices that assume other proofs fill the chip
  bool pow_armed = false;     // pow_result[0] holds ~0 and pow_result[1] (the non-canonical-input flag) 0: re-armed by the last proof's gather kernel
  bool batch_mode = false;    // set for workers of qpgpu_prove_batch when several proofs share the chip: fused quotient kernel
  bool cores_to_spare = true; // the process may use at least two host cores per batch worker: workers keep busy-polling their hand-offs
  u64 *wires_vals, *wires_coeffs, *wires_lde, *wires_dig, *zpp_scratch, *zs_vals, *zs_coeffs, *zs_lde, *zs_dig, *q_vals, *q_lde, *q_dig;
  u64 *alpha_pows, *zp0, *zp1, *zi0, *zi1, *openings, *open_partial, *comp0, *comp1, *compose_partial, *div_scratch, *pow_result;
  u64* fri_coeffs[MAX_FRI_LAYERS + 1]; u64* fri_vals[MAX_FRI_LAYERS]; u64* fri_dig[MAX_FRI_LAYERS];
  uint8_t* query_bytes;
};

} // namespace

struct qpgpu_circuit {
  qpgpu_ctx* ctx; int device;
  qpgpu_circuit_desc d; std::vector<qpgpu_gate> gates; std::vector<uint32_t> sel_idx, groups; std::vector<u64> k_is_h;
  CircuitDev cd; NttPlan pl;
  int log_n, log_N, rate, S, W, Zp, Q, salt, nch, cap_h, nlayers; size_t n, N;
  int arity[MAX_FRI_LAYERS]; int fri_log_n[MAX_FRI_LAYERS + 1];   // log2 of coefficient count at each layer
  u64 *d_rt = nullptr, *d_lde_scale = nullptr, *d_qpost = nullptr, *d_kis = nullptr, *d_l0 = nullptr;
  u64* d_fri_scale[MAX_FRI_LAYERS + 1] = {nullptr};
  u64 *d_cs_vals = nullptr, *d_cs_coeffs = nullptr, *d_cs_lde = nullptr, *d_cs_dig = nullptr;
  std::vector<u64> cs_cap;
  bool split_quotient = false, leaf_gates_only = true, quotient_ls = false;
  Layout lay; size_t ws_bytes = 0;
  std::mutex mu; std::vector<Workspace*> free_ws;
};

struct ProfScope {   // records an event pair around the launches made inside its lifetime
  Workspace* w; hipStream_t s; bool on;
  ProfScope(Workspace* w_, hipStream_t s_, int id, double bytes) : w(w_), s(s_), on(false) {
    const int mode = g_prof_on.load(std::memory_order_relaxed);       // 0 off, 1 leaf-hash group only (cheap enough for timed runs), 2 all groups
    if (!mode || (mode == 1 && id != PF_LEAVES) || w->pev_used + 2 > PROF_EVENTS) return;
    while (w->pev_made < w->pev_used + 2) { if (hipEventCreate(&w->pev[w->pev_made]) != hipSuccess) return; w->pev_made++; }
    on = true; w->pev_id[w->pev_used / 2] = id; w->pev_bytes[w->pev_used / 2] = bytes;
    (void)hipEventRecord(w->pev[w->pev_used], s);
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(w->pev[w->pev_used + 1], s); w->pev_used += 2; } }
};
static void prof_flush(Workspace* w) {   // call after the stream has been synchronised
  if (!w->pev_used) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < w->pev_used; i += 2) {
    float ms = 0; if (hipEventElapsedTime(&ms, w->pev[i], w->pev[i + 1]) != hipSuccess) continue;
    ProfAcc& a = g_prof[w->pev_id[i / 2]]; a.launches++; a.ms += ms; a.bytes += w->pev_bytes[i / 2];
  }
  w->pev_used = 0;
}


// ------------------------------------------------------------------------------------------------ workspace
static size_t plan_workspace(qpgpu_circuit* c, Workspace* w) {
  DevArena& a = w->arena; a.used = 0;
  const size_t n = c->n, N = c->N; const int s = c->salt;
  w->wires_vals = a.take((size_t)c->W * n); w->wires_coeffs = a.take((size_t)c->W * n);
  w->wires_lde = a.take((size_t)(c->W + s) * N); w->wires_dig = a.take(merkle_total_digests(N, c->cap_h) * 4);
  w->zpp_scratch = a.take(zpp_scratch_elems(c->nch, n, c->d.num_partial_products));
  w->zs_vals = a.take((size_t)c->Zp * n); w->zs_coeffs = a.take((size_t)c->Zp * n);
  w->zs_lde = a.take((size_t)(c->Zp + s) * N); w->zs_dig = a.take(merkle_total_digests(N, c->cap_h) * 4);
  w->q_vals = a.take((size_t)c->nch * N); w->q_lde = a.take((size_t)(c->Q + s) * N); w->q_dig = a.take(merkle_total_digests(N, c->cap_h) * 4);
  w->alpha_pows = a.take((size_t)c->nch * 256);
  if (c->split_quotient) { w->q_gate_terms = a.take((size_t)c->d.num_gate_constraints * N); w->q_perm_terms = a.take((size_t)c->nch * (2 + c->d.num_partial_products) * N); }
  w->zp0 = a.take(2 * (n + 1)); w->zp1 = a.take(2 * (n + 1)); w->zi0 = a.take(2 * (n + 1)); w->zi1 = a.take(2 * (n + 1));
  w->openings = a.take(2 * c->lay.n_openings);
  w->open_partial = a.take(2 * c->lay.n_openings * open_chunks(c->log_n));
  w->compose_partial = a.take((size_t)FRI_COMPOSE_GROUPS * 2 * n);
  w->comp0 = a.take(2 * n); w->comp1 = a.take(2 * n); w->div_scratch = a.take(4 * n + 4 * (n / 2048 + 2)); w->pow_result = a.take(4); w->pub_counter = a.take(4);
  for (int l = 0; l <= c->nlayers; l++) w->fri_coeffs[l] = a.take((size_t)2 << c->fri_log_n[l]);
  for (int l = 0; l < c->nlayers; l++) {
    size_t Nl = (size_t)1 << (c->fri_log_n[l] + c->rate);
    w->fri_vals[l] = a.take(2 * Nl);
    w->fri_dig[l] = a.take(merkle_total_digests(Nl >> c->arity[l], c->cap_h) * 4);
  }
  w->query_bytes = (uint8_t*)a.take((c->lay.bytes_per_query * c->d.fri_num_queries + 7) / 8 + 1);
  return a.used;
}

static void ws_destroy(Workspace* w);
static int ws_acquire(qpgpu_circuit* c, Workspace** out) {
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->free_ws.empty()) { *out = c->free_ws.back(); c->free_ws.pop_back(); return 0; }
  }
  Workspace* w = new Workspace();
  HIP_TRY(hipSetDevice(c->device));
  hipError_t e = hipMalloc((void**)&w->arena.base, c->ws_bytes);
  if (e != hipSuccess) { delete w; return set_err(QPGPU_ERR_OOM, std::string("workspace hipMalloc: ") + hipGetErrorString(e)); }
  w->arena.size = c->ws_bytes;
  plan_workspace(c, w);
  int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // numerically lower = higher priority
  e = hipStreamCreateWithPriority(&w->stream, hipStreamNonBlocking, prio_hi);
  if (e != hipSuccess) { (void)hipFree(w->arena.base); delete w; return set_err(QPGPU_ERR_HIP, "hipStreamCreate failed"); }
  {
    // the side stream carries work that runs beside the main stream (the quotient's second half; the opt-in split quotient)
    if (hipStreamCreateWithPriority(&w->side, hipStreamNonBlocking, prio_lo) != hipSuccess || hipEventCreateWithFlags(&w->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&w->ev_gates, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&w->ev_perm, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&w->ev_canon, hipEventDisableTiming) != hipSuccess) {
      ws_destroy(w); return set_err(QPGPU_ERR_HIP, "side stream / event creation failed");
    }
  }
  // the largest thing fetch_to_host ever brings over: a cap, the opening set, the final polynomial, the serialized query rounds (+1 extra word)
  { size_t m = (size_t(4) << c->cap_h) + 1;
    m = std::max(m, 2 * c->lay.n_openings); m = std::max(m, 2 * c->lay.final_poly_len);
    m = std::max(m, (c->lay.bytes_per_query * c->d.fri_num_queries + 7) / 8); w->pinned_elems = m + 64; }
  e = hipHostMalloc((void**)&w->pinned, (w->pinned_elems + 8) * 8, hipHostMallocMapped);
  if (e != hipSuccess) { (void)hipStreamDestroy(w->stream); (void)hipFree(w->arena.base); delete w; return set_err(QPGPU_ERR_OOM, "hipHostMalloc failed"); }
  if (hipHostGetDevicePointer((void**)&w->pinned_dev, w->pinned, 0) != hipSuccess) w->pinned_dev = nullptr;
  w->pinned[w->pinned_elems] = 0;
  (void)hipMemsetAsync(w->pub_counter, 0, 8, w->stream);
  *out = w; return 0;
}
static void ws_release(qpgpu_circuit* c, Workspace* w) { std::lock_guard<std::mutex> lk(c->mu); c->free_ws.push_back(w); }
static void ws_destroy(Workspace* w) {
  if (!w) return;
  for (int i = 0; i < w->pev_made; i++) (void)hipEventDestroy(w->pev[i]);
  if (w->pinned) (void)hipHostFree(w->pinned);
  if (w->stream) (void)hipStreamDestroy(w->stream);
  if (w->side) (void)hipStreamDestroy(w->side);
  for (hipEvent_t e : {w->ev_fork, w->ev_gates, w->ev_perm, w->ev_in, w->ev_canon}) if (e) (void)hipEventDestroy(e);
  if (w->arena.base) (void)hipFree(w->arena.base);
  delete w;
}

using qp::usable_cpus;          // csrc/hostcpus.h: affinity mask, capped by the cgroup CPU quota, divided by the ranks of this node
static std::atomic<uint64_t> g_handoff_fallbacks{0};
static std::atomic<int> g_proofs_in_flight{0};          // qpgpu_prove calls running right now in this process
extern "C" int qpgpu_usable_host_cpus(void) { return qp::usable_cpus_uncached(); }
extern "C" uint64_t qpgpu_handoff_fallbacks(void) { return g_handoff_fallbacks.load(std::memory_order_relaxed); }

// Brings `words` u64 from device memory to w->pinned and waits for them. Fast path: publish kernel + spin on the mapped flag;
// fallback (no mapped pointer, or the flag does not arrive): copy + stream synchronise, which also surfaces kernel faults.
// `extra` (optional) is one more device word, delivered at w->pinned[words].
// The wait is adaptive: busy-poll the flag for the spin budget (Options::handoff_spin_us; a hand-off usually lands within it), then nap
// 20 us between polls so that a stage that runs for a millisecond (the wires leaf hash) does not pin a host core, which matters
// with 4 worker streams x 8 ranks on one host. Every hand-off also surfaces launch errors of the stage it ends (hipGetLastError).
static int fetch_to_host(Workspace* w, hipStream_t s, const u64* src, size_t words, u64* extra = nullptr, bool clear_extra = false) {
  if (words + (extra ? 1 : 0) > w->pinned_elems) return set_err(QPGPU_ERR_BUFFER, "internal: hand-off larger than the pinned buffer");
  { hipError_t le = hipGetLastError(); if (le != hipSuccess) return set_err(QPGPU_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le)); }
  if (w->pinned_dev && !opts().no_publish) {
    const u64 seq = ++w->pub_seq;
    volatile u64* flag = w->pinned + w->pinned_elems;
    publish_to_host(src, w->pinned_dev, words, w->pinned_dev + w->pinned_elems, w->pub_counter, seq, s, extra, clear_extra);
    const int budget_us = opts().handoff_spin_us >= 0 ? opts().handoff_spin_us : (w->batch_mode && !w->cores_to_spare ? 50 : 2000);
    const auto t0 = std::chrono::steady_clock::now();
    bool napping = false; uint64_t naps = 0;
    for (uint64_t spins = 0;; spins++) {
      if (*flag == seq) { std::atomic_thread_fence(std::memory_order_acquire); return 0; }
      if (!napping) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0x3F) == 0x3F && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget_us)) napping = true;
      } else {
        struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr);
        if ((++naps & 0x3F) == 0) {                  // every ~few ms: has the stream died?
          hipError_t q = hipStreamQuery(s);
          if (q != hipSuccess && q != hipErrorNotReady) return set_err(QPGPU_ERR_HIP, std::string("stream error while waiting: ") + hipGetErrorString(q));
          if (q == hipSuccess && *flag != seq && naps > 4096) { g_handoff_fallbacks.fetch_add(1, std::memory_order_relaxed); break; }   // kernel finished but the flag never arrived: fall back (counted)
        }
      }
    }
  }
  HIP_TRY(hipMemcpyAsync(w->pinned, src, words * 8, hipMemcpyDeviceToHost, s));
  if (extra) HIP_TRY(hipMemcpyAsync(w->pinned + words, extra, 8, hipMemcpyDeviceToHost, s));
  if (extra && clear_extra) HIP_TRY(hipMemsetAsync(extra, 0, 8, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

// Commit to `cols` coefficient columns already sitting in `coeffs` ([cols][n]): LDE -> salt -> leaves -> levels; the cap is
// is fetched by the caller (cap_ptr + fetch_to_host).
// `fork` (optional) forks work onto the side stream: by default after the leaf hash, beside the latency-bound tree levels
// (which run at raised wave priority); QPGPU_SPLIT_FORK_LATE=0 forks it before the leaf hash instead (the leaf kernel at
// N = 2^17 only offers 2 waves/SIMD and leaves issue slots unused, but sharing slows it by more than the overlap gains).
struct SaltSrc { const u64* host = nullptr; bool have_key = false; SaltKey key{}; u64 stream = 0; };   // explicit columns, or ChaCha8(key, stream)
template <class F>
static void commit_coeffs(qpgpu_circuit* c, Workspace* w, const u64* coeffs, int cols, u64* lde, u64* dig, int oracle,
                          const SaltSrc& salt, hipStream_t s, F fork) {
  const size_t N = c->N;
  { ProfScope ps(w, s, PF_LDE, 8.0 * cols * (c->n + N));
    lde_batch(coeffs, c->n, lde, N, cols, c->log_n, c->rate, c->d_lde_scale, c->pl, s, w->shared_gpu); }
  int width = cols;
  if (c->salt && oracle > 0) {
    if (salt.host) {
      (void)hipMemcpyAsync(lde + (size_t)cols * N, salt.host + (size_t)(oracle - 1) * 4 * N, 4 * N * 8, hipMemcpyHostToDevice, s);
      check_canonical(lde + (size_t)cols * N, 4 * N, w->pow_result + 1, s);      // explicit salts are caller data: must be canonical
    } else salt_fill(lde + (size_t)cols * N, N, salt.key, salt.stream, oracle - 1, s);
    width += 4;
  }
  // side-stream work is forked after the leaf hash, beside the latency-bound tree levels (3.827 vs 3.852 ms on config B; the level
  // kernels run at raised wave priority); QPGPU_SPLIT_FORK_LATE=0 forks it before the leaf hash instead
  const int fork_late = opts().split_fork_late;
  if (!fork_late) fork();
  int done = 0;
  { ProfScope ps(w, s, PF_LEAVES, (8.0 * width + 32.0) * N);
    done = merkle_leaves_colmajor(lde, N, width, N, c->cap_h, dig, s, w->batch_mode); }
  if (fork_late) fork();
  { ProfScope ps(w, s, PF_LEVELS, 32.0 * 3 * N);
    merkle_levels(dig, N, c->cap_h, s, done, w->batch_mode); }
}
static void commit_coeffs(qpgpu_circuit* c, Workspace* w, const u64* coeffs, int cols, u64* lde, u64* dig, int oracle,
                          const SaltSrc& salt, hipStream_t s) {
  commit_coeffs(c, w, coeffs, cols, lde, dig, oracle, salt, s, [] {});
}
static const u64* cap_ptr(qpgpu_circuit* c, const u64* dig, size_t leaves) {
  return dig + (merkle_total_digests(leaves, c->cap_h) - (size_t(1) << c->cap_h)) * 4;
}
// values [cols][n] (natural order) -> coefficients [cols][n]; `tmp` needs cols*n elements and is clobbered.
static void values_to_coeffs(qpgpu_circuit* c, Workspace* w, const u64* vals, int cols, u64* tmp, u64* coeffs, hipStream_t s) {
  ProfScope ps(w, s, PF_INTT, 8.0 * cols * c->n * 2);
  if (ntt16_supported(c->log_n)) { ntt16_run(vals, c->n, coeffs, c->n, cols, c->log_n, 0, true, true, nullptr, gl_inv((u64)c->n), c->pl, s); return; }
  if (!w->shared_gpu && lde_strided_supported(c->log_n) && cols >= 16) {        // same kernel choice rule as lde_batch
    lde_strided_run(vals, c->n, tmp, c->n, cols, c->log_n, 0, nullptr, c->pl, s, true);
    bitrev_permute(tmp, c->n, coeffs, c->n, cols, c->log_n, gl_inv((u64)c->n), s);
    return;
  }
  ntt_copy(vals, c->n, tmp, c->n, cols, c->log_n, false, true, c->pl, s);             // inverse DIF: bit-reversed, unscaled
  bitrev_permute(tmp, c->n, coeffs, c->n, cols, c->log_n, gl_inv((u64)c->n), s);
}

// No C++ exception may cross the C ABI (include/qpgpu.h): host allocation failures and the like become status codes.
template <class F> static int abi_guard(F&& f) {
  try { return f(); }
  catch (const std::bad_alloc&) { return set_err(QPGPU_ERR_OOM, "host memory allocation failed"); }
  catch (const std::exception& e) { return set_err(QPGPU_ERR_HIP, std::string("internal error: ") + e.what()); }
  catch (...) { return set_err(QPGPU_ERR_HIP, "internal error"); }
}

// ------------------------------------------------------------------------------------------------ API
extern "C" {

const char* qpgpu_last_error(void) { return g_last_error.c_str(); }
int qpgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int qpgpu_ctx_create(int device, qpgpu_ctx** out) { return abi_guard([&]() -> int {
  if (!out) return set_err(QPGPU_ERR_ARG, "null out");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return set_err(QPGPU_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return set_err(QPGPU_ERR_ARG, "device index out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return set_err(QPGPU_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  HIP_TRY(hipSetDevice(device));
  options_refresh_from_env();        // the one read of the QPGPU_* knobs (options.h)
  *out = new qpgpu_ctx{device};
  return 0;
}); }
void qpgpu_ctx_destroy(qpgpu_ctx* ctx) { delete ctx; }

void qpgpu_circuit_destroy(qpgpu_circuit* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (Workspace* w : c->free_ws) ws_destroy(w);
  for (u64* p : {c->d_rt, c->d_lde_scale, c->d_qpost, c->d_kis, c->d_l0, c->d_cs_vals, c->d_cs_coeffs, c->d_cs_lde, c->d_cs_dig}) if (p) (void)hipFree(p);
  for (int l = 0; l <= MAX_FRI_LAYERS; l++) if (c->d_fri_scale[l]) (void)hipFree(c->d_fri_scale[l]);
  delete c;
}

int qpgpu_circuit_create(qpgpu_ctx* ctx, const qpgpu_circuit_desc* desc, qpgpu_circuit** out) { return abi_guard([&]() -> int {
  if (!ctx || !desc || !out) return set_err(QPGPU_ERR_ARG, "null argument");
  const qpgpu_circuit_desc& d = *desc;
  if (!d.gates || !d.selector_indices || !d.selector_group_ranges || !d.k_is || !d.constants_sigmas) return set_err(QPGPU_ERR_ARG, "null array in the circuit descriptor");
  if (d.degree_bits < 2 || d.degree_bits + d.fri_rate_bits > 28) return set_err(QPGPU_ERR_ARG, "degree_bits out of range");
  if (d.num_wires < 1 || d.num_wires > 4096 || d.num_routed < 1 || d.num_routed > d.num_wires) return set_err(QPGPU_ERR_ARG, "need 1 <= num_routed <= num_wires <= 4096");
  if (d.fri_cap_height > d.degree_bits + d.fri_rate_bits || d.fri_cap_height > 20) return set_err(QPGPU_ERR_ARG, "cap height exceeds tree");
  if (d.fri_pow_bits > 40) return set_err(QPGPU_ERR_ARG, "proof_of_work_bits out of range");
  if (d.num_public_inputs > (1u << 20)) return set_err(QPGPU_ERR_ARG, "too many public inputs");
  if (d.num_challenges < 1 || d.num_challenges > 2) return set_err(QPGPU_ERR_ARG, "num_challenges must be 1 or 2");
  if (d.num_gates > MAX_GATES || d.num_gates == 0) return set_err(QPGPU_ERR_ARG, "bad gate count");
  if (d.fri_num_layers > MAX_FRI_LAYERS) return set_err(QPGPU_ERR_ARG, "too many FRI layers");
  if (d.fri_num_queries > MAX_QUERIES) return set_err(QPGPU_ERR_ARG, "too many FRI queries");
  if (d.fri_rate_bits < 1 || d.fri_rate_bits > 3 || (1u << d.fri_rate_bits) != d.quotient_degree_factor)
    return set_err(QPGPU_ERR_ARG, "this backend needs quotient_degree_factor == 2^rate_bits <= 8");
  if (d.num_partial_products + 1 > 16 || (d.num_partial_products + 1) * d.quotient_degree_factor < d.num_routed) return set_err(QPGPU_ERR_ARG, "bad partial product count");
  if (d.num_selector_groups < 1 || d.num_constants < d.num_selector_groups + 2) return set_err(QPGPU_ERR_ARG, "bad constants/selectors");
  if (d.num_challenges * (1 + d.num_partial_products + d.num_gate_constraints) > 2 * 256 || d.num_gate_constraints + 40 > 256) return set_err(QPGPU_ERR_ARG, "too many constraint terms");
  { uint32_t sum = 0; for (uint32_t i = 0; i < d.fri_num_layers; i++) { if (d.fri_arity_bits[i] < 1 || d.fri_arity_bits[i] > 4) return set_err(QPGPU_ERR_ARG, "FRI arity bits must be 1..4"); sum += d.fri_arity_bits[i]; }
    if (sum > d.degree_bits) return set_err(QPGPU_ERR_ARG, "FRI reductions exceed degree"); }
  for (uint32_t g = 0; g < d.num_selector_groups; g++) {
    const uint32_t lo = d.selector_group_ranges[2 * g], hi = d.selector_group_ranges[2 * g + 1];
    if (lo >= hi || hi > d.num_gates) return set_err(QPGPU_ERR_ARG, "selector group range out of order or beyond the gate list");
  }
  for (uint32_t i = 0; i < d.num_gates; i++) {
    uint32_t t = d.gates[i].tag;
    const qpgpu_gate& gt = d.gates[i];
    const uint64_t W = d.num_wires, Rt = d.num_routed;
    uint64_t ncons = 0;          // constraints this gate writes: slots [0, ncons) of the alpha-power table must exist
    switch (t) {
      case QPGPU_GATE_NOOP: break;
      case QPGPU_GATE_CONSTANT: ncons = gt.param; break;
      case QPGPU_GATE_PUBLIC_INPUT: ncons = 4; if (W < 4) return set_err(QPGPU_ERR_ARG, "PublicInput gate needs 4 wires"); break;
      case QPGPU_GATE_BASESUM2: ncons = gt.param + 1; break;
      case QPGPU_GATE_ARITHMETIC: ncons = gt.param; break;
      case QPGPU_GATE_POSEIDON: ncons = 123; break;
      case QPGPU_GATE_ARITHMETIC_EXTENSION: if (8 * gt.param > Rt) return set_err(QPGPU_ERR_ARG, "ArithmeticExtension ops exceed routed wires"); ncons = 2 * gt.param; break;
      case QPGPU_GATE_MUL_EXTENSION: if (6 * gt.param > Rt) return set_err(QPGPU_ERR_ARG, "MulExtension ops exceed routed wires"); ncons = 2 * gt.param; break;
      case QPGPU_GATE_REDUCING: if (gt.param < 1 || 6 + 3 * gt.param - 2 > W) return set_err(QPGPU_ERR_ARG, "Reducing coefficients exceed wires"); ncons = 2 * gt.param; break;
      case QPGPU_GATE_REDUCING_EXTENSION: if (gt.param < 1 || 6 + 4 * gt.param - 2 > W) return set_err(QPGPU_ERR_ARG, "ReducingExtension coefficients exceed wires"); ncons = 2 * gt.param; break;
      case QPGPU_GATE_RANDOM_ACCESS: {
        if (gt.param < 1 || gt.param > 4) return set_err(QPGPU_ERR_ARG, "RandomAccess bits must be 1..4");
        uint64_t vec = 1ull << gt.param, routed = (2 + vec) * gt.param2 + gt.param3;
        if (routed > Rt || routed + gt.param * gt.param2 > W || gt.param3 + d.num_selector_groups > d.num_constants) return set_err(QPGPU_ERR_ARG, "RandomAccess copies exceed wires/constants");
        ncons = (gt.param + 2) * gt.param2 + gt.param3;
        break;
      }
      case QPGPU_GATE_POSEIDON_MDS: if (W < 48) return set_err(QPGPU_ERR_ARG, "PoseidonMds gate needs 48 wires"); ncons = 24; break;
      case QPGPU_GATE_EXPONENTIATION: if (gt.param < 1 || 2 + 2 * gt.param > W) return set_err(QPGPU_ERR_ARG, "Exponentiation bits exceed wires"); ncons = gt.param + 1; break;
      case QPGPU_GATE_COSET_INTERPOLATION: {
        if (gt.param < 1 || gt.param > 4 || gt.param2 < 2 || !gt.table || gt.table_len != (1u << gt.param)) return set_err(QPGPU_ERR_ARG, "CosetInterpolation needs subgroup_bits 1..4, degree >= 2 and 2^bits weights");
        uint64_t np = 1ull << gt.param, ni = (np - 2) / (gt.param2 - 1);
        if (1 + 2 * np + 4 + 4 * ni + 2 > W) return set_err(QPGPU_ERR_ARG, "CosetInterpolation exceeds wires");
        for (uint32_t k = 0; k < gt.table_len; k++) if (gt.table[k] >= GL_P) return set_err(QPGPU_ERR_ARG, "non-canonical barycentric weight");
        for (uint32_t j = 0; j < i; j++) if (d.gates[j].tag == QPGPU_GATE_COSET_INTERPOLATION) return set_err(QPGPU_ERR_ARG, "one CosetInterpolation gate per circuit");
        ncons = 4 + 4 * ni;
        break;
      }
      default: return set_err(QPGPU_ERR_ARG, "gate tag " + std::to_string(t) + " is not supported by this backend (lookup gates)");
    }
    if (ncons > d.num_gate_constraints) return set_err(QPGPU_ERR_ARG, "num_gate_constraints (" + std::to_string(d.num_gate_constraints) + ") is smaller than gate " + std::to_string(i) + "'s " + std::to_string(ncons) + " constraints");
    if (d.selector_indices[i] >= d.num_selector_groups) return set_err(QPGPU_ERR_ARG, "selector index out of range");
    if (t == QPGPU_GATE_POSEIDON && d.num_wires < 135) return set_err(QPGPU_ERR_ARG, "Poseidon gate needs 135 wires");
    if (t == QPGPU_GATE_BASESUM2 && d.gates[i].param + 1 > d.num_wires) return set_err(QPGPU_ERR_ARG, "BaseSum limbs exceed wires");
    if (t == QPGPU_GATE_ARITHMETIC && 4 * d.gates[i].param > d.num_wires) return set_err(QPGPU_ERR_ARG, "Arithmetic ops exceed wires");
    if (t == QPGPU_GATE_CONSTANT && (d.gates[i].param + d.num_selector_groups > d.num_constants || d.gates[i].param > d.num_wires)) return set_err(QPGPU_ERR_ARG, "Constant gate exceeds constants / wires");
  }
  HIP_TRY(hipSetDevice(ctx->device));
  struct CircuitGuard { qpgpu_circuit* p; ~CircuitGuard() { if (p) qpgpu_circuit_destroy(p); } } guard{new qpgpu_circuit()};   // freed (device memory too) on every early exit, exceptions included
  qpgpu_circuit* c = guard.p;
  c->ctx = ctx; c->device = ctx->device; c->d = d;
  c->gates.assign(d.gates, d.gates + d.num_gates); for (auto& g : c->gates) g.table = nullptr;   /* weights live in cd.coset_w */ c->sel_idx.assign(d.selector_indices, d.selector_indices + d.num_gates);
  c->groups.assign(d.selector_group_ranges, d.selector_group_ranges + 2 * d.num_selector_groups);
  c->k_is_h.assign(d.k_is, d.k_is + d.num_routed);
  c->d.gates = c->gates.data(); c->d.selector_indices = c->sel_idx.data(); c->d.selector_group_ranges = c->groups.data();
  c->d.k_is = c->k_is_h.data(); c->d.constants_sigmas = nullptr;
  c->log_n = d.degree_bits; c->rate = d.fri_rate_bits; c->log_N = c->log_n + c->rate; c->n = size_t(1) << c->log_n; c->N = size_t(1) << c->log_N;
  c->S = d.num_constants + d.num_routed; c->W = d.num_wires; c->nch = d.num_challenges; c->Zp = c->nch * (1 + d.num_partial_products);
  c->Q = c->nch * d.quotient_degree_factor; c->salt = d.zero_knowledge ? 4 : 0; c->cap_h = d.fri_cap_height; c->nlayers = d.fri_num_layers;
  if ((size_t(1) << c->cap_h) > c->N) return set_err(QPGPU_ERR_ARG, "cap height exceeds tree");
  c->fri_log_n[0] = c->log_n;
  for (int l = 0; l < c->nlayers; l++) {
    c->arity[l] = d.fri_arity_bits[l]; c->fri_log_n[l + 1] = c->fri_log_n[l] - c->arity[l];
    if (c->fri_log_n[l] + c->rate - c->arity[l] < c->cap_h) return set_err(QPGPU_ERR_ARG, "FRI layer tree smaller than its cap");
  }
  // ---- layout
  compute_layout(d, c->lay);
  const size_t capn = size_t(1) << c->cap_h;
  // ---- device constants
  hipStream_t s = 0;
#define CC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::string m = std::string(#x) + ": " + hipGetErrorString(e_); return set_err(e_ == hipErrorOutOfMemory ? QPGPU_ERR_OOM : QPGPU_ERR_HIP, m); } } while (0)
  const size_t n = c->n, N = c->N;
  CC_TRY(hipMalloc((void**)&c->d_rt, (N / 2 + 1) * 8));
  build_root_table(c->d_rt, c->log_N, s);
  c->pl = NttPlan{c->d_rt, c->log_N};
  CC_TRY(hipMalloc((void**)&c->d_lde_scale, N * 8));
  build_lde_scale(c->d_lde_scale, c->log_n, c->rate, GL_GEN, c->pl, s);
  c->d_fri_scale[0] = nullptr;
  { u64 shift = GL_GEN;
    for (int l = 1; l < c->nlayers; l++) {
      for (int i = 0; i < c->arity[l - 1]; i++) shift = gl_sqr(shift);
      size_t Nl = size_t(1) << (c->fri_log_n[l] + c->rate);
      CC_TRY(hipMalloc((void**)&c->d_fri_scale[l], Nl * 8));
      build_lde_scale(c->d_fri_scale[l], c->fri_log_n[l], c->rate, shift, c->pl, s);
    } }
  CC_TRY(hipMalloc((void**)&c->d_qpost, N * 8));
  build_powers(c->d_qpost, N, gl_inv(GL_GEN), gl_inv((u64)N), s);
  CC_TRY(hipMalloc((void**)&c->d_kis, d.num_routed * 8));
  CC_TRY(hipMemcpy(c->d_kis, d.k_is, d.num_routed * 8, hipMemcpyHostToDevice));
  // CircuitDev
  memset(&c->cd, 0, sizeof c->cd);
  c->cd.log_n = c->log_n; c->cd.rate_bits = c->rate; c->cd.num_wires = c->W; c->cd.num_routed = d.num_routed; c->cd.num_constants = d.num_constants;
  c->cd.num_challenges = c->nch; c->cd.qdf = d.quotient_degree_factor; c->cd.num_pp = d.num_partial_products; c->cd.num_gates = d.num_gates;
  c->cd.num_selectors = d.num_selector_groups; c->cd.num_gate_constraints = d.num_gate_constraints; c->cd.k_is = c->d_kis;
  for (uint32_t i = 0; i < d.num_gates; i++) {
    uint32_t si = d.selector_indices[i];
    c->cd.gates[i] = GateDev{d.gates[i].tag, si, d.selector_group_ranges[2 * si], d.selector_group_ranges[2 * si + 1], d.gates[i].param, d.gates[i].param2, d.gates[i].param3};
    if (d.gates[i].tag == QPGPU_GATE_COSET_INTERPOLATION) {
      const u64 wg = gl_root((int)d.gates[i].param); u64 x = 1;
      for (uint32_t k = 0; k < d.gates[i].table_len; k++) { c->cd.coset_w[k] = d.gates[i].table[k]; c->cd.coset_dom[k] = x; x = gl_mul(x, wg); }
    }
    const uint32_t t = d.gates[i].tag;
    if (!(t == QPGPU_GATE_NOOP || t == QPGPU_GATE_CONSTANT || t == QPGPU_GATE_PUBLIC_INPUT || t == QPGPU_GATE_BASESUM2 || t == QPGPU_GATE_ARITHMETIC || t == QPGPU_GATE_POSEIDON))
      c->leaf_gates_only = false;
  }
  // L_0 on the LDE coset (circuit constant; filled once the root table exists, below)
  CC_TRY(hipMalloc((void**)&c->d_l0, N * 8));
  l0_table(c->cd, c->d_l0, c->pl, s);
  c->cd.l0_lde = opts().l0_table ? c->d_l0 : nullptr;      // 0: recompute L_0 per point (A/B switch)
  // preprocessed oracle: constants/sigmas values -> coeffs -> LDE -> Merkle (kept resident)
  CC_TRY(hipMalloc((void**)&c->d_cs_vals, (size_t)c->S * n * 8));
  CC_TRY(hipMalloc((void**)&c->d_cs_coeffs, (size_t)c->S * n * 8));
  CC_TRY(hipMalloc((void**)&c->d_cs_lde, (size_t)c->S * N * 8));
  CC_TRY(hipMalloc((void**)&c->d_cs_dig, merkle_total_digests(N, c->cap_h) * 32));
  CC_TRY(hipMemcpy(c->d_cs_vals, d.constants_sigmas, (size_t)c->S * n * 8, hipMemcpyHostToDevice));
  { struct WsGuard { Workspace* p; ~WsGuard() { if (p) { p->pev_used = 0; ws_destroy(p); } } } nows{new Workspace()};   // no profiling events for the one-off preprocessing
    values_to_coeffs(c, nows.p, c->d_cs_vals, c->S, c->d_cs_lde, c->d_cs_coeffs, s);
    c->cs_cap.resize(capn * 4);
    commit_coeffs(c, nows.p, c->d_cs_coeffs, c->S, c->d_cs_lde, c->d_cs_dig, 0, SaltSrc(), s);
    CC_TRY(hipStreamSynchronize(s)); }
  CC_TRY(hipGetLastError());
  { size_t cap_off = merkle_total_digests(N, c->cap_h) - capn; CC_TRY(hipMemcpy(c->cs_cap.data(), c->d_cs_dig + cap_off * 4, capn * 32, hipMemcpyDeviceToHost)); }
  c->split_quotient = quotient_split_supported(c->cd) && c->leaf_gates_only;
  c->quotient_ls = quotient_leaf_set(c->cd);
  { Workspace tmp; tmp.arena.base = nullptr; c->ws_bytes = plan_workspace(c, &tmp) + 4096; }
  guard.p = nullptr;
  *out = c; return 0;
}); }

// ---- CommonCircuitData / VerifierOnlyCircuitData parsers (byte layouts: SURVEY.md App. B.1-B.2)
namespace {
struct ByteReader {
  const uint8_t* b; size_t len, pos = 0; bool ok = true;
  uint64_t u64_() { if (pos + 8 > len) { ok = false; return 0; } uint64_t v; memcpy(&v, b + pos, 8); pos += 8; return v; }
  uint32_t u32_() { if (pos + 4 > len) { ok = false; return 0; } uint32_t v; memcpy(&v, b + pos, 4); pos += 4; return v; }
  uint8_t u8_() { if (pos + 1 > len) { ok = false; return 0; } return b[pos++]; }
};
static bool read_fri_config(ByteReader& r, uint32_t out[4]) {
  out[0] = (uint32_t)r.u64_(); out[1] = (uint32_t)r.u64_(); out[2] = (uint32_t)r.u64_(); out[3] = r.u32_();
  uint8_t tag = r.u8_();
  if (tag == 0) { uint64_t k = r.u64_(); for (uint64_t i = 0; i < k && r.ok; i++) r.u64_(); }
  else if (tag == 1) { r.u64_(); r.u64_(); }
  else if (tag == 2) { r.u64_(); }
  else return false;
  return r.ok;
}
}
int qpgpu_parse_common(const uint8_t* bytes, size_t len, qpgpu_circuit_desc* d, qpgpu_gate* gates, size_t gates_cap,
                       uint32_t* selector_indices, uint32_t* selector_group_ranges, size_t groups_cap, uint64_t* k_is, size_t k_is_cap) { return abi_guard([&]() -> int {
  if (!bytes || !d || !gates || !selector_indices || !selector_group_ranges || !k_is) return set_err(QPGPU_ERR_ARG, "null argument");
  ByteReader r{bytes, len};
  memset(d, 0, sizeof *d);
  d->num_wires = (uint32_t)r.u64_(); d->num_routed = (uint32_t)r.u64_(); r.u64_() /*config.num_constants*/; r.u64_() /*security_bits*/;
  d->num_challenges = (uint32_t)r.u64_(); r.u64_() /*max_quotient_degree_factor*/;
  r.u8_() /*use_base_arithmetic_gate*/; d->zero_knowledge = r.u8_();
  uint32_t fc[4];
  if (!read_fri_config(r, fc) || !read_fri_config(r, fc)) return set_err(QPGPU_ERR_ARG, "malformed FRI config");
  d->fri_rate_bits = fc[0]; d->fri_cap_height = fc[1]; d->fri_num_queries = fc[2]; d->fri_pow_bits = fc[3];
  uint64_t na = r.u64_(); if (na > 8) return set_err(QPGPU_ERR_ARG, "too many FRI reduction layers");
  d->fri_num_layers = (uint32_t)na; for (uint64_t i = 0; i < na; i++) d->fri_arity_bits[i] = (uint32_t)r.u64_();
  d->degree_bits = (uint32_t)r.u64_(); uint8_t hiding = r.u8_();
  if ((hiding != 0) != (d->zero_knowledge != 0)) return set_err(QPGPU_ERR_ARG, "fri_params.hiding disagrees with config.zero_knowledge");
  uint64_t ns = r.u64_(); if (ns > gates_cap) return set_err(QPGPU_ERR_ARG, "selector_indices exceed capacity");
  for (uint64_t i = 0; i < ns; i++) selector_indices[i] = (uint32_t)r.u64_();
  uint64_t ng = r.u64_(); if (ng > groups_cap) return set_err(QPGPU_ERR_ARG, "selector groups exceed capacity");
  for (uint64_t i = 0; i < ng; i++) { selector_group_ranges[2 * i] = (uint32_t)r.u64_(); selector_group_ranges[2 * i + 1] = (uint32_t)r.u64_(); }
  d->num_selector_groups = (uint32_t)ng;
  d->quotient_degree_factor = (uint32_t)r.u64_(); d->num_gate_constraints = (uint32_t)r.u64_(); d->num_constants = (uint32_t)r.u64_();
  d->num_public_inputs = (uint32_t)r.u64_();
  uint64_t nk = r.u64_(); if (nk > k_is_cap || nk != d->num_routed) return set_err(QPGPU_ERR_ARG, "k_is count mismatch");
  for (uint64_t i = 0; i < nk; i++) { k_is[i] = r.u64_(); if (k_is[i] >= GL_P) return set_err(QPGPU_ERR_ARG, "non-canonical k_i"); }
  d->num_partial_products = (uint32_t)r.u64_();
  uint64_t nlp = r.u64_(), nls = r.u64_(), nluts = r.u64_();
  if (nlp || nls || nluts) return set_err(QPGPU_ERR_ARG, "lookup arguments are not supported by this backend");
  uint64_t tables_used = 0;
  uint64_t ngates = r.u64_(); if (ngates > gates_cap || ngates != ns) return set_err(QPGPU_ERR_ARG, "gate count mismatch");
  for (uint64_t i = 0; i < ngates && r.ok; i++) {
    memset(&gates[i], 0, sizeof gates[i]); gates[i].tag = r.u32_();
    switch (gates[i].tag) {
      case QPGPU_GATE_NOOP: case QPGPU_GATE_PUBLIC_INPUT: case QPGPU_GATE_POSEIDON: case QPGPU_GATE_POSEIDON_MDS: break;
      case QPGPU_GATE_CONSTANT: case QPGPU_GATE_BASESUM2: case QPGPU_GATE_ARITHMETIC: case QPGPU_GATE_ARITHMETIC_EXTENSION: case QPGPU_GATE_MUL_EXTENSION:
      case QPGPU_GATE_REDUCING: case QPGPU_GATE_REDUCING_EXTENSION: case QPGPU_GATE_EXPONENTIATION: gates[i].param = r.u64_(); break;
      // field order of the two multi-parameter gates is recalled, not pinned by any shipped artefact
      case QPGPU_GATE_RANDOM_ACCESS: gates[i].param = r.u64_(); gates[i].param2 = (uint32_t)r.u64_(); gates[i].param3 = (uint32_t)r.u64_(); break;
      case QPGPU_GATE_COSET_INTERPOLATION: {
        gates[i].param = r.u64_(); gates[i].param2 = (uint32_t)r.u64_(); uint64_t nw = r.u64_();
        if (nw > 16 || nk + tables_used + nw > k_is_cap) return set_err(QPGPU_ERR_ARG, "barycentric weights exceed the k_is buffer tail");
        gates[i].table = k_is + nk + tables_used; gates[i].table_len = (uint32_t)nw;
        for (uint64_t k = 0; k < nw; k++) k_is[nk + tables_used + k] = r.u64_();
        tables_used += nw; break;
      }
      default: return set_err(QPGPU_ERR_ARG, "gate tag " + std::to_string(gates[i].tag) + " is not supported (lookup gates)");
    }
  }
  d->num_gates = (uint32_t)ngates;
  if (!r.ok) return set_err(QPGPU_ERR_ARG, "common data truncated");
  if (r.pos != len) return set_err(QPGPU_ERR_ARG, "trailing bytes after common data");
  d->gates = gates; d->selector_indices = selector_indices; d->selector_group_ranges = selector_group_ranges; d->k_is = k_is;
  return 0;
}); }
int qpgpu_parse_verifier_only(const uint8_t* bytes, size_t len, uint64_t* cap_out, size_t cap_words, uint64_t digest_out[4]) { return abi_guard([&]() -> int {
  if (!bytes || !cap_out || !digest_out) return set_err(QPGPU_ERR_ARG, "null argument");
  ByteReader r{bytes, len};
  uint64_t ch = r.u64_(); if (!r.ok || ch > 20) return set_err(QPGPU_ERR_ARG, "bad cap height");
  size_t words = (size_t(4)) << ch; if (words > cap_words) return set_err(QPGPU_ERR_BUFFER, "cap buffer too small");
  for (size_t i = 0; i < words; i++) { cap_out[i] = r.u64_(); if (cap_out[i] >= GL_P) return set_err(QPGPU_ERR_ARG, "non-canonical cap element"); }
  for (int i = 0; i < 4; i++) digest_out[i] = r.u64_();
  if (!r.ok) return set_err(QPGPU_ERR_ARG, "verifier data truncated");
  return (int)ch;
}); }

int qpgpu_circuit_constants_sigmas_cap(const qpgpu_circuit* c, uint64_t* out_cap) {
  if (!c || !out_cap) return set_err(QPGPU_ERR_ARG, "null argument");
  memcpy(out_cap, c->cs_cap.data(), c->cs_cap.size() * 8); return 0;
}
size_t qpgpu_proof_size(const qpgpu_circuit* c) { return c ? c->lay.total : 0; }

} // extern "C"

// ------------------------------------------------------------------------------------------------ prove
static thread_local double g_stage_ms[16]; static thread_local int g_stage_n = 0;
int qpgpu_last_stage_ms(double* out, int cap) { int k = g_stage_n < cap ? g_stage_n : cap; for (int i = 0; i < k; i++) out[i] = g_stage_ms[i]; return g_stage_n; }

static int prove_one(qpgpu_circuit* c, Workspace* w, hipStream_t s, const uint64_t* wires, const uint64_t* pis, const qpgpu_prove_opts* opts,
                     const SaltSrc& salt, uint8_t* out) {
  const size_t n = c->n, N = c->N; const qpgpu_circuit_desc& d = c->d; const Layout& L = c->lay;
  const int nch = c->nch, npp = d.num_partial_products; const size_t capn = size_t(1) << c->cap_h;
  const bool timing = qp::opts().timing != 0;
  struct StageEvents { hipEvent_t ev[16]; int n = 0; ~StageEvents() { for (int i = 0; i < n; i++) (void)hipEventDestroy(ev[i]); } } se;
  hipEvent_t* ev = se.ev; int& nev = se.n;
  auto mark = [&]() { if (timing && nev < 16) { if (hipEventCreate(&ev[nev]) == hipSuccess) { (void)hipEventRecord(ev[nev], s); nev++; } } };
  for (uint32_t i = 0; i < d.num_public_inputs; i++) if (pis[i] >= GL_P) return set_err(QPGPU_ERR_WITNESS, "non-canonical public input");
  if (c->salt && !salt.host && !salt.have_key) return set_err(QPGPU_ERR_ARG, "zero-knowledge circuit: opts.salts or opts.salt_key (32 random bytes) is required");
  u64* pin = w->pinned;
  mark();
  // the PoW result slot and the non-canonical-input flag behind it are re-armed by the previous proof's query_gather; an explicit
  // reset is needed only after workspace creation or an aborted proof
  if (!w->pow_armed) { HIP_TRY(hipMemsetAsync(w->pow_result, 0xFF, 8, s)); HIP_TRY(hipMemsetAsync(w->pow_result + 1, 0, 8, s)); }
  w->pow_armed = false;
  // ---- 1/2: witness in, wires commitment
  const u64* wires_vals = (const u64*)wires;
  if (!(opts && opts->wires_on_device)) { HIP_TRY(hipMemcpyAsync(w->wires_vals, wires, (size_t)c->W * n * 8, hipMemcpyHostToDevice, s)); wires_vals = w->wires_vals; }
  (void)hipEventRecord(w->ev_in, s);                                  // the wires are on the device
  values_to_coeffs(c, w, wires_vals, c->W, w->wires_lde, w->wires_coeffs, s);        // the critical path's first kernels go out first
  // A wire >= p would flow through the lazy arithmetic and be copied raw into query rows: the whole matrix is checked on the device, on
  // the side stream (the wires are read-only), and the flag is collected with the proof's last hand-off — before any byte is released.
  (void)hipStreamWaitEvent(w->side, w->ev_in, 0);
  check_canonical(wires_vals, (size_t)c->W * n, w->pow_result + 1, w->side);
  (void)hipEventRecord(w->ev_canon, w->side);
  u64 pi_hash[4]; hash_no_pad((const u64*)pis, d.num_public_inputs, pi_hash);
  const bool split = c->split_quotient && !w->batch_mode;
  const bool split_serial = qp::opts().split_serial != 0;
  commit_coeffs(c, w, w->wires_coeffs, c->W, w->wires_lde, w->wires_dig, 1, salt, s, [&] {
    if (!split || split_serial) return;   // gate-constraint terms: only wires + constants, run beside the wires leaf hash
    (void)hipEventRecord(w->ev_fork, s); (void)hipStreamWaitEvent(w->side, w->ev_fork, 0);
    quotient_gate_terms(c->cd, c->d_cs_lde, w->wires_lde, pi_hash, w->q_gate_terms, w->side);
    (void)hipEventRecord(w->ev_gates, w->side);
  });
  { int rc_ = fetch_to_host(w, s, cap_ptr(c, w->wires_dig, N), capn * 4); if (rc_) return rc_; }
  mark();
  memcpy(out, pin, L.cap_bytes);
  Transcript tr(nch); Challenger& ch = tr.ch;
  Chal2 betas{{0, 0}}, gammas{{0, 0}}, alphas{{0, 0}};
  tr.start((const u64*)d.circuit_digest, pi_hash, pin, capn * 4, betas.v, gammas.v);
  // ---- 4: Z / partial products
  { ProfScope ps(w, s, PF_ZPP, 8.0 * n * (2.0 * d.num_routed + c->Zp));
    zpp_compute(c->cd, wires_vals, c->d_cs_vals + (size_t)d.num_constants * n, betas, gammas, w->zpp_scratch, w->zs_vals, c->pl, s); }
  values_to_coeffs(c, w, w->zs_vals, c->Zp, w->zs_lde, w->zs_coeffs, s);
  commit_coeffs(c, w, w->zs_coeffs, c->Zp, w->zs_lde, w->zs_dig, 2, salt, s, [&] {
    if (!split || split_serial) return;   // permutation terms: beta/gamma + Z oracle, run beside the Z leaf hash
    (void)hipEventRecord(w->ev_fork, s); (void)hipStreamWaitEvent(w->side, w->ev_fork, 0);
    quotient_perm_terms(c->cd, c->d_cs_lde, w->wires_lde, w->zs_lde, betas, gammas, w->q_perm_terms, c->pl, w->side);
    (void)hipEventRecord(w->ev_perm, w->side);
  });
  { int rc_ = fetch_to_host(w, s, cap_ptr(c, w->zs_dig, N), capn * 4); if (rc_) return rc_; }
  mark();
  memcpy(out + L.cap_bytes, pin, L.cap_bytes);
  tr.after_zs_cap(pin, capn * 4, alphas.v);
  // ---- 6: quotient
  const int nterms = nch + nch * (npp + 1) + d.num_gate_constraints;
  alpha_powers(w->alpha_pows, alphas, nch, nterms, s);
  { ProfScope ps(w, s, PF_QUOTIENT, 8.0 * N * (c->S + c->W + c->Zp + 2 * nch));
    if (split && split_serial) {        // diagnostic: the three split kernels back to back on the main stream
      quotient_gate_terms(c->cd, c->d_cs_lde, w->wires_lde, pi_hash, w->q_gate_terms, s);
      quotient_perm_terms(c->cd, c->d_cs_lde, w->wires_lde, w->zs_lde, betas, gammas, w->q_perm_terms, c->pl, s);
      quotient_combine(c->cd, w->q_perm_terms, w->q_gate_terms, w->alpha_pows, nterms, w->q_vals, s);
    } else if (split) {
      (void)hipStreamWaitEvent(s, w->ev_gates, 0); (void)hipStreamWaitEvent(s, w->ev_perm, 0);
      quotient_combine(c->cd, w->q_perm_terms, w->q_gate_terms, w->alpha_pows, nterms, w->q_vals, s);
    } else if (!w->shared_gpu && qp::opts().quotient_halves) {
      // single proof: the permutation half and the gate half run at once on two streams (each alone offers only 2 waves/SIMD
      // and stalls on its loads); q_lde is free until the quotient commitment and serves as scratch for the two partial sums
      u64* pa = w->q_lde; u64* pb = w->q_lde + (size_t)nch * N;
      (void)hipEventRecord(w->ev_fork, s); (void)hipStreamWaitEvent(w->side, w->ev_fork, 0);
      quotient_eval_part(2, c->cd, c->d_cs_lde, w->wires_lde, w->zs_lde, betas, gammas, w->alpha_pows, nterms, pi_hash, pb, c->pl, w->side);
      (void)hipEventRecord(w->ev_gates, w->side);
      quotient_eval_part(1, c->cd, c->d_cs_lde, w->wires_lde, w->zs_lde, betas, gammas, w->alpha_pows, nterms, pi_hash, pa, c->pl, s);
      (void)hipStreamWaitEvent(s, w->ev_gates, 0);
      quotient_sum(c->cd, pa, pb, w->q_vals, s);
    } else quotient_eval(c->cd, c->d_cs_lde, w->wires_lde, w->zs_lde, betas, gammas, w->alpha_pows, nterms, pi_hash, w->q_vals, c->pl, s, c->quotient_ls && qp::opts().quotient_shared_loads != 0); }
  { ProfScope ps(w, s, PF_QUOTIENT_INTT, 8.0 * N * nch * 3);
    ntt_inplace(w->q_vals, N, nch, c->log_N, true, true, c->d_qpost, c->pl, s); }     // coset iNTT: leaf order in, coefficients out
  commit_coeffs(c, w, w->q_vals, c->Q, w->q_lde, w->q_dig, 3, salt, s);
  { int rc_ = fetch_to_host(w, s, cap_ptr(c, w->q_dig, N), capn * 4); if (rc_) return rc_; }
  mark();
  memcpy(out + 2 * L.cap_bytes, pin, L.cap_bytes);
  Ext zeta = tr.after_quotient_cap(pin, capn * 4);
  if (ex_eq(ex_pow(zeta, n), ex(1, 0))) return set_err(QPGPU_ERR_ZETA, "zeta lies in the trace subgroup");
  // ---- 8: openings
  const Ext gzeta = ex_scale(zeta, gl_root(c->log_n));
  const Ext zinv = ex_inv(zeta), gzinv = ex_inv(gzeta);
  u64 *zp0a = w->zp0, *zp0b = w->zp0 + n + 1, *zp1a = w->zp1, *zp1b = w->zp1 + n + 1;
  u64 *zi0a = w->zi0, *zi0b = w->zi0 + n + 1, *zi1a = w->zi1, *zi1b = w->zi1 + n + 1;
  {
    ProfScope ps(w, s, PF_OPENINGS, 8.0 * n * (c->S + c->W + c->Zp + c->Q + nch + 8));
    ExtPowArgs pa; pa.count = n + 1; pa.nbits = c->log_n + 1;
    pa.out_a[0] = zp0a; pa.out_b[0] = zp0b; pa.out_a[1] = zp1a; pa.out_b[1] = zp1b; pa.out_a[2] = zi0a; pa.out_b[2] = zi0b; pa.out_a[3] = zi1a; pa.out_b[3] = zi1b;
    const Ext bases[4] = {zeta, gzeta, zinv, gzinv};
    for (int t = 0; t < 4; t++) { Ext x = bases[t]; for (int k = 0; k < pa.nbits; k++) { pa.pw[t][k] = x; x = ex_mul(x, x); } }
    ext_powers4(pa, s);
    // proof order: constants|sigmas (S), wires (W), zs (nch), zs_next (nch, at g*zeta), pp (nch*npp), quotient (Q)
    OpenArgs oa; oa.nseg = 6; oa.zp0_a = zp0a; oa.zp0_b = zp0b; oa.zp1_a = zp1a; oa.zp1_b = zp1b;
    oa.seg[0] = OpenSeg{c->d_cs_coeffs, n, c->S, 0, 0};
    oa.seg[1] = OpenSeg{w->wires_coeffs, n, c->W, 0, 0};
    oa.seg[2] = OpenSeg{w->zs_coeffs, n, nch, 0, 0};
    oa.seg[3] = OpenSeg{w->zs_coeffs, n, nch, 1, 0};
    oa.seg[4] = OpenSeg{w->zs_coeffs + (size_t)nch * n, n, nch * npp, 0, 0};
    oa.seg[5] = OpenSeg{w->q_vals, n, c->Q, 0, 0};
    open_polys(oa, c->log_n, w->open_partial, w->openings, s);
  }
  { int rc_ = fetch_to_host(w, s, w->openings, 2 * L.n_openings); if (rc_) return rc_; }
  mark();
  memcpy(out + L.openings_off, pin, 16 * L.n_openings);
  // ---- 9: FRI
  const Ext fa = tr.after_openings(pin, (size_t)c->S + c->W + nch, (size_t)nch * npp + c->Q);
  {
    ProfScope ps(w, s, PF_FRI_COMBINE, 8.0 * n * (c->S + c->W + c->Zp + c->Q + nch + 24));
    fri_compose(PolyGroup{c->d_cs_coeffs, n, c->S}, PolyGroup{w->wires_coeffs, n, c->W}, PolyGroup{w->zs_coeffs, n, c->Zp},
                PolyGroup{w->q_vals, n, c->Q}, nch, c->log_n, fa, w->compose_partial, w->comp0, w->comp1, s);
    fri_divide_combine(w->comp0, w->comp1, c->log_n, zp0a, zp0b, zp1a, zp1b, zi0a, zi0b, zi1a, zi1b, ex_pow(fa, nch), w->div_scratch, w->fri_coeffs[0], s);
  }
  Ext fri_betas[MAX_FRI_LAYERS];
  for (int l = 0; l < c->nlayers; l++) {
    const int lg = c->fri_log_n[l]; const size_t nl = size_t(1) << lg, Nl = nl << c->rate;
    const size_t leaves = Nl >> c->arity[l];
    {
      ProfScope ps(w, s, PF_FRI_LAYERS, 8.0 * (2 * nl + 2 * Nl * 2 + 12 * leaves));
      lde_batch(w->fri_coeffs[l], nl, w->fri_vals[l], Nl, 2, lg, c->rate, l == 0 ? c->d_lde_scale : c->d_fri_scale[l], c->pl, s, w->shared_gpu);
      merkle_leaves_fri(w->fri_vals[l], w->fri_vals[l] + Nl, c->arity[l], leaves, w->fri_dig[l], s);
      merkle_levels(w->fri_dig[l], leaves, c->cap_h, s, 0, w->batch_mode);
    }
    { int rc_ = fetch_to_host(w, s, cap_ptr(c, w->fri_dig[l], leaves), capn * 4); if (rc_) return rc_; }
    memcpy(out + L.fri_caps_off + (size_t)l * L.cap_bytes, pin, L.cap_bytes);
    fri_betas[l] = tr.fri_layer_cap(pin, capn * 4);
    const size_t nn = nl >> c->arity[l];
    fri_fold(w->fri_coeffs[l], w->fri_coeffs[l] + nl, nn, c->arity[l], fri_betas[l], w->fri_coeffs[l + 1], w->fri_coeffs[l + 1] + nn, s);
  }
  {
    const size_t fl = L.final_poly_len;
    { int rc_ = fetch_to_host(w, s, w->fri_coeffs[c->nlayers], 2 * fl); if (rc_) return rc_; }
    u64* fp = (u64*)(out + L.final_poly_off);
    for (size_t i = 0; i < fl; i++) { u64 a = pin[i], b = pin[fl + i]; memcpy(&fp[2 * i], &a, 8); memcpy(&fp[2 * i + 1], &b, 8); tr.final_poly_coeff(a, b); }
  }
  mark();
  // ---- PoW
  u64 pow_witness;
  if (opts && opts->have_pow_witness) pow_witness = opts->pow_witness;
  else {
    Sponge12 sp; for (int i = 0; i < 12; i++) sp.s[i] = ch.st[i];
    for (int i = 0; i < ch.n_in; i++) sp.s[i] = ch.in[i];
    const int pos = ch.n_in;
    u64 none = ~0ULL; pow_witness = none;
    // first launch = the expected number of candidates (2^pow_bits: 63 % of proofs stop there), then doubling spans; the smallest valid
    // witness wins whatever the span schedule (atomicMin over a range that is complete before the next one starts)
    u64 base = 0, span = 1ULL << (d.fri_pow_bits < 12 ? 12 : (d.fri_pow_bits > 22 ? 22 : d.fri_pow_bits));
    { const int sh = qp::opts().pow_first_shift; if (sh > 0 && sh <= 4) span <<= sh; }
    for (int iter = 0; iter < 24 && pow_witness == none; iter++) {
      { ProfScope ps(w, s, PF_POW, 0.0); pow_grind(sp, pos, base, span, (int)d.fri_pow_bits, w->pow_result, s); }
      { int rc_ = fetch_to_host(w, s, w->pow_result, 1); if (rc_) return rc_; }
      pow_witness = pin[0]; base += span; if (span < (1ULL << 26)) span <<= 1;
    }
    if (pow_witness == none) return set_err(QPGPU_ERR_POW, "proof-of-work grind found no witness");
  }
  {
    u64 resp = tr.pow_response(pow_witness);
    int lz = resp == 0 ? 64 : __builtin_clzll(resp);
    if (lz < (int)d.fri_pow_bits) return set_err(QPGPU_ERR_POW, "supplied pow_witness does not meet proof_of_work_bits");
  }
  memcpy(out + L.pow_off, &pow_witness, 8);
  mark();
  // ---- queries
  GatherArgs ga; memset(&ga, 0, sizeof ga);
  ga.trees[0] = GatherTree{c->d_cs_lde, N, c->S, c->d_cs_dig};
  ga.trees[1] = GatherTree{w->wires_lde, N, c->W + c->salt, w->wires_dig};
  ga.trees[2] = GatherTree{w->zs_lde, N, c->Zp + c->salt, w->zs_dig};
  ga.trees[3] = GatherTree{w->q_lde, N, c->Q + c->salt, w->q_dig};
  ga.log_N = c->log_N; ga.cap_height = c->cap_h; ga.num_layers = c->nlayers;
  for (int l = 0; l < c->nlayers; l++) {
    const size_t Nl = size_t(1) << (c->fri_log_n[l] + c->rate);
    ga.layers[l] = GatherFri{w->fri_vals[l], w->fri_vals[l] + Nl, w->fri_dig[l], c->arity[l], c->fri_log_n[l] + c->rate - c->arity[l]};
  }
  ga.nq = d.fri_num_queries; ga.bytes_per_query = L.bytes_per_query; ga.pow_reset = w->pow_result;
  for (int q = 0; q < ga.nq; q++) ga.idx[q] = tr.query_index(N);
  { ProfScope ps(w, s, PF_GATHER, 2.0 * L.bytes_per_query * ga.nq); query_gather(ga, w->query_bytes, s); }
  const size_t qwords = (L.bytes_per_query * ga.nq + 7) / 8;
  (void)hipStreamWaitEvent(s, w->ev_canon, 0);                   // the input check (side stream) must have finished before its flag is read
  { int rc_ = fetch_to_host(w, s, (const u64*)w->query_bytes, qwords, w->pow_result + 1, true); if (rc_) return rc_; }
  HIP_TRY(hipGetLastError());
  w->pow_armed = true;            // the gather kernel has run and the flag was collected and cleared: both slots are armed for the next proof
  if (pin[qwords] != 0) return set_err(QPGPU_ERR_WITNESS, "non-canonical field element (>= p) in the wire matrix" + std::string(salt.host ? " or the salts" : ""));
  memcpy(out + L.queries_off, pin, L.bytes_per_query * ga.nq);
  { u64 npi = d.num_public_inputs; memcpy(out + L.pis_off, &npi, 8); memcpy(out + L.pis_off + 8, pis, 8 * npi); }
  mark();
  if (w->pev_used || timing) HIP_TRY(hipStreamSynchronize(s));     // event timestamps need the stream drained (the hand-offs above do not sync)
  prof_flush(w);
  if (timing) {
    g_stage_n = nev - 1;
    for (int i = 0; i + 1 < nev; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); g_stage_ms[i] = ms; }
  }
  return 0;
}

// proof `index` of a call: explicit salts advance by one proof's columns, a keyed stream by one nonce
static SaltSrc salt_source(const qpgpu_prove_opts* o, size_t index, size_t N) {
  SaltSrc s;
  if (!o) return s;
  if (o->salts) s.host = o->salts + index * 3 * 4 * N;
  else if (o->salt_key) { s.have_key = true; s.key = salt_key_from_bytes(o->salt_key); s.stream = o->salt_stream + index; }
  return s;
}

extern "C" {

// Re-derives the Fiat-Shamir challenges of a finished proof with the prover's own Transcript (host only, no device work).
// out: pi_hash[4] | betas[nch] | gammas[nch] | alphas[nch] | zeta[2] | fri_alpha[2] | fri_betas[2*layers] | pow_response | indices[queries].
// Returns the number of words written, or a negative status. Declared in include/qpgpu_internal.h.
int qpgpu_replay_transcript(const qpgpu_circuit_desc* dp, const uint8_t* proof, size_t len, uint64_t* out, size_t out_cap) { return abi_guard([&]() -> int {
  if (!dp || !proof || !out) return set_err(QPGPU_ERR_ARG, "null argument");
  const qpgpu_circuit_desc& d = *dp;
  if (d.num_challenges < 1 || d.num_challenges > 2 || d.fri_num_layers > MAX_FRI_LAYERS || d.fri_cap_height > 20 || d.degree_bits + d.fri_rate_bits > 28)
    return set_err(QPGPU_ERR_ARG, "unsupported shape");
  { uint32_t sum = 0; for (uint32_t i = 0; i < d.fri_num_layers; i++) sum += d.fri_arity_bits[i]; if (sum > d.degree_bits) return set_err(QPGPU_ERR_ARG, "FRI reductions exceed degree"); }
  Layout L; compute_layout(d, L);
  if (len != L.total) return set_err(QPGPU_ERR_ARG, "proof length " + std::to_string(len) + " != " + std::to_string(L.total) + " for this circuit shape");
  const int nch = d.num_challenges; const size_t capw = size_t(4) << d.fri_cap_height, N = size_t(1) << (d.degree_bits + d.fri_rate_bits);
  const size_t need = 4 + 3 * (size_t)nch + 4 + 2 * (size_t)d.fri_num_layers + 1 + d.fri_num_queries;
  if (out_cap < need) return set_err(QPGPU_ERR_BUFFER, "output too small");
  std::vector<u64> words((len + 7) / 8 + 1); memcpy(words.data(), proof, len);       // caps and openings (everything before the query rounds) are 8-byte aligned
  auto at = [&](size_t off) { return words.data() + off / 8; };
  u64 npi; memcpy(&npi, proof + L.pis_off, 8);
  if (npi != d.num_public_inputs) return set_err(QPGPU_ERR_ARG, "public input count mismatch");
  size_t o = 0;
  std::vector<u64> pis(d.num_public_inputs ? d.num_public_inputs : 1); memcpy(pis.data(), proof + L.pis_off + 8, 8 * (size_t)d.num_public_inputs);   // unaligned: follows the query rounds
  u64 pi_hash[4]; hash_no_pad(pis.data(), d.num_public_inputs, pi_hash);
  for (int i = 0; i < 4; i++) out[o++] = pi_hash[i];
  Transcript tr(nch);
  u64 betas[2], gammas[2], alphas[2];
  tr.start((const u64*)d.circuit_digest, pi_hash, at(0), capw, betas, gammas);
  for (int i = 0; i < nch; i++) out[o++] = betas[i];
  for (int i = 0; i < nch; i++) out[o++] = gammas[i];
  tr.after_zs_cap(at(L.cap_bytes), capw, alphas);
  for (int i = 0; i < nch; i++) out[o++] = alphas[i];
  Ext zeta = tr.after_quotient_cap(at(2 * L.cap_bytes), capw); out[o++] = zeta.a; out[o++] = zeta.b;
  const size_t S = (size_t)d.num_constants + d.num_routed;
  Ext fa = tr.after_openings(at(L.openings_off), S + d.num_wires + nch, (size_t)nch * d.num_partial_products + (size_t)nch * d.quotient_degree_factor);
  out[o++] = fa.a; out[o++] = fa.b;
  for (uint32_t l = 0; l < d.fri_num_layers; l++) { Ext b = tr.fri_layer_cap(at(L.fri_caps_off + l * L.cap_bytes), capw); out[o++] = b.a; out[o++] = b.b; }
  for (size_t i = 0; i < L.final_poly_len; i++) { u64 a, b; memcpy(&a, proof + L.final_poly_off + 16 * i, 8); memcpy(&b, proof + L.final_poly_off + 16 * i + 8, 8); tr.final_poly_coeff(a, b); }
  u64 pw; memcpy(&pw, proof + L.pow_off, 8);
  out[o++] = tr.pow_response(pw);
  for (uint32_t q = 0; q < d.fri_num_queries; q++) out[o++] = tr.query_index(N);
  return (int)o;
}); }

// Per-kernel-group timing for bench.py's roofline object (HIP events on the proving stream; off by default).
void qpgpu_profile_enable(int on) { g_prof_on.store(on); }
void qpgpu_profile_reset(void) { std::lock_guard<std::mutex> lk(g_prof_mu); for (auto& a : g_prof) a = ProfAcc(); }
int qpgpu_profile_count(void) { return PF_COUNT; }
const char* qpgpu_profile_name(int id) { return id >= 0 && id < PF_COUNT ? PROF_NAMES[id] : ""; }
int qpgpu_profile_get(int id, uint64_t* launches, double* ms, double* alg_bytes) {
  if (id < 0 || id >= PF_COUNT) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  *launches = g_prof[id].launches; *ms = g_prof[id].ms; *alg_bytes = g_prof[id].bytes; return 0;
}

int qpgpu_host_alloc(size_t bytes, void** out) { return abi_guard([&]() -> int {
  if (!out || bytes == 0) return set_err(QPGPU_ERR_ARG, "null argument");
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocPortable));
  *out = p; return 0;
}); }
void qpgpu_host_free(void* p) { if (p) (void)hipHostFree(p); }

int qpgpu_prove(qpgpu_circuit* c, const uint64_t* wires, const uint64_t* public_inputs, size_t num_public_inputs,
                const qpgpu_prove_opts* opts, uint8_t* out, size_t out_cap, size_t* out_len) { return abi_guard([&]() -> int {
  if (!c || !wires || !public_inputs || !out) return set_err(QPGPU_ERR_ARG, "null argument");
  if (num_public_inputs != c->d.num_public_inputs) return set_err(QPGPU_ERR_ARG, "public input count mismatch");
  if (out_cap < c->lay.total) return set_err(QPGPU_ERR_BUFFER, "output buffer too small: need " + std::to_string(c->lay.total));
  HIP_TRY(hipSetDevice(c->device));
  Workspace* w = nullptr; int rc = ws_acquire(c, &w); if (rc) return rc;
  hipStream_t s = (opts && opts->stream) ? (hipStream_t)opts->stream : w->stream;
  // A host that keeps several proofs in flight by calling qpgpu_prove from several threads (the reference's rayon workers, tree.rs:100;
  // aggregator.py's dependency-driven tree) gets the kernel choices of qpgpu_prove_batch: what is best for a lone proof (side-stream
  // quotient halves, the 16-lane form for mid-sized tree levels) is not when other proofs fill the chip. Same bytes either way.
  struct InFlight { int others; InFlight() : others(g_proofs_in_flight.fetch_add(1, std::memory_order_relaxed)) {} ~InFlight() { g_proofs_in_flight.fetch_sub(1, std::memory_order_relaxed); } } fl;
  w->batch_mode = fl.others > 0;
  w->cores_to_spare = usable_cpus() >= 2 * (fl.others + 1);
  rc = prove_one(c, w, s, wires, public_inputs, opts, salt_source(opts, 0, c->N), out);
  w->batch_mode = false; w->cores_to_spare = true;
  if (rc != 0) {                              // nothing of this proof may still run when the workspace is recycled, and no byte derived from
    (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(w->side); w->pev_used = 0;      // a refused witness may stay in the caller's buffer
    memset(out, 0, c->lay.total);
  }
  ws_release(c, w);
  if (rc == 0 && out_len) *out_len = c->lay.total;
  return rc;
}); }

int qpgpu_prove_batch(qpgpu_circuit* c, size_t batch, const uint64_t* const* wires, const uint64_t* const* public_inputs,
                      size_t num_public_inputs, const qpgpu_prove_opts* opts, uint8_t* out, size_t out_stride, size_t* out_lens) { return abi_guard([&]() -> int {
  if (!c || !wires || !public_inputs || !out) return set_err(QPGPU_ERR_ARG, "null argument");
  if (num_public_inputs != c->d.num_public_inputs) return set_err(QPGPU_ERR_ARG, "public input count mismatch");
  if (out_stride < c->lay.total) return set_err(QPGPU_ERR_BUFFER, "out_stride smaller than proof size");
  if (batch == 0) return 0;
  int workers = qp::opts().streams; if (workers < 1) workers = 1; if ((size_t)workers > batch) workers = (int)batch;
  std::atomic<size_t> next(0); std::atomic<int> first_rc(0); std::mutex emu; std::string emsg;
  auto body_inner = [&]() {
    if (hipSetDevice(c->device) != hipSuccess) { first_rc = QPGPU_ERR_HIP; return; }
    Workspace* w = nullptr; int rc = ws_acquire(c, &w);
    if (rc) { std::lock_guard<std::mutex> lk(emu); if (!first_rc) { first_rc = rc; emsg = g_last_error; } return; }
    w->batch_mode = workers > 1;
    w->cores_to_spare = usable_cpus() >= 2 * workers;      // busy-polling 4 workers on a 16-core share costs nothing and is 2-3 % faster than napping
    w->shared_gpu = qp::opts().shared_mode != 0;   // the batch-mode kernel choices stopped paying once the field arithmetic got cheaper (profiles/r01_notes_ab_experiments.md)
    for (;;) {
      size_t i = next.fetch_add(1); if (i >= batch || first_rc.load()) break;
      qpgpu_prove_opts o; if (opts) o = *opts; else memset(&o, 0, sizeof o);
      o.stream = nullptr;
      rc = prove_one(c, w, w->stream, wires[i], public_inputs[i], &o, salt_source(opts, i, c->N), out + i * out_stride);
      if (rc) { (void)hipStreamSynchronize(w->stream); (void)hipStreamSynchronize(w->side); w->pev_used = 0; std::lock_guard<std::mutex> lk(emu); if (!first_rc) { first_rc = rc; emsg = g_last_error; } break; }
      if (out_lens) out_lens[i] = c->lay.total;
    }
    w->shared_gpu = false; w->batch_mode = false; w->cores_to_spare = true;
    ws_release(c, w);
  };
  auto body = [&]() {
    try { body_inner(); }
    catch (...) { std::lock_guard<std::mutex> lk(emu); if (!first_rc) { first_rc = QPGPU_ERR_OOM; emsg = "internal error in a batch worker (host allocation?)"; } }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < workers; t++) { try { th.emplace_back(body); } catch (...) { break; } }     // fewer workers if threads cannot be created
  body();
  for (auto& t : th) t.join();
  if (first_rc.load()) {                       // a failed batch releases nothing: every slot is cleared, also those of proofs that had completed
    for (size_t i = 0; i < batch; i++) { memset(out + i * out_stride, 0, c->lay.total); if (out_lens) out_lens[i] = 0; }
    return set_err(first_rc.load(), emsg);
  }
  return 0;
}); }

} // extern "C"
