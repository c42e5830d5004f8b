// Per-kernel test hooks (host buffers in, host buffers out). Not part of the drop-in ABI: they exist so tests/ can
// compare each HIP kernel with its oracle twin bit for bit (SURVEY.md §4 implication (4)).
#include "../../include/qpgpu_internal.h"
#include "kernels.h"
#include <vector>
#include <string>
#include <chrono>
#include "poseidon.h"
#include "salt.h"

namespace qp { extern thread_local std::string g_last_error; int set_err(int code, const std::string& m); }
using namespace qp;

#define HOOK_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(-3, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  u64* u() { return (u64*)p; }
};

extern "C" {

int qpgpu_test_permute(uint64_t* states, size_t count) { return qpgpu_test_permute_form(states, count, -1); }
// form: 0 literal partial rounds, 1 fast partial rounds, -1 the library's default
int qpgpu_test_permute_form(uint64_t* states, size_t count, int form) {
  if (!states || form < -1 || form > 3) return set_err(-1, "bad argument");
  DevBuf d; HOOK_CHECK(d.alloc(count * 96));
  HOOK_CHECK(hipMemcpy(d.p, states, count * 96, hipMemcpyHostToDevice));
  poseidon_permute_batch(d.u(), count, 0, form);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(states, d.p, count * 96, hipMemcpyDeviceToHost));
  return 0;
}

// mode 0: forward DIF (natural in, bit-reversed out); 1: forward DIT (bit-reversed in, natural out);
// 2: inverse DIF (unscaled); 3: inverse DIT (unscaled). data: [count][n] in place.
int qpgpu_test_ntt(uint64_t* data, int count, int log_n, int mode) {
  size_t n = size_t(1) << log_n;
  DevBuf d, rt; HOOK_CHECK(d.alloc(count * n * 8)); HOOK_CHECK(rt.alloc((n / 2 + 1) * 8));
  HOOK_CHECK(hipMemcpy(d.p, data, count * n * 8, hipMemcpyHostToDevice));
  build_root_table(rt.u(), log_n, 0);
  NttPlan pl{rt.u(), log_n};
  ntt_inplace(d.u(), n, count, log_n, (mode & 1) != 0, (mode & 2) != 0, nullptr, pl, 0);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(data, d.p, count * n * 8, hipMemcpyDeviceToHost));
  return 0;
}

// coeffs [count][n] -> out [count][n << log_cosets], leaf order, coset shift `shift`.
int qpgpu_test_lde(const uint64_t* coeffs, int count, int log_n, int log_cosets, uint64_t shift, uint64_t* out) {
  size_t n = size_t(1) << log_n, N = n << log_cosets; int log_N = log_n + log_cosets;
  DevBuf in, o, rt, sc; HOOK_CHECK(in.alloc(count * n * 8)); HOOK_CHECK(o.alloc(count * N * 8)); HOOK_CHECK(rt.alloc((N / 2 + 1) * 8)); HOOK_CHECK(sc.alloc(N * 8));
  HOOK_CHECK(hipMemcpy(in.p, coeffs, count * n * 8, hipMemcpyHostToDevice));
  build_root_table(rt.u(), log_N, 0);
  NttPlan pl{rt.u(), log_N};
  build_lde_scale(sc.u(), log_n, log_cosets, shift, pl, 0);
  lde_batch(in.u(), n, o.u(), N, count, log_n, log_cosets, sc.u(), pl, 0);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(out, o.p, count * N * 8, hipMemcpyDeviceToHost));
  return 0;
}

// cols: [width][num_leaves] column-major. digests out: all levels concatenated down to the cap.
int qpgpu_test_merkle(const uint64_t* cols, int width, size_t num_leaves, int cap_height, uint64_t* digests) {
  size_t tot = merkle_total_digests(num_leaves, cap_height);
  DevBuf c, d; HOOK_CHECK(c.alloc((size_t)width * num_leaves * 8)); HOOK_CHECK(d.alloc(tot * 32));
  HOOK_CHECK(hipMemcpy(c.p, cols, (size_t)width * num_leaves * 8, hipMemcpyHostToDevice));
  int done = merkle_leaves_colmajor(c.u(), num_leaves, width, num_leaves, cap_height, d.u(), 0);
  merkle_levels(d.u(), num_leaves, cap_height, 0, done);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(digests, d.p, tot * 32, hipMemcpyDeviceToHost));
  return 0;
}

// ---- golden-vector replay hooks: the device sponge / 2-to-1 / FRI-leaf kernels on caller-supplied rows, one kernel form at a
// time, so tests can pin each form directly to vectors the reference holds (KATs, the Merkle paths of bench-data/proof.bin).
// Leaf digests only (hash_or_noop per row). cols: [width][num_leaves] column-major. mode 0: per-lane kernel (plain), 1: 2-lane pair
// kernel, 2: per-lane kernel with fused levels 1-2 (needs num_leaves % 256 == 0; only the leaf digests are returned); 3 and 4: forms 0 and 2
// with the fast form of the partial rounds (poseidon_dev.h).
int qpgpu_test_leaf_hash(const uint64_t* cols, int width, size_t num_leaves, int mode, uint64_t* digests) {
  if (!cols || !digests || width < 1 || num_leaves < 1 || mode < 0 || mode > 4 || ((mode == 2 || mode == 4) && num_leaves % 256)) return set_err(-1, "bad argument");
  DevBuf c, d; HOOK_CHECK(c.alloc((size_t)width * num_leaves * 8)); HOOK_CHECK(d.alloc(num_leaves * 32 * 2));
  HOOK_CHECK(hipMemcpy(c.p, cols, (size_t)width * num_leaves * 8, hipMemcpyHostToDevice));
  merkle_leaves_colmajor_mode(c.u(), num_leaves, width, num_leaves, d.u(), mode, 0);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(digests, d.p, num_leaves * 32, hipMemcpyDeviceToHost));
  return 0;
}
// out[i] = two_to_one(pairs[i][0..4], pairs[i][4..8]). mode 0: per-lane, 1: 2-lane pair, 2: 16-lane cooperative, 3: fused multi-level
// kernel (one level).
int qpgpu_test_two_to_one(const uint64_t* pairs, size_t count, int mode, uint64_t* out) {
  if (!pairs || !out || count < 1 || mode < 0 || mode > 3) return set_err(-1, "bad argument");
  DevBuf a, b; HOOK_CHECK(a.alloc(count * 64)); HOOK_CHECK(b.alloc(count * 32));
  HOOK_CHECK(hipMemcpy(a.p, pairs, count * 64, hipMemcpyHostToDevice));
  merkle_level_mode(a.u(), b.u(), count, mode, 0);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(out, b.p, count * 32, hipMemcpyDeviceToHost));
  return 0;
}
// FRI layer leaves: leaf L = 2^arity_bits extension values (a[k], b[k]). mode 0: per-lane kernel, 1: 16-lane cooperative kernel.
int qpgpu_test_fri_leaf_hash(const uint64_t* a, const uint64_t* b, int arity_bits, size_t num_leaves, int mode, uint64_t* digests) {
  if (!a || !b || !digests || arity_bits < 1 || arity_bits > 4 || num_leaves < 1 || mode < 0 || mode > 1) return set_err(-1, "bad argument");
  const size_t n = num_leaves << arity_bits;
  DevBuf da, db, d; HOOK_CHECK(da.alloc(n * 8)); HOOK_CHECK(db.alloc(n * 8)); HOOK_CHECK(d.alloc(num_leaves * 32));
  HOOK_CHECK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice)); HOOK_CHECK(hipMemcpy(db.p, b, n * 8, hipMemcpyHostToDevice));
  merkle_leaves_fri_mode(da.u(), db.u(), arity_bits, num_leaves, d.u(), mode, 0);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(digests, d.p, num_leaves * 32, hipMemcpyDeviceToHost));
  return 0;
}

// host-side costs that sit on the proof's critical path: ns per host permutation, and us per (tiny kernel + 512 B D2H + sync)
double qpgpu_test_host_permute_ns(int iters) {
  u64 st[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) poseidon_permute(st);
  auto t1 = std::chrono::steady_clock::now();
  volatile u64 sink = st[0]; (void)sink;
  return std::chrono::duration<double, std::nano>(t1 - t0).count() / iters;
}
double qpgpu_test_sync_roundtrip_us2(int iters, int mode) {
  // mode 0: hipStreamSynchronize; 1: spin on hipStreamQuery; 2: spin on hipEventQuery; 3: kernel writes pinned host memory, host spins on it
  if (mode == 10) { return (double)hipSetDeviceFlags(hipDeviceScheduleSpin); }
  DevBuf d; if (d.alloc(4096) != hipSuccess) return -1;
  u64* pin = nullptr; if (hipHostMalloc((void**)&pin, 4096, hipHostMallocMapped) != hipSuccess) return -1;
  u64* pin_dev = nullptr; (void)hipHostGetDevicePointer((void**)&pin_dev, pin, 0);
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  salt_fill(d.u(), 16, SaltKey{{1, 2, 3, 4, 5, 6, 7, 8}}, 1, 0, s); (void)hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) {
    if (mode == 3) {
      volatile u64* flag = pin + 64; *flag = 0;
      salt_fill(pin_dev, 18, SaltKey{{1, 2, 3, 4, 5, 6, 7, 8}}, (u64)i + 1, 0, s);   // writes 72 u64 incl. slot 64 (non-zero with overwhelming probability)
      while (*flag == 0) { }
    } else {
      salt_fill(d.u(), 16, SaltKey{{1, 2, 3, 4, 5, 6, 7, 8}}, 1, 0, s); (void)hipMemcpyAsync(pin, d.p, 512, hipMemcpyDeviceToHost, s);
      if (mode == 0) (void)hipStreamSynchronize(s);
      else if (mode == 1) { while (hipStreamQuery(s) == hipErrorNotReady) { } }
      else { (void)hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { } }
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  (void)hipStreamSynchronize(s);
  (void)hipEventDestroy(ev); (void)hipStreamDestroy(s); (void)hipHostFree(pin);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
}
double qpgpu_test_sync_roundtrip_us(int iters) {
  DevBuf d; if (d.alloc(4096) != hipSuccess) return -1;
  u64* pin = nullptr; if (hipHostMalloc((void**)&pin, 4096, hipHostMallocDefault) != hipSuccess) return -1;
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  salt_fill(d.u(), 16, SaltKey{{1, 2, 3, 4, 5, 6, 7, 8}}, 1, 0, s); (void)hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; i++) { salt_fill(d.u(), 16, SaltKey{{1, 2, 3, 4, 5, 6, 7, 8}}, 1, 0, s); (void)hipMemcpyAsync(pin, d.p, 512, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
  auto t1 = std::chrono::steady_clock::now();
  (void)hipStreamDestroy(s); (void)hipHostFree(pin);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
}

// values [count][n] natural -> coefficients [count][n] natural (the production witness/Z iNTT path)
int qpgpu_test_intt_natural(const uint64_t* vals, int count, int log_n, uint64_t* coeffs) {
  size_t n = size_t(1) << log_n;
  DevBuf a, b, t, rt; HOOK_CHECK(a.alloc(count * n * 8)); HOOK_CHECK(b.alloc(count * n * 8)); HOOK_CHECK(t.alloc(count * n * 8)); HOOK_CHECK(rt.alloc((n / 2 + 1) * 8));
  HOOK_CHECK(hipMemcpy(a.p, vals, count * n * 8, hipMemcpyHostToDevice));
  build_root_table(rt.u(), log_n, 0);
  NttPlan pl{rt.u(), log_n};
  if (ntt16_supported(log_n)) ntt16_run(a.u(), n, b.u(), n, count, log_n, 0, true, true, nullptr, gl_inv((u64)n), pl, 0);
  else { ntt_copy(a.u(), n, t.u(), n, count, log_n, false, true, pl, 0); bitrev_permute(t.u(), n, b.u(), n, count, log_n, gl_inv((u64)n), 0); }
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(coeffs, b.p, count * n * 8, hipMemcpyDeviceToHost));
  return 0;
}

// PMC calibration kernel: streams a [width][n] column-major matrix with the leaf kernel's access pattern (8 B per lane per
// column) and writes one u64 per row, so FETCH_SIZE / WRITE_SIZE can be scaled against a known byte count
// (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16 B/lane streams).
static __global__ void __launch_bounds__(256) k_calib_colsum(const u64* cols, size_t stride, int width, size_t n, u64* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 acc = 0;
  for (int k = 0; k < width; k++) acc ^= cols[(size_t)k * stride + i];
  out[i] = acc;
}
int qpgpu_test_calib_colsum(int width, size_t n, int iters) {
  DevBuf c, o; HOOK_CHECK(c.alloc((size_t)width * n * 8)); HOOK_CHECK(o.alloc(n * 8));
  HOOK_CHECK(hipMemset(c.p, 3, (size_t)width * n * 8));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_calib_colsum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, c.u(), n, width, n, o.u());
  HOOK_CHECK(hipDeviceSynchronize());
  return 0;
}

// Timing hook: G permutations/s of the one-permutation-per-lane kernel at a given grid size (occupancy sweep)
double qpgpu_bench_permute(size_t count, int iters) { return qpgpu_bench_permute_form(count, iters, -1); }
double qpgpu_bench_permute_form(size_t count, int iters, int form) {
  DevBuf d; if (d.alloc(count * 96) != hipSuccess) return -1;
  (void)hipMemset(d.p, 0, count * 96);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  poseidon_permute_batch(d.u(), count, 0, form); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0); for (int i = 0; i < iters; i++) poseidon_permute_batch(d.u(), count, 0, form); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)count * iters / (ms * 1e-3) / 1e9;
}

// Timing hook: ms per launch of the x8 coset LDE of `count` columns of 2^log_n coefficients (device-resident data).
double qpgpu_bench_lde(int count, int log_n, int iters, int force_generic) {
  size_t n = size_t(1) << log_n, N = n << 3; int log_N = log_n + 3;
  DevBuf in, o, rt, sc; if (in.alloc(count * n * 8) != hipSuccess || o.alloc(count * N * 8) != hipSuccess || rt.alloc((N / 2 + 1) * 8) != hipSuccess || sc.alloc(N * 8) != hipSuccess) return -1;
  (void)hipMemset(in.p, 1, count * n * 8);
  build_root_table(rt.u(), log_N, 0);
  NttPlan pl{rt.u(), log_N};
  build_lde_scale(sc.u(), log_n, 3, GL_GEN, pl, 0);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&]() { if (force_generic) lde_batch_generic(in.u(), n, o.u(), N, count, log_n, 3, sc.u(), pl, 0); else lde_batch(in.u(), n, o.u(), N, count, log_n, 3, sc.u(), pl, 0); };
  run(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0); for (int i = 0; i < iters; i++) run(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / iters;
}

int qpgpu_test_bitrev(const uint64_t* in, int count, int log_n, uint64_t scale, uint64_t* out) {
  size_t n = size_t(1) << log_n;
  DevBuf a, b; HOOK_CHECK(a.alloc(count * n * 8)); HOOK_CHECK(b.alloc(count * n * 8));
  HOOK_CHECK(hipMemcpy(a.p, in, count * n * 8, hipMemcpyHostToDevice));
  bitrev_permute(a.u(), n, b.u(), n, count, log_n, scale, 0);
  HOOK_CHECK(hipGetLastError());
  HOOK_CHECK(hipMemcpy(out, b.p, count * n * 8, hipMemcpyDeviceToHost));
  return 0;
}

} // extern "C"
