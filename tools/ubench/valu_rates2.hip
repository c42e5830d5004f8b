// Micro-benchmark #2: issue cost of the gfx950 VALU instructions a Poseidon/Goldilocks kernel could be built from.
// Every kernel runs ITERS x REP instances of ONE instruction over eight independent register chains (no dependency stalls);
// cost is reported in SHADER CYCLES (s_memtime inside the kernel, so DVFS does not distort it) per wave-instruction per SIMD,
// at 1, 2, 4 and 8 resident waves per SIMD, next to the wall-clock figure.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define ITERS 1000
#define REP 8          // asm blocks per loop body; each block holds 8 instructions

#define DEFK(name, asmtext, ...) \
__global__ void name(uint32_t* out, unsigned long long* cyc, uint32_t a0, uint32_t b0) { \
  uint32_t r0 = a0 + threadIdx.x, r1 = b0 | 1, r2 = r0 ^ b0, r3 = r0 + 7, r4 = r1 * 3, r5 = r0 * 5, r6 = r2 + 11, r7 = r3 ^ 0x55; \
  uint64_t q0 = ((uint64_t)r0 << 32) | r1, q1 = ((uint64_t)r2 << 32) | r3, q2 = q0 ^ q1, q3 = q0 + q1, q4 = q0 * 3, q5 = q1 * 5, q6 = q2 + 9, q7 = q3 ^ 0x77; \
  uint32_t s = b0 | 5, t = a0 | 3; \
  unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int r = 0; r < REP; r++) { \
    asm volatile(asmtext : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
                 "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(s), "v"(t) : __VA_ARGS__); } } \
  unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32); \
  if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0; \
}
// operand map: %0-%7 = 32-bit chains, %8-%15 = 64-bit chains, %16 = s, %17 = t
#define I8(op, fmt) op " " fmt(0) "\n" op " " fmt(1) "\n" op " " fmt(2) "\n" op " " fmt(3) "\n" op " " fmt(4) "\n" op " " fmt(5) "\n" op " " fmt(6) "\n" op " " fmt(7)
#define F2(i) "%" #i ", %" #i ", %16"                 // d = d op s
#define F2s(i) "%" #i ", %16, %" #i                   // d = s op d   (shift amount first)
#define F3(i) "%" #i ", %" #i ", %16, %17"            // d = f(d, s, t)
#define F3acc(i) "%" #i ", %16, %17, %" #i            // d = s*t + d
#define F1(i) "%" #i ", %" #i

DEFK(k_add_u32, I8("v_add_u32", F2), "memory")
DEFK(k_sub_u32, I8("v_sub_u32", F2), "memory")
DEFK(k_and_b32, I8("v_and_b32", F2), "memory")
DEFK(k_xor_b32, I8("v_xor_b32", F2), "memory")
DEFK(k_lshlrev_b32, I8("v_lshlrev_b32", F2s), "memory")
DEFK(k_lshrrev_b32, I8("v_lshrrev_b32", F2s), "memory")
DEFK(k_mov_b32, I8("v_mov_b32", F1), "memory")
DEFK(k_lshl_or_b32, I8("v_lshl_or_b32", F3), "memory")
DEFK(k_lshl_add_u32, I8("v_lshl_add_u32", F3), "memory")
DEFK(k_add_lshl_u32, I8("v_add_lshl_u32", F3), "memory")
DEFK(k_and_or_b32, I8("v_and_or_b32", F3), "memory")
DEFK(k_or3_b32, I8("v_or3_b32", F3), "memory")
DEFK(k_add3_u32, I8("v_add3_u32", F3), "memory")
DEFK(k_xad_u32, I8("v_xad_u32", F3), "memory")
DEFK(k_bfe_u32, I8("v_bfe_u32", F3), "memory")
DEFK(k_bfi_b32, I8("v_bfi_b32", F3), "memory")
DEFK(k_alignbit_b32, I8("v_alignbit_b32", F3), "memory")
DEFK(k_perm_b32, I8("v_perm_b32", F3), "memory")
DEFK(k_mul_u32_u24, I8("v_mul_u32_u24", F2), "memory")
DEFK(k_mul_hi_u32_u24, I8("v_mul_hi_u32_u24", F2), "memory")
DEFK(k_mad_u32_u24, I8("v_mad_u32_u24", F3acc), "memory")
DEFK(k_mad_i32_i24, I8("v_mad_i32_i24", F3acc), "memory")
DEFK(k_mul_lo_u32, I8("v_mul_lo_u32", F2), "memory")
DEFK(k_mul_hi_u32, I8("v_mul_hi_u32", F2), "memory")
DEFK(k_mad_u32_u16, I8("v_mad_u32_u16", F3acc), "memory")
DEFK(k_dot2_u32_u16, I8("v_dot2_u32_u16", F3acc), "memory")
DEFK(k_dot4_u32_u8, I8("v_dot4_u32_u8", F3acc), "memory")
DEFK(k_dot8_u32_u4, I8("v_dot8_u32_u4", F3acc), "memory")
DEFK(k_dot2_i32_i16, I8("v_dot2_i32_i16", F3acc), "memory")
DEFK(k_dot4_i32_i8, I8("v_dot4_i32_i8", F3acc), "memory")
DEFK(k_pk_mad_u16, I8("v_pk_mad_u16", F3acc), "memory")
DEFK(k_pk_mul_lo_u16, I8("v_pk_mul_lo_u16", F2), "memory")
DEFK(k_pk_add_u16, I8("v_pk_add_u16", F2), "memory")
DEFK(k_fma_f32, I8("v_fma_f32", F3acc), "memory")
DEFK(k_sad_u32, I8("v_sad_u32", F3acc), "memory")
DEFK(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %16, vcc\n v_cndmask_b32 %1, %1, %16, vcc\n v_cndmask_b32 %2, %2, %16, vcc\n v_cndmask_b32 %3, %3, %16, vcc\n"
                    "v_cndmask_b32 %4, %4, %16, vcc\n v_cndmask_b32 %5, %5, %16, vcc\n v_cndmask_b32 %6, %6, %16, vcc\n v_cndmask_b32 %7, %7, %16, vcc", "memory")

#define G2(i) "%" #i ", %" #i ", %" #i                // 64-bit d = d op d'
#define Q(i) "%" #i
DEFK(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n v_mad_u64_u32 %9, vcc, %16, %17, %9\n v_mad_u64_u32 %10, vcc, %16, %17, %10\n v_mad_u64_u32 %11, vcc, %16, %17, %11\n"
                    "v_mad_u64_u32 %12, vcc, %16, %17, %12\n v_mad_u64_u32 %13, vcc, %16, %17, %13\n v_mad_u64_u32 %14, vcc, %16, %17, %14\n v_mad_u64_u32 %15, vcc, %16, %17, %15", "vcc")
DEFK(k_mad_u64_u32_sgprco, "v_mad_u64_u32 %8, s[20:21], %16, %17, %8\n v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n v_mad_u64_u32 %10, s[20:21], %16, %17, %10\n v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n"
                    "v_mad_u64_u32 %12, s[20:21], %16, %17, %12\n v_mad_u64_u32 %13, s[22:23], %16, %17, %13\n v_mad_u64_u32 %14, s[20:21], %16, %17, %14\n v_mad_u64_u32 %15, s[22:23], %16, %17, %15", "s20", "s21", "s22", "s23")
DEFK(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %12\n"
                     "v_lshl_add_u64 %12, %12, 0, %13\n v_lshl_add_u64 %13, %13, 0, %14\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %15, %15, 0, %8", "memory")
DEFK(k_lshlrev_b64, "v_lshlrev_b64 %8, 1, %8\n v_lshlrev_b64 %9, 1, %9\n v_lshlrev_b64 %10, 1, %10\n v_lshlrev_b64 %11, 1, %11\n"
                    "v_lshlrev_b64 %12, 1, %12\n v_lshlrev_b64 %13, 1, %13\n v_lshlrev_b64 %14, 1, %14\n v_lshlrev_b64 %15, 1, %15", "memory")
DEFK(k_add_co_vcc, "v_add_co_u32 %0, vcc, %0, %16\n v_add_co_u32 %1, vcc, %1, %16\n v_add_co_u32 %2, vcc, %2, %16\n v_add_co_u32 %3, vcc, %3, %16\n"
                   "v_add_co_u32 %4, vcc, %4, %16\n v_add_co_u32 %5, vcc, %5, %16\n v_add_co_u32 %6, vcc, %6, %16\n v_add_co_u32 %7, vcc, %7, %16", "vcc")
DEFK(k_addc_co_vcc, "v_addc_co_u32 %0, vcc, %0, %16, vcc\n v_addc_co_u32 %1, vcc, %1, %16, vcc\n v_addc_co_u32 %2, vcc, %2, %16, vcc\n v_addc_co_u32 %3, vcc, %3, %16, vcc\n"
                    "v_addc_co_u32 %4, vcc, %4, %16, vcc\n v_addc_co_u32 %5, vcc, %5, %16, vcc\n v_addc_co_u32 %6, vcc, %6, %16, vcc\n v_addc_co_u32 %7, vcc, %7, %16, vcc", "vcc")
DEFK(k_add_co_sgpr, "v_add_co_u32_e64 %0, s[20:21], %0, %16\n v_add_co_u32_e64 %1, s[22:23], %1, %16\n v_add_co_u32_e64 %2, s[20:21], %2, %16\n v_add_co_u32_e64 %3, s[22:23], %3, %16\n"
                    "v_add_co_u32_e64 %4, s[20:21], %4, %16\n v_add_co_u32_e64 %5, s[22:23], %5, %16\n v_add_co_u32_e64 %6, s[20:21], %6, %16\n v_add_co_u32_e64 %7, s[22:23], %7, %16", "s20", "s21", "s22", "s23")
DEFK(k_addc_co_sgpr, "v_addc_co_u32_e64 %0, s[20:21], %0, %16, s[20:21]\n v_addc_co_u32_e64 %1, s[22:23], %1, %16, s[22:23]\n v_addc_co_u32_e64 %2, s[20:21], %2, %16, s[20:21]\n v_addc_co_u32_e64 %3, s[22:23], %3, %16, s[22:23]\n"
                     "v_addc_co_u32_e64 %4, s[20:21], %4, %16, s[20:21]\n v_addc_co_u32_e64 %5, s[22:23], %5, %16, s[22:23]\n v_addc_co_u32_e64 %6, s[20:21], %6, %16, s[20:21]\n v_addc_co_u32_e64 %7, s[22:23], %7, %16, s[22:23]", "s20", "s21", "s22", "s23")
DEFK(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %16, s[20:21]\n v_cndmask_b32_e64 %1, %1, %16, s[20:21]\n v_cndmask_b32_e64 %2, %2, %16, s[20:21]\n v_cndmask_b32_e64 %3, %3, %16, s[20:21]\n"
                     "v_cndmask_b32_e64 %4, %4, %16, s[20:21]\n v_cndmask_b32_e64 %5, %5, %16, s[20:21]\n v_cndmask_b32_e64 %6, %6, %16, s[20:21]\n v_cndmask_b32_e64 %7, %7, %16, s[20:21]", "s20", "s21")
DEFK(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %16\n v_cmp_lt_u32 vcc, %1, %16\n v_cmp_lt_u32 vcc, %2, %16\n v_cmp_lt_u32 vcc, %3, %16\n"
                   "v_cmp_lt_u32 vcc, %4, %16\n v_cmp_lt_u32 vcc, %5, %16\n v_cmp_lt_u32 vcc, %6, %16\n v_cmp_lt_u32 vcc, %7, %16", "vcc")
DEFK(k_fma_f64, "v_fma_f64 %8, %8, %9, %8\n v_fma_f64 %9, %9, %10, %9\n v_fma_f64 %10, %10, %11, %10\n v_fma_f64 %11, %11, %12, %11\n"
                "v_fma_f64 %12, %12, %13, %12\n v_fma_f64 %13, %13, %14, %13\n v_fma_f64 %14, %14, %15, %14\n v_fma_f64 %15, %15, %8, %15", "memory")
DEFK(k_pk_fma_f32, "v_pk_fma_f32 %8, %8, %9, %8\n v_pk_fma_f32 %9, %9, %10, %9\n v_pk_fma_f32 %10, %10, %11, %10\n v_pk_fma_f32 %11, %11, %12, %11\n"
                   "v_pk_fma_f32 %12, %12, %13, %12\n v_pk_fma_f32 %13, %13, %14, %13\n v_pk_fma_f32 %14, %14, %15, %14\n v_pk_fma_f32 %15, %15, %8, %15", "memory")
DEFK(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
// a 2:1 mix: two v_mad_u64_u32 and one v_add_u32 / v_perm_b32 — do cheap 32-bit ops hide beside the multiplier?
DEFK(k_mix_mad_add, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n v_add_u32 %0, %0, %16\n v_mad_u64_u32 %9, vcc, %16, %17, %9\n v_add_u32 %1, %1, %16\n"
                    "v_mad_u64_u32 %10, vcc, %16, %17, %10\n v_add_u32 %2, %2, %16\n v_mad_u64_u32 %11, vcc, %16, %17, %11\n v_add_u32 %3, %3, %16", "vcc")


// ---- does the VALU-writes-SGPR -> VALU-reads-it hazard nop cost issue time? Each pair is a v_mad_u64_u32 with an SGPR carry-out and a
// v_addc_co_u32 consuming it. A: consumer right behind its producer with the s_nop the compiler inserts; B: two pairs interleaved so that an
// independent instruction sits in the gap (no nop needed); C: as A without the nop (wrong carries, timing reference only).
DEFK(k_haz_nop, "v_mad_u64_u32 %8, s[20:21], %16, %17, %8\n s_nop 0\n v_addc_co_u32_e64 %0, s[20:21], %0, %16, s[20:21]\n v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n s_nop 0\n v_addc_co_u32_e64 %1, s[22:23], %1, %16, s[22:23]\n"
                "v_mad_u64_u32 %10, s[20:21], %16, %17, %10\n s_nop 0\n v_addc_co_u32_e64 %2, s[20:21], %2, %16, s[20:21]\n v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n s_nop 0\n v_addc_co_u32_e64 %3, s[22:23], %3, %16, s[22:23]", "s20", "s21", "s22", "s23")
DEFK(k_haz_interleaved, "v_mad_u64_u32 %8, s[20:21], %16, %17, %8\n v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n v_addc_co_u32_e64 %0, s[20:21], %0, %16, s[20:21]\n v_addc_co_u32_e64 %1, s[22:23], %1, %16, s[22:23]\n"
                        "v_mad_u64_u32 %10, s[20:21], %16, %17, %10\n v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n v_addc_co_u32_e64 %2, s[20:21], %2, %16, s[20:21]\n v_addc_co_u32_e64 %3, s[22:23], %3, %16, s[22:23]", "s20", "s21", "s22", "s23")
DEFK(k_haz_nonop, "v_mad_u64_u32 %8, s[20:21], %16, %17, %8\n v_addc_co_u32_e64 %0, s[20:21], %0, %16, s[20:21]\n v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n v_addc_co_u32_e64 %1, s[22:23], %1, %16, s[22:23]\n"
                  "v_mad_u64_u32 %10, s[20:21], %16, %17, %10\n v_addc_co_u32_e64 %2, s[20:21], %2, %16, s[20:21]\n v_mad_u64_u32 %11, s[22:23], %16, %17, %11\n v_addc_co_u32_e64 %3, s[22:23], %3, %16, s[22:23]", "s20", "s21", "s22", "s23")

struct Entry { const char* name; void (*k)(uint32_t*, unsigned long long*, uint32_t, uint32_t); };
#define E(n) {#n, n}

int main() {
  std::vector<Entry> ks = {E(k_add_u32), E(k_sub_u32), E(k_and_b32), E(k_xor_b32), E(k_lshlrev_b32), E(k_lshrrev_b32), E(k_mov_b32), E(k_lshl_or_b32),
    E(k_lshl_add_u32), E(k_add_lshl_u32), E(k_and_or_b32), E(k_or3_b32), E(k_add3_u32), E(k_xad_u32), E(k_bfe_u32), E(k_bfi_b32), E(k_alignbit_b32), E(k_perm_b32),
    E(k_mul_u32_u24), E(k_mul_hi_u32_u24), E(k_mad_u32_u24), E(k_mad_i32_i24), E(k_mul_lo_u32), E(k_mul_hi_u32), E(k_mad_u32_u16), E(k_dot2_u32_u16), E(k_dot4_u32_u8),
    E(k_dot8_u32_u4), E(k_dot2_i32_i16), E(k_dot4_i32_i8), E(k_pk_mad_u16), E(k_pk_mul_lo_u16), E(k_pk_add_u16), E(k_fma_f32), E(k_sad_u32), E(k_cndmask_vcc),
    E(k_mad_u64_u32), E(k_mad_u64_u32_sgprco), E(k_lshl_add_u64), E(k_lshlrev_b64), E(k_add_co_vcc), E(k_addc_co_vcc), E(k_add_co_sgpr), E(k_addc_co_sgpr), E(k_cndmask_sgpr),
    E(k_cmp_lt_u32), E(k_fma_f64), E(k_pk_fma_f32), E(k_mov_dpp), E(k_mix_mad_add), E(k_haz_nop), E(k_haz_interleaved), E(k_haz_nonop)};
  uint32_t* d; unsigned long long* dc;
  const int max_blocks = 256 * 8;
  hipMalloc(&d, (size_t)max_blocks * 256 * 4); hipMalloc(&dc, (size_t)max_blocks * 4 * 8);
  std::vector<unsigned long long> hc(max_blocks * 4);
  printf("%-22s", "instruction");
  for (int wps : {1, 2, 4, 8}) printf("  wps=%d cyc  (ns)  ", wps);
  printf("   [shader cycles / wall ns per wave-instruction per SIMD; 8 instr per asm block]\n");
  for (auto& e : ks) {
    printf("%-22s", e.name + 2);
    for (int wps : {1, 2, 4, 8}) {
      int blocks = 256 * wps;      // 256-thread blocks: 4 waves = one per SIMD of a CU
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, dc, 12345u, 6789u);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, dc, 12345u, 6789u); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc.data(), dc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> v(hc.begin(), hc.begin() + blocks * 4); std::sort(v.begin(), v.end());
      const double instr = (double)ITERS * REP * 8;
      // s_memtime counts at a fixed 100 MHz-derived rate on some parts; report it raw per instruction, times the resident waves
      double cyc = (double)v[v.size() / 2] / instr / wps;
      double ns = ms * 1e6 / instr / wps;
      printf("  %8.2f (%5.2f)  ", cyc, ns);
      hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n");
  }
  return 0;
}
