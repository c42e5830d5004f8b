// Micro-benchmark: issue rate of the integer VALU instructions the Goldilocks kernels lean on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; prints ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITERS 2000

#define KERNEL(name, body) \
__global__ void name(uint32_t* out, uint32_t a0, uint32_t b0) { \
  uint32_t a = a0 + threadIdx.x, b = b0 | 1, c = a ^ b, d = a + 7; \
  uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d, z = x ^ y, w = x + y; \
  for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int r = 0; r < REP / 4; r++) { body } } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)x ^ (uint32_t)y ^ (uint32_t)z ^ (uint32_t)w ^ (uint32_t)(x >> 32) ^ (uint32_t)(y >> 32) ^ (uint32_t)(z>>32) ^ (uint32_t)(w>>32); \
}

// four independent chains per body to avoid dependency stalls
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a0));)
KERNEL(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b0));)
KERNEL(k_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b0));)
KERNEL(k_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b0));)
KERNEL(k_mul_hi_u32_u24, asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b0));)
KERNEL(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a0), "v"(b0) : "vcc");)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(x));)
KERNEL(k_add_co_addc, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a0) : "vcc");)
KERNEL(k_cmp_cndmask, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(a0) : "vcc");)
KERNEL(k_cmp_u64, asm volatile("v_cmp_lt_u64 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u64 vcc, %2, %4\n v_cndmask_b32 %3, %3, %5, vcc" : "+v"(x), "+v"(a), "+v"(z), "+v"(c) : "v"(y), "v"(a0) : "vcc");)
KERNEL(k_lshlrev_b64, asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));)
KERNEL(k_mul_u32_u24, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b0));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(b0));)

template <class K> void run(const char* name, K k, uint32_t* d_out, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD of a CU; blocks = CUs * wps
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 6789u);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 6789u); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)ITERS * REP;
  double waves_per_simd_total = waves_per_simd;       // each SIMD runs wps waves
  double ns_per_instr = ms * 1e6 / (instr_per_wave * waves_per_simd_total);
  printf("%-18s wps=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4);
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int wps : {1, 4, 8}) {
    run("v_add_u32", k_add_u32, d, wps); run("v_add3_u32", k_add3, d, wps); run("v_mul_lo_u32", k_mul_lo_u32, d, wps); run("v_mul_hi_u32", k_mul_hi_u32, d, wps);
    run("v_mul_u32_u24", k_mul_u32_u24, d, wps); run("v_mad_u32_u24", k_mad_u32_u24, d, wps); run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, d, wps);
    run("v_mad_u64_u32", k_mad_u64_u32, d, wps); run("v_lshl_add_u64", k_lshl_add_u64, d, wps); run("add_co+addc", k_add_co_addc, d, wps);
    run("cmp_u32+cndmask", k_cmp_cndmask, d, wps); run("cmp_u64+cndmask", k_cmp_u64, d, wps); run("v_lshlrev_b64", k_lshlrev_b64, d, wps);
    printf("\n");
  }
  return 0;
}
