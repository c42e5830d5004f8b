#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <thread>
__global__ void k_add(unsigned long long* p, const unsigned long long* src) { *p += *src; }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long* sig = nullptr; 
  hipError_t e = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
  printf("signal malloc: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) { e = hipHostMalloc((void**)&sig, 64, hipHostMallocMapped); printf("fallback hostmalloc: %s\n", hipGetErrorString(e)); }
  unsigned long long* host; hipHostMalloc((void**)&host, 64, hipHostMallocMapped); unsigned long long* hdev; hipHostGetDevicePointer((void**)&hdev, host, 0);
  unsigned long long* d; hipMalloc((void**)&d, 8); hipMemset(d, 0, 8);
  *sig = 0; host[0] = 5;
  double tot = 0; int ok = 1;
  for (int it = 1; it <= 20; it++) {
    e = hipStreamWaitValue64(s, sig, (uint64_t)it, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
    if (e != hipSuccess) { printf("waitvalue: %s\n", hipGetErrorString(e)); ok = 0; break; }
    hipLaunchKernelGGL(k_add, dim3(1), dim3(1), 0, s, d, hdev);
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    host[0] = it;                                 // "challenge" written by the host after the kernel was enqueued
    auto t0 = std::chrono::high_resolution_clock::now();
    __atomic_store_n(sig, (unsigned long long)it, __ATOMIC_RELEASE);
    hipStreamSynchronize(s);
    tot += std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
  }
  unsigned long long r = 0; hipMemcpy(&r, d, 8, hipMemcpyDeviceToHost);
  printf("ok=%d result=%llu (expect 210) avg release->done %.1f us\n", ok, r, tot / 20);
  // baseline: launch after the fact
  tot = 0;
  for (int it = 0; it < 20; it++) { auto t0 = std::chrono::high_resolution_clock::now(); hipLaunchKernelGGL(k_add, dim3(1), dim3(1), 0, s, d, hdev); hipStreamSynchronize(s); tot += std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count(); }
  printf("baseline launch->done %.1f us\n", tot / 20);
  return 0;
}
