import sys; sys.path.insert(0,"tests")
import torch; torch.cuda.init()
import harness as H, ctypes
L=H.qpgpu(); L.qpgpu_bench_permute.restype=ctypes.c_double; L.qpgpu_bench_permute.argtypes=[ctypes.c_size_t, ctypes.c_int]
print("G perm/s:", [(cnt, round(L.qpgpu_bench_permute(cnt, 10),3)) for cnt in (1<<16, 1<<17, 1<<18, 1<<20, 1<<22)])
