// Throughput of v_mfma_f64_16x16x4_f64 against v_fma_f64 on this GPU (what bounds the wavefront-per-element
// Schwarz kernel next).   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_f64_rate scripts/mfma_f64_rate.hip && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double* out, int n) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void k_mfma4(double* out, int n) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int n) {
  double a[8];
  for (int q = 0; q < 8; ++q) a[q] = q;
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1e-9 * threadIdx.x;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = __builtin_fma(a[q], x, y);
  }
  double s = 0; for (int q = 0; q < 8; ++q) s += a[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = ncu * 8, n = 20000;
  double* out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fl_m = (double)blocks * 4 /*waves*/ * n * 4.0 * 2048.0;
  printf("v_mfma_f64_16x16x4_f64: %.1f TFLOP/s (%d CUs, %.2f ms)\n", fl_m / ms / 1e9, ncu, ms);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k_mfma4, dim3(blocks), dim3(256), 0, 0, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fl_4 = (double)blocks * 4 * n * 4.0 * 512.0;       // four 4x4x4 blocks per instruction
  printf("v_mfma_f64_4x4x4_4b_f64: %.1f TFLOP/s (%.2f ms)\n", fl_4 / ms / 1e9, ms);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double fl_v = (double)blocks * 256 * n * 8.0 * 2.0;
  printf("v_fma_f64:              %.1f TFLOP/s (%.2f ms)\n", fl_v / ms / 1e9, ms);
  return 0;
}
