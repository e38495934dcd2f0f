// Operand layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks), found by experiment: A = 1 in lane la only, B = 1 in
// lane lb only -> which lane of D becomes 1?   hipcc -O3 --offload-arch=gfx950 -o /tmp/l44 scripts/mfma_f64_4x4_layout.hip && /tmp/l44
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) out[la * 64 + lb] = lane;
    }
}
int main() {
  int* out; hipMalloc(&out, 4096 * sizeof(int)); hipMemset(out, 0xff, 4096 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out); hipDeviceSynchronize();
  static int h[4096]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int n = 0;
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) { ++n; if (la < 20 || la % 16 == 0) printf("A lane %2d x B lane %2d -> D lane %2d\n", la, lb, h[la * 64 + lb]); }
  printf("%d nonzero pairs\n", n);
  // hypothesis check: A lane = i + 4 k + 16 blk, B lane = j + 4 k + 16 blk, D lane = j + 4 i + 16 blk  (or i + 4 j)
  int ok1 = 1, ok2 = 1;
  for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int kk = 0; kk < 4; ++kk) {
    const int la = i + 4 * kk + 16 * blk, lb = j + 4 * kk + 16 * blk, d = h[la * 64 + lb];
    if (d != j + 4 * i + 16 * blk) ok1 = 0;
    if (d != i + 4 * j + 16 * blk) ok2 = 0;
  }
  printf("hypothesis D lane = j + 4 i + 16 blk: %d;  D lane = i + 4 j + 16 blk: %d\n", ok1, ok2);
  return 0;
}
