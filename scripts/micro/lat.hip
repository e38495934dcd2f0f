// micro-benchmark: what does a dependent global read of data written by the previous kernel cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
__global__ void k_empty(double* p) { if (p == nullptr) p[0] = 1; }
// producer/consumer of per-block partials
__global__ void k_part(const double* __restrict__ in, double* __restrict__ out, int n, int mode) {
  __shared__ double sred[16];
  const int tid = threadIdx.x;
  double s = 0.0;
  if (mode & 1) for (int k = tid; k < n; k += 256) s += in[k];     // read previous kernel's partials
  if (mode & 2) {   // block reduce via LDS
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = s;
    __syncthreads();
    s = sred[0] + sred[1] + sred[2] + sred[3];
  }
  if (tid == 0) out[blockIdx.x] = s + 1.0;
}
// streaming: read 3 arrays write 1 (1 MB each)
__global__ void k_stream(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c, double* __restrict__ o, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = a[i] + 2.0 * b[i] - c[i];
}
// gather through an index table (1 level) then value
__global__ void k_gather(const int* __restrict__ tab, const double* __restrict__ a, double* __restrict__ o, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = a[tab[i]];
}
int main() {
  const int n = 127744, nblk = 499;
  double *a, *b, *c, *o, *p0, *p1; int* tab;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&o, n * 8));
  CK(hipMalloc(&p0, 4096 * 8)); CK(hipMalloc(&p1, 4096 * 8)); CK(hipMalloc(&tab, n * 4));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8)); CK(hipMemset(c, 0, n * 8)); CK(hipMemset(p0, 0, 4096 * 8)); CK(hipMemset(p1, 0, 4096 * 8));
  std::vector<int> t(n); for (int i = 0; i < n; ++i) t[i] = (i * 7919) % n; CK(hipMemcpy(tab, t.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto timeit = [&](const char* name, auto launch, int reps) {
    // capture a graph of `reps` launches, replay 20 times
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int r = 0; r < reps; ++r) launch(r);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int k = 0; k < 20; ++k) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (20.0 * reps);
    printf("%-44s %.2f us per kernel\n", name, us);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  };
  timeit("empty kernel (1 block)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, a); }, 200);
  timeit("empty kernel (499 blocks x 256)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(nblk), dim3(256), 0, st, a); }, 200);
  timeit("partials: write only", [&](int r) { hipLaunchKernelGGL(k_part, dim3(nblk), dim3(256), 0, st, (r & 1) ? p0 : p1, (r & 1) ? p1 : p0, nblk, 0); }, 200);
  timeit("partials: read prev + no reduce", [&](int r) { hipLaunchKernelGGL(k_part, dim3(nblk), dim3(256), 0, st, (r & 1) ? p0 : p1, (r & 1) ? p1 : p0, nblk, 1); }, 200);
  timeit("partials: read prev + shuffle reduce", [&](int r) { hipLaunchKernelGGL(k_part, dim3(nblk), dim3(256), 0, st, (r & 1) ? p0 : p1, (r & 1) ? p1 : p0, nblk, 3); }, 200);
  timeit("stream 3r1w 1MB arrays (ping-pong o<->a)", [&](int r) { hipLaunchKernelGGL(k_stream, dim3(nblk), dim3(256), 0, st, (r & 1) ? o : a, b, c, (r & 1) ? a : o, n); }, 200);
  timeit("gather a[tab[i]] (ping-pong)", [&](int r) { hipLaunchKernelGGL(k_gather, dim3(nblk), dim3(256), 0, st, tab, (r & 1) ? o : a, (r & 1) ? a : o, n); }, 200);
  return 0;
}
