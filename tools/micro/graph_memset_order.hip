// Reproducer for DESIGN.md §6 "memset / memcpy graph nodes ran out of order with neighbouring kernel nodes": capture
//   K1: x[i] = tag      memset(x, 0)      K2: y[i] = x[i] + tag      memcpy(z <- y)      K3: w[i] += |z[i] - tag + x[i]|
// CHAINS times (tag = 1 … CHAINS) on one stream, replay the graph REPLAYS times, and count elements where w != 0 (the memset
// must land between K1 and K2, else y = 2·tag; the memcpy between K2 and K3, else z holds the previous chain's tag). Build: hipcc --offload-arch=gfx950 -O2 -o
// graph_memset_order graph_memset_order.hip ; run on the GPU box. Exit status 1 on any mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void k_set(int* x, int n, int tag) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = tag; }
__global__ void k_acc(int* y, const int* x, int n, int tag) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = x[i] + tag; }
__global__ void k_out(int* w, const int* z, const int* x, int n, int tag) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) w[i] += abs(z[i] - tag + x[i]); }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : (64 << 20), CHAINS = 24, REPLAYS = 50;   // 256 MB buffers: long memsets / kernels
  int *x, *y, *z, *w; hipStream_t s; hipGraph_t g; hipGraphExec_t ge;
  CK(hipMalloc(&x, 4L * n)); CK(hipMalloc(&y, 4L * n)); CK(hipMalloc(&z, 4L * n)); CK(hipMalloc(&w, 4L * n));
  CK(hipStreamCreate(&s)); CK(hipMemset(y, 0, 4L * n));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int c = 0; c < CHAINS; ++c) {
    k_set<<<(n + 255) / 256, 256, 0, s>>>(x, n, c + 1);
    CK(hipMemsetAsync(x, 0, 4L * n, s));
    k_acc<<<(n + 255) / 256, 256, 0, s>>>(y, x, n, c + 1);
    CK(hipMemcpyAsync(z, y, 4L * n, hipMemcpyDeviceToDevice, s));
    k_out<<<(n + 255) / 256, 256, 0, s>>>(w, z, x, n, c + 1);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  long bad = 0; int* h = (int*)malloc(4L * n);
  for (int r = 0; r < REPLAYS; ++r) {
    CK(hipMemset(w, 0, 4L * n)); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, w, 4L * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i += 97) bad += h[i] != 0;
  }
  printf("graph_memset_order: %d chains x %d replays, n = %d ints: %ld sampled mismatches\n", CHAINS, REPLAYS, n, bad);
  return bad ? 1 : 0;
}
