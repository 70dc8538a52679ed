#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f4* o, long n) { f4 v = {1.f,2.f,3.f,4.f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = v; }
__global__ void k_read(const f4* a, long n, f4* o) { f4 s = {0,0,0,0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i];
  if (s[0] == 123.f) o[0] = s; }
__global__ void k_copy(const f4* a, long n, f4* o) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i]; }
// 2 reads + 1 write (residual-add pattern)
__global__ void k_add(const f4* a, const f4* b, long n, f4* o) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] + b[i]; }
// strided half-sector stores: lane writes 16 B at 32 B stride, two instructions
__global__ void k_write_half(f4* o, long n) { f4 v = {1.f,2.f,3.f,4.f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n / 2; i += (long)gridDim.x * blockDim.x) { o[2 * i] = v; o[2 * i + 1] = v; } }
int main() {
  const long n = (1L << 30) / 16 * 4;  // 4 GiB
  f4 *a, *b, *o; CK(hipMalloc((void**)&a, n * 16)); CK(hipMalloc((void**)&b, n * 16)); CK(hipMalloc((void**)&o, n * 16));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 1, n * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192, 65536}) {
    for (int which = 0; which < 5; ++which) {
      float best = 1e9;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, 0);
        if (which == 0) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, o, n);
        if (which == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, o);
        if (which == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, n, o);
        if (which == 3) hipLaunchKernelGGL(k_add, dim3(grid), dim3(256), 0, 0, a, b, n, o);
        if (which == 4) hipLaunchKernelGGL(k_write_half, dim3(grid), dim3(256), 0, 0, o, n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double bytes = n * 16.0 * (which == 2 ? 2 : which == 3 ? 3 : 1);
      const char* nm[] = {"write", "read", "copy", "add(2r1w)", "write_half"};
      printf("grid %d %s: %.3f ms %.2f TB/s\n", grid, nm[which], best, bytes / best / 1e9);
    }
  }
  return 0;
}
