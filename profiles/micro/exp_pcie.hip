// PCIe micro-benchmark: what an asynchronous copy between pinned host memory and HBM costs on this box -
// host time to ISSUE the call and time until it has landed, per direction and size.  (jg_step_node uploads
// ~16 MB of command rows and downloads ~10 MB of columns and rows per 100 k-partition tick.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (size_t mb : {1, 4, 16, 64}) {
    const size_t bytes = mb << 20;
    void *h = nullptr, *d = nullptr;
    hipHostMalloc(&h, bytes, hipHostMallocDefault);
    hipMalloc(&d, bytes);
    memset(h, 1, bytes);
    for (int dir = 0; dir < 2; dir++) {
      double issue = 0, total = 0;
      const int reps = 10;
      for (int i = 0; i < reps + 2; i++) {
        hipStreamSynchronize(st);
        const double t0 = us();
        if (dir == 0) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
        else hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
        const double t1 = us();
        hipStreamSynchronize(st);
        const double t2 = us();
        if (i >= 2) issue += t1 - t0, total += t2 - t0;
      }
      printf("%s %3zu MB: issue %7.1f us, landed after %7.1f us  = %5.1f GB/s\n", dir ? "D2H" : "H2D", mb, issue / reps, total / reps,
             bytes / (total / reps) / 1e3);
    }
    hipHostFree(h);
    hipFree(d);
  }
  // both directions at once, each on its own stream: what several event loops side by side can draw from the bus
  {
    hipStream_t s2;
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const size_t bytes = 64u << 20;
    void *h1 = nullptr, *h2 = nullptr, *d1 = nullptr, *d2 = nullptr;
    hipHostMalloc(&h1, bytes, hipHostMallocDefault), hipHostMalloc(&h2, bytes, hipHostMallocDefault);
    hipMalloc(&d1, bytes), hipMalloc(&d2, bytes);
    memset(h1, 1, bytes), memset(h2, 2, bytes);
    double total = 0;
    const int reps = 10;
    for (int i = 0; i < reps + 2; i++) {
      hipStreamSynchronize(st), hipStreamSynchronize(s2);
      const double t0 = us();
      hipMemcpyAsync(d1, h1, bytes, hipMemcpyHostToDevice, st);
      hipMemcpyAsync(h2, d2, bytes, hipMemcpyDeviceToHost, s2);
      hipStreamSynchronize(st), hipStreamSynchronize(s2);
      if (i >= 2) total += us() - t0;
    }
    printf("H2D + D2H at once, 64 MB each on two streams: both landed after %7.1f us  = %5.1f GB/s per direction, %5.1f GB/s over the bus\n",
           total / reps, bytes / (total / reps) / 1e3, 2 * bytes / (total / reps) / 1e3);
    // eight streams, 8 MB each, alternating directions (eight loops' copies in flight together)
    hipStream_t ss[8];
    for (auto& x : ss) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    total = 0;
    for (int i = 0; i < reps + 2; i++) {
      for (auto& x : ss) hipStreamSynchronize(x);
      const double t0 = us();
      for (int k = 0; k < 8; k++) {
        const size_t off = (size_t)(k / 2) * (8u << 20);
        if (k & 1) hipMemcpyAsync((char*)h2 + off, (char*)d2 + off, 8u << 20, hipMemcpyDeviceToHost, ss[k]);
        else hipMemcpyAsync((char*)d1 + off, (char*)h1 + off, 8u << 20, hipMemcpyHostToDevice, ss[k]);
      }
      for (auto& x : ss) hipStreamSynchronize(x);
      if (i >= 2) total += us() - t0;
    }
    printf("8 streams x 8 MB, alternating directions: all landed after %7.1f us  = %5.1f GB/s over the bus\n", total / reps,
           64.0 * (1 << 20) / (total / reps) / 1e3);
  }
  return 0;
}
