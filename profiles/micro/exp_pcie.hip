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
  return 0;
}
