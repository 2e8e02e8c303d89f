// Micro-experiment for the next round (DESIGN.md "What comes next"): does the follower half's time follow
// the NUMBER OF ADDRESS STREAMS of its state?  Same bytes per group, two layouts:
//   columns : ten state columns (4 x u64, 6 x u32 = 56 B) read, three written (head u64, commit u64, flags u32)
//   record  : one 64-byte record per group (four 16-byte loads per lane), 24 B of it written back
// plus the same 24 B of inbox (16 + 8) and 8 B of outbox in both.  hipcc --offload-arch=gfx950 -O3 exp_layout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));
struct Cols { u64 *term, *head, *commit, *etime; uint32_t *flags, *voted, *leader, *queued, *draws, *eto; };
__global__ __launch_bounds__(256) void k_columns(Cols c, const v2* beat, const u64* ae, u64* answer, uint32_t G) {
  for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
    const uint32_t f = c.flags[g], vf = c.voted[g], ld = c.leader[g], q = c.queued[g], dr = c.draws[g], eto = c.eto[g];
    const u64 term = c.term[g], head = c.head[g], commit = c.commit[g], et = c.etime[g];
    const v2 b = beat[g];
    const u64 a = __builtin_nontemporal_load(&ae[g]);
    const u64 nh = (a >> 8) + (a & 0xff) + (head & 1) + (vf == ld) + q + dr + (et > eto);
    const u64 nc = b.y <= nh && b.y > commit ? b.y : commit;
    c.head[g] = nh + (term == b.x);
    c.commit[g] = nc;
    c.flags[g] = f ^ 1u;
    answer[g] = nh << 8 | 1;
  }
}
struct Rec { u64 term, head, commit, etime; uint32_t flags, voted, leader, queued, draws, eto, pad0, pad1; };
__global__ __launch_bounds__(256) void k_record(Rec* recs, const v2* beat, const u64* ae, u64* answer, uint32_t G) {
  for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
    const v2* p = (const v2*)&recs[g];
    const v2 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
    const v2 b = beat[g];
    const u64 a = __builtin_nontemporal_load(&ae[g]);
    const u64 term = r0.x, head = r0.y, commit = r1.x, et = r1.y;
    const uint32_t f = (uint32_t)r2.x, vf = (uint32_t)(r2.x >> 32), ld = (uint32_t)r2.y, q = (uint32_t)(r2.y >> 32);
    const uint32_t dr = (uint32_t)r3.x, eto = (uint32_t)(r3.x >> 32);
    const u64 nh = (a >> 8) + (a & 0xff) + (head & 1) + (vf == ld) + q + dr + (et > eto);
    const u64 nc = b.y <= nh && b.y > commit ? b.y : commit;
    v2* w = (v2*)&recs[g];
    w[0] = v2{term, nh + (term == b.x)};               // 16 B: term (unchanged), head
    ((u64*)&recs[g])[2] = nc;                           // 8 B: commit
    ((uint32_t*)&recs[g])[8] = f ^ 1u;                  // 4 B: flags
    answer[g] = nh << 8 | 1;
  }
}
int main(int argc, char** argv) {
  const uint32_t G = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1000000u;
  const int K = 60;
  Cols c{};
  CHECK(hipMalloc(&c.term, 8ull * G)); CHECK(hipMalloc(&c.head, 8ull * G)); CHECK(hipMalloc(&c.commit, 8ull * G)); CHECK(hipMalloc(&c.etime, 8ull * G));
  CHECK(hipMalloc(&c.flags, 4ull * G)); CHECK(hipMalloc(&c.voted, 4ull * G)); CHECK(hipMalloc(&c.leader, 4ull * G));
  CHECK(hipMalloc(&c.queued, 4ull * G)); CHECK(hipMalloc(&c.draws, 4ull * G)); CHECK(hipMalloc(&c.eto, 4ull * G));
  Rec* recs; v2* beat; u64 *ae, *answer;
  CHECK(hipMalloc(&recs, sizeof(Rec) * (size_t)G)); CHECK(hipMalloc(&beat, 16ull * G)); CHECK(hipMalloc(&ae, 8ull * G)); CHECK(hipMalloc(&answer, 8ull * G));
  for (void* p : {(void*)c.term, (void*)c.head, (void*)c.commit, (void*)c.etime}) CHECK(hipMemset(p, 0, 8ull * G));
  for (void* p : {(void*)c.flags, (void*)c.voted, (void*)c.leader, (void*)c.queued, (void*)c.draws, (void*)c.eto}) CHECK(hipMemset(p, 0, 4ull * G));
  CHECK(hipMemset(recs, 0, sizeof(Rec) * (size_t)G)); CHECK(hipMemset(beat, 0, 16ull * G)); CHECK(hipMemset(ae, 0, 8ull * G));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const uint32_t grid = (G + 255) / 256;
  for (int which = 0; which < 2; which++) {
    for (int rep = 0; rep < 2; rep++) {  // first pass warms up
      CHECK(hipEventRecord(e0));
      for (int k = 0; k < K; k++) {
        if (which == 0) hipLaunchKernelGGL(k_columns, dim3(grid), dim3(256), 0, 0, c, (const v2*)beat, (const u64*)ae, answer, G);
        else hipLaunchKernelGGL(k_record, dim3(grid), dim3(256), 0, 0, recs, (const v2*)beat, (const u64*)ae, answer, G);
      }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) {
        const double bytes = which == 0 ? (56 + 24 + 20 + 8) : (64 + 24 + 28 + 8);  // read state + inbox, written state + outbox
        std::printf("%s G=%u: %.2f us per launch, %.0f B per group -> %.2f TB/s\n", which ? "record " : "columns", G, ms * 1000 / K, bytes,
                    bytes * G / (ms / K * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
