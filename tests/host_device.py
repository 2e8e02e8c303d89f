"""The ENGINE's own translation unit (josefine_amd/csrc/josefine_gpu.hip: the C ABI, its host code, every kernel) compiled
for the HOST against an emulated HIP runtime (CPU): tests/host_workgroups.py's stand-in for the kernel language - a
workgroup is 256 cooperative fibers - plus the forty runtime calls the engine makes (allocations are malloc filled with a
pattern, a stream executes in order at the call, events are moments that have passed, a captured graph is the list of
what was launched while capturing).

TEST INFRASTRUCTURE, and nothing else.  The library is built at test time into a temporary directory; the package can only
be pointed at it from outside, by JOSEFINE_GPU_LIB (the hook for an A/B of two builds), which tests/test_host_device.py
sets for a CHILD process - nothing under josefine_amd/ builds it, looks for it or falls back to it, and the product still
fails loudly without the gfx950 library.  What it is for: host-side wiring that no GPU-minute was left for (the routed
round under JG_ROUTE_VOTE_WORDS=1: job tables, step numbers, the mail's double buffering, the repeated delivering pass)
runs - through the C ABI, by the GPU suite's own tests - before a device sees it.  It says nothing about the memory
system, about races between workgroups, or about time.  (Host threads - the pipelined drain's worker, the router's pool -
are serialised on the one emulated device: every call happens at the call, one at a time.)

One thing it only approximates: reconvergence.  The hardware brings a wave's lanes back together behind a divergent branch
(the compiler's immediate post-dominator); here the lanes that skipped the branch have run ahead to their next shuffle or
ballot when the others stop inside it.  The scheduler lets the side go first that is still inside: the rendezvous from which
lanes have been SEEN to come to the other one (and not the other way round), else the one that whole waves rarely reach,
else the higher address - cold code is laid out last (tests/host_workgroups.py).  Code that works under any active mask - the deferral bitmaps, the
queues' ballots, the staging - is indifferent; a wave REDUCTION behind a divergent region (the decision counters:
jg_wave_count adds from lane 0) needs the right choice, and the first time a pair of rendezvous meets there is nothing to go
by: the leader tick's counters were off by 20 in 316 930 in the FIRST tick of a fresh process
(tests/test_dense_node.py::test_dense_leader_tick_parity; the same forty ticks again in that process: exact; exact in every
other test run here).  So
tests that run here compare state, rows, faults and applies, and leave the exact
decision counters to the device (JG_EMULATED_DEVICE=1 tells them)."""
import os
import subprocess
import tempfile

from host_compiled import CSRC, ROOT
from host_workgroups import WG_SHIM

RUNTIME = r'''
// ---- the runtime calls of josefine_gpu.hip (tests/host_device.py) -----------------------------------------------------
#include <string>
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct wgStreamT { int id; }* hipStream_t;
typedef struct wgEventT { int id; }* hipEvent_t;
struct wgGraphT { std::vector<std::function<void()>> ops; };
typedef wgGraphT* hipGraph_t;
typedef wgGraphT* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
#define hipHostMallocDefault 0u
#define hipEventDisableTiming 2u
#define hipStreamNonBlocking 1u
#include <mutex>
namespace wg {
static wgGraphT* capturing = nullptr;  // (one stream captures at a time: the engine's closed loop)
// the engine's host threads (the pipelined drain's worker, an event loop's tasks) share ONE device here: whatever they
// ask of it happens at the call, one call at a time - which is an order the streams and events of the real runtime allow
static std::recursive_mutex mu;
template <class F> static inline void op(F&& f) {
  std::lock_guard<std::recursive_mutex> lock(mu);
  if (capturing) capturing->ops.push_back(std::function<void()>(f));
  else f();
}
}  // namespace wg
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated runtime: error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t wg_alloc(void** p, size_t n) {
  *p = std::malloc(n ? n : 1);
  if (!*p) return hipErrorInvalidValue;
  std::memset(*p, 0xA5, n);  // (device memory is not zero: whoever relies on it finds out here)
  return hipSuccess;
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return wg_alloc((void**)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return wg_alloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  std::lock_guard<std::recursive_mutex> lock(wg::mu);
  std::memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
  wg::op([=] { std::memmove(d, s, n); });
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) {
  wg::op([=] { std::memset(d, v, n); });
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; r++) std::memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new wgStreamT{0}; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new wgStreamT{0}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new wgEventT{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new wgEventT{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1e-3f; return hipSuccess; }  // (no clock here)
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
  if (wg::capturing) return hipErrorInvalidValue;
  wg::capturing = new wgGraphT;
  return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = wg::capturing; wg::capturing = nullptr; return *g ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, void*, void*, size_t) { *x = new wgGraphT(*g); return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t) {
  std::lock_guard<std::recursive_mutex> lock(wg::mu);
  for (auto& f : x->ops) f();
  return hipSuccess;
}
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t x) { delete x; return hipSuccess; }
// a launch: every workgroup, one after the other, now (or when the captured graph is replayed)
template <class... KA, class... A>
static inline void wg_launch(void (*k)(KA...), dim3 grid, dim3 block, size_t /*dynamic LDS: none in this engine*/, hipStream_t, A... a) {
  wg::op([=] { wg::launch(grid, block.x, [=] { k(a...); }); });
}
#define hipLaunchKernelGGL(...) wg_launch(__VA_ARGS__)
// what this build is, for whoever loads it: josefine_amd refuses it unless the tests' child process says it means to
extern "C" __attribute__((visibility("default"))) int jg_emulated_device() { return 1; }
'''

ROCPRIM_SORT = r'''
#pragma once
// stand-in for rocPRIM's device radix sort (the library-sort A/B and the drained fault records): a stable host sort on the key bits
#include <hip/hip_runtime.h>
#include <algorithm>
#include <numeric>
namespace rocprim {
template <class K, class V>
static inline hipError_t radix_sort_pairs(void* tmp, size_t& bytes, K* kin, K* kout, V* vin, V* vout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t = nullptr,
                                          bool = false) {
  if (!tmp) { bytes = 64; return hipSuccess; }
  wg::op([=] {
    std::vector<size_t> o(n);
    std::iota(o.begin(), o.end(), 0);
    const K mask = end_bit >= 8 * sizeof(K) ? ~K(0) : (K(1) << end_bit) - 1;
    std::stable_sort(o.begin(), o.end(), [&](size_t a, size_t b) { return ((kin[a] & mask) >> begin_bit) < ((kin[b] & mask) >> begin_bit); });
    for (size_t i = 0; i < n; i++) kout[i] = kin[o[i]], vout[i] = vin[o[i]];
  });
  return hipSuccess;
}
}  // namespace rocprim
'''

_path = None


def build():
    """g++ -> libjosefine_gpu_emulated.so in a temporary directory (once per process); returns its path"""
    global _path
    if _path is not None:
        return _path
    tmp = tempfile.mkdtemp(prefix="jg_host_device_")
    for d in (("shim", "hip"), ("shim", "rocprim", "device"), ("shim", "rocprim", "iterator")):
        os.makedirs(os.path.join(tmp, *d))
    open(os.path.join(tmp, "shim", "hip", "hip_runtime.h"), "w").write(WG_SHIM + RUNTIME)
    open(os.path.join(tmp, "shim", "rocprim", "device", "device_radix_sort.hpp"), "w").write(ROCPRIM_SORT)
    open(os.path.join(tmp, "shim", "rocprim", "device", "device_select.hpp"), "w").write("#pragma once\n")
    open(os.path.join(tmp, "shim", "rocprim", "iterator", "counting_iterator.hpp"), "w").write("#pragma once\n")
    so = os.path.join(tmp, "libjosefine_gpu_emulated.so")
    # (clang, as hipcc's host pass: the translation unit is written for it - e.g. helpers declared before the extern "C" block that defines them)
    cxx = os.environ.get("JG_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cc = [cxx, "-std=c++17", "-O1", "-shared", "-fPIC", "-pthread", "-x", "c++", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
          f"-I{os.path.join(tmp, 'shim')}", f"-I{CSRC}", f"-I{os.path.join(ROOT, 'include')}", os.path.join(CSRC, "josefine_gpu.hip"), "-o", so]
    if os.environ.get("JG_HOST_SANITIZE"):  # (as tests/host_compiled.py; clang's runtime: LD_PRELOAD its libclang_rt.asan-x86_64.so for the child's python)
        cc += ["-g", "-fno-omit-frame-pointer", "-DWG_UCONTEXT", f"-fsanitize={os.environ['JG_HOST_SANITIZE']}", "-fno-sanitize-recover=all", "-shared-libsan"]
    r = subprocess.run(cc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-6000:]
    _path = so
    return so


def run_pytest(args, env=None, timeout=1800):
    """the given tests in a child process whose josefine_amd loads the emulated build"""
    import sys
    e = dict(os.environ)
    e.update(JOSEFINE_GPU_LIB=build(), JG_EMULATED_DEVICE="1", PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), e.get("PYTHONPATH", "")]))
    e.update(env or {})
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + list(args), cwd=ROOT, env=e, capture_output=True, text=True,
                          timeout=timeout)
