// Internal declarations shared by the HIP sources of libautoinst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <atomic>
#include <thread>
#include <string>
#include <vector>

#include "../../include/autoinst_hip.h"

// ----------------------------------------------------------------------------- errors
void ai_set_error(const char* fmt, ...);

#define AI_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ai_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return (_e == hipErrorOutOfMemory) ? AI_ERR_OOM : AI_ERR_HIP;                      \
    }                                                                                     \
  } while (0)

#define AI_TRY(expr)            \
  do {                          \
    int _s = (expr);            \
    if (_s != AI_OK) return _s; \
  } while (0)

#define AI_KERNEL_CHECK() AI_HIP(hipGetLastError())

// ----------------------------------------------------------------------------- workspace arena
// Call-scoped device workspace: a grow-only list of large hipMalloc'd blocks owned by the context,
// bump-allocated during one API call and rewound when the call ends.  After the first chunk no
// call allocates or frees device memory (hipMalloc / hipFree cost ~0.1 ms each and stall the stream).
struct ai_arena {
  struct Block {
    char* base;
    size_t cap;
  };
  std::vector<Block> blocks;
  size_t cur = 0, off = 0;
  size_t min_block = (size_t)256 << 20;
  size_t need = 0, need_max = 0;  // bytes handed out by the running call / by the largest call so far
  void* alloc(size_t bytes);      // nullptr on out-of-memory
  void rewind() {
    if (need > need_max) need_max = need;
    need = 0, cur = 0, off = 0;
  }
  // A call that does not fit the blocks it finds appends one, and calls of different shapes leave a list whose sum
  // is well above what any one of them needs.  At the START of a call (nothing of this context is in flight on the
  // arena: every call drains its streams' use of it before it returns, and hipFree waits for the device anyway) such a
  // list is replaced by one block of the largest need seen + 6 %.
  void consolidate();
  size_t capacity() const {
    size_t t = 0;
    for (const auto& b : blocks) t += b.cap;
    return t;
  }
  void release_all();
};
ai_arena* ai_current_arena();             // arena of the API call running on this thread (or nullptr)
void ai_set_current_arena(ai_arena* a);

struct ArenaScope {
  ai_arena* prev;
  explicit ArenaScope(ai_arena* a) : prev(ai_current_arena()) {
    if (a) {
      a->rewind();
      a->consolidate();
    }
    ai_set_current_arena(a);
  }
  ~ArenaScope() {
    if (ai_current_arena()) ai_current_arena()->rewind();
    ai_set_current_arena(prev);
  }
};

// ----------------------------------------------------------------------------- handles
#define AI_PINNED_INTS 4096
#define AI_CHECK_DEPTH 16
#define AI_STAGE_BYTES ((size_t)8 << 20)
#define AI_FLOW_UPSLOTS 32                       // pinned upload ring of the asynchronous frontier (ai_flow.inc)
#define AI_FLOW_EVENTS (AI_FLOW_UPSLOTS + 2)     // one event per upload slot + wave stage + main-stream mark
// Device buffers of the graphs a context hands out (ai_csr): hipFree synchronises the whole device,
// which with several host threads stalls every other thread's stream, so freed buffers are kept and
// re-used by the next graph of similar size.
struct ai_graph_cache {
  struct Block {
    void* p;
    size_t cap;
  };
  std::vector<Block> free_list, live;
  std::mutex mu;  // a graph may be released (e.g. by a garbage collector) on another thread than its context's
  size_t cached_bytes = 0;
  size_t max_cached_bytes = (size_t)32 << 30;
  hipError_t alloc(void** out, size_t bytes);
  void release(void* p);
  void release_all();
};

// One helper thread per context for host-side work of a call that must not hold up the thread that feeds the streams (the
// Ritz coefficients of a harvest wave: ai_flow.inc).  One job at a time; created on first use, joined with the context.
struct ai_helper {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, quit = false;
  std::atomic<bool> done{true};   // store-release by the worker after the job's last write, load-acquire by the poller (Flow::run)
  std::atomic<bool> failed{false};  // the job threw (std::bad_alloc of a vector, ...): the submitter turns that into an error status
  bool try_done() const { return done.load(std::memory_order_acquire); }
  void start();
  void submit(std::function<void()> f);  // the previous job must have been waited for
  void wait();
  void stop();
};

struct ai_csr;
struct ai_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev[8];
  int num_cu;
  int32_t* pinned;                     // AI_PINNED_INTS host-pinned ints: results of in-flight convergence checks
  hipEvent_t chk_ev[AI_CHECK_DEPTH];   // one event per in-flight check (recorded on `side`)
  hipEvent_t chk_ev1[AI_CHECK_DEPTH];  // main stream -> side stream hand-off of a check's inputs
  hipStream_t side;                    // convergence checks run here, beside the Lanczos steps
  hipStream_t wave = nullptr;          // harvest waves of the asynchronous frontier (created on first use)
  char* fpin = nullptr;                // its pinned host memory (uploads, check results, wave results)
  hipEvent_t fev[AI_FLOW_EVENTS] = {};  // its events
  ai_arena arena;                      // call-scoped device workspace, kept between calls
  ai_graph_cache graphs;               // buffers of the graphs this context built
  char* stage;                         // AI_STAGE_BYTES of pinned host memory for packed small uploads / downloads
  std::mutex graphs_mu;                // guards live_graphs (a graph may be freed from another thread)
  std::vector<ai_csr*> live_graphs;    // graphs this context built and that are still alive: their buffers are the context's
  ai_helper helper;                    // host-side helper thread (started on first use)
};
void ai_register_graph(ai_ctx* ctx, ai_csr* g);   // sets g->owner
// a graph whose context was destroyed keeps a valid handle (to free) but no buffers
#define AI_CHECK_GRAPH(g, who)                                          \
  do {                                                                  \
    if ((g) && !(g)->rowptr) {                                          \
      ai_set_error("%s: the graph's context was destroyed", (who));     \
      return AI_ERR_BAD_ARG;                                            \
    }                                                                   \
  } while (0)

struct ai_csr {
  int64_t n;
  int64_t nnz;
  int32_t* rowptr;  // n + 1 (device)
  int32_t* col;     // nnz    (device) internal row ids
  double* val;      // nnz    (device) raw affinities w_ij
  int32_t* orig;    // n      (device) internal row -> caller's original id; nullptr = identity
  int device;
  ai_ctx* owner = nullptr;  // the context whose buffer cache holds rowptr / col / val / orig; nullptr: buffers are not the handle's
                            // (a temporary view, or the owner was destroyed: then the pointers above are null as well)
};

// Device buffer of one API call: carved from the context's arena when one is active (then
// release is a no-op and the memory is reclaimed when the call ends), else hipMalloc / hipFree.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t count = 0;
  bool owned = false;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    count = 0;
    owned = false;
  }
  int alloc(size_t n) {
    release();
    if (n == 0) n = 1;
    if (ai_arena* a = ai_current_arena()) {
      p = (T*)a->alloc(n * sizeof(T));
      if (!p) {
        ai_set_error("device workspace of %zu bytes could not be allocated", n * sizeof(T));
        return AI_ERR_OOM;
      }
      count = n;
      return AI_OK;
    }
    hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      ai_set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
      return AI_ERR_OOM;
    }
    owned = true;
    count = n;
    return AI_OK;
  }
  int ensure(size_t n) { return (n <= count && p) ? AI_OK : alloc(n); }
};

// ----------------------------------------------------------------------------- scan (ai_scan.hip)
// out[0..n] = exclusive prefix sums of in[0..n-1]; out[n] = total.  in may alias out.
// tmp must hold ai_scan_tmp_elems(n) int32.
size_t ai_scan_tmp_elems(int64_t n);
// total_out (may be null): pinned host address that also receives the total
int ai_exclusive_scan_i32(hipStream_t stream, const int32_t* in, int32_t* out, int64_t n, int32_t* tmp, int32_t* total_out = nullptr);

// ----------------------------------------------------------------------------- device helpers
#ifdef __HIPCC__

#define AI_BLOCK 256
#define AI_LPR 16   // lanes per row in the row-parallel kernels (neighbour counts are ~30-40)

// Lane exchanges inside a 16-lane DPP row run on the VALU (v_mov_b32 with a dpp modifier); __shfl_xor compiles to
// ds_bpermute_b32, i.e. goes through the LDS pipe: in the Lanczos SpMV 56 of 71 LDS instructions were such shuffles and
// kept the LDS pipe busy for 30 % of the kernel (SQ_LDS_IDX_ACTIVE, profiles/r02_pmc_lds_pipe.txt).
template <int CTRL>
__device__ __forceinline__ double ai_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a row; every lane gets the total.  Tree: pairs, quads (quad_perm), halves (row_half_mirror),
// row (row_mirror) -- a fixed order, so results stay reproducible run to run.
__device__ __forceinline__ double ai_group16_sum(double v) {
  v += ai_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += ai_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += ai_dpp<0x141>(v);  // row_half_mirror
  v += ai_dpp<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ double ai_wave_sum(double v) {
  v = ai_group16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// Every thread of the block returns the same value; fixed summation order -> reproducible.
__device__ __forceinline__ double ai_block_sum(double v, double* sm /* AI_BLOCK/64 doubles */) {
  v = ai_wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < AI_BLOCK / 64; ++i) r += sm[i];
  return r;
}

// the same when `sm` has not been used before in the block (no barrier in front of the write)
__device__ __forceinline__ double ai_block_sum_first(double v, double* sm /* AI_BLOCK/64 doubles */) {
  v = ai_wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sm[w] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < AI_BLOCK / 64; ++i) r += sm[i];
  return r;
}

// Deterministic start-vector entry in (-1, 1) from the ORIGINAL point id (splitmix64 finaliser).
// tests/gpu_model.py::start_vector is the same function.
__device__ __forceinline__ double ai_hash_unit(uint32_t id) {
  uint64_t x = (uint64_t)id + 0x9E3779B97F4A7C15ull;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// THE RULE for memory that another stream of the call, or the host, rewrites while the call runs (pool tables and their ring of
// versions, SegRec tables in pinned memory, per-slot check state, activity flags, Lanczos histories, step counters): no PLAIN load.
// A plain load of a block-uniform address becomes a scalar load through the scalar cache, which is not coherent with another
// stream's or the host's writes; round 3 lost one chunk in ~700 to exactly that (fk_check).  ai_ld_agent (device-written data) and
// ai_ld_sys (host-written pinned data) are relaxed atomic loads: vector loads that bypass the non-coherent caches.
// DESIGN.md section 6 lists the audited sites; grep for these two names to find them.
template <typename T>
__device__ __forceinline__ T ai_ld_agent(const T* p) {
  return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T ai_ld_sys(const T* p) {
  return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename T>
__device__ __forceinline__ void ai_st_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a record of 32-bit words from host-written pinned memory, for the whole block: thread k loads word k ONCE (a system-scope load
// crosses the bus: 256 threads loading the same 16 words each made fk_expand_pool ten times longer), LDS hands it round
template <typename R>
__device__ __forceinline__ R ai_ld_sys_record_block(const R* p, int32_t* sm /* sizeof(R) / 4 words of LDS */) {
  static_assert(sizeof(R) % 4 == 0, "record of 32-bit words");
  constexpr int NW = (int)(sizeof(R) / 4);
  if ((int)threadIdx.x < NW) sm[threadIdx.x] = ai_ld_sys(reinterpret_cast<const int32_t*>(p) + threadIdx.x);
  __syncthreads();
  R r;
  int32_t* d = reinterpret_cast<int32_t*>(&r);
#pragma unroll
  for (int k = 0; k < NW; ++k) d[k] = sm[k];
  return r;
}
// the same for one thread (tables copied element-wise: each element is loaded by exactly one thread)
template <typename R>
__device__ __forceinline__ R ai_ld_sys_record(const R* p) {
  static_assert(sizeof(R) % 4 == 0, "record of 32-bit words");
  R r;
  int32_t* d = reinterpret_cast<int32_t*>(&r);
  const int32_t* q = reinterpret_cast<const int32_t*>(p);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(R) / 4); ++k) d[k] = ai_ld_sys(q + k);
  return r;
}

// blockIdx -> task remap so that each XCD (blocks b, b+8, ... share one) works on one
// contiguous eighth of the rows: its L2 then holds one slice of the gathered vector.
__device__ __forceinline__ int ai_xcd_task(int bid, int nblk) {
  const int per = nblk >> 3;  // blocks per XCD in the evenly divisible part
  const int body = per << 3;
  if (bid >= body) return bid;  // ragged tail keeps its identity
  return (bid & 7) * per + (bid >> 3);
}
#endif
