// Context, error reporting and CSR hand-off (upload / export / free) of libautoinst_hip.so.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <stdlib.h>
#include <hip/hip_ext.h>
#include "ai_common.h"

static thread_local char g_err[1024] = "";

void ai_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ai_last_error(void) { return g_err; }
extern "C" int ai_abi_version(void) { return AI_ABI_VERSION; }
extern "C" int64_t ai_abi_sizeof(int which) {
  return which == 0 ? (int64_t)sizeof(ai_ncut_opts) : which == 1 ? (int64_t)sizeof(ai_ncut_stats) : -1;
}

// ----------------------------------------------------------------------------- arena
static thread_local ai_arena* g_arena = nullptr;
ai_arena* ai_current_arena() { return g_arena; }
void ai_set_current_arena(ai_arena* a) { g_arena = a; }

void* ai_arena::alloc(size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  need += bytes;
  // first block, from the current one on, that still has room
  for (size_t b = cur; b < blocks.size(); ++b) {
    const size_t o = (b == cur) ? off : 0;
    if (o + bytes <= blocks[b].cap) {
      cur = b;
      off = o + bytes;
      return blocks[b].base + o;
    }
  }
  Block nb;
  nb.cap = bytes > min_block ? bytes : min_block;
  if (hipMalloc((void**)&nb.base, nb.cap) != hipSuccess) return nullptr;
  blocks.push_back(nb);
  cur = blocks.size() - 1;
  off = bytes;
  return nb.base;
}

void ai_arena::consolidate() {
  if (blocks.size() <= 1) return;
  const size_t want = need_max + need_max / 16 + ((size_t)16 << 20);
  for (auto& b : blocks) (void)hipFree(b.base);
  blocks.clear();
  cur = off = 0;
  Block nb;
  nb.cap = want > min_block ? want : min_block;
  if (hipMalloc((void**)&nb.base, nb.cap) == hipSuccess) blocks.push_back(nb);  // on failure the call's own requests report it
}

void ai_arena::release_all() {
  for (auto& b : blocks) (void)hipFree(b.base);
  blocks.clear();
  cur = off = 0;
}
hipError_t ai_graph_cache::alloc(void** out, size_t bytes) {
  std::lock_guard<std::mutex> lock(mu);
  bytes = (bytes + ((size_t)1 << 16) - 1) & ~(((size_t)1 << 16) - 1);
  // best fit among the kept buffers, at most 25 % (+1 MB) larger than asked for
  int best = -1;
  for (int i = 0; i < (int)free_list.size(); ++i)
    if (free_list[i].cap >= bytes && free_list[i].cap <= bytes + bytes / 4 + ((size_t)1 << 20) &&
        (best < 0 || free_list[i].cap < free_list[best].cap))
      best = i;
  if (best >= 0) {
    live.push_back(free_list[best]);
    cached_bytes -= free_list[best].cap;
    *out = free_list[best].p;
    free_list.erase(free_list.begin() + best);
    return hipSuccess;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess && !free_list.empty()) {  // give the kept buffers back and try once more
    for (auto& b : free_list) (void)hipFree(b.p);
    free_list.clear();
    cached_bytes = 0;
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) return e;
  live.push_back(Block{p, bytes});
  *out = p;
  return hipSuccess;
}

void ai_graph_cache::release(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(mu);
  for (size_t i = 0; i < live.size(); ++i)
    if (live[i].p == p) {
      const Block b = live[i];
      live.erase(live.begin() + i);
      if (cached_bytes + b.cap > max_cached_bytes) {
        (void)hipFree(b.p);
      } else {
        free_list.push_back(b);
        cached_bytes += b.cap;
      }
      return;
    }
  // not one of this cache's buffers: leave it alone (ai_csr_free always goes through the owning context)
}

void ai_graph_cache::release_all() {
  std::lock_guard<std::mutex> lock(mu);
  for (auto& b : free_list) (void)hipFree(b.p);
  for (auto& b : live) (void)hipFree(b.p);
  free_list.clear();
  live.clear();
  cached_bytes = 0;
}

extern "C" int ai_version(void) { return 100; }

extern "C" int ai_ctx_create(int device, ai_ctx** out) {
  if (!out) {
    ai_set_error("ai_ctx_create: out is null");
    return AI_ERR_BAD_ARG;
  }
  *out = nullptr;
  int count = 0;
  AI_HIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) {
    ai_set_error("ai_ctx_create: device %d not present (%d visible)", device, count);
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  AI_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    ai_set_error("ai_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return AI_ERR_BAD_ARG;
  }
  ai_ctx* c = new ai_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  c->stream = nullptr;
  c->side = nullptr;
  c->pinned = nullptr;
  c->stage = nullptr;
  for (int i = 0; i < 8; ++i) c->ev[i] = nullptr;
  for (int i = 0; i < AI_CHECK_DEPTH; ++i) c->chk_ev[i] = c->chk_ev1[i] = nullptr;
  // any failure below releases what has been created so far
  auto build = [&]() -> int {
    AI_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i) AI_HIP(hipEventCreate(&c->ev[i]));
    // (confining this stream and the wave stream to 32 / 64 / 128 CUs with hipExtStreamCreateWithCUMask was measured in round 4:
    // 105 / 91 / 96 chunks/s against 119 without a mask -- the checks and waves are on the critical path of every solve)
    AI_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    for (int i = 0; i < AI_CHECK_DEPTH; ++i) {
      AI_HIP(hipEventCreateWithFlags(&c->chk_ev[i], hipEventDisableTiming));
      AI_HIP(hipEventCreateWithFlags(&c->chk_ev1[i], hipEventDisableTiming));
    }
    AI_HIP(hipHostMalloc((void**)&c->pinned, AI_PINNED_INTS * sizeof(int32_t), hipHostMallocDefault));
    AI_HIP(hipHostMalloc((void**)&c->stage, AI_STAGE_BYTES, hipHostMallocDefault));
    return AI_OK;
  };
  const int st = build();
  if (st != AI_OK) {
    (void)ai_ctx_destroy(c);
    return st;
  }
  *out = c;
  return AI_OK;
}

// ----------------------------------------------------------------------------- stream-rate hook (bench.py)
// every block streams its own contiguous range, four 16-byte accesses in flight per thread (tools/micro/copy_rate.hip: 5.66 TB/s
// with 65 536 blocks on an MI355X, against 4.4-5.1 for a grid-stride loop over 2 048-16 384 blocks and 5.1 for hipMemcpyAsync)
__global__ __launch_bounds__(256) void k_copy16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  size_t i = b0 + threadIdx.x;
  for (; i + 3 * 256 < b1; i += 4 * 256) {
    const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a;
    dst[i + 256] = b;
    dst[i + 512] = c;
    dst[i + 768] = d;
  }
  for (; i < b1; i += 256) dst[i] = src[i];
}

extern "C" int ai_bench_copy(ai_ctx* ctx, int64_t bytes, int32_t reps, double* gbps) {
  if (!ctx || !gbps || bytes < 16 || reps < 1) {
    ai_set_error("ai_bench_copy: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  const size_t n16 = (size_t)bytes / 16;
  float4 *src = nullptr, *dst = nullptr;
  AI_HIP(hipMalloc((void**)&src, n16 * 16));
  if (hipMalloc((void**)&dst, n16 * 16) != hipSuccess) {
    (void)hipFree(src);
    ai_set_error("ai_bench_copy: out of device memory");
    return AI_ERR_OOM;
  }
  hipStream_t st = ctx->stream;
  (void)hipMemsetAsync(src, 1, n16 * 16, st);
  const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(65536, (n16 + 1023) / 1024));
  hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const float4*)src, dst, n16);  // warm-up
  hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1];
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const float4*)src, dst, n16);
  (void)hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  if (e != hipSuccess || !(ms > 0.f)) {
    ai_set_error("ai_bench_copy: %s", hipGetErrorString(e));
    return AI_ERR_HIP;
  }
  *gbps = (double)reps * 2.0 * (double)(n16 * 16) / ((double)ms * 1e-3) / 1e9;
  return AI_OK;
}

void ai_helper::start() {
  if (th.joinable()) return;
  quit = false;
  th = std::thread([this] {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [this] { return has_job || quit; });
      if (quit) return;
      std::function<void()> f = std::move(job);
      has_job = false;
      lk.unlock();
      try {
        f();
      } catch (...) {  // nothing may escape a std::thread body (std::terminate): the submitter reads `failed` after wait() / try_done()
        failed.store(true, std::memory_order_relaxed);
      }
      lk.lock();
      done.store(true, std::memory_order_release);
      cv.notify_all();
    }
  });
}
void ai_helper::submit(std::function<void()> f) {
  start();
  std::lock_guard<std::mutex> lk(mu);
  job = std::move(f);
  has_job = true;
  failed.store(false, std::memory_order_relaxed);
  done.store(false, std::memory_order_relaxed);
  cv.notify_all();
}
void ai_helper::wait() {
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [this] { return done.load(std::memory_order_acquire); });
}
void ai_helper::stop() {
  if (!th.joinable()) return;
  {
    std::lock_guard<std::mutex> lk(mu);
    quit = true;
    cv.notify_all();
  }
  th.join();
}

void ai_register_graph(ai_ctx* ctx, ai_csr* g) {
  g->owner = ctx;
  std::lock_guard<std::mutex> lock(ctx->graphs_mu);
  ctx->live_graphs.push_back(g);
}

extern "C" int ai_ctx_mem_info(ai_ctx* ctx, int64_t out[4]) {
  if (!ctx || !out) {
    ai_set_error("ai_ctx_mem_info: null argument");
    return AI_ERR_BAD_ARG;
  }
  out[0] = (int64_t)ctx->arena.capacity();
  out[1] = (int64_t)ctx->arena.blocks.size();
  std::lock_guard<std::mutex> lock(ctx->graphs.mu);
  size_t live = 0;
  for (const auto& b : ctx->graphs.live) live += b.cap;
  out[2] = (int64_t)live;
  out[3] = (int64_t)ctx->graphs.cached_bytes;
  return AI_OK;
}

extern "C" int ai_ctx_destroy(ai_ctx* ctx) {
  if (!ctx) return AI_OK;
  ctx->helper.stop();
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 8; ++i)
    if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  if (ctx->side) (void)hipStreamSynchronize(ctx->side);
  for (int i = 0; i < AI_CHECK_DEPTH; ++i) {
    if (ctx->chk_ev[i]) (void)hipEventDestroy(ctx->chk_ev[i]);
    if (ctx->chk_ev1[i]) (void)hipEventDestroy(ctx->chk_ev1[i]);
  }
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->wave) {
    (void)hipStreamSynchronize(ctx->wave);
    (void)hipStreamDestroy(ctx->wave);
  }
  for (int i = 0; i < AI_FLOW_EVENTS; ++i)
    if (ctx->fev[i]) (void)hipEventDestroy(ctx->fev[i]);
  if (ctx->fpin) (void)hipHostFree(ctx->fpin);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  {
    // graphs that outlive their context lose their buffers with it: their handles stay valid to free, and every
    // other call on them fails with "the graph's context was destroyed" instead of touching freed memory
    std::lock_guard<std::mutex> lock(ctx->graphs_mu);
    for (ai_csr* g : ctx->live_graphs) {
      g->rowptr = nullptr;
      g->col = nullptr;
      g->val = nullptr;
      g->orig = nullptr;
      g->owner = nullptr;
    }
    ctx->live_graphs.clear();
  }
  ctx->arena.release_all();
  ctx->graphs.release_all();
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return AI_OK;
}

// The buffers go back to the cache of the context that BUILT the graph, whichever context (or none) is passed.
extern "C" int ai_csr_free(ai_ctx* ctx, ai_csr* csr) {
  (void)ctx;
  if (!csr) return AI_OK;
  if (ai_ctx* o = csr->owner) {
    (void)hipSetDevice(o->device);
    {
      std::lock_guard<std::mutex> lock(o->graphs_mu);
      for (size_t i = 0; i < o->live_graphs.size(); ++i)
        if (o->live_graphs[i] == csr) {
          o->live_graphs.erase(o->live_graphs.begin() + i);
          break;
        }
    }
    o->graphs.release(csr->rowptr);
    o->graphs.release(csr->col);
    o->graphs.release(csr->val);
    o->graphs.release(csr->orig);
  }
  delete csr;
  return AI_OK;
}

extern "C" int ai_csr_dims(const ai_csr* csr, int64_t* n, int64_t* nnz) {
  if (!csr) {
    ai_set_error("ai_csr_dims: csr is null");
    return AI_ERR_BAD_ARG;
  }
  if (n) *n = csr->n;
  if (nnz) *nnz = csr->nnz;
  return AI_OK;
}

extern "C" int ai_csr_from_host(ai_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                                ai_csr** out) {
  if (!ctx || !indptr || !out || n <= 0) {
    ai_set_error("ai_csr_from_host: bad argument");
    return AI_ERR_BAD_ARG;
  }
  const int64_t nnz = indptr[n];
  if (indptr[0] != 0 || nnz < 0 || nnz >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 30) || (nnz > 0 && (!indices || !data))) {
    ai_set_error("ai_csr_from_host: indptr[0] = %lld, nnz = %lld out of range", (long long)indptr[0], (long long)nnz);
    return AI_ERR_BAD_ARG;
  }
  std::vector<int32_t> rp((size_t)n + 1);
  for (int64_t i = 0; i <= n; ++i) {
    if (i > 0 && indptr[i] < indptr[i - 1]) {
      ai_set_error("ai_csr_from_host: indptr decreases at row %lld", (long long)i);
      return AI_ERR_BAD_ARG;
    }
    rp[(size_t)i] = (int32_t)indptr[i];
  }
  for (int64_t e = 0; e < nnz; ++e)
    if (indices[e] < 0 || indices[e] >= n) {
      ai_set_error("ai_csr_from_host: column index %d out of range at entry %lld", indices[e], (long long)e);
      return AI_ERR_BAD_ARG;
    }
  AI_HIP(hipSetDevice(ctx->device));
  ai_csr* A = new ai_csr();
  A->n = n;
  A->nnz = nnz;
  A->rowptr = nullptr;
  A->col = nullptr;
  A->val = nullptr;
  A->orig = nullptr;
  A->device = ctx->device;
  ai_register_graph(ctx, A);
  hipError_t e1 = ctx->graphs.alloc((void**)&A->rowptr, (size_t)(n + 1) * sizeof(int32_t));
  hipError_t e2 = ctx->graphs.alloc((void**)&A->col, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
  hipError_t e3 = ctx->graphs.alloc((void**)&A->val, (size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    ai_csr_free(ctx, A);
    ai_set_error("ai_csr_from_host: device allocation failed");
    return AI_ERR_OOM;
  }
  hipStream_t st = ctx->stream;
  hipError_t c1 = hipMemcpyAsync(A->rowptr, rp.data(), (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st);
  hipError_t c2 = nnz ? hipMemcpyAsync(A->col, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, st) : hipSuccess;
  hipError_t c3 = nnz ? hipMemcpyAsync(A->val, data, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, st) : hipSuccess;
  hipError_t c4 = hipStreamSynchronize(st);
  if (c1 != hipSuccess || c2 != hipSuccess || c3 != hipSuccess || c4 != hipSuccess) {
    ai_csr_free(ctx, A);
    ai_set_error("ai_csr_from_host: copy to device failed");
    return AI_ERR_HIP;
  }
  *out = A;
  return AI_OK;
}

namespace {
__global__ __launch_bounds__(AI_BLOCK) void k_rowlen_to_orig(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ orig,
                                                             int64_t n, int32_t* __restrict__ len_orig) {
  const int64_t p = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (p >= n) return;
  len_orig[orig[p]] = rowptr[p + 1] - rowptr[p];
}

// one thread per row: relabel columns to original ids and insertion-sort them ascending
__global__ __launch_bounds__(AI_BLOCK) void k_export_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const double* __restrict__ val, const int32_t* __restrict__ orig,
                                                          int64_t n, const int32_t* __restrict__ out_ptr, int32_t* __restrict__ out_col,
                                                          double* __restrict__ out_val) {
  const int64_t p = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (p >= n) return;
  const int32_t s = rowptr[p], e = rowptr[p + 1];
  const int32_t base = out_ptr[orig[p]];
  for (int32_t k = s; k < e; ++k) {
    const int32_t c = orig[col[k]];
    const double v = val[k];
    int32_t pos = base + (k - s);
    while (pos > base && out_col[pos - 1] > c) {
      out_col[pos] = out_col[pos - 1];
      out_val[pos] = out_val[pos - 1];
      --pos;
    }
    out_col[pos] = c;
    out_val[pos] = v;
  }
}
}  // namespace

extern "C" int ai_csr_export(ai_ctx* ctx, const ai_csr* csr, int64_t* indptr, int32_t* indices, double* data) {
  if (!ctx || !csr || !indptr || (csr->nnz > 0 && (!indices || !data))) {
    ai_set_error("ai_csr_export: bad argument");
    return AI_ERR_BAD_ARG;
  }
  if (!csr->rowptr) {
    ai_set_error("ai_csr_export: the graph's context was destroyed");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n = csr->n, nnz = csr->nnz;
  std::vector<int32_t> rp((size_t)n + 1);
  if (!csr->orig) {
    AI_HIP(hipMemcpyAsync(rp.data(), csr->rowptr, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (nnz) {
      AI_HIP(hipMemcpyAsync(indices, csr->col, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipMemcpyAsync(data, csr->val, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    AI_HIP(hipStreamSynchronize(st));
  } else {
    ArenaScope arena_scope(&ctx->arena);
    DevBuf<int32_t> len, optr, ocol, tmp;
    DevBuf<double> oval;
    AI_TRY(len.alloc(n + 1));
    AI_TRY(optr.alloc(n + 1));
    AI_TRY(ocol.alloc(nnz));
    AI_TRY(oval.alloc(nnz));
    AI_TRY(tmp.alloc(ai_scan_tmp_elems(n)));
    const unsigned gb = (unsigned)((n + AI_BLOCK - 1) / AI_BLOCK);
    hipLaunchKernelGGL(k_rowlen_to_orig, dim3(gb), dim3(AI_BLOCK), 0, st, (const int32_t*)csr->rowptr, (const int32_t*)csr->orig, n, len.p);
    AI_KERNEL_CHECK();
    AI_TRY(ai_exclusive_scan_i32(st, len.p, optr.p, n, tmp.p));
    hipLaunchKernelGGL(k_export_rows, dim3(gb), dim3(AI_BLOCK), 0, st, (const int32_t*)csr->rowptr, (const int32_t*)csr->col,
                       (const double*)csr->val, (const int32_t*)csr->orig, n, (const int32_t*)optr.p, ocol.p, oval.p);
    AI_KERNEL_CHECK();
    AI_HIP(hipMemcpyAsync(rp.data(), optr.p, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (nnz) {
      AI_HIP(hipMemcpyAsync(indices, ocol.p, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipMemcpyAsync(data, oval.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    AI_HIP(hipStreamSynchronize(st));
  }
  for (int64_t i = 0; i <= n; ++i) indptr[i] = rp[(size_t)i];
  return AI_OK;
}
