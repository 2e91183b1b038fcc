// nfp_launch.h — host-side launch machinery shared by the translation units of libnfp_hip.so (nfp_hip.hip: C ABI,
// table kernels, any-geometry kernels; nfp_tile.hip: the row-band kernels).  Two units so that they compile in
// parallel; everything here is `inline` (one instance per shared library).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "nfp_common.h"

namespace nfp_host {
using namespace nfp;

inline thread_local char g_err[512] = "";
// The dispatcher names the variant it picked in a buffer of the CALLING thread (forward and autograd's backward
// thread dispatch concurrently); a finished nfp_forward / nfp_backward publishes it under a lock as the process's
// "last variant", which nfp_last_variant() copies back into the reader's own thread.
inline thread_local char g_variant[64] = "";
inline thread_local char t_variant_out[64] = "";
inline std::mutex g_variant_mu;
inline char g_variant_last[64] = "";
inline std::atomic<uint64_t> g_launches{0};
inline std::atomic<void*> g_time_start{nullptr}, g_time_stop{nullptr};  // nfp_time_next_launch (process-wide: backward launches from autograd's thread)

inline void publish_variant() {
  std::lock_guard<std::mutex> lock(g_variant_mu);
  memcpy(g_variant_last, g_variant, sizeof(g_variant_last));
}

// Test / A-B switches, read from the environment ONCE when the library is loaded (and again only when a test
// calls nfp_reload_env): the launch path itself never reads it.
struct Switches {
  std::atomic<int> fwd_scalar{0}, bwd_atomic{0}, bwd_bands{0}, force_generic{0}, mfma{1}, tile_first{0};
  std::atomic<int> tile_wgs{0}, tile_lds_kb{0}, tile_cap{0};   // A/B overrides of the row-band launchers' constants (0: built-in)
  std::atomic<int> tile_max_grid{0};                           // (tests: a smaller launch-splitting threshold)
  std::atomic<int> pool_ticket{0};   // A/B: fused pooling tail, several row bands per image combined inside ONE launch by the
                                     // band that arrives last (nfp_common.h::pool_last_band).  Measured slower than what it
                                     // replaces (profiles/r04_e_…: 8.7 vs 6.9 us at the headline shape, 103 vs 91 us at
                                     // [256,16,112,112]): off by default
  std::atomic<int> gemm3{1};         // the matrix-core backward's second form (nfp_fast.h::bwd_gemm_phase3: every row tile's densified
                                     // weights written by phase A, x in double-buffered chunks) where its LDS fits; 0 = round 3's form
  std::atomic<int> gemm2{1};         // A/B: the matrix-core backward with the table-free phase A (nfp_gemm2.h; -DNFP_GEMM2_ARM builds only)
  std::atomic<int> tile_dma{-1};     // the channels-last row-band forward staged by LDS-DMA into a position-major slab
                                     // (nfp_tile.h::fwd_tile, DMA): -1 = where it measured faster — bf16 maps of 64 channels and
                                     // more per pixel (8 pieces per position: [256,128,28,28] 23.9 -> 18.8 us, [256,64,56,56] 44.2 ->
                                     // 42.2) — 0 = never, 1 = wherever it applies (float32 loses: 72 -> 121 us at
                                     // [256,16,112,112]; profiles/r04_h_…)
};
inline Switches g_sw;
#ifndef NFP_MFMA_DEFAULT
#define NFP_MFMA_DEFAULT 1
#endif
#ifndef NFP_GEMM3_DEFAULT
#define NFP_GEMM3_DEFAULT 1
#endif
inline void read_env() {
  auto flag = [](const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? (e[0] == '1' ? 1 : 0) : dflt;
  };
  g_sw.fwd_scalar = flag("NFP_FWD_SCALAR", 0);
  g_sw.bwd_atomic = flag("NFP_BWD_ATOMIC", 0);
  g_sw.force_generic = flag("NFP_FORCE_GENERIC", 0);
  g_sw.mfma = flag("NFP_MFMA", NFP_MFMA_DEFAULT);
  g_sw.pool_ticket = flag("NFP_POOL_TICKET", 0);
  g_sw.gemm2 = flag("NFP_GEMM2", 1);
  {
    const char* e = getenv("NFP_GEMM3");   // 0 = never, 1 = where the first form needs several rounds, 2 = wherever it fits (tests)
    g_sw.gemm3 = e ? (e[0] == '2' ? 2 : (e[0] == '1' ? 1 : 0)) : NFP_GEMM3_DEFAULT;
  }
  {
    const char* e = getenv("NFP_TILE_DMA");
    g_sw.tile_dma = e ? (e[0] == '1' ? 1 : 0) : -1;
  }
  g_sw.tile_first = flag("NFP_TILE_FIRST", 0);   // A/B: the row-band kernels of nfp_tile.h also for maps the table kernels serve
  auto num = [](const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
  };
  g_sw.bwd_bands = num("NFP_BWD_BANDS");
  g_sw.tile_wgs = num("NFP_TILE_WGS");
  g_sw.tile_lds_kb = num("NFP_TILE_LDS_KB");
  g_sw.tile_cap = num("NFP_TILE_CAP");
  g_sw.tile_max_grid = num("NFP_TILE_MAX_GRID");
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return NFP_OK;
  return fail(NFP_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

constexpr int kLdsMax = 160 * 1024;
constexpr int kNotApplicable = 1;  // internal: a hot-path launcher declined, use the generic kernels

// floats per input pixel that forward hands to backward
inline int stats_of(int measure) {
  switch (measure) {
    case NFP_COSINE: case NFP_GFC: case NFP_SMITH: return 1;
    case NFP_PEARSON: return 2;
    default: return 0;
  }
}

// Kernels that want more than 64 KiB of dynamic LDS must be told so once (per kernel and size class);
// remembered here so that steady-state launches make no extra runtime call.
template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return NFP_OK;
  // hipFuncSetAttribute applies to the kernel ON THE CURRENT DEVICE: remembered per (device, kernel)
  static std::mutex mu;
  static std::unordered_map<uintptr_t, size_t> granted;
  int dev = 0;
  if (int rc = hip_ok(hipGetDevice(&dev), "hipGetDevice")) return rc;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[(uintptr_t)(const void*)kernel * 64 + (uintptr_t)(dev & 63)];
  if (have >= bytes) return NFP_OK;
  if (int rc = hip_ok(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax),
                      "hipFuncSetAttribute(max dynamic LDS)"))
    return rc;
  have = kLdsMax;
  return NFP_OK;
}

// One way to launch: host-side limits first (a launch the hardware would refuse or fault on is never
// attempted), then the LDS opt-in, the launch and its error.  In plan mode (nfp_plan) nothing touches the
// GPU: the launch is only described, so the dispatcher's decisions are testable without a device.
inline thread_local bool t_dry = false;
inline thread_local char t_plan[512] = "";

template <typename K, typename... A>
int launch(const char* name, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
  const unsigned threads = block.x * block.y * block.z;   // (the row-band kernels launch (groups, columns, rows) blocks)
  if (lds > (size_t)kLdsMax || threads < 1 || threads > 1024 || grid.x < 1 || grid.y < 1 || grid.z < 1 ||
      grid.y > 65535 || grid.z > 65535)
    return fail(NFP_E_UNSUPPORTED, "%s: launch shape grid (%u,%u,%u) block %u lds %zu outside the device limits", name,
                grid.x, grid.y, grid.z, threads, lds);
  if (t_dry) {
    const size_t n = strlen(t_plan);
    snprintf(t_plan + n, sizeof(t_plan) - n, "%s%s grid=(%u,%u,%u) block=%u lds=%zu", n ? "; " : "", name, grid.x, grid.y,
             grid.z, threads, lds);
    return NFP_OK;
  }
  if (int rc = set_lds(kernel, lds)) return rc;
  // telemetry (nfp_time_next_launch): bracket this one kernel with the caller's events — recorded by the command
  // processor at the kernel's own start and end, like a profiler's kernel trace
  hipEvent_t ev0 = name[0] != '#' ? (hipEvent_t)g_time_start.exchange(nullptr) : nullptr;
  if (ev0 != nullptr) {
    hipEvent_t ev1 = (hipEvent_t)g_time_stop.exchange(nullptr);
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ev0, ev1, 0, args...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
  if (name[0] != '#') g_launches++;  // ('#': one-time setup kernels, not part of a forward / backward)
  return hip_ok(hipGetLastError(), name);
}

// The hot-path kernels (nfp_band.h / nfp_fast.h / nfp_mfma.h / nfp_tile.h) are instantiated for the product form (Cosine) and
// the squared-difference form (L2); DotProduct, GFC and RMSE ride on them through KP's run-time constants (nfp_common.h).
inline bool hot_product(const KP& g) { return g.measure == NFP_COSINE || g.measure == NFP_DOT || g.measure == NFP_GFC; }
inline bool hot_measure(const KP& g) {
  return hot_product(g) || (g.measure == NFP_NORM && g.p == 2.f) || g.measure == NFP_RMSE;
}
// Norm p = 1 (the class default, nfp.py:16) — and EMD (nfp.py:207-216), the same sum, which make_kp turns into it:
// instantiations of their own of the table kernels, the radius-(1, 2) form and the row-band kernels (sums of |a - b|; a
// gradient in sign(a - b)); plain maps only — no fused pooling tail.
inline bool hot_l1(const KP& g) { return g.measure == NFP_NORM && g.p == 1.f; }
// Geman-McClure, Canberra, Hellinger, Jeffrey, squared chord, chi-squared 1 (nfp.py:181-193, 218-227, 229-241, 295-308, 310-324,
// 243-252): sums of a symmetric per-channel term, no per-pixel statistic — one shared instantiation of the row-band and of
// the table kernels (nfp_measures.h::kSymTerm); plain maps only.  (Chi-squared 2 is not symmetric in the pair.)
inline bool hot_sym(const KP& g) {
  return g.measure == NFP_GEMAN || g.measure == NFP_CANBERRA || g.measure == NFP_SQUAREDCHORD || g.measure == NFP_CHISQUARED1 ||
         g.measure == NFP_HELLINGER || g.measure == NFP_JEFFREY;
}
inline const char* hot_name(const KP& g) {
  if (hot_l1(g)) return "l1";
  if (hot_sym(g))
    return g.measure == NFP_GEMAN ? "geman" : (g.measure == NFP_CANBERRA ? "canberra" : (g.measure == NFP_SQUAREDCHORD ? "sqchord" : (g.measure == NFP_HELLINGER ? "hellinger" : (g.measure == NFP_JEFFREY ? "jeffrey" : "chisq1"))));
  return g.measure == NFP_COSINE ? "cos" : (g.measure == NFP_DOT ? "dot" : (g.measure == NFP_GFC ? "gfc" : (g.measure == NFP_RMSE ? "rmse" : "l2")));
}
inline bool force_generic() { return g_sw.force_generic.load(std::memory_order_relaxed) != 0; }
inline int round4(int v) { return (v + 3) & ~3; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// fewest groups (<= gmax) that still finish ncq channel quads in ceil(ncq/gmax) rounds
inline int even_groups(int ncq, int gmax) {
  int rounds = (ncq + gmax - 1) / gmax;
  return (ncq + rounds - 1) / rounds;
}

// ---- the row-band kernels of nfp_tile.h (defined in nfp_tile.hip) -------------------------------------------------------
// Each returns NFP_OK, kNotApplicable (this descriptor is not theirs / does not fit) or an NFP_E_* code; g_variant names
// the launch.  pool: the fused nfp_pooling tail (part = scratch for the bands' partial sums; *nb = bands per image).
bool tile_ok(const KP& g, const void* x, const void* gx);
int tile_forward(const KP& g, const void* x, void* out, float* saved, hipStream_t st, bool pool, float* part, int* nb,
                 float* gap, float* nfpm);
int tile_backward(const KP& g, const void* x, const void* go, const void* out, const float* saved, void* gx, hipStream_t st,
                  bool pool, const float* ggap, const float* gnfpm);
int tile_pool_fold(const KP& g, const float* part, float* gap, float* nfpm, int nb, hipStream_t st);

}  // namespace nfp_host
