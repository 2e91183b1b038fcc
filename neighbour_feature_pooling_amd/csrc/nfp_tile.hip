// nfp_tile.hip — second translation unit of libnfp_hip.so: the row-band kernels of nfp_tile.h and their launchers
// (compiled in parallel with nfp_hip.hip; shared host-side machinery in nfp_launch.h).
#include "nfp_launch.h"
#include "nfp_tile.h"

using namespace nfp;

namespace nfp_host {
namespace {

// ---- row-band kernels on a padded slab (nfp_tile.h): "same" maps of any size, no tables ----------------------------
#ifndef NFP_TILE_WGS
#define NFP_TILE_WGS 512
#endif
#ifndef NFP_TILE_LDS_KB
#define NFP_TILE_LDS_KB 78     // per workgroup, so that two share a compute unit
#endif
bool tile_geometry(const KP& g) {
  if (g.stride != 1 || g.dil != 1 || g.pad != g.R || g.mode == NFP_PAD_CIRCULAR || g.rs == 12) return false;
  if (g.R != 1 && g.R != 2) return false;
  // a band of one row must fit the forward's threads
  if (g.W < 4 || g.W < 2 * g.R + 2 || (g.W + 2 * g.R) * (1 + 2 * g.R) > 1024) return false;
  return g.W <= 512;                                                        // ... and the backward's
}
}  // namespace

bool tile_ok(const KP& g, const void* x, const void* gx) {
  if (force_generic() || !tile_geometry(g) || (g.C & 3)) return false;
  if (!hot_measure(g)) return false;
  const bool nhwc = g.sC == 1 && g.sW == g.C && g.sH == (long long)g.W * g.C;
  if (!g.contig && !nhwc) return false;
  const int es = g.dtype == NFP_F32 ? 4 : 2;
  if ((long long)g.C * g.P * es >= 0x7ffffff0LL) return false;             // 32-bit buffer offsets inside an image
  if (!g.contig) {
    const uintptr_t m = g.dtype == NFP_F32 ? 15 : 7;
    if (((uintptr_t)x & m) || ((uintptr_t)gx & m) || ((g.sB * es) & m) || ((g.gB * es) & m)) return false;
  }
  return true;
}

namespace {

template <int R, int M, bool BF, bool NHWC, bool POOL = false>
int launch_fwd_tile_t(KP g, const void* x, void* out, float* saved, hipStream_t st, float* part = nullptr, int* nb_out = nullptr) {
  constexpr int NF = Win<R>::NF, N = Win<R>::N;
  const int Wp = nfp::tile_row_stride(g.W, R), Wu = g.W + 2 * R;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // first choice: two workgroups per compute unit — half of LDS each; k = 3 runs in 64 registers (up to 1024 threads),
    // k = 5 in up to 128 (at most 512 threads).  Second: one workgroup of up to 1024 threads and all of LDS.
    const size_t budget = attempt == 0 ? (size_t)NFP_TILE_LDS_KB * 1024 : (size_t)kLdsMax;
    const int kCap = (R == 1 || attempt == 1) ? 1024 : 512;   // threads per workgroup
    const int rb_max = std::min(g.H, kCap / Wu - 2 * R);
    for (int rb0 = rb_max; rb0 >= 1; --rb0) {
      int nb = std::max(ceil_div(g.H, rb0), std::min(g.H, ceil_div(NFP_TILE_WGS, g.B)));
      const int rb = ceil_div(g.H, nb);
      nb = ceil_div(g.H, rb);
      const int rows = rb + 2 * R, npos = rows * Wp, npu = rows * Wu, nbp = rb * g.W;
      int lg = 0;
      while (lg < 5 && (2 << lg) * npu <= kCap && (2 << lg) <= g.C / 4) ++lg;
      const int G = 1 << lg, T = ((G * npu + 63) / 64) * 64;
      const int ppb = nfp::band_row_slots(npos, lg);
      const size_t tail = (size_t)(NF + 1) * npu * 4 + (POOL ? (size_t)N * nbp * 4 : 0);
      if (tail + (size_t)ppb * 16 > budget) continue;
      int ncq = (int)((budget - tail) / ((size_t)ppb * 16));
      if (NHWC) {
        ncq = std::min(ncq, nfp::kFwdKN * T / npu);
      } else {
        ncq = std::min(ncq, nfp::kFwdKB * T / (rows * ((g.W + 3) / 4)));
        ncq = std::min(ncq, nfp::kFwdKR * T / (rows * 2 * R));
      }
      if (ncq < 1) continue;
      const int total = g.C / 4, nch = ceil_div(total, ncq);
      g.Cc = 4 * ceil_div(total, nch);
      g.G = G;
      g.Tc = lg;
      const size_t lds = (size_t)(g.Cc / 4) * ppb * 16 + tail;
      nfp::TileGeo tg = {rb, nb, Wp, 1};
      if (nb_out) *nb_out = nb;
      snprintf(g_variant, sizeof(g_variant), "fwd_tile<R%d,%s,%s,%s%s>x%d", R, hot_name(g), BF ? "bf16" : "f32",
               NHWC ? "nhwc" : "nchw", POOL ? ",pool" : "", nb);
      return launch("fwd_tile", fwd_tile<R, M, BF, NHWC, POOL>, dim3((unsigned)(g.B * nb)), dim3(T), lds, st, g, tg, x, out, saved,
                    part);
    }
  }
  return kNotApplicable;
}

template <int R, int M, bool BF, bool NHWC, bool POOL = false>
int launch_bwd_tile_t(KP g, const void* x, const void* go, const void* out, const float* saved, void* gx, hipStream_t st,
                      const float* ggap = nullptr, const float* gnfpm = nullptr) {
  constexpr int N = Win<R>::N;
  const int Wp = nfp::tile_row_stride(g.W, R), Wu = g.W + 2 * R, pvw = M == NFP_COSINE ? 2 : 1;
  const int rb_max = std::min(g.H, 512 / g.W);
  if (rb_max < 1) return kNotApplicable;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t budget = attempt == 0 ? (size_t)NFP_TILE_LDS_KB * 1024 : (size_t)kLdsMax;
    for (int rb0 = rb_max; rb0 >= 1; --rb0) {
      int nb = std::max(ceil_div(g.H, rb0), std::min(g.H, ceil_div(NFP_TILE_WGS, g.B)));
      const int rb = ceil_div(g.H, nb);
      nb = ceil_div(g.H, rb);
      const int rows = rb + 2 * R, npos = rows * Wp, npu = rows * Wu, nbp = rb * g.W, npA = std::min(g.H, rows) * g.W;
      const size_t fixed = ((size_t)(npu + nbp) * 4 + 15) & ~(size_t)15;   // ipn, dfn (the window weights live in registers)
      const size_t pv = (size_t)N * npA * pvw * 4;
      const int ppb = npos | 1;
      if (fixed + std::max(pv, (size_t)ppb * 16) > budget) continue;
      // channel blocks: enough workgroups to fill the chip when images x bands do not
      int S = ceil_div(NFP_TILE_WGS, g.B * nb);
      if (S > g.C / 4) S = g.C / 4;
      if (S < 1) S = 1;
      g.Cwg = round4(ceil_div(g.C, S));
      S = ceil_div(g.C, g.Cwg);
      g.G = std::max(1, std::min(512 / nbp, g.Cwg / 4));
      const int T = ((nbp * g.G + 63) / 64) * 64;
      int ncq = (int)((budget - fixed) / ((size_t)ppb * 16));
      if (NHWC) {
        ncq = std::min(ncq, nfp::kBwdKN * T / npu);
      } else {
        ncq = std::min(ncq, nfp::kBwdKB * T / (rows * ((g.W + 3) / 4)));
        ncq = std::min(ncq, nfp::kBwdKR * T / (rows * 2 * R));
      }
      if (ncq < 1) continue;
      const int total = g.Cwg / 4, nch = ceil_div(total, ncq);
      g.Cc = 4 * ceil_div(total, nch);
      g.G = even_groups(g.Cc / 4, g.G);
      const size_t lds = fixed + std::max(pv, (size_t)(g.Cc / 4) * ppb * 16);
      nfp::TileGeo tg = {rb, nb, Wp, S};
      snprintf(g_variant, sizeof(g_variant), "bwd_tile<R%d,%s,%s,%s%s>x%d", R, hot_name(g), BF ? "bf16" : "f32",
               NHWC ? "nhwc" : "nchw", POOL ? ",pool" : "", nb);
      if constexpr (M == NFP_COSINE) {
        if (g.gfc)
          return launch("bwd_tile", bwd_tile<R, M, BF, NHWC, POOL, true>, dim3((unsigned)(g.B * nb * S)), dim3(T), lds, st, g, tg, x, go,
                        out, saved, gx, ggap, gnfpm);
      }
      return launch("bwd_tile", bwd_tile<R, M, BF, NHWC, POOL>, dim3((unsigned)(g.B * nb * S)), dim3(T), lds, st, g, tg, x, go, out,
                    saved, gx, ggap, gnfpm);
    }
  }
  return kNotApplicable;
}

template <int R, int M, bool POOL>
int fwd_rm(const KP& g, const void* x, void* out, float* saved, hipStream_t st, float* part, int* nb) {
  const bool bf = g.dtype == NFP_BF16, nhwc = !g.contig;
  if (bf) return nhwc ? launch_fwd_tile_t<R, M, true, true, POOL>(g, x, out, saved, st, part, nb)
                      : launch_fwd_tile_t<R, M, true, false, POOL>(g, x, out, saved, st, part, nb);
  return nhwc ? launch_fwd_tile_t<R, M, false, true, POOL>(g, x, out, saved, st, part, nb)
              : launch_fwd_tile_t<R, M, false, false, POOL>(g, x, out, saved, st, part, nb);
}
template <int R, int M, bool POOL>
int bwd_rm(const KP& g, const void* x, const void* go, const void* out, const float* saved, void* gx, hipStream_t st,
           const float* ggap, const float* gnfpm) {
  const bool bf = g.dtype == NFP_BF16, nhwc = !g.contig;
  if (bf) return nhwc ? launch_bwd_tile_t<R, M, true, true, POOL>(g, x, go, out, saved, gx, st, ggap, gnfpm)
                      : launch_bwd_tile_t<R, M, true, false, POOL>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  return nhwc ? launch_bwd_tile_t<R, M, false, true, POOL>(g, x, go, out, saved, gx, st, ggap, gnfpm)
              : launch_bwd_tile_t<R, M, false, false, POOL>(g, x, go, out, saved, gx, st, ggap, gnfpm);
}

}  // namespace

int tile_forward(const KP& g, const void* x, void* out, float* saved, hipStream_t st, bool pool, float* part, int* nb) {
  if (!tile_ok(g, x, x)) return kNotApplicable;
  const bool cosv = hot_product(g);
  if (pool) {
    if (cosv) return g.R == 1 ? fwd_rm<1, NFP_COSINE, true>(g, x, out, saved, st, part, nb) : fwd_rm<2, NFP_COSINE, true>(g, x, out, saved, st, part, nb);
    return g.R == 1 ? fwd_rm<1, NFP_NORM, true>(g, x, out, saved, st, part, nb) : fwd_rm<2, NFP_NORM, true>(g, x, out, saved, st, part, nb);
  }
  if (cosv) return g.R == 1 ? fwd_rm<1, NFP_COSINE, false>(g, x, out, saved, st, part, nb) : fwd_rm<2, NFP_COSINE, false>(g, x, out, saved, st, part, nb);
  return g.R == 1 ? fwd_rm<1, NFP_NORM, false>(g, x, out, saved, st, part, nb) : fwd_rm<2, NFP_NORM, false>(g, x, out, saved, st, part, nb);
}

int tile_backward(const KP& g, const void* x, const void* go, const void* out, const float* saved, void* gx, hipStream_t st,
                  bool pool, const float* ggap, const float* gnfpm) {
  if (!tile_ok(g, x, gx)) return kNotApplicable;
  const bool cosv = hot_product(g);
  if (pool) {
    if (cosv) return g.R == 1 ? bwd_rm<1, NFP_COSINE, true>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_COSINE, true>(g, x, go, out, saved, gx, st, ggap, gnfpm);
    return g.R == 1 ? bwd_rm<1, NFP_NORM, true>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_NORM, true>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  }
  if (cosv) return g.R == 1 ? bwd_rm<1, NFP_COSINE, false>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_COSINE, false>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  return g.R == 1 ? bwd_rm<1, NFP_NORM, false>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_NORM, false>(g, x, go, out, saved, gx, st, ggap, gnfpm);
}

int tile_pool_fold(const KP& g, const float* part, float* gap, float* nfpm, int nb, hipStream_t st) {
  const long long n = (long long)g.B * (g.C + g.N);
  return launch("pool_fold", pool_fold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, gap, nfpm, g.B, nb, g.C, g.N,
                g.invP);
}

}  // namespace nfp_host

#ifdef NFP_STAMPS
// (every translation unit has its own copy of the device-side stamp pointer)
extern "C" int nfp_debug_set_stamp_buffer_tile(void* dev_ptr) {
  unsigned long long* p = (unsigned long long*)dev_ptr;
  return hipMemcpyToSymbol(HIP_SYMBOL(nfp::nfp_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif
