// nfp_tile.hip — second translation unit of libnfp_hip.so: the row-band kernels of nfp_tile.h and their launchers
// (compiled in parallel with nfp_hip.hip; shared host-side machinery in nfp_launch.h).
#include "nfp_launch.h"
#include "nfp_tile.h"

#include <type_traits>

using namespace nfp;

namespace nfp_host {
namespace {

// ---- row-band kernels on a padded slab (nfp_tile.h): "same" maps of any size, no tables ----------------------------
#ifndef NFP_TILE_WGS
#define NFP_TILE_WGS 512
#endif
#ifndef NFP_TILE_THREADS
#define NFP_TILE_THREADS 480   // threads per workgroup the band height aims at
#endif
#ifndef NFP_TILE_LDS_KB
#define NFP_TILE_LDS_KB 78     // per workgroup, so that two share a compute unit
#endif
int tile_wgs() { return g_sw.tile_wgs > 0 ? (int)g_sw.tile_wgs : NFP_TILE_WGS; }
size_t tile_lds() { return (size_t)(g_sw.tile_lds_kb > 0 ? (int)g_sw.tile_lds_kb : NFP_TILE_LDS_KB) * 1024; }
int tile_max_grid() { return g_sw.tile_max_grid > 0 ? std::min((int)g_sw.tile_max_grid, nfp::kTileMaxGrid) : nfp::kTileMaxGrid; }
int tile_cap(int dflt) { return g_sw.tile_cap > 0 ? std::min((int)g_sw.tile_cap, 1024) : dflt; }
bool tile_geometry(const KP& g) {
  if (g.stride != 1 || g.dil != 1 || g.pad != g.R || g.mode == NFP_PAD_CIRCULAR || g.rs == 12) return false;
  if (g.R != 1 && g.R != 2) return false;
  // one thread per padded position: the smallest band (R + 1 rows and its halo) must fit a workgroup
  if (g.W < 2 * g.R + 2 || g.H < g.R + 1) return false;
  return (g.W + 2 * g.R) * (3 * g.R + 1) <= 1024;
}
}  // namespace

bool tile_ok(const KP& g, const void* x, const void* gx) {
  if (force_generic() || !tile_geometry(g) || (g.C & 3)) return false;
  if (!hot_measure(g) && !hot_l1(g) && !hot_sym(g)) return false;
  const bool nhwc = g.sC == 1 && g.sW == g.C && g.sH == (long long)g.W * g.C;
  if (!g.contig && !nhwc) return false;
  const int es = g.dtype == NFP_F32 ? 4 : 2;
  // 32-bit buffer offsets inside an image, and room for the "reads zero" offset (nfp_tile.h::Oob)
  if ((long long)std::max(g.C, g.N) * g.P * es >= 0x7ffffff0LL) return false;
  if (!g.contig) {
    const uintptr_t m = g.dtype == NFP_F32 ? 15 : 7;
    if (((uintptr_t)x & m) || ((uintptr_t)gx & m) || ((g.sB * es) & m) || ((g.gB * es) & m)) return false;
  }
  return true;
}

namespace {

// Bands.  Every band re-reads and re-sums its 2R halo rows, so rows per band >= 6R keeps that at a third; beyond that,
// SMALLER workgroups win (one thread per padded position: ~480 threads, three or four workgroups per compute unit
// overlapping each other's load / sum / store phases) — measured on the MultiStage maps at B = 256
// (profiles/r03_v_tile_launch_constants.jsonl): 112x112 best at 6 rows (912 threads), 56x56 at 6 (464), 28x28 at 14 (480).
// At least tile_wgs() workgroups on the chip; every band at least R + 1 rows (the ring rows a border pixel collects
// from lie in its own band's halo).  Returns the rows of the largest band, or 0.
int tile_bands(const KP& g, int rb_cap, int& nb) {
  if (rb_cap < std::min(g.H, g.R + 1)) return 0;
  const int nb_max = std::max(1, g.H / (g.R + 1));
  nb = std::max(ceil_div(g.H, rb_cap), std::min(nb_max, ceil_div(tile_wgs(), g.B)));
  if (nb > nb_max) return 0;
  return ceil_div(g.H, nb);
}
int tile_rows_wanted(const KP& g, int cap) {
  const int Wu = g.W + 2 * g.R;
  const int want = std::max(6 * g.R, NFP_TILE_THREADS / Wu - 2 * g.R);
  return std::max(1, std::min(std::min(g.H, cap / Wu - 2 * g.R), want));
}
// channel groups per position: one, unless the chip would be short of threads (small batches)
int tile_groups(const KP& g, int nb, int npu, int cap, int cq) {
  int lg = 0;
  while (lg < 5 && (2 << lg) * npu <= cap && (2 << lg) <= cq && (long long)g.B * nb * npu * (1 << lg) < 128LL * 1024) ++lg;
  return lg;
}

template <int R, int M, bool BF, bool NHWC, bool POOL = false>
int launch_fwd_tile_t(KP g, const void* x, void* out, float* saved, hipStream_t st, float* part = nullptr, int* nb_out = nullptr,
                      float* gap = nullptr, float* nfpm = nullptr) {
  constexpr int NF = Win<R>::NF, N = Win<R>::N;
  const int Wu = g.W + 2 * R;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // first choice: two workgroups of up to 1024 threads per compute unit — half of LDS each, 64 registers.
    // Second: one workgroup and all of LDS.
    const size_t budget = attempt == 0 ? tile_lds() : (size_t)kLdsMax;
    const int kCap = tile_cap(1024);                           // threads per workgroup (64 registers either radius)
    for (int rb0 = tile_rows_wanted(g, kCap); rb0 >= 1; --rb0) {
      int nb = 1;
      const int rb = tile_bands(g, rb0, nb);
      if (rb < 1) break;
      const int rows = rb + 2 * R, npu = rows * Wu, nbp = rb * g.W;
      const int lg = tile_groups(g, nb, npu, kCap, g.C / 4), G = 1 << lg;
      const int ppb = nfp::band_row_slots((npu + 3) & ~3, lg);
      const size_t tail = 16 + (size_t)(NF + 1) * npu * 4;                      // spare slot, pair sums, factors
      const size_t vmb = POOL ? (((size_t)N * nbp + 3) / 4) * 16 : 0;            // pooled: the band's maps, over the slab
      constexpr size_t SB = BF ? 8 : 16;   // bytes of a slab slot (four channels of a position in the storage type: nfp_tile.h::Quad)
      if (tail + std::max((size_t)ppb * SB, vmb) > budget) continue;
      int ncq = (int)((budget - tail) / ((size_t)ppb * SB));
      ncq = std::min(ncq, std::min(nfp::Quad<BF>::KQ * G, g.C / 4));
      if (ncq < 1) continue;
      const int total = g.C / 4, nch = ceil_div(total, ncq);
      g.Cc = 4 * ceil_div(total, nch);
      g.G = G;
      g.Tc = -1;
      if (NHWC && G == 1) {   // wavefront-shared staging: chunks of 1, 2, 4 (bf16: 8) quads (nfp_tile.h::TileStage)
        g.Tc = (BF && ncq >= 8) ? 3 : (ncq >= 4 ? 2 : (ncq >= 2 ? 1 : 0));
        g.Cc = 4 << g.Tc;
      }
      size_t lds = std::max((size_t)(g.Cc / 4) * ppb * SB, vmb) + tail;
      // Round 4 (default: bf16 maps of >= 64 channels, where it measured faster; NFP_TILE_DMA=1 / 0 force it on / off —
      // nfp_launch.h): channels-last, one thread per position, plain maps -> LDS-DMA into a position-major slab in the
      // storage type (nfp_tile.h::fwd_tile, DMA): PC =
      // 16-byte pieces per position and chunk, the largest power of two <= 8 that divides a pixel's pieces and fits (two
      // slabs when there are several chunks).
      bool dma = false;
      const int dma_sw = g_sw.tile_dma.load(std::memory_order_relaxed);
      const bool dma_auto = BF && (g.C * 2) % 128 == 0;   // (8 pieces per position and chunk: where it measured faster)
      if ((dma_sw == 1 || (dma_sw < 0 && dma_auto)) && M != nfp::kSymTerm && NHWC && !POOL && G == 1 && !g.gfc && (g.C * (BF ? 2 : 4)) % 16 == 0 &&
          !(((uintptr_t)x) & 15) && !((g.sB * (BF ? 2 : 4)) & 15)) {
        const int tp = g.C * (BF ? 2 : 4) / 16, npu64 = (npu + 63) & ~63;
        for (int lpc = 3; lpc >= 0; --lpc) {
          const int pc = 1 << lpc;
          if (tp % pc) continue;
          const size_t slabs = (size_t)(tp > pc ? 2 : 1) * npu64 * pc * 16;
          const size_t slack = (size_t)(R * Wu + R + 1) * pc * 16;   // (taps past the band read whatever lies there)
          if (slabs + tail + slack > budget) continue;
          g.Tc = lpc;
          g.Cc = pc * (BF ? 8 : 4);
          lds = slabs + tail + slack;
          dma = true;
          break;
        }
      }
      nfp::TileGeo tg = {nb, rows, Wu, ppb, 1, g.H / nb, g.H % nb, (int)(lds / 4)};
      if (nb_out) *nb_out = POOL ? nb * nfp::kPoolSub : nb;   // (pooled: rows of partial sums per image)
      snprintf(g_variant, sizeof(g_variant), "fwd_tile<R%d,%s,%s,%s%s%s>x%d", R, hot_name(g), BF ? "bf16" : "f32",
               NHWC ? "nhwc" : "nchw", dma ? ",dma" : "", POOL ? ",pool" : "", nb);
      const dim3 block(G, Wu, rows);
      // (tile_ids' exact range: batches beyond kTileMaxGrid workgroups go out as several launches, images in order)
      const int es = BF ? 2 : 4, bmax = std::max(8, (tile_max_grid() / nb) & ~7);
      for (int b0 = 0; b0 < g.B; b0 += bmax) {
        KP gs = g;
        gs.B = std::min(bmax, g.B - b0);
        const dim3 grid((unsigned)(gs.B * nb));
        const void* xs = (const char*)x + (long long)b0 * g.sB * es;
        void* os = (char*)out + (long long)b0 * N * g.P * es;
        float* ss = saved ? saved + (long long)b0 * g.P : nullptr;
        float* ps = part ? part + (long long)b0 * nb * nfp::kPoolSub * (g.C + N) : nullptr;
        float* gs_ = gap ? gap + (long long)b0 * g.C : nullptr;
        float* ns_ = nfpm ? nfpm + (long long)b0 * N : nullptr;
        int rc;
        if (M == NFP_COSINE && g.gfc) {
          rc = launch("fwd_tile", fwd_tile<R, M, BF, NHWC, POOL, M == NFP_COSINE>, grid, block, lds, st, gs, tg, xs, os, ss, ps, gs_, ns_);
        } else if (dma) {
          if constexpr (NHWC && !POOL)
            rc = launch("fwd_tile_dma", fwd_tile<R, M, BF, true, false, false, true>, grid, block, lds, st, gs, tg, xs, os, ss, ps, gs_, ns_);
          else
            rc = kNotApplicable;
        } else {
          rc = launch("fwd_tile", fwd_tile<R, M, BF, NHWC, POOL, false>, grid, block, lds, st, gs, tg, xs, os, ss, ps, gs_, ns_);
        }
        if (rc != NFP_OK) return rc;
      }
      return NFP_OK;
    }
  }
  return kNotApplicable;
}

template <int R, int M, bool BF, bool NHWC, bool POOL = false>
int launch_bwd_tile_t(KP g, const void* x, const void* go, const void* out, const float* saved, void* gx, hipStream_t st,
                      const float* ggap = nullptr, const float* gnfpm = nullptr) {
  constexpr int N = Win<R>::N, K2 = Win<R>::K2;
  const int Wu = g.W + 2 * R;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // first choice: two workgroups per compute unit (k = 3: 64 registers, up to 1024 threads each; k = 5: 128, up to 512)
    const size_t budget = attempt == 0 ? tile_lds() : (size_t)kLdsMax;
    const int kCap = tile_cap((R == 1 || attempt == 1) ? 1024 : 512);
    for (int rb0 = tile_rows_wanted(g, kCap); rb0 >= 1; --rb0) {
      int nb = 1;
      const int rb = tile_bands(g, rb0, nb);
      if (rb < 1) break;
      const int rows = rb + 2 * R, npu = rows * Wu, PL = (rows + 2 * R) * Wu;
      const size_t fixed = (size_t)((PL + 3) & ~3) * 4;                       // ipn
      const size_t pv = (size_t)(N * PL + 4) * 4;                             // pair values, every plane with its zero rows; guard
      const size_t wr = 16 + (size_t)((rows * 2 * R + 2 * R * g.W) * K2 + 4 * K2) * 4;   // spare slot, ring rows (+ slack)
      if (fixed + pv > budget) continue;
      // channel blocks: enough workgroups to fill the chip when images x bands do not
      int S = ceil_div(tile_wgs(), g.B * nb);
      if (S > g.C / 4) S = g.C / 4;
      if (S < 1) S = 1;
      g.Cwg = round4(ceil_div(g.C, S));
      S = ceil_div(g.C, g.Cwg);
      const int lg = tile_groups(g, nb * S, npu, kCap, g.Cwg / 4), G = 1 << lg;
      const int ppb = nfp::band_row_slots((npu + 3) & ~3, lg);
      constexpr size_t SB = BF ? 8 : 16;   // bytes of a slab slot (nfp_tile.h::Quad)
      if (fixed + wr + (size_t)ppb * SB > budget) continue;
      int ncq = (int)((budget - fixed - wr) / ((size_t)ppb * SB));
      ncq = std::min(ncq, std::min(nfp::Quad<BF>::KQ * G, g.Cwg / 4));
      if (ncq < 1) continue;
      const int total = g.Cwg / 4, nch = ceil_div(total, ncq);
      g.Cc = 4 * ceil_div(total, nch);
      g.G = G;
      g.Tc = -1;
      if (NHWC && G == 1) {
        g.Tc = (BF && ncq >= 8) ? 3 : (ncq >= 4 ? 2 : (ncq >= 2 ? 1 : 0));
        g.Cc = 4 << g.Tc;
      }
      const size_t lds = fixed + std::max(pv, (size_t)(g.Cc / 4) * ppb * SB + wr);
      nfp::TileGeo tg = {nb, rows, Wu, ppb, S, g.H / nb, g.H % nb, (int)(lds / 4)};
      const dim3 block(G, Wu, rows);
      // dense grad_x stores through LDS (nfp_tile.h, phase B): channels-last with one thread per position, pixels of 256
      // bytes and more (below, the scattered 16-byte stores of a pixel complete their cache lines soon enough: measured
      // at 96 / 160 bytes, profiles/r03_w_tile_backward_stores_ab.txt), workgroups of up to 640 threads (the variant
      // takes 96 registers: two such workgroups per compute unit)
      // (round 4, PMC: the scattered stores of bf16 pixels leave as partial sectors — WRITE_SIZE 2.6x the bytes of grad_x at
      // [256,24,56,56] bf16, profiles/r04_j_…; same-box A/B of the dense form per pixel size, profiles/r04_k_…: bf16 128-byte
      // pixels 123.6 -> 102.5 us, 80-byte 23.7 -> 20.9; 48-byte pixels and float32 below 256 bytes lose or tie)
#ifndef NFP_TILE_CST_MINB
#define NFP_TILE_CST_MINB (BF ? 80 : 256)
#endif
      const bool cst = NHWC && G == 1 && g.C * (BF ? 2 : 4) >= NFP_TILE_CST_MINB && npu <= 640;
      snprintf(g_variant, sizeof(g_variant), "bwd_tile<R%d,%s,%s,%s%s%s>x%d", R, hot_name(g), BF ? "bf16" : "f32",
               NHWC ? "nhwc" : "nchw", cst ? ",dense" : "", POOL ? ",pool" : "", nb);
      const std::true_type T_;
      const std::false_type F_;
      const int es = BF ? 2 : 4, bmax = std::max(8, (tile_max_grid() / (nb * S)) & ~7);
      for (int b0 = 0; b0 < g.B; b0 += bmax) {   // (tile_ids' exact range: see launch_fwd_tile_t)
        KP gs = g;
        gs.B = std::min(bmax, g.B - b0);
        const dim3 grid((unsigned)(gs.B * nb * S));
        const void* xs = (const char*)x + (long long)b0 * g.sB * es;
        const void* gos = go ? (const char*)go + (long long)b0 * N * g.P * es : nullptr;
        const void* os = (const char*)out + (long long)b0 * N * g.P * es;
        const float* ss = saved ? saved + (long long)b0 * g.P : nullptr;
        void* gxs = (char*)gx + (long long)b0 * g.gB * es;
        const float* ggs = ggap ? ggap + (long long)b0 * g.C : nullptr;
        const float* gns = gnfpm ? gnfpm + (long long)b0 * N : nullptr;
        auto go_ = [&](auto gfc, auto cstv) {
          return launch("bwd_tile", bwd_tile<R, M, BF, NHWC, POOL, decltype(gfc)::value, decltype(cstv)::value>, grid, block, lds, st,
                        gs, tg, xs, gos, os, ss, gxs, ggs, gns);
        };
        int rc;
        if (M == NFP_COSINE && g.gfc) {
          if constexpr (M == NFP_COSINE) {
            if constexpr (NHWC) rc = cst ? go_(T_, T_) : go_(T_, F_);
            else rc = go_(T_, F_);
          } else {
            rc = go_(F_, F_);
          }
        } else {
          if constexpr (NHWC) rc = cst ? go_(F_, T_) : go_(F_, F_);
          else rc = go_(F_, F_);
        }
        if (rc != NFP_OK) return rc;
      }
      return NFP_OK;
    }
  }
  return kNotApplicable;
}

template <int R, int M, bool POOL>
int fwd_rm(const KP& g, const void* x, void* out, float* saved, hipStream_t st, float* part, int* nb, float* gap = nullptr,
           float* nfpm = nullptr) {
  const bool bf = g.dtype == NFP_BF16, nhwc = !g.contig;
  if (bf) return nhwc ? launch_fwd_tile_t<R, M, true, true, POOL>(g, x, out, saved, st, part, nb, gap, nfpm)
                      : launch_fwd_tile_t<R, M, true, false, POOL>(g, x, out, saved, st, part, nb, gap, nfpm);
  return nhwc ? launch_fwd_tile_t<R, M, false, true, POOL>(g, x, out, saved, st, part, nb, gap, nfpm)
              : launch_fwd_tile_t<R, M, false, false, POOL>(g, x, out, saved, st, part, nb, gap, nfpm);
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

int tile_forward(const KP& g, const void* x, void* out, float* saved, hipStream_t st, bool pool, float* part, int* nb,
                 float* gap, float* nfpm) {
  if (!tile_ok(g, x, x)) return kNotApplicable;
  const bool cosv = hot_product(g);
  if (hot_l1(g)) {   // Norm p = 1 (the class default, nfp.py:16) and EMD (make_kp): plain maps
    if (pool) return kNotApplicable;
    return g.R == 1 ? fwd_rm<1, kNormP1, false>(g, x, out, saved, st, part, nb) : fwd_rm<2, kNormP1, false>(g, x, out, saved, st, part, nb);
  }
  if (hot_sym(g)) {   // one instantiation for the four of them (nfp_measures.h::kSymTerm)
    // (below 14 x 14 the any-geometry forward spreads the same arithmetic over more threads: [64,512,7,7] 9 vs 14 us; the
    // backward wins at every size: 17.6-22.9 vs 20.9-28.6 us there — profiles/r04_zd_…)
    if (pool || g.P < 196) return kNotApplicable;
    return g.R == 1 ? fwd_rm<1, kSymTerm, false>(g, x, out, saved, st, part, nb) : fwd_rm<2, kSymTerm, false>(g, x, out, saved, st, part, nb);
  }
  if (pool) {
    if (cosv) return g.R == 1 ? fwd_rm<1, NFP_COSINE, true>(g, x, out, saved, st, part, nb, gap, nfpm)
                              : fwd_rm<2, NFP_COSINE, true>(g, x, out, saved, st, part, nb, gap, nfpm);
    return g.R == 1 ? fwd_rm<1, NFP_NORM, true>(g, x, out, saved, st, part, nb, gap, nfpm)
                    : fwd_rm<2, NFP_NORM, true>(g, x, out, saved, st, part, nb, gap, nfpm);
  }
  if (cosv) return g.R == 1 ? fwd_rm<1, NFP_COSINE, false>(g, x, out, saved, st, part, nb) : fwd_rm<2, NFP_COSINE, false>(g, x, out, saved, st, part, nb);
  return g.R == 1 ? fwd_rm<1, NFP_NORM, false>(g, x, out, saved, st, part, nb) : fwd_rm<2, NFP_NORM, false>(g, x, out, saved, st, part, nb);
}

int tile_backward(const KP& g, const void* x, const void* go, const void* out, const float* saved, void* gx, hipStream_t st,
                  bool pool, const float* ggap, const float* gnfpm) {
  if (!tile_ok(g, x, gx)) return kNotApplicable;
  const bool cosv = hot_product(g);
  if (hot_l1(g)) {
    if (pool) return kNotApplicable;
    return g.R == 1 ? bwd_rm<1, kNormP1, false>(g, x, go, out, saved, gx, st, ggap, gnfpm)
                    : bwd_rm<2, kNormP1, false>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  }
  if (hot_sym(g)) {
    if (pool) return kNotApplicable;
    return g.R == 1 ? bwd_rm<1, kSymTerm, false>(g, x, go, out, saved, gx, st, ggap, gnfpm)
                    : bwd_rm<2, kSymTerm, false>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  }
  if (pool) {
    if (cosv) return g.R == 1 ? bwd_rm<1, NFP_COSINE, true>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_COSINE, true>(g, x, go, out, saved, gx, st, ggap, gnfpm);
    return g.R == 1 ? bwd_rm<1, NFP_NORM, true>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_NORM, true>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  }
  if (cosv) return g.R == 1 ? bwd_rm<1, NFP_COSINE, false>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_COSINE, false>(g, x, go, out, saved, gx, st, ggap, gnfpm);
  return g.R == 1 ? bwd_rm<1, NFP_NORM, false>(g, x, go, out, saved, gx, st, ggap, gnfpm) : bwd_rm<2, NFP_NORM, false>(g, x, go, out, saved, gx, st, ggap, gnfpm);
}

int tile_pool_fold(const KP& g, const float* part, float* gap, float* nfpm, int nb, hipStream_t st) {
  const long long n = (long long)g.B * (g.C + g.N);
  return launch("pool_fold", pool_fold<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, gap, nfpm, g.B, nb, g.C, g.N,
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
