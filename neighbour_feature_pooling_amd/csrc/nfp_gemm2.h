// nfp_gemm2.h — the matrix-core backward with a TABLE-FREE phase A (round 4; config 5: ViT-Tiny tokens, bf16, k = 5).
//
// bwd_fast<…,GEMM> (nfp_fast.h) builds the window weights W[r][j] from the workspace's link tables: per-pair values to
// LDS, then one gathered entry per thread and round (config 5: 4 704 pair values, 2 548 entries of up to 16 links), a fold
// phase — 6.0 of the kernel's 15.9 us (in-kernel stamps, profiles/r02_j_matrix_core_backward_ab.txt).  The row-band kernels
// of nfp_tile.h build the same weights with ONE THREAD PER PADDED POSITION and no tables: a position loads its own N
// gradients / similarities, publishes one float per tap, and reads the opposite taps of the positions under its window at
// CONSTANT offsets; ring positions (copies of the pixel they fold onto) hand their windows over through LDS.  That is a
// few hundred instructions per position and three barriers.  Here the two are joined: phase A as in bwd_tile (the whole
// image is one "band"), phase B = nfp_fast.h::bwd_gemm_phase unchanged — W·X on the matrix cores, W split hi / lo into
// two bf16 images.  The wavefronts that hold no position have their share of the x block in flight meanwhile.
//
// LDS: Wt [P][K2] (live to the end) | phase A's planes: ipn [PL] | pair values [N][PL] | guard | ring windows — all dead
// when phase B starts, whose operand images (Xt, Wd, grad(GAP) of the block) lie over them.
#pragma once
#include "nfp_tile.h"

namespace nfp {

// floats of phase A's LDS behind Wt (host and device agree through this one function)
template <int R>
__host__ __device__ inline int gemm2_plane_floats(int H, int W) {
  constexpr int N = Win<R>::N, K2 = Win<R>::K2;
  const int Wu = W + 2 * R, rows = H + 2 * R, PL = (rows + 2 * R) * Wu;
  return ((PL + 3) & ~3) + N * PL + 4 + 4 * K2 + (rows * 2 * R + 2 * R * W) * K2 + 4 * K2;
}

template <int R, int M, bool NHWC, bool POOL>
__global__ void __launch_bounds__(1024) bwd_gemm2(const KP g, const void* __restrict__ x, const void* __restrict__ go,
                                                   const void* __restrict__ out, const float* __restrict__ saved,
                                                   void* __restrict__ gx, const float* __restrict__ ggap,
                                                   const float* __restrict__ gnfpm) {
  constexpr int N = Win<R>::N, K = Win<R>::K, K2 = Win<R>::K2;
  constexpr bool BF = true;
  constexpr int ES = 2;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  lds_poison(lds4, g.Ow);   // (-DNFP_LDS_POISON test build only; g.Ow: 32-bit words of this launch's LDS)
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int cb0 = blockIdx.y * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  const int W = g.W, H = g.H, P = g.P;
  const int Wu = W + 2 * R, rows = H + 2 * R, npu = rows * Wu, PL = (rows + 2 * R) * Wu;
  float* Wt = (float*)lds4;                                   // [P][K2] window weights for phase B
  float* A0 = Wt + ((P * K2 + 3) & ~3);                       // phase A's planes
  float* ipn = A0 + R * Wu;                                   // ipn[v], margins at v < 0 and v >= npu
  float* pvb = A0 + ((PL + 3) & ~3);
  float* PV = pvb + R * Wu;                                   // plane n at PV + n * PL
  float* Wr = pvb + N * PL + 4 + 2 * K2;                      // ring windows (2 K2 floats of slack either side)
  // phase B's operand images over phase A's planes
  uint4* gemm_Xt = (uint4*)A0;
  uint4* gemm_Wd = gemm_Xt + (long long)(cb1 - cb0) * gemm_xq(P);
  const uint16_t* x16 = (const uint16_t*)x + (long long)b * g.sB;
  float* gg_s = nullptr;
  if constexpr (POOL) {
    const int band = R * W + R, KW = gemm_kw(band);
    gg_s = (float*)(gemm_Wd + (long long)g.Tc * 2 * 32 * odd_up(2 * KW + 1));
  }
  const Rsrc gob = make_rsrc((const char*)go + (long long)b * N * P * ES, POOL ? 0 : (long long)N * P * ES);
  const Rsrc outb = make_rsrc((const char*)out + (long long)b * N * P * ES, (long long)N * P * ES);
  const Rsrc svb = make_rsrc((const char*)saved + (long long)b * P * 4, (M == NFP_COSINE && !g.unit) ? (long long)P * 4 : 0);
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  // ---- the thread's padded position (threads past the last position hold none: they only stage x) -------------------
  const Fold fo(g);
  const bool haspos = t < npu;
  const int vy = fdivi(min(t, npu - 1), Wu), vx = min(t, npu - 1) - vy * Wu, v = vy * Wu + vx;
  const int py = vy - R, px = vx - R;                          // image coordinates (outside the image for ring positions)
  const int sy = fo.y(py, H), sx = fo.x(px, W);
  const bool real = haspos && (unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W;
  const bool zero = !haspos || (sy | sx) < 0;
  const int src = min(max(sy, 0), H - 1) * W + max(sx, 0);
  const int zf = zero ? Oob<false>::e : 0;

  // ---- A1: the position's own pairs (every tap) and its norm factor; then this thread's share of the x block ----------
  float w[K2], sv[N];
  float dfn = 0.f, ipr = 1.f;
  GemmX<NHWC> gxr;
  {
    const int ep = real ? py * W + px : Oob<BF>::e;
    float gov[N];
    const bool ring = M == NFP_COSINE && haspos && !real && !zero;
    const int ea = ring ? py * W + px : ep, rm = ring ? -1 : 0;
    bool rbad[K], cbad[K];
#pragma unroll
    for (int d = 0; d < K; ++d) {
      rbad[d] = ring && (unsigned)(py + d - R) >= (unsigned)H;
      cbad[d] = ring && (unsigned)(px + d - R) >= (unsigned)W;
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1, dy = j / K - R, dx = j % K - R;
      int e = ep + n * P;
      if (M == NFP_COSINE) {
        e = ea + n * P + (((N - 1 - 2 * n) * P + dy * W + dx) & rm);
        e = (rbad[dy + R] || cbad[dx + R]) ? Oob<BF>::e : e;
      }
      sv[n] = load_1<BF>(outb, e, 0);
      gov[n] = POOL ? 0.f : load_1<BF>(gob, ep, n * P);
    }
    float nrm = 0.f;
    if (M == NFP_COSINE) nrm = load_1<false>(svb, src | zf, 0);
    float ggv0 = 0.f, ggv1 = 0.f;
    if constexpr (POOL) {
      const int ncw = cb1 - cb0;
      if (g.pool_gap && t < ncw) ggv0 = ggap[(long long)b * g.C + cb0 + t];
      if (g.pool_gap && t + T < ncw) ggv1 = ggap[(long long)b * g.C + cb0 + t + T];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int part = 0; part < 3; ++part) gemm_x_issue<NHWC>(gxr, g, x16, cb0, cb1 - cb0, t, T, part);
    __builtin_amdgcn_sched_barrier(0);
    NFP_STAMP(1);
    const float sa = g.osa, sb = -g.osa * g.osb;   // s = osa * (out - osb)
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1;
      float gc = gov[n];
      if constexpr (POOL) gc = real ? gnfpm[(long long)b * N + n] * g.invP : 0.f;
      if (M == NFP_COSINE) {
        sv[n] = fmaf(sa, sv[n], sb);
        w[j] = sa * gc;
      } else {
        w[j] = real ? dist_coef(g, gc, sv[n]) : 0.f;
      }
      if (haspos) PV[n * PL + v] = w[j];
    }
    if (M == NFP_COSINE) {
      const float ip = unit_or(g, __builtin_amdgcn_rcpf(fmaxf(nrm, g.eps)));
      ipr = ip;
      if (haspos) ipn[v] = ipr;
      dfn = nrm > 0.f ? -(g.nuf * ip) * __builtin_amdgcn_rcpf(nrm) : 0.f;
    }
    // the zero rows above and below the image's padded band, every plane (and the guard words)
    if (haspos && (vy < R || vy >= rows - R)) {
      const int gw = min(v, 3);
      A0[PL + gw] = 0.f;
      pvb[N * PL + gw] = 0.f;
      const int m = vy < R ? v - R * Wu : v + R * Wu;
      if (M == NFP_COSINE) ipn[m] = 0.f;
#pragma unroll
      for (int n = 0; n < N; ++n) PV[n * PL + m] = 0.f;
    }
    if constexpr (POOL) {
      // (gg_s lies behind Wd, beyond phase A's planes: launcher)
      const int ncw = cb1 - cb0;
      if (t < ncw) gg_s[t] = ggv0 * g.invP;
      if (t + T < ncw) gg_s[t + T] = ggv1 * g.invP;
      for (int i = t + 2 * T; i < ncw; i += T) gg_s[i] = g.pool_gap ? ggap[(long long)b * g.C + cb0 + i] * g.invP : 0.f;
    }
  }
  __syncthreads();
  NFP_STAMP(2);

  // ---- A2: the window weights of this thread's position, in registers ---------------------------------------------------
  const float dneg = g.diff ? -1.f : 0.f;
  float D = 0.f;
  if (haspos) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1, dy = j / K - R, dx = j % K - R, opp = N - 1 - n;
      const float c2 = (PV + opp * PL + v + dy * Wu - R)[dx + R];
      if (M == NFP_COSINE) {
        const float ipq = (ipn + v + dy * Wu - R)[dx + R];
        const float S = w[j] + c2;
        D = fmaf(S, sv[n], D);
        w[j] = ipr * ipq * S;
      } else {
        const float c1 = w[j];
        D += fmaf(-dneg, c1, c2);
        w[j] = dneg * (c1 + c2);
      }
    }
  }
  w[K2 / 2] = (M == NFP_COSINE ? dfn : 1.f) * D;
  __syncthreads();  // the pair values are dead
  NFP_STAMP(3);
  // a ring position is a copy of the image pixel it folds onto: its window goes to LDS for that pixel's thread
  if (haspos && !real && !zero) {
    float* wrow = Wr + ring_slot<R>(vy, py, px, rows, W, H) * K2;
#pragma unroll
    for (int j = 0; j < K2; ++j) wrow[j] = w[j];
  }
  __syncthreads();
  if (real) {
    constexpr int KK = 2 * R + 1;
    unsigned my = 1u, mx = 1u;
#pragma unroll
    for (int i = 1; i < KK; ++i) {
      const int ry = i <= R ? -i : H - 1 + (i - R), rx = i <= R ? -i : W - 1 + (i - R);
      my |= (fo.y(ry, H) == py ? 1u : 0u) << i;
      mx |= (fo.x(rx, W) == px ? 1u : 0u) << i;
    }
    unsigned mask = 0u;
#pragma unroll
    for (int c = 1; c < KK * KK; ++c) mask |= (((my >> (c / KK)) & (mx >> (c % KK)) & 1u)) << c;
    while (mask != 0u) {
      const int c = __builtin_ctz(mask);
      mask &= mask - 1u;
      const int iy = fdivi(c, KK), ix = c - iy * KK;
      const int uy = iy == 0 ? py : (iy <= R ? -iy : H - 1 + (iy - R));
      const int ux = ix == 0 ? px : (ix <= R ? -ix : W - 1 + (ix - R));
      const int dy_ = py - uy, dx_ = px - ux;                       // r - u: slot j of r is slot j + (dy_, dx_) of u
      const float* wu = Wr + ring_slot<R>(uy + R, uy, ux, rows, W, H) * K2 + dy_ * K + dx_;
      bool oky[K], okx[K];
#pragma unroll
      for (int d = 0; d < K; ++d) {
        oky[d] = (unsigned)(d + dy_) < (unsigned)K;
        okx[d] = (unsigned)(d + dx_) < (unsigned)K;
      }
#pragma unroll
      for (int j = 0; j < K2; ++j) {
        const bool ok = oky[j / K] && okx[j % K] && !(j / K + dy_ == R && j % K + dx_ == R);
        const float val = wu[j];
        w[j] += ok ? val : 0.f;
      }
      w[K2 / 2] += wu[K2 / 2 - dy_ * K - dx_];
    }
    // The window table of phase B.  Phase B multiplies the UNPADDED image (bwd_tile's slab is padded: there a ring slot's
    // weight meets the ring copy of the pixel it folds onto), so the padding adjoint is applied here: the weight of a slot
    // whose position lies outside the image is added to the slot of its fold source — always inside the window, the centre
    // when the position is a copy of the pixel itself (replicate; reflect with R = 2) — and dropped under zero padding.
    // One thread owns the row: its read-modify-writes are ordered, the sum has a fixed order.
    float* wt = Wt + (py * W + px) * K2;
#pragma unroll
    for (int j = 0; j < K2; ++j) {
      const int uy = py + j / K - R, ux = px + j % K - R;
      const bool inimg = (unsigned)uy < (unsigned)H && (unsigned)ux < (unsigned)W;
      wt[j] = inimg ? w[j] : 0.f;
    }
    if (vy < 2 * R || vy >= rows - 2 * R || vx < 2 * R || vx >= Wu - 2 * R) {   // (a pixel within R of the border)
#pragma unroll
      for (int j = 0; j < K2; ++j) {
        const int uy = py + j / K - R, ux = px + j % K - R;
        const bool inimg = (unsigned)uy < (unsigned)H && (unsigned)ux < (unsigned)W;
        const int ty = fo.y(uy, H), tx = fo.x(ux, W);
        if (!inimg && (ty | tx) >= 0) {
          const int jp = (ty - py + R) * K + (tx - px + R);
          wt[jp] += w[j];
        }
      }
    }
  }
  __syncthreads();
  NFP_STAMP(4);
  bwd_gemm_phase<R, NHWC>(g, Wt, gemm_Xt, gemm_Wd, gxr, x16, (uint16_t*)gx + (long long)b * g.gB, cb0, cb1, t, T,
                          POOL ? gg_s : nullptr);
}

}  // namespace nfp
