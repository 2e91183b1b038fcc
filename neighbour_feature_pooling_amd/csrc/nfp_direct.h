// nfp_direct.h — the kernels of last resort: every geometry, every measure, any map size, NO shared-memory tables.
//
// They stand behind fwd_pairs / bwd_gather / bwd_gather_banded (nfp_gather.h), whose index tables live in LDS and
// therefore bound the map: a single row window or a band's pair table that does not fit (maps thousands of pixels
// wide, k >= 5 on maps wider than ~150 pixels, more than 65 534 pixels per map for the forward).  Here a thread
// reads what it needs straight from global memory (L2 serves the neighbours' re-reads) and sums in a fixed order:
// slower, but with no size limit and bitwise reproducible — there is no atomic kernel anywhere in the library.
//
// Forward  (nfp.py:132-134 for any measure): thread = (output o, group of 8 neighbours); it walks all channels,
//          keeping the pair sums and the per-pixel statistics of the centre and of its <= 8 neighbours.
// Backward (autograd of the same): thread = (input pixel r, block of 8 channels).  The pairs that involve r are
//          enumerated analytically — the inverse of pad -> strided / dilated taps is separable, axis_reader_raw
//          (nfp_gather.h) lists per axis which (tap, output) reads a coordinate and through which fold of the
//          padding: r as the CENTRE of an output contributes through its N pairs, r as NEIGHBOUR n of an output
//          through that one pair.
#pragma once
#include "nfp_gather.h"

namespace nfp {

constexpr int kGroup = 8;    // neighbours per forward thread
constexpr int kDirectCB = 8; // channels per backward thread

template <int M>
__global__ void __launch_bounds__(256) fwd_direct(const KP g, const void* __restrict__ x, void* __restrict__ out,
                                                  float* __restrict__ saved) {
  const long long b = blockIdx.z;
  const int grp = blockIdx.y;
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= g.O) return;
  const int pc = tap_pixel(g, (int)o, g.R, g.R);
  int q[kGroup];
#pragma unroll
  for (int j = 0; j < kGroup; ++j) {
    const int n = grp * kGroup + j;
    q[j] = n < g.N ? nbr_pixel(g, (int)o, n) : -1;
  }
  auto at = [&](int p, int c) {  // x[b, c, pixel p]
    const int y = p / g.W, xx = p - y * g.W;
    return ldx(x, b * g.sB + (long long)c * g.sC + (long long)y * g.sH + (long long)xx * g.sW, g.dtype);
  };
  float acc[kGroup], sb0[kGroup], sb1[kGroup], pb[kGroup];
  float sa0 = 0.f, sa1 = 0.f, pa = 0.f;
#pragma unroll
  for (int j = 0; j < kGroup; ++j) acc[j] = sb0[j] = sb1[j] = pb[j] = 0.f;
  if constexpr (Pivot<M>::v) {  // (nfp_measures.h::Pivot) channel 0 of each pixel; 0 for zero-padded taps
    if (pc >= 0) pa = at(pc, 0);
#pragma unroll
    for (int j = 0; j < kGroup; ++j)
      if (q[j] >= 0) pb[j] = at(q[j], 0);
  }
  for (int c = 0; c < g.C; ++c) {
    float a = pc >= 0 ? at(pc, c) : 0.f;
    if constexpr (Pivot<M>::v) a -= pa;
    Meas<M>::stat(a, sa0, sa1);
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
      float bv = q[j] >= 0 ? at(q[j], c) : 0.f;
      if constexpr (Pivot<M>::v) bv -= pb[j];
      acc[j] += Meas<M>::term(a, bv, g);
      Meas<M>::stat(bv, sb0[j], sb1[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < kGroup; ++j) {
    const int n = grp * kGroup + j;
    if (n < g.N) stx(out, (b * g.N + n) * g.O + o, Meas<M>::fin(acc[j], sa0, sa1, sb0[j], sb1[j], g), g.odtype);
  }
  if constexpr (Meas<M>::NSTAT > 0) if (saved != nullptr) {
    // per-input-pixel statistics for the backward, [B][NSTAT][P]; several threads may store the same pixel, all
    // with a value summed in the same order (channel 0 upwards): benign duplicate stores
    float* sv = saved + b * Meas<M>::NSTAT * g.P;
    if (pc >= 0) {
      sv[pc] = Meas<M>::save0(sa0, sa1, g) + (Pivot<M>::v ? pa : 0.f);
      if (Meas<M>::NSTAT > 1) sv[g.P + pc] = Meas<M>::save1(sa0, sa1, g);
    }
#pragma unroll
    for (int j = 0; j < kGroup; ++j)
      if (q[j] >= 0) {
        sv[q[j]] = Meas<M>::save0(sb0[j], sb1[j], g) + (Pivot<M>::v ? pb[j] : 0.f);
        if (Meas<M>::NSTAT > 1) sv[g.P + q[j]] = Meas<M>::save1(sb0[j], sb1[j], g);
      }
  }
}

template <int M>
__global__ void __launch_bounds__(256) bwd_direct(const KP g, const void* __restrict__ x, const void* __restrict__ go,
                                                  const void* __restrict__ out, const float* __restrict__ saved,
                                                  void* __restrict__ gx) {
  constexpr int CB = kDirectCB;
  const long long b = blockIdx.z;
  const int c0 = blockIdx.y * CB;
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= g.P) return;
  const int ry = (int)(r / g.W), rx = (int)(r - (long long)ry * g.W);
  const float* sv = (Meas<M>::NSTAT > 0) ? saved + b * Meas<M>::NSTAT * g.P : nullptr;
  auto at = [&](int p, int c) {
    const int y = p / g.W, xx = p - y * g.W;
    return ldx(x, b * g.sB + (long long)c * g.sC + (long long)y * g.sH + (long long)xx * g.sW, g.dtype);
  };
  auto st0 = [&](int p) { return (Meas<M>::NSTAT > 0 && p >= 0) ? sv[p] : 0.f; };
  auto st1 = [&](int p) { return (Meas<M>::NSTAT > 1 && p >= 0) ? sv[g.P + p] : 0.f; };
  float xr[CB], acc[CB];
#pragma unroll
  for (int u = 0; u < CB; ++u) {
    xr[u] = c0 + u < g.C ? at((int)r, c0 + u) : 0.f;
    acc[u] = 0.f;
  }
  const int nslot = (2 * g.pad + 1) * g.k;
  for (int sy = 0; sy < nslot; ++sy) {
    const uint2 ey = axis_reader_raw(g, ry, g.H, g.Ho, sy / g.k, sy % g.k);
    if (ey.x == 0xFFFFFFFFu) continue;
    const int dy = (int)(ey.x & 0xFFu), oy = (int)(ey.x >> 8);
    for (int sx = 0; sx < nslot; ++sx) {
      const uint2 ex = axis_reader_raw(g, rx, g.W, g.Wo, sx / g.k, sx % g.k);
      if (ex.x == 0xFFFFFFFFu) continue;
      const int dx = (int)(ex.x & 0xFFu), ox = (int)(ex.x >> 8);
      const int o = oy * g.Wo + ox, tap = dy * g.k + dx;
      if (tap == (g.k * g.k) / 2) {
        // r is the centre of output o: its N pairs pull on x_r through d/da
        const float sp0 = st0((int)r), sp1 = st1((int)r);
        for (int n = 0; n < g.N; ++n) {
          const int q = nbr_pixel(g, o, n);
          const long long oi = (b * g.N + n) * g.O + o;
          const Coef cf = Meas<M>::coef(ldx(go, oi, g.godtype), ldx(out, oi, g.dtype), sp0, sp1, st0(q), st1(q), g);
#pragma unroll
          for (int u = 0; u < CB; ++u) {
            const float bv = (q >= 0 && c0 + u < g.C) ? at(q, c0 + u) : 0.f;
            float da, db;
            Meas<M>::grad(xr[u], bv, cf, g, da, db);
            acc[u] += da;
          }
        }
      } else {
        // r is neighbour n of output o: that one pair pulls on x_r through d/db
        const int n = tap < (g.k * g.k) / 2 ? tap : tap - 1;
        const int pc = tap_pixel(g, o, g.R, g.R);
        const long long oi = (b * g.N + n) * g.O + o;
        const Coef cf = Meas<M>::coef(ldx(go, oi, g.godtype), ldx(out, oi, g.dtype), st0(pc), st1(pc), st0((int)r),
                                      st1((int)r), g);
#pragma unroll
        for (int u = 0; u < CB; ++u) {
          const float a = (pc >= 0 && c0 + u < g.C) ? at(pc, c0 + u) : 0.f;
          float da, db;
          Meas<M>::grad(a, xr[u], cf, g, da, db);
          acc[u] += db;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < CB; ++u)
    if (c0 + u < g.C)
      stx(gx, b * g.gB + (long long)(c0 + u) * g.sC + (long long)ry * g.sH + (long long)rx * g.sW, acc[u], g.dtype);
}

// ---- Attention (nfp.py:195-205): softmax over the N neighbour maps of the raw dots `src` (f32: in place on out
// for float32 maps, the scratch of nfp_saved_floats for bf16 maps, so that the dots are rounded to bf16 never and
// the probabilities once) ------
__global__ void __launch_bounds__(256) attn_softmax_fwd(const KP g, const float* src, void* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (b, o)
  if (i >= (long long)g.B * g.O) return;
  const long long b = i / g.O, o = i - b * g.O;
  const long long base = b * g.N * g.O + o;
  float mx = -INFINITY;
  for (int n = 0; n < g.N; ++n) mx = fmaxf(mx, src[base + (long long)n * g.O]);
  float s = 0.f;
  for (int n = 0; n < g.N; ++n) s += expf(src[base + (long long)n * g.O] - mx);
  const float sg = g.similarity ? 1.f : -1.f;
  for (int n = 0; n < g.N; ++n) {
    float y = expf(src[base + (long long)n * g.O] - mx) / s;
    stx(out, base + (long long)n * g.O, sg * y, g.dtype);
  }
}
// grad wrt the dots: gd_n = y_n * (gy_n - sum_m gy_m y_m), out = +-y  ->  scratch (f32)
__global__ void __launch_bounds__(256) attn_softmax_bwd(const KP g, const void* __restrict__ go,
                                                        const void* __restrict__ out, float* __restrict__ gd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)g.B * g.O) return;
  const long long b = i / g.O, o = i - b * g.O;
  const long long base = b * g.N * g.O + o;
  const float sg = g.similarity ? 1.f : -1.f;
  float dot = 0.f;
  for (int n = 0; n < g.N; ++n) {
    long long k = base + (long long)n * g.O;
    dot += sg * ldx(go, k, g.dtype) * sg * ldx(out, k, g.dtype);
  }
  for (int n = 0; n < g.N; ++n) {
    long long k = base + (long long)n * g.O;
    float y = sg * ldx(out, k, g.dtype);
    gd[k] = y * (sg * ldx(go, k, g.dtype) - dot);
  }
}

}  // namespace nfp
