// nfp_gather.h — any-geometry, any-measure backward in GATHER form (no atomics, bitwise deterministic).
//
// Replaces, for every measure of nfp.py:141-342 and every nn.Conv2d geometry, the autograd graph of
// measure -> view -> frozen depthwise conv -> pad (nfp.py:132-159).  grad_x[c][r] is the sum of
//   * "centre role":    for every output o whose centre tap lands on r, and each of its N neighbours:
//                       d out[n,o] / d a   with a = x[c][r],  b = x[c][q(o,n)]
//   * "neighbour role": for every (o, n) whose neighbour tap lands on r:
//                       d out[n,o] / d b   with a = x[c][centre(o)],  b = x[c][r]
// The adjoint of pad / stride / dilation is an inverse index map.  It depends on the geometry only and
// is separable, so each workgroup inverts it analytically into one short reader list per ROW and per
// COLUMN in LDS (H + W threads, a few dozen integer steps each); the readers of pixel (ry, rx) are the
// product of the two lists, walked in a fixed order.  Next to them sits a table of the per-pair backward
// scalars Meas<M>::coef (grad_out, saved output and saved per-pixel stats enter only here).  The channel loop then reads x from an LDS slab laid out
// float4[channel quad][pixel] and accumulates grad_x in registers: one ds_read_b128 serves four
// pair-gradients, nothing is scattered, and grad_x is stored straight from registers.
// The LDS-atomic kernel (nfp_generic.h::bwd_generic) stays as the fallback for maps whose tables do not
// fit in LDS.
#pragma once
#include "nfp_measures.h"

namespace nfp {

// floats of Coef that Meas<M>::grad reads
template <int M> struct NCoef { static constexpr int v = 1; };
template <> struct NCoef<NFP_COSINE> { static constexpr int v = 3; };
template <> struct NCoef<NFP_GFC> { static constexpr int v = 3; };
template <> struct NCoef<NFP_SMITH> { static constexpr int v = 3; };
template <> struct NCoef<NFP_PEARSON> { static constexpr int v = 5; };

// LDS word offsets of the tables (host: launch_bwd_gather in nfp_hip.hip)
struct GatherLds {
  int ON;          // O * N pairs
  int cf;          // float [NC][ON]  coefficient k of pair j = n*O + o at cf + k*ON + j
  int nbq;         // u16   [ON]      neighbour pixel of pair j (P = zero padding: the slab's zero pixel)
  int yl, xl;      // uint2 [H][capY] / [W][capX]  per-row / per-column reader lists (see axis_reader)
  int yc, xc;      // u32   [H] / [W]              their lengths | number of leading centre-tap readers << 16
  int capY, capX;
  int xs;          // float4 slab [Cq][P + 1] (word offset, multiple of 4); pixel P of every quad is 0.
                   // The uncompacted reader slots live here until the first slab is staged.
  int Cq;          // channel quads per slab
  int Qwg;         // channel quads per workgroup
};

constexpr unsigned kNoCentre = 0x10000u;  // added to a centre-pixel part: the sum then exceeds every pixel index

// The inverse of pad -> strided, dilated taps is separable.  Along one axis of size n (no outputs), slot
// (ic, d) of coordinate i asks: does tap d of some output oa read i through padded coordinate tc(ic)?
// tc(0) = i, then the left and right margins.  A reader is stored as the two addends the channel loop
// needs, so that a (row reader, column reader) pair costs a handful of integer instructions:
//   .x  part of the pair index j = n*O + o:      rows d*k*O + oa*Wo,       columns d*O + oa
//   .y  part of the centre pixel of output o:    rows ca*W,  columns ca    (kNoCentre: zero padding)
//       | part of the tap index << 20:           rows d*k,   columns d
// (j still needs "- O if tap > centre tap", the centre itself is not a neighbour).
__device__ __forceinline__ uint2 axis_reader(const KP& g, int i, int n, int no, int ic, int d, bool rows) {
  uint2 e = make_uint2(0xFFFFFFFFu, 0u);
  const int tc = ic == 0 ? i : (ic <= g.pad ? -ic : n - 1 + (ic - g.pad));
  if (ic > 0 && map_index(tc, n, g.mode) != i) return e;
  const int nn = tc + g.pad - d * g.dil;
  if (nn < 0) return e;
  const int oa = g.stride == 1 ? nn : nn / g.stride;
  if (oa * g.stride != nn || oa >= no) return e;
  const int ca = map_index(oa * g.stride + g.R * g.dil - g.pad, n, g.mode);
  e.x = rows ? (unsigned)(d * g.k * g.O + oa * g.Wo) : (unsigned)(d * g.O + oa);
  e.y = (ca < 0 ? kNoCentre : (unsigned)(rows ? ca * g.W : ca)) | ((unsigned)(rows ? d * g.k : d) << 20);
  return e;
}

template <int M>
__device__ __forceinline__ Coef load_coef(const float* cf, int ON, int j) {
  constexpr int NC = NCoef<M>::v;
  Coef c = {0.f, 0.f, 0.f, 0.f, 0.f};
  c.k0 = cf[j];
  if (NC > 1) c.k1 = cf[ON + j];
  if (NC > 2) c.k2 = cf[2 * ON + j];
  if (NC > 3) c.k3 = cf[3 * ON + j];
  if (NC > 4) c.k4 = cf[4 * ON + j];
  return c;
}

template <int M, int QB>
__global__ void __launch_bounds__(512) bwd_gather(const KP g, const GatherLds L, const void* __restrict__ x,
                                                  const void* __restrict__ go, const void* __restrict__ out,
                                                  const float* __restrict__ saved, void* __restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NC = NCoef<M>::v;
  float* cf = lds + L.cf;
  unsigned short* nbq = (unsigned short*)(lds + L.nbq);
  uint2* yl = (uint2*)(lds + L.yl);
  uint2* xl = (uint2*)(lds + L.xl);
  unsigned* yc = (unsigned*)(lds + L.yc);
  unsigned* xc = (unsigned*)(lds + L.xc);
  float4* xs = (float4*)(lds + L.xs);
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int ON = L.ON, PS = g.P + 1;
  const float* sv = (Meas<M>::NSTAT > 0) ? saved + (long long)b * Meas<M>::NSTAT * g.P : nullptr;
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  // ---- tables: reader slots of every row and column (geometry only), one slot per thread ---------------
  const int nslot = (2 * g.pad + 1) * g.k;
  uint2* slots = (uint2*)xs;
  for (int s = t; s < (g.H + g.W) * nslot; s += T) {
    const int i = s / nslot, w = s - i * nslot, ic = w / g.k, d = w - ic * g.k;
    slots[s] = i < g.H ? axis_reader(g, i, g.H, g.Ho, ic, d, true) : axis_reader(g, i - g.H, g.W, g.Wo, ic, d, false);
  }
  NFP_STAMP(1);
  // ---- tables: per-pair coefficients and neighbour pixels (this image) ----------------------------------
  for (int j = t; j < ON; j += T) {
    const int n = j / g.O, o = j - n * g.O;
    const int q = nbr_pixel(g, o, n), pc = tap_pixel(g, o, g.R, g.R);
    const long long oi = ((long long)b * g.N + n) * g.O + o;
    const float sp0 = (Meas<M>::NSTAT > 0 && pc >= 0) ? sv[pc] : 0.f;
    const float sp1 = (Meas<M>::NSTAT > 1 && pc >= 0) ? sv[g.P + pc] : 0.f;
    const float sq0 = (Meas<M>::NSTAT > 0 && q >= 0) ? sv[q] : 0.f;
    const float sq1 = (Meas<M>::NSTAT > 1 && q >= 0) ? sv[g.P + q] : 0.f;
    const Coef c = Meas<M>::coef(ldx(go, oi, g.godtype), ldx(out, oi, g.dtype), sp0, sp1, sq0, sq1, g);
    cf[j] = c.k0;
    if (NC > 1) cf[ON + j] = c.k1;
    if (NC > 2) cf[2 * ON + j] = c.k2;
    if (NC > 3) cf[3 * ON + j] = c.k3;
    if (NC > 4) cf[4 * ON + j] = c.k4;
    nbq[j] = (unsigned short)(q < 0 ? g.P : q);
  }
  NFP_STAMP(2);
  __syncthreads();
  // compact the slots of each row / column into its reader list: readers through the CENTRE tap first (a
  // pair of two of them makes this pixel the centre of an output), then the others, each in slot order
  for (int i = t; i < g.H + g.W; i += T) {
    uint2* list = i < g.H ? yl + i * L.capY : xl + (i - g.H) * L.capX;
    const unsigned ctap = (unsigned)(i < g.H ? g.R * g.k : g.R);
    unsigned cnt = 0;
    for (int w = 0; w < nslot; ++w) {
      const uint2 e = slots[i * nslot + w];
      if (e.x != 0xFFFFFFFFu && (e.y >> 20) == ctap) list[cnt++] = e;
    }
    const unsigned ncentre = cnt;
    for (int w = 0; w < nslot; ++w) {
      const uint2 e = slots[i * nslot + w];
      if (e.x != 0xFFFFFFFFu && (e.y >> 20) != ctap) list[cnt++] = e;
    }
    if (i < g.H)
      yc[i] = cnt | (ncentre << 16);
    else
      xc[i - g.H] = cnt | (ncentre << 16);
  }
  NFP_STAMP(3);

  // ---- channel loop ------------------------------------------------------------------------------------
  const int Q = (g.C + 3) >> 2;
  const int q_begin = blockIdx.y * L.Qwg, q_end = min(Q, q_begin + L.Qwg);
  const unsigned mid = (unsigned)((g.k * g.k) >> 1);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int qc0 = q_begin; qc0 < q_end; qc0 += L.Cq) {
    const int cqn = min(L.Cq, q_end - qc0);
    __syncthreads();  // tables complete, slots consumed / previous slab consumed
    if (qc0 == q_begin) NFP_STAMP(4);
    // stage x[b, 4*qc0 : 4*(qc0+cqn), :, :] as float4[cq][p]; channels past C read as 0
    if (g.sC == 1) {  // channels-last: channel quad fastest -> 16 contiguous bytes per lane
      for (int i = t; i < cqn * g.P; i += T) {
        const int p = i / cqn, cq = i - p * cqn;
        const int y = p / g.W, xx = p - y * g.W, c = 4 * (qc0 + cq);
        const long long base = (long long)b * g.sB + (long long)y * g.sH + (long long)xx * g.sW + c;
        float4 v;
        v.x = ldx(x, base, g.dtype);
        v.y = c + 1 < g.C ? ldx(x, base + 1, g.dtype) : 0.f;
        v.z = c + 2 < g.C ? ldx(x, base + 2, g.dtype) : 0.f;
        v.w = c + 3 < g.C ? ldx(x, base + 3, g.dtype) : 0.f;
        xs[cq * PS + p] = v;
      }
    } else {
      for (int i = t; i < cqn * g.P; i += T) {
        const int cq = i / g.P, p = i - cq * g.P;
        const int y = p / g.W, xx = p - y * g.W, c = 4 * (qc0 + cq);
        const long long base = (long long)b * g.sB + (long long)c * g.sC + (long long)y * g.sH + (long long)xx * g.sW;
        float4 v;
        v.x = ldx(x, base, g.dtype);
        v.y = c + 1 < g.C ? ldx(x, base + g.sC, g.dtype) : 0.f;
        v.z = c + 2 < g.C ? ldx(x, base + 2 * g.sC, g.dtype) : 0.f;
        v.w = c + 3 < g.C ? ldx(x, base + 3 * g.sC, g.dtype) : 0.f;
        xs[cq * PS + p] = v;
      }
    }
    for (int cq = t; cq < cqn; cq += T) xs[cq * PS + g.P] = zero4;  // what a zero-padded tap reads
    __syncthreads();
    if (qc0 == q_begin) NFP_STAMP(5);
    const int nqb = (cqn + QB - 1) / QB;
    for (int i = t; i < nqb * g.P; i += T) {
      const int qb = i / g.P, r = i - qb * g.P;
      const int ry = r / g.W, rx = r - ry * g.W;
      const float4* slab[QB];
      float4 a[QB], acc[QB];
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        slab[u] = xs + min(qb * QB + u, cqn - 1) * PS;  // clamped: the surplus quad is computed, never stored
        a[u] = slab[u][r];
        acc[u] = zero4;
      }
      const uint2* yrow = yl + ry * L.capY;
      const uint2* xrow = xl + rx * L.capX;
      const unsigned ycw = yc[ry], xcw = xc[rx];
      const unsigned ny = ycw & 0xFFFFu, nx = xcw & 0xFFFFu, nyc = ycw >> 16, nxc = xcw >> 16;
      // centre role: r is the centre of output o (exactly one for the usual geometries, none or several
      // when stride > 1 or the padding exceeds the kernel radius)
      for (unsigned iy = 0; iy < nyc; ++iy) {
        const uint2 ey = yrow[iy];
        for (unsigned ix = 0; ix < nxc; ++ix) {
          const uint2 ex = xrow[ix];
          for (int n = 0, j = (int)(ey.x + ex.x) - (int)mid * g.O; n < g.N; ++n, j += g.O) {
            const unsigned q = nbq[j];
            const Coef c = load_coef<M>(cf, ON, j);
#pragma unroll
            for (int u = 0; u < QB; ++u) {
              const float4 bv = slab[u][q];
              float da, db;
              Meas<M>::grad(a[u].x, bv.x, c, g, da, db); acc[u].x += da;
              Meas<M>::grad(a[u].y, bv.y, c, g, da, db); acc[u].y += da;
              Meas<M>::grad(a[u].z, bv.z, c, g, da, db); acc[u].z += da;
              Meas<M>::grad(a[u].w, bv.w, c, g, da, db); acc[u].w += da;
            }
          }
        }
      }
      // neighbour role: r is neighbour n of output o; a = x at o's centre (the zero pixel if padded)
      for (unsigned iy = 0; iy < ny; ++iy) {
        const uint2 ey = yrow[iy];
        for (unsigned ix = 0; ix < nx; ++ix) {
          if (iy < nyc && ix < nxc) continue;  // centre tap x centre tap: handled above
          const uint2 ex = xrow[ix];
          const unsigned tap = (ey.y >> 20) + (ex.y >> 20);
          const int j = (int)(ey.x + ex.x) - (tap > mid ? g.O : 0);
          const unsigned pc = min((ey.y & 0xFFFFFu) + (ex.y & 0xFFFFFu), (unsigned)g.P);
          const Coef c = load_coef<M>(cf, ON, j);
#pragma unroll
          for (int u = 0; u < QB; ++u) {
            const float4 av = slab[u][pc];
            float da, db;
            Meas<M>::grad(av.x, a[u].x, c, g, da, db); acc[u].x += db;
            Meas<M>::grad(av.y, a[u].y, c, g, da, db); acc[u].y += db;
            Meas<M>::grad(av.z, a[u].z, c, g, da, db); acc[u].z += db;
            Meas<M>::grad(av.w, a[u].w, c, g, da, db); acc[u].w += db;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        const int cq = qb * QB + u;
        if (cq < cqn) {
          const int c = 4 * (qc0 + cq);
          const long long base = (long long)b * g.sB + (long long)c * g.sC + (long long)ry * g.sH + (long long)rx * g.sW;
          stx(gx, base, acc[u].x, g.dtype);
          if (c + 1 < g.C) stx(gx, base + g.sC, acc[u].y, g.dtype);
          if (c + 2 < g.C) stx(gx, base + 2 * g.sC, acc[u].z, g.dtype);
          if (c + 3 < g.C) stx(gx, base + 3 * g.sC, acc[u].w, g.dtype);
        }
      }
    }
  }
  NFP_STAMP(6);
}

}  // namespace nfp
