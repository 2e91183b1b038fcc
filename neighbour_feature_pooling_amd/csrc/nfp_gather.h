// nfp_gather.h — the kernels for EVERY measure and EVERY nn.Conv2d geometry (bitwise deterministic, no atomics):
//   fwd_pairs           forward, one workgroup per (image, tile of outputs)      [second half of this file]
//   bwd_gather          backward in gather form, one workgroup per (image, channel block)
//   bwd_gather_banded   the same per band of grad_x rows, for maps whose whole-image tables exceed LDS
// Shared: the LDS slab float4[channel quad][pixel + 1] (the extra pixel is the zero a zero-padded tap reads) and
// its staging from any strides / f32 / bf16 (stage_quads).
//
// Backward.  It replaces the autograd graph of measure -> view -> frozen depthwise conv -> pad
// (nfp.py:132-159).  grad_x[c][r] is the sum of
//   * "centre role":    for every output o whose centre tap lands on r, and each of its N neighbours:
//                       d out[n,o] / d a   with a = x[c][r],  b = x[c][q(o,n)]
//   * "neighbour role": for every (o, n) whose neighbour tap lands on r:
//                       d out[n,o] / d b   with a = x[c][centre(o)],  b = x[c][r]
// The adjoint of pad / stride / dilation is an inverse index map.  It depends on the geometry only and is
// separable, so each workgroup inverts it analytically into one short reader list per ROW and per COLUMN in LDS;
// the readers of pixel (ry, rx) are the product of the two lists, walked in a fixed order.  Next to them sits a
// table of the per-pair backward scalars Meas<M>::coef (grad_out, saved output and saved per-pixel stats enter only
// here).  Thread (pixel, 16 channels) then accumulates grad_x in registers — one ds_read_b128 serves four
// pair-gradients, nothing is scattered — and stores it directly.  The LDS-atomic kernel
// (nfp_generic.h::bwd_generic) stays as the last resort (circular / over-padded maps beyond the LDS tables).
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
  int sl;          // uint2 [(H + W)][(2*pad + 1)*k]  uncompacted reader slots
  int Ts;          // threads (whole wavefronts) that stage the first slab while the others build the tables
  int xs;          // float4 slab [Cq][P + 1] (word offset, multiple of 4); pixel P of every quad is 0
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
  const int oa = g.stride == 1 ? nn : fdivi(nn, g.stride);
  if (oa * g.stride != nn || oa >= no) return e;
  const int ca = map_index(oa * g.stride + g.R * g.dil - g.pad, n, g.mode);
  e.x = rows ? (unsigned)(d * g.k * g.O + oa * g.Wo) : (unsigned)(d * g.O + oa);
  e.y = (ca < 0 ? kNoCentre : (unsigned)(rows ? ca * g.W : ca)) | ((unsigned)(rows ? d * g.k : d) << 20);
  return e;
}


// x[b, 4*qc0 : 4*(qc0+cqn), pixels p0 .. p0+np-1] -> LDS float4[cq][np + 1] (local pixel np of every quad is
// the zero a zero-padded tap reads); any strides, f32 or bf16; channels past C are 0.  With a pivot table the slab
// holds x - piv[pixel] (nfp_measures.h::Pivot), so padding channels and the zero pixel stay exact zeros.
// The four loads of a quad are unconditional (addresses clamped into the tensor, surplus lanes zeroed
// afterwards): a select or branch around a load makes hipcc wait for each one before issuing the next.
template <bool BF>
__device__ __forceinline__ float ld_elem(const void* x, long long i) {
  if (BF) return bf16_to_f32(((const uint16_t*)x)[i]);
  return ((const float*)x)[i];
}
template <bool BF>
__device__ __forceinline__ float4 load_quad(const void* x, const KP& g, long long base, int left) {
  float4 v;
  v.x = ld_elem<BF>(x, base);
  v.y = ld_elem<BF>(x, base + min(1, left) * g.sC);
  v.z = ld_elem<BF>(x, base + min(2, left) * g.sC);
  v.w = ld_elem<BF>(x, base + min(3, left) * g.sC);
  return v;
}
__device__ __forceinline__ float4 finish_quad(float4 v, int left, float pv) {
  v.x -= pv;
  v.y = left >= 1 ? v.y - pv : 0.f;
  v.z = left >= 2 ? v.z - pv : 0.f;
  v.w = left >= 3 ? v.w - pv : 0.f;
  return v;
}
template <bool BF>
__device__ __forceinline__ void stage_quads_t(float4* xs, const void* x, const KP& g, int b, int qc0, int cqn,
                                              const float* piv, int p0, int np, int t, int T) {
  const int PS = np + 1;  // t of T threads take part (a kernel may keep a wavefront back for its index tables)
  if (np <= 0) {
    for (int cq = t; cq < cqn; cq += T) xs[cq] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  if (!BF && g.sW == 1 && g.sC != 1 && g.sH == (long long)g.W && np >= 4) {
    // Dense NCHW float32: 4 channels x 4 consecutive pixels per step = four 16-byte loads along the pixel axis
    // (global_load_dwordx4 needs only 4-byte alignment), transposed in registers into four float4 slots.
    // The last (partial) group of a row re-reads the final four pixels.  Two blocks in flight per thread.
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int npg = (np + 3) >> 2;
    const int TF = min(npg, T), TS = fdivi(T, TF);
    const int ts = fdivi(t, TF), tf = t - ts * TF;
    const float* xf = (const float*)x + (long long)b * g.sB + p0;
    if (ts < TS) {
      for (int s0 = ts; s0 < cqn; s0 += 2 * TS) {
        for (int f = tf; f < npg; f += TF) {
          const int pb = min(4 * f, np - 4);  // first pixel of the block (the last block is shifted back)
          f4u v[2][4];
          int cqs[2], lefts[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            cqs[u] = min(s0 + u * TS, cqn - 1);
            const int c = 4 * (qc0 + cqs[u]);
            lefts[u] = g.C - 1 - c;
            const float* row = xf + (long long)c * g.sC + pb;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[u][i] = *(const f4u*)(row + min(i, lefts[u]) * g.sC);
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (s0 + u * TS < cqn) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float pv = piv ? piv[pb + j] : 0.f;
                float4 w;
                w.x = v[u][0][j] - pv;
                w.y = lefts[u] >= 1 ? v[u][1][j] - pv : 0.f;
                w.z = lefts[u] >= 2 ? v[u][2][j] - pv : 0.f;
                w.w = lefts[u] >= 3 ? v[u][3][j] - pv : 0.f;
                xs[cqs[u] * PS + pb + j] = w;
              }
            }
          }
        }
      }
    }
    for (int cq = t; cq < cqn; cq += T) xs[cq * PS + np] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  // 2-D thread grid, fast axis = what is contiguous in memory: pixels (NCHW) or channel quads (channels-last);
  // one integer division per thread per call, two quads in flight per thread and iteration.
  const bool nhwc = g.sC == 1;
  const bool dense = g.sH == (long long)g.W * g.sW;  // pixel p sits at p * sW: no row / column split needed
  const int nf = nhwc ? cqn : np, ns = nhwc ? np : cqn;
  const int TF = min(nf, T), TS = fdivi(T, TF);
  const int ts = fdivi(t, TF), tf = t - ts * TF;
  const long long img = (long long)b * g.sB;
  constexpr int U = 8;  // quads in flight per thread: staging is latency-bound, not issue-bound
  if (ts < TS) {
    for (int s0 = ts; s0 < ns; s0 += U * TS) {
      for (int f = tf; f < nf; f += TF) {
        float4 v[U];
        int cqs[U], ps[U], lefts[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int su = min(s0 + u * TS, ns - 1);
          cqs[u] = nhwc ? f : su;
          ps[u] = nhwc ? su : f;
          long long off;
          if (dense) {
            off = (long long)(p0 + ps[u]) * g.sW;
          } else {
            const int pg = p0 + ps[u], y = fdivi(pg, g.W);
            off = (long long)y * g.sH + (long long)(pg - y * g.W) * g.sW;
          }
          const int c = 4 * (qc0 + cqs[u]);
          lefts[u] = g.C - 1 - c;
          v[u] = load_quad<BF>(x, g, img + (long long)c * g.sC + off, lefts[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float4 w = finish_quad(v[u], lefts[u], piv ? piv[ps[u]] : 0.f);
          if (s0 + u * TS < ns) xs[cqs[u] * PS + ps[u]] = w;
        }
      }
    }
  }
  for (int cq = t; cq < cqn; cq += T) xs[cq * PS + np] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void stage_quads(float4* xs, const void* x, const KP& g, int b, int qc0, int cqn,
                                            const float* piv, int p0, int np, int t, int T) {
  if (g.dtype == NFP_F32)
    stage_quads_t<false>(xs, x, g, b, qc0, cqn, piv, p0, np, t, T);
  else
    stage_quads_t<true>(xs, x, g, b, qc0, cqn, piv, p0, np, t, T);
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

// grad_x of one (pixel, QB channel quads) item: centre role + neighbour role over the pixel's reader lists.
// Two pairs are in flight per step (index -> coefficient / slab reads -> arithmetic is a chain of LDS latencies;
// N = k*k - 1 is a multiple of 8, so the centre loop needs no remainder).
template <int M, int QB>
__device__ __forceinline__ void gather_item(const KP& g, const float4* const (&slab)[QB], const float4 (&a)[QB],
                                            float4 (&acc)[QB], const uint2* yrow, const uint2* xrow, unsigned ycw,
                                            unsigned xcw, const float* cf, int ON, const unsigned short* nbq, int Os,
                                            unsigned zpix, unsigned mid) {
  const unsigned ny = ycw & 0xFFFFu, nx = xcw & 0xFFFFu, nyc = ycw >> 16, nxc = xcw >> 16;
  // centre role: r is the centre of output o (exactly one for the usual geometries, none or several when
  // stride > 1 or the padding exceeds the kernel radius)
  for (unsigned iy = 0; iy < nyc; ++iy) {
    const uint2 ey = yrow[iy];
    for (unsigned ix = 0; ix < nxc; ++ix) {
      const uint2 ex = xrow[ix];
      for (int n = 0, j = (int)(ey.x + ex.x) - (int)mid * Os; n < g.N; n += 2, j += 2 * Os) {
        const unsigned q0 = nbq[j], q1 = nbq[j + Os];
        const Coef c0 = load_coef<M>(cf, ON, j), c1 = load_coef<M>(cf, ON, j + Os);
        float4 b0[QB], b1[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
          b0[u] = slab[u][q0];
          b1[u] = slab[u][q1];
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) {
          float da, db;
          Meas<M>::grad(a[u].x, b0[u].x, c0, g, da, db); acc[u].x += da;
          Meas<M>::grad(a[u].y, b0[u].y, c0, g, da, db); acc[u].y += da;
          Meas<M>::grad(a[u].z, b0[u].z, c0, g, da, db); acc[u].z += da;
          Meas<M>::grad(a[u].w, b0[u].w, c0, g, da, db); acc[u].w += da;
          Meas<M>::grad(a[u].x, b1[u].x, c1, g, da, db); acc[u].x += da;
          Meas<M>::grad(a[u].y, b1[u].y, c1, g, da, db); acc[u].y += da;
          Meas<M>::grad(a[u].z, b1[u].z, c1, g, da, db); acc[u].z += da;
          Meas<M>::grad(a[u].w, b1[u].w, c1, g, da, db); acc[u].w += da;
        }
      }
    }
  }
  // neighbour role: r is neighbour n of output o; a = x at o's centre (the zero pixel if padded).  The
  // (row reader, column reader) pairs are walked as one flat list, two at a time.
  const unsigned tot = ny * nx;
  for (unsigned e = 0; e < tot; e += 2) {
    int jj[2];
    unsigned pcs[2];
    bool ok[2];
    Coef cc[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const unsigned ee = min(e + v, tot - 1);
      const unsigned iy = (unsigned)fdivi((int)ee, (int)nx), ix = ee - iy * nx;
      const uint2 ey = yrow[iy], ex = xrow[ix];
      const unsigned tap = (ey.y >> 20) + (ex.y >> 20);
      ok[v] = e + v < tot && !(iy < nyc && ix < nxc);  // centre tap x centre tap was the centre role
      jj[v] = (int)(ey.x + ex.x) - (tap > mid ? Os : 0);
      jj[v] = tap == mid ? 0 : jj[v];                    // (that pair's index is not a neighbour index: stay in range)
      pcs[v] = min((ey.y & 0xFFFFFu) + (ex.y & 0xFFFFFu), zpix);
      cc[v] = load_coef<M>(cf, ON, jj[v]);
    }
    float4 a0[QB], a1[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
      a0[u] = slab[u][pcs[0]];
      a1[u] = slab[u][pcs[1]];
    }
    if (ok[0]) {
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        float da, db;
        Meas<M>::grad(a0[u].x, a[u].x, cc[0], g, da, db); acc[u].x += db;
        Meas<M>::grad(a0[u].y, a[u].y, cc[0], g, da, db); acc[u].y += db;
        Meas<M>::grad(a0[u].z, a[u].z, cc[0], g, da, db); acc[u].z += db;
        Meas<M>::grad(a0[u].w, a[u].w, cc[0], g, da, db); acc[u].w += db;
      }
    }
    if (ok[1]) {
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        float da, db;
        Meas<M>::grad(a1[u].x, a[u].x, cc[1], g, da, db); acc[u].x += db;
        Meas<M>::grad(a1[u].y, a[u].y, cc[1], g, da, db); acc[u].y += db;
        Meas<M>::grad(a1[u].z, a[u].z, cc[1], g, da, db); acc[u].z += db;
        Meas<M>::grad(a1[u].w, a[u].w, cc[1], g, da, db); acc[u].w += db;
      }
    }
  }
}

#ifndef NFP_GATHER_T
#define NFP_GATHER_T 512
#endif
template <int M, int QB>
__global__ void __launch_bounds__(NFP_GATHER_T) bwd_gather(const KP g, const GatherLds L, const void* __restrict__ x,
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

  // The first L.Ts threads stage the first slab; the others build the tables meanwhile.  Each part is one or two
  // passes of dependent work (index arithmetic; global load -> arithmetic -> LDS), i.e. latency, not throughput:
  // side by side they cost the longer of the two instead of the sum.
  const int Q = (g.C + 3) >> 2;
  const int q_begin = blockIdx.y * L.Qwg, q_end = min(Q, q_begin + L.Qwg);
  const int nslot = (2 * g.pad + 1) * g.k;
  uint2* slots = (uint2*)(lds + L.sl);
  if (t < L.Ts) {
    stage_quads(xs, x, g, b, q_begin, min(L.Cq, q_end - q_begin), nullptr, 0, g.P, t, L.Ts);
  } else {
    const int tt = t - L.Ts, TT = T - L.Ts;
    // ---- tables: reader slots of every row and column (geometry only), one slot per thread ---------------
    for (int s = tt; s < (g.H + g.W) * nslot; s += TT) {
      const int i = fdivi(s, nslot), w = s - i * nslot, ic = fdivi(w, g.k), d = w - ic * g.k;
      slots[s] = i < g.H ? axis_reader(g, i, g.H, g.Ho, ic, d, true) : axis_reader(g, i - g.H, g.W, g.Wo, ic, d, false);
    }
    // ---- tables: per-pair coefficients and neighbour pixels (this image) ----------------------------------
    for (int j = tt; j < ON; j += TT) {
      const int n = fdivi(j, g.O), o = j - n * g.O;
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
  }
  NFP_STAMP(2);
  __syncthreads();
  // compact the slots of each row / column into its reader list: readers through the CENTRE tap first (a
  // pair of two of them makes this pixel the centre of an output), then the others, each in slot order.
  // One wavefront per list: lane = slot, position = running count + popcount of the ballot below the lane.
  for (int i = t >> 6; i < g.H + g.W; i += T >> 6) {
    uint2* list = i < g.H ? yl + i * L.capY : xl + (i - g.H) * L.capX;
    const unsigned ctap = (unsigned)(i < g.H ? g.R * g.k : g.R);
    const int lane = t & 63;
    unsigned cnt = 0, ncentre = 0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int w0 = 0; w0 < nslot; w0 += 64) {
        const int w = w0 + lane;
        uint2 e = make_uint2(0xFFFFFFFFu, 0u);
        if (w < nslot) e = slots[i * nslot + w];
        const bool pred = e.x != 0xFFFFFFFFu && (((e.y >> 20) == ctap) == (pass == 0));
        const unsigned long long m = __ballot(pred);
        if (pred) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = e;
        cnt += (unsigned)__popcll(m);
      }
      if (pass == 0) ncentre = cnt;
    }
    if (lane == 0) {
      if (i < g.H)
        yc[i] = cnt | (ncentre << 16);
      else
        xc[i - g.H] = cnt | (ncentre << 16);
    }
  }
  NFP_STAMP(3);

  // ---- channel loop ------------------------------------------------------------------------------------
  const unsigned mid = (unsigned)((g.k * g.k) >> 1);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int qc0 = q_begin; qc0 < q_end; qc0 += L.Cq) {
    const int cqn = min(L.Cq, q_end - qc0);
    __syncthreads();  // tables and first slab complete / previous slab consumed
    if (qc0 == q_begin) NFP_STAMP(4);
    if (qc0 != q_begin) {
      stage_quads(xs, x, g, b, qc0, cqn, nullptr, 0, g.P, t, T);
      __syncthreads();
    }
    if (qc0 == q_begin) NFP_STAMP(5);
    const int nqb = (cqn + QB - 1) / QB;
    for (int i = t; i < nqb * g.P; i += T) {
      const int qb = fdivi(i, g.P), r = i - qb * g.P;
      const int ry = fdivi(r, g.W), rx = r - ry * g.W;
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
      gather_item<M, QB>(g, slab, a, acc, yrow, xrow, yc[ry], xc[rx], cf, ON, nbq, g.O, (unsigned)g.P, mid);
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        const int cq = qb * QB + u;
        if (cq < cqn) {
          const int c = 4 * (qc0 + cq);
          const long long base = (long long)b * g.gB + (long long)c * g.sC + (long long)ry * g.sH + (long long)rx * g.sW;
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

// ---- the same backward for maps whose tables do not fit LDS whole: one workgroup per BAND of input rows ---
// A band [ra, rb) of rows of grad_x needs: the reader lists of its rows (and of every column); the pair
// table of the OUTPUT rows oya..oyb that read the band (local pair index j' = n*O' + o', O' = their count);
// and the rows w0..w1 of x that those pairs pair the band with (the band itself, the centres of the outputs
// that read it, the neighbours of the outputs centred in it).  All three ranges are found in LDS (min / max
// over the lists and over the pairs), the launcher only bounds them: for non-circular padding with
// pad <= R*dilation an output reads rows within R*dilation of its centre and padding folds back inside that
// window, so oyb - oya + 1 <= (RB - 1 + 2*R*dil)/stride + 2 and w1 - w0 + 1 <= RB + 2*R*dil.
struct BandLds {
  int RB;          // input rows per band
  int ONm;         // bound on N * O'
  int cf, nbq;     // as GatherLds, sized for ONm
  int yl, xl, yc, xc, capY, capX;  // row lists for RB rows, column lists for W columns
  int mm;          // int [4]: min / max output row reading the band, min / max x row needed
  int xs;          // float4 slab [Cq][PSm]; the raw reader slots live here until the first slab is staged
  int PSm;         // bound on window pixels + 1
  int Cq, Qwg;
};

// raw reader of axis coordinate i: {d | oa << 8, ca} or {0xFFFFFFFF, -}; ca = -1: centre in zero padding
__device__ __forceinline__ uint2 axis_reader_raw(const KP& g, int i, int n, int no, int ic, int d) {
  uint2 e = make_uint2(0xFFFFFFFFu, 0u);
  const int tc = ic == 0 ? i : (ic <= g.pad ? -ic : n - 1 + (ic - g.pad));
  if (ic > 0 && map_index(tc, n, g.mode) != i) return e;
  const int nn = tc + g.pad - d * g.dil;
  if (nn < 0) return e;
  const int oa = g.stride == 1 ? nn : fdivi(nn, g.stride);
  if (oa * g.stride != nn || oa >= no) return e;
  e.x = (unsigned)d | ((unsigned)oa << 8);
  e.y = (unsigned)map_index(oa * g.stride + g.R * g.dil - g.pad, n, g.mode);
  return e;
}

template <int M, int QB>
__global__ void __launch_bounds__(512) bwd_gather_banded(const KP g, const BandLds L, const void* __restrict__ x,
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
  int* mm = (int*)(lds + L.mm);
  float4* xs = (float4*)(lds + L.xs);
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int ra = blockIdx.z * L.RB, rb = min(g.H, ra + L.RB), nr = rb - ra;  // this band's rows
  const float* sv = (Meas<M>::NSTAT > 0) ? saved + (long long)b * Meas<M>::NSTAT * g.P : nullptr;

  if (t == 0) {
    mm[0] = g.Ho;
    mm[1] = -1;
    mm[2] = ra;        // the band itself is part of the window
    mm[3] = rb - 1;
  }
  __syncthreads();
  // ---- raw reader slots of the band's rows and of every column; which output rows read the band? --------
  const int nslot = (2 * g.pad + 1) * g.k;
  uint2* slots = (uint2*)xs;
  for (int s = t; s < (nr + g.W) * nslot; s += T) {
    const int i = fdivi(s, nslot), w = s - i * nslot, ic = fdivi(w, g.k), d = w - ic * g.k;
    uint2 e;
    if (i < nr) {
      e = axis_reader_raw(g, ra + i, g.H, g.Ho, ic, d);
      if (e.x != 0xFFFFFFFFu) {
        const int oa = (int)(e.x >> 8), ca = (int)e.y;
        atomicMin(&mm[0], oa);
        atomicMax(&mm[1], oa);
        if (ca >= 0) {  // the centre row of an output that reads the band
          atomicMin(&mm[2], ca);
          atomicMax(&mm[3], ca);
        }
      }
    } else {
      e = axis_reader_raw(g, i - nr, g.W, g.Wo, ic, d);
    }
    slots[s] = e;
  }
  __syncthreads();
  const int oya = mm[0], oyb = mm[1];
  const int Ol = oyb < oya ? 0 : (oyb - oya + 1) * g.Wo, ON = Ol * g.N;  // local outputs and pairs (<= L.ONm)
  const int o_base = oya * g.Wo;
  // ---- which rows of x do the outputs centred in the band pair it with? -----------------------------------
  for (int j = t; j < ON; j += T) {
    const int n = fdivi(j, Ol), o = o_base + (j - n * Ol);
    const int pc = tap_pixel(g, o, g.R, g.R);
    if (pc >= ra * g.W && pc < rb * g.W) {
      const int q = nbr_pixel(g, o, n);
      if (q >= 0) {
        atomicMin(&mm[2], fdivi(q, g.W));
        atomicMax(&mm[3], fdivi(q, g.W));
      }
    }
  }
  __syncthreads();
  const int w0 = mm[2], w1 = mm[3];
  const int p0 = w0 * g.W, np = (w1 - w0 + 1) * g.W, PS = np + 1;  // window pixels (np + 1 <= L.PSm)
  if (ON > L.ONm || PS > L.PSm) {
    // cannot happen if the launcher's bounds hold; never write past LDS: poison this band's gradient instead
    const int Q0 = (g.C + 3) >> 2, qa = blockIdx.y * L.Qwg, qe = min(Q0, qa + L.Qwg);
    for (int i = t; i < (qe - qa) * 4 * nr * g.W; i += T) {
      const int c = 4 * qa + i / (nr * g.W), rl = i % (nr * g.W);
      if (c < g.C)
        stx(gx, (long long)b * g.gB + (long long)c * g.sC + (long long)(ra + rl / g.W) * g.sH +
                (long long)(rl % g.W) * g.sW, __builtin_nanf(""), g.dtype);
    }
    return;
  }
  // ---- final tables -------------------------------------------------------------------------------------------
  for (int i = t; i < nr + g.W; i += T) {
    const bool rows = i < nr;
    uint2* list = rows ? yl + i * L.capY : xl + (i - nr) * L.capX;
    const unsigned cd = (unsigned)g.R;
    unsigned cnt = 0, ncentre = 0;
    for (int pass = 0; pass < 2; ++pass) {  // readers through the centre tap first, then the others, in slot order
      for (int w = 0; w < nslot; ++w) {
        const uint2 e = slots[i * nslot + w];
        if (e.x == 0xFFFFFFFFu) continue;
        const unsigned d = e.x & 0xFFu;
        if ((d == cd) != (pass == 0)) continue;
        const int oa = (int)(e.x >> 8), ca = (int)e.y;
        uint2 f;
        if (rows) {
          f.x = (unsigned)((int)d * g.k * Ol + (oa - oya) * g.Wo);
          f.y = (ca < 0 ? kNoCentre : (unsigned)((ca - w0) * g.W)) | ((d * (unsigned)g.k) << 20);
        } else {
          f.x = (unsigned)((int)d * Ol + oa);
          f.y = (ca < 0 ? kNoCentre : (unsigned)ca) | (d << 20);
        }
        list[cnt++] = f;
      }
      if (pass == 0) ncentre = cnt;
    }
    if (rows)
      yc[i] = cnt | (ncentre << 16);
    else
      xc[i - nr] = cnt | (ncentre << 16);
  }
  for (int j = t; j < ON; j += T) {
    const int n = fdivi(j, Ol), o = o_base + (j - n * Ol);
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
    // neighbour pixel in window coordinates; read only for outputs centred in the band, whose neighbours are
    // inside the window by construction (anything else, and zero padding, -> the zero pixel)
    const int ql = q - p0;
    nbq[j] = (unsigned short)((q < 0 || ql < 0 || ql >= np) ? np : ql);
  }

  // ---- channel loop ------------------------------------------------------------------------------------
  const int Q = (g.C + 3) >> 2;
  const int q_begin = blockIdx.y * L.Qwg, q_end = min(Q, q_begin + L.Qwg);
  const unsigned mid = (unsigned)((g.k * g.k) >> 1);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nbp = nr * g.W;  // band pixels
  for (int qc0 = q_begin; qc0 < q_end; qc0 += L.Cq) {
    const int cqn = min(L.Cq, q_end - qc0);
    __syncthreads();  // tables complete, slots consumed / previous slab consumed
    stage_quads(xs, x, g, b, qc0, cqn, nullptr, p0, np, t, T);
    __syncthreads();
    const int nqb = (cqn + QB - 1) / QB;
    for (int i = t; i < nqb * nbp; i += T) {
      const int qb = fdivi(i, nbp), rl = i - qb * nbp;
      const int ryl = fdivi(rl, g.W), rx = rl - ryl * g.W, ry = ra + ryl;
      const int r = (ry - w0) * g.W + rx;  // window pixel of this band pixel
      const float4* slab[QB];
      float4 a[QB], acc[QB];
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        slab[u] = xs + min(qb * QB + u, cqn - 1) * PS;
        a[u] = slab[u][r];
        acc[u] = zero4;
      }
      const uint2* yrow = yl + ryl * L.capY;
      const uint2* xrow = xl + rx * L.capX;
      gather_item<M, QB>(g, slab, a, acc, yrow, xrow, yc[ryl], xc[rx], cf, ON, nbq, Ol, (unsigned)np, mid);
#pragma unroll
      for (int u = 0; u < QB; ++u) {
        const int cq = qb * QB + u;
        if (cq < cqn) {
          const int c = 4 * (qc0 + cq);
          const long long base = (long long)b * g.gB + (long long)c * g.sC + (long long)ry * g.sH + (long long)rx * g.sW;
          stx(gx, base, acc[u].x, g.dtype);
          if (c + 1 < g.C) stx(gx, base + g.sC, acc[u].y, g.dtype);
          if (c + 2 < g.C) stx(gx, base + 2 * g.sC, acc[u].z, g.dtype);
          if (c + 3 < g.C) stx(gx, base + 3 * g.sC, acc[u].w, g.dtype);
        }
      }
    }
  }
}

// ---- forward --------------------------------------------------------------------------------------------
// Replaces nfp.py:132-159 (pad -> two frozen depthwise convs -> view -> measure over C) for every measure
// and geometry.  One workgroup = one image x one tile of outputs.  x is staged once per channel chunk into
// the float4[quad][pixel] slab; thread (output o, channel group) keeps the NN pair sums of o in registers
// and walks its channel quads with one ds_read_b128 per (quad, neighbour) = four channel terms; the
// channel groups are then combined through LDS in a fixed order (bitwise deterministic).  What a measure
// needs per PIXEL (norms, means: Meas::stat) is summed once per pixel, not once per pair.
struct PairsLds {
  int xs;    // float4 slab (word offset, multiple of 4)
  int st;    // float [2][PSm]  per-pixel stat sums of the window (the zero pixel's entry = 0)
  int piv;   // float [PSm]     per-pixel pivot (nfp_measures.h::Pivot), 0 where unused
  int tap;   // u16   [N + 1][Ot] window pixel each tap of each output of the tile reads (row N = centre)
  int red;   // float scratch for the cross-group sums
  int mm;    // int [2] min / max input pixel the tile reads
  int PSm;   // bound on window pixels + 1 (the launcher's row-window bound)
  int Cq;    // channel quads per slab
  int Ot;    // outputs per workgroup tile
  int G;     // channel groups in the pair loop (G * Ot <= blockDim)
  int Gs;    // channel groups in the per-pixel stat loop
  int Tt;    // threads (whole wavefronts) that build the index tables while the others stage the first slab
};

template <int M, int NN>
__global__ void __launch_bounds__(512) fwd_pairs(const KP g, const PairsLds L, const void* __restrict__ x,
                                                 void* __restrict__ out, float* __restrict__ saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NS = Meas<M>::NSTAT;
  float4* xs = (float4*)(lds + L.xs);
  float* st = lds + L.st;
  float* piv = lds + L.piv;
  unsigned short* tap = (unsigned short*)(lds + L.tap);
  float* red = lds + L.red;
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int Q = (g.C + 3) >> 2;
  const int o0 = blockIdx.y * L.Ot, on = min(L.Ot, g.O - o0);
  const int cg = fdivi(t, L.Ot), ol = t - cg * L.Ot;
  const bool active = cg < L.G && ol < on;
  const int nchunk = fdivi(Q + L.Cq - 1, L.Cq);
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  // ---- which rows of x does this tile of outputs read?  Only those are staged.  Its output rows oyA..oyB read
  // padded rows oyA*s - pad .. oyB*s + 2*R*dil - pad, and what padding folds back lies inside the clamped
  // window (the launcher bounds PSm with the same formula, or with the whole image for circular / over-padding).
  const bool whole = L.PSm == g.P + 1;
  int r0 = 0, r1 = g.H - 1;
  if (!whole) {
    const int oyA = fdivi(o0, g.Wo), oyB = fdivi(o0 + on - 1, g.Wo);
    r0 = max(0, oyA * g.stride - g.pad);
    r1 = min(g.H - 1, oyB * g.stride + 2 * g.R * g.dil - g.pad);
  }
  const int p0 = r0 * g.W, np = (r1 - r0 + 1) * g.W, PS = np + 1;
  if (Pivot<M>::v) {  // the slab holds x - pivot: the pivots come first
    for (int p = t; p < PS; p += T) {
      float v = 0.f;
      if (p < np) {
        const int pg = p0 + p, y = fdivi(pg, g.W), xx = pg - y * g.W;
        v = ldx(x, (long long)b * g.sB + (long long)y * g.sH + (long long)xx * g.sW, g.dtype);
      }
      piv[p] = v;
    }
    __syncthreads();
  }
  // The last wavefront builds the index tables (a few hundred dependent integer instructions: microseconds
  // for one wavefront, nothing for the chip) while the others stage the first slab.
  const int tw = T - L.Tt;
  if (t >= tw) {
    const int lane = t - tw;
    for (int i = lane; i < L.Ot * g.k; i += L.Tt) {  // thread (output, kernel row) walks a row of taps
      const int ky = fdivi(i, L.Ot), l = i - ky * L.Ot;
      const int oy = fdivi(o0 + l, g.Wo), ox = (o0 + l) - oy * g.Wo, mid = (g.k * g.k) >> 1;
      const int yy = map_index(oy * g.stride + ky * g.dil - g.pad, g.H, g.mode);
      for (int kx = 0, tp = ky * g.k; kx < g.k; ++kx, ++tp) {
        const int xx = map_index(ox * g.stride + kx * g.dil - g.pad, g.W, g.mode);
        const int px = (l < on && yy >= 0 && xx >= 0) ? yy * g.W + xx - p0 : np;  // window pixel; np = the zero
        tap[(tp == mid ? g.N : (tp < mid ? tp : tp - 1)) * L.Ot + l] = (unsigned short)px;
      }
    }
    for (int i = lane; i < 2 * PS; i += L.Tt) st[i] = 0.f;
    if (!Pivot<M>::v)
      for (int p = lane; p < PS; p += L.Tt) piv[p] = 0.f;
  } else {
    stage_quads(xs, x, g, b, 0, min(L.Cq, Q), Pivot<M>::v ? piv : nullptr, p0, np, t, tw);
  }
  NFP_STAMP(1);
  __syncthreads();
  NFP_STAMP(2);
  const int pcz = tap[g.N * L.Ot + ol];

  for (int n0 = 0; n0 < g.N; n0 += NN) {
    int q[NN];
    float acc[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) {
      acc[j] = 0.f;
      q[j] = n0 + j < g.N ? (int)tap[(n0 + j) * L.Ot + ol] : np;
    }
    for (int ch = 0; ch < nchunk; ++ch) {
      const int qc0 = ch * L.Cq, cqn = min(L.Cq, Q - qc0);
      if (n0 == 0 || nchunk > 1) {
        if (!(n0 == 0 && ch == 0)) {  // (the prologue staged slab 0)
          __syncthreads();            // previous slab consumed
          stage_quads(xs, x, g, b, qc0, cqn, Pivot<M>::v ? piv : nullptr, p0, np, t, T);
          __syncthreads();
        }
        if (NS > 0 && n0 == 0) {
          // per-pixel stat sums of this chunk: thread (pixel, group) -> scratch -> st
          for (int i = t; i < L.Gs * np; i += T) {
            const int gs = fdivi(i, np), p = i - gs * np;
            float s0 = 0.f, s1 = 0.f;
            for (int cq = gs; cq < cqn; cq += L.Gs) {
              const float4 a = xs[cq * PS + p];  // padding channels are 0 and stat(0) adds nothing
              Meas<M>::stat(a.x, s0, s1);
              Meas<M>::stat(a.y, s0, s1);
              Meas<M>::stat(a.z, s0, s1);
              Meas<M>::stat(a.w, s0, s1);
            }
            red[(gs * 2) * np + p] = s0;
            red[(gs * 2 + 1) * np + p] = s1;
          }
          __syncthreads();
          for (int i = t; i < 2 * np; i += T) {
            const int k = fdivi(i, np), p = i - k * np;
            float s = st[k * PS + p];
            for (int gs = 0; gs < L.Gs; ++gs) s += red[(gs * 2 + k) * np + p];
            st[k * PS + p] = s;
          }
          __syncthreads();
        }
      }
      if (ch == 0 && n0 == 0) NFP_STAMP(3);
      if (active) {
        for (int cq = cg; cq < cqn; cq += L.G) {
          const float4* sl = xs + cq * PS;
          const float4 a = sl[pcz];
#pragma unroll
          for (int j = 0; j < NN; ++j) {
            const float4 bv = sl[q[j]];  // padding channels: term(0, 0) = 0 for every measure
            acc[j] += (Meas<M>::term(a.x, bv.x, g) + Meas<M>::term(a.y, bv.y, g)) +
                      (Meas<M>::term(a.z, bv.z, g) + Meas<M>::term(a.w, bv.w, g));
          }
        }
      }
    }
    // combine the channel groups (fixed order), finalise, store
    if (n0 == 0) NFP_STAMP(4);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < NN; ++j) red[(cg * NN + j) * L.Ot + ol] = acc[j];
    }
    __syncthreads();
    // thread (output ol, row cg): one division per thread for the whole kernel (ol, cg above).  With many
    // channel groups and spare rows of threads the sum over groups is split into `parts` interleaved partial
    // sums first (written back over group rows 0..parts-1, which only their own thread reads).
    const int rows = fdivi(T, L.Ot), NNv = min(NN, g.N - n0);
    const int parts = max(1, min(min(fdivi(rows, NNv), 8), L.G));
    if (parts > 1) {
      float s = 0.f;
      const int part = fdivi(cg, NNv), j = cg - part * NNv;
      const bool mine = ol < on && part < parts;
      if (mine)
        for (int c2 = part; c2 < L.G; c2 += parts) s += red[(c2 * NN + j) * L.Ot + ol];
      if (mine) red[(part * NN + j) * L.Ot + ol] = s;
      __syncthreads();
      if (ol < on && cg < NNv) {
        const int n = n0 + cg;
        float v = 0.f;
        for (int p2 = 0; p2 < parts; ++p2) v += red[(p2 * NN + cg) * L.Ot + ol];
        const int qz = tap[n * L.Ot + ol];
        stx(out, ((long long)b * g.N + n) * g.O + o0 + ol,
            Meas<M>::fin(v, st[pcz], st[PS + pcz], st[qz], st[PS + qz], g), g.odtype);
      }
    } else if (ol < on) {
      for (int j = cg; j < NNv; j += rows) {
        const int n = n0 + j;
        float s = 0.f;
        for (int c2 = 0; c2 < L.G; ++c2) s += red[(c2 * NN + j) * L.Ot + ol];
        const int qz = tap[n * L.Ot + ol];
        stx(out, ((long long)b * g.N + n) * g.O + o0 + ol,
            Meas<M>::fin(s, st[pcz], st[PS + pcz], st[qz], st[PS + qz], g), g.odtype);
      }
    }
  }
  NFP_STAMP(5);
  if constexpr (NS > 0) {
    // per-input-pixel stats for backward, [B][NSTAT][P]: every tile stores the pixels of its window (tiles that
    // share rows store identical values: the same channel order in every workgroup)
    if (saved != nullptr) {
      float* sv = saved + (long long)b * NS * g.P + p0;
      for (int p = t; p < np; p += T) {
        sv[p] = Meas<M>::save0(st[p], st[PS + p], g) + (Pivot<M>::v ? piv[p] : 0.f);
        if (NS > 1) sv[g.P + p] = Meas<M>::save1(st[p], st[PS + p], g);
      }
    }
  }
}

}  // namespace nfp
