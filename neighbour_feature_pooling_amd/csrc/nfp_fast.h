// nfp_fast.h — the hot-path kernels: stride 1, dilation 1, padding == R ("same" maps, every
// in-tree caller: NFP_Pooling.py:14, resnet18.py:20, texture_pooling.py:232,302), padding_mode
// zeros / reflect / replicate, cosine (nfp.py:150-159) and L2 = Norm p=2 (nfp.py:141-148),
// C % 4 == 0, H*W <= 1024, NCHW or channels-last.
//
// HBM/LDS layout.  A workgroup owns one image (forward) or one image x channel block
// (backward).  x[b, c0:c0+cc] is staged ONCE into LDS as float4[cc/4][P]: four consecutive
// channels of one pixel share a 16-byte slot, so every neighbour access is one ds_read_b128
// at a compile-time pixel offset from the thread's own slot.  Thread (p, lane) = (t % P, t / P)
// owns pixel p for channel quads lane, lane+G, ...; the same mapping stages, computes and
// (backward) stores, and consecutive threads touch consecutive pixels (coalesced, no bank
// conflicts).
//
// Forward: half stencil.  sim(p,q) is symmetric, so only the N/2 "forward" in-image pairs
// (dy>0, or dy==0 && dx>0) plus |x_p|^2 are summed over channels; the k*k-1 outputs of a
// pixel, including its reflect/replicate/zero-padded taps, are table lookups afterwards.
// Per output pixel: C*e bytes read, N*e written, nothing else touches HBM.
//
// Backward: for cosine and L2 the gradient is LINEAR in x once the per-pair scalars are
// known:  grad_x[c][r] = sum_t Wm[r][t] * x[c][t],  t in the (2R+1)^2 window of r, with
//   cosine: pair (p,q), g=grad_out, s=out:  Wm[p][q]+=g ip iq, Wm[q][p]+=g ip iq,
//           Wm[p][p]-=g s /(|p| m_p), Wm[q][q]-=g s /(|q| m_q)   (m = max(|.|,eps), ip = 1/m)
//   L2:     c = -+g/d:  Wm[p][p]+=c, Wm[q][q]+=c, Wm[p][q]-=c, Wm[q][p]-=c
// Wm (P x (2R+1)^2 floats) is built in LDS in GATHER form — every entry is summed by one
// thread in a fixed order, so the result is bitwise reproducible and no atomics are used;
// the padding adjoint is folded into Wm.  Then one pass: read x slab, write grad_x.
#pragma once
#include "nfp_measures.h"

namespace nfp {

constexpr int kMaxK = 8;  // channel quads staged per thread per chunk
constexpr int kFwdThreads = 1024;
constexpr int kBwdThreads = 512;  // backward keeps (2R+1)^2 weights + offsets + staged x in registers

template <int R>
struct Win {
  static constexpr int K = 2 * R + 1, K2 = K * K, N = K2 - 1, NF = N / 2;
};

// forward direction d in [0, NF): (0,1..R), then rows dy=1..R with dx=-R..R
template <int R>
__device__ __forceinline__ void fdir(int d, int& dy, int& dx) {
  if (d < R) {
    dy = 0;
    dx = d + 1;
  } else {
    int e = d - R;
    dy = 1 + e / (2 * R + 1);
    dx = e % (2 * R + 1) - R;
  }
}
template <int R>
__device__ __forceinline__ int fidx(int dy, int dx) {
  return dy == 0 ? dx - 1 : R + (dy - 1) * (2 * R + 1) + (dx + R);
}

__device__ __forceinline__ float4 ld4(const void* x, const KP& g, bool nhwc, long long img, int c, int p) {
  // four consecutive channels c..c+3 of pixel p
  float4 v;
  if (g.dtype == NFP_F32) {
    const float* s = (const float*)x + img;
    if (nhwc) {
      v = *(const float4*)(s + (long long)p * g.C + c);
    } else {
      const float* q = s + (long long)c * g.P + p;
      v.x = q[0];
      v.y = q[g.P];
      v.z = q[2 * g.P];
      v.w = q[3 * g.P];
    }
  } else {
    const uint16_t* s = (const uint16_t*)x + img;
    if (nhwc) {
      uint2 u = *(const uint2*)(s + (long long)p * g.C + c);
      v.x = __uint_as_float(u.x << 16);
      v.y = __uint_as_float(u.x & 0xffff0000u);
      v.z = __uint_as_float(u.y << 16);
      v.w = __uint_as_float(u.y & 0xffff0000u);
    } else {
      const uint16_t* q = s + (long long)c * g.P + p;
      v.x = bf16_to_f32(q[0]);
      v.y = bf16_to_f32(q[g.P]);
      v.z = bf16_to_f32(q[2 * g.P]);
      v.w = bf16_to_f32(q[3 * g.P]);
    }
  }
  return v;
}

__device__ __forceinline__ void st4(void* x, const KP& g, bool nhwc, long long img, int c, int p, float4 v) {
  if (g.dtype == NFP_F32) {
    float* s = (float*)x + img;
    if (nhwc) {
      *(float4*)(s + (long long)p * g.C + c) = v;
    } else {
      float* q = s + (long long)c * g.P + p;
      q[0] = v.x;
      q[g.P] = v.y;
      q[2 * g.P] = v.z;
      q[3 * g.P] = v.w;
    }
  } else {
    uint16_t* s = (uint16_t*)x + img;
    if (nhwc) {
      uint2 u;
      u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
      u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
      *(uint2*)(s + (long long)p * g.C + c) = u;
    } else {
      uint16_t* q = s + (long long)c * g.P + p;
      q[0] = f32_to_bf16(v.x);
      q[g.P] = f32_to_bf16(v.y);
      q[2 * g.P] = f32_to_bf16(v.z);
      q[3 * g.P] = f32_to_bf16(v.w);
    }
  }
}

// pair sum of two distinct in-image pixels from the half-stencil table Tt[(NF+1)][P]
template <int R>
__device__ __forceinline__ float pair_lookup(const float* Tt, const KP& g, int p, int q) {
  int py = p / g.W, px = p - py * g.W, qy = q / g.W, qx = q - qy * g.W;
  int dy = qy - py, dx = qx - px;
  if (dy > 0 || (dy == 0 && dx > 0)) return Tt[fidx<R>(dy, dx) * g.P + p];
  return Tt[fidx<R>(-dy, -dx) * g.P + q];
}

// ---- forward --------------------------------------------------------------------------------
template <int R, int M>
__global__ void __launch_bounds__(kFwdThreads) fwd_fast(const KP g, const void* __restrict__ x, void* __restrict__ out,
                                                 float* __restrict__ saved) {
  constexpr int N = Win<R>::N, NF = Win<R>::NF;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  float4* slab = lds4;
  const int P = g.P;
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int p = t % P, gl = t / P;
  const bool active = gl < g.G;
  const bool nhwc = !g.contig;
  const long long img = (long long)b * g.sB;

  int off[NF];
  {
    const int py = p / g.W, px = p - py * g.W;
#pragma unroll
    for (int d = 0; d < NF; ++d) {
      int dy, dx;
      fdir<R>(d, dy, dx);
      bool ok = (px + dx >= 0) && (px + dx < g.W) && (py + dy < g.H);
      off[d] = ok ? dy * g.W + dx : 0;  // invalid pairs read the own slot: finite junk, never looked up
    }
  }
  float acc[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = 0.f;
  float nrm = 0.f;

  for (int c0 = 0; c0 < g.C; c0 += g.Cc) {
    const int ncq = min(g.Cc, g.C - c0) >> 2;
    float4 v[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      int cq = gl + k * g.G;
      if (active && cq < ncq) v[k] = ld4(x, g, nhwc, img, c0 + 4 * cq, p);
    }
    if (c0 > 0) __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      int cq = gl + k * g.G;
      if (active && cq < ncq) slab[cq * P + p] = v[k];
    }
    __syncthreads();
    if (active) {
      for (int cq = gl; cq < ncq; cq += g.G) {
        const float4* row = slab + cq * P + p;
        const float4 a = row[0];
        nrm = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, fmaf(a.w, a.w, nrm))));
#pragma unroll
        for (int d = 0; d < NF; ++d) {
          const float4 q = row[off[d]];
          if (M == NFP_COSINE) {
            acc[d] = fmaf(a.x, q.x, fmaf(a.y, q.y, fmaf(a.z, q.z, fmaf(a.w, q.w, acc[d]))));
          } else {
            float e0 = a.x - q.x, e1 = a.y - q.y, e2 = a.z - q.z, e3 = a.w - q.w;
            acc[d] = fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, fmaf(e3, e3, acc[d]))));
          }
        }
      }
    }
  }
  // cross-lane (channel group) reduction through LDS, fixed order
  __syncthreads();
  float* red = (float*)lds4;                 // [G][NF+1][P]
  float* Tt = red + g.G * (NF + 1) * P;      // [NF+1][P]
  if (active) {
#pragma unroll
    for (int d = 0; d < NF; ++d) red[(gl * (NF + 1) + d) * P + p] = acc[d];
    red[(gl * (NF + 1) + NF) * P + p] = nrm;
  }
  __syncthreads();
  for (int i = t; i < (NF + 1) * P; i += T) {
    float s = 0.f;
    for (int gg = 0; gg < g.G; ++gg) s += red[gg * (NF + 1) * P + i];
    Tt[i] = s;
  }
  __syncthreads();
  const float* n2 = Tt + NF * P;
  for (int i = t; i < N * P; i += T) {
    const int n = i / P, pp = i - n * P;
    const int q = nbr_pixel(g, pp, n);
    float v;
    if (M == NFP_COSINE) {
      float s = 0.f;
      if (q >= 0) {
        float ip = 1.f / fmaxf(sqrtf(n2[pp]), g.eps), iq = 1.f / fmaxf(sqrtf(n2[q]), g.eps);
        float dot = (q == pp) ? n2[pp] : pair_lookup<R>(Tt, g, pp, q);
        s = dot * ip * iq;
      }
      v = g.similarity ? s : 1.f - s;
    } else {
      float d2;
      if (g.diff)
        d2 = q < 0 ? n2[pp] : (q == pp ? 0.f : pair_lookup<R>(Tt, g, pp, q));
      else
        d2 = q < 0 ? 0.f : n2[q];  // 'Norm' quirk (nfp.py:74 vs 85): |neighbour|
      float dd = sqrtf(d2);
      v = g.similarity ? -dd : dd;
    }
    stx(out, ((long long)b * N + n) * P + pp, v, g.dtype);
  }
  if (M == NFP_COSINE && saved != nullptr)
    for (int i = t; i < P; i += T) saved[(long long)b * P + i] = sqrtf(n2[i]);
}

// ---- backward -------------------------------------------------------------------------------
template <int R, int M>
__global__ void __launch_bounds__(kBwdThreads) bwd_fast(const KP g, const void* __restrict__ x, const void* __restrict__ go,
                                                 const void* __restrict__ out, const float* __restrict__ saved,
                                                 void* __restrict__ gx) {
  constexpr int K = Win<R>::K, K2 = Win<R>::K2, N = Win<R>::N;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int P = g.P;
  // LDS: Wt (lives to the end) | union { x slab , coefficient tables (dead once Wt is built) }
  float* Wt = (float*)lds4;                             // [P][K2] gathered weights
  float4* slab = lds4 + ((P * K2 + 3) >> 2);            // [Cc/4][P]
  float* CR = (float*)slab;                             // [P][N] cross coefficient of pair (p, n)
  float* SP = CR + P * N;                               // [P][N] self coefficient on the centre
  float* SQ = SP + P * N;                               // [P][N] self coefficient on the neighbour
  float* Sq2 = SQ + P * N;                              // [P][K2] neighbour-role self terms per window slot
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int cb0 = blockIdx.y * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  const int p = t % P, gl = t / P;
  const bool active = gl < g.G;
  const bool nhwc = !g.contig;
  const long long img = (long long)b * g.sB;

  // x loads of the first chunk go out before the coefficient phase so they fly under it
  float4 v[kMaxK];
  {
    const int ncq = min(g.Cc, cb1 - cb0) >> 2;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      int cq = gl + k * g.G;
      if (active && cq < ncq) v[k] = ld4(x, g, nhwc, img, cb0 + 4 * cq, p);
    }
  }
  // A1: per-pair coefficients
  for (int i = t; i < P * N; i += T) {
    const int pp = i / N, n = i - pp * N;
    const long long oi = ((long long)b * N + n) * P + pp;
    const float gv = ldx(go, oi, g.dtype), ov = ldx(out, oi, g.dtype);
    const int q = nbr_pixel(g, pp, n);
    float cr = 0.f, sp = 0.f, sq = 0.f;
    if (M == NFP_COSINE) {
      if (q >= 0) {
        const float np_ = saved[(long long)b * P + pp], nq = saved[(long long)b * P + q];
        const float s = g.similarity ? ov : 1.f - ov;
        const float sg = g.similarity ? gv : -gv;
        const float ip = 1.f / fmaxf(np_, g.eps), iq = 1.f / fmaxf(nq, g.eps);
        cr = sg * ip * iq;
        sp = np_ > 0.f ? -sg * s * ip / np_ : 0.f;
        sq = nq > 0.f ? -sg * s * iq / nq : 0.f;
      }
    } else {
      const float d = fabsf(ov);
      const float c = d == 0.f ? 0.f : (g.similarity ? -gv : gv) / d;
      if (g.diff) {
        sp = c;
        if (q >= 0) {
          cr = -c;
          sq = c;
        }
      } else if (q >= 0) {
        sq = c;  // d|x_q| / dx_q only
      }
    }
    CR[i] = cr;
    SP[i] = sp;
    SQ[i] = sq;
  }
  __syncthreads();
  // A2: gather.  Entry (r, slot j) with t = r + delta_j in the image.
  for (int i = t; i < P * K2; i += T) {
    const int r = i / K2, j = i - r * K2;
    const int ry = r / g.W, rx = r - ry * g.W;
    const int ty = ry + j / K - R, tx = rx + j % K - R;
    float w = 0.f, s2 = 0.f;
    if (ty >= 0 && ty < g.H && tx >= 0 && tx < g.W) {
      const int tt = ty * g.W + tx;
      for (int n = 0; n < N; ++n)  // r as centre, tt as its neighbour n
        if (nbr_pixel(g, r, n) == tt) w += CR[r * N + n];
      for (int n = 0; n < N; ++n)  // tt as centre, r as its neighbour n
        if (nbr_pixel(g, tt, n) == r) {
          w += CR[tt * N + n];
          s2 += SQ[tt * N + n];
        }
      if (tt == r)
        for (int n = 0; n < N; ++n) w += SP[r * N + n];
    }
    Wt[i] = w;
    Sq2[i] = s2;
  }
  __syncthreads();
  // A3: fold the neighbour-role self terms into the diagonal, fixed order
  for (int r = t; r < P; r += T) {
    float s = Wt[r * K2 + K2 / 2];
    for (int j = 0; j < K2; ++j) s += Sq2[r * K2 + j];
    Wt[r * K2 + K2 / 2] = s;
  }
  __syncthreads();
  float w[K2];
  int off[K2];
  {
    const int py = p / g.W, px = p - py * g.W;
#pragma unroll
    for (int j = 0; j < K2; ++j) {
      const int dy = j / K - R, dx = j % K - R;
      const bool ok = py + dy >= 0 && py + dy < g.H && px + dx >= 0 && px + dx < g.W;
      off[j] = ok ? dy * g.W + dx : 0;
      w[j] = ok ? Wt[p * K2 + j] : 0.f;
    }
  }
  // B: one pass over the channel block
  for (int c0 = cb0; c0 < cb1; c0 += g.Cc) {
    const int ncq = min(g.Cc, cb1 - c0) >> 2;
    if (c0 > cb0) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kMaxK; ++k) {
        int cq = gl + k * g.G;
        if (active && cq < ncq) v[k] = ld4(x, g, nhwc, img, c0 + 4 * cq, p);
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      int cq = gl + k * g.G;
      if (active && cq < ncq) slab[cq * P + p] = v[k];
    }
    __syncthreads();
    if (active) {
      for (int cq = gl; cq < ncq; cq += g.G) {
        const float4* row = slab + cq * P + p;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < K2; ++j) {
          const float4 q = row[off[j]];
          r4.x = fmaf(w[j], q.x, r4.x);
          r4.y = fmaf(w[j], q.y, r4.y);
          r4.z = fmaf(w[j], q.z, r4.z);
          r4.w = fmaf(w[j], q.w, r4.w);
        }
        st4(gx, g, nhwc, img, c0 + 4 * cq, p, r4);
      }
    }
  }
}

}  // namespace nfp
