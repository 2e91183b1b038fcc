// nfp_mfma.h — forward on the matrix cores for bf16 channels-last feature maps (ViT tokens: config 5).
//
// Replaces nfp.py:132-159 like fwd_fast, for the case where the matrix cores are the right tool: bf16 storage
// (products of two bf16 are exact in f32, so the MFMA's f32 accumulation is the same arithmetic as the vector
// path, in a different order) and a channels-last layout, which IS the A / B fragment layout of
// v_mfma_f32_32x32x16_bf16 — lane (r, h) holds 8 consecutive channels of pixel r: one 16-byte global load, no
// LDS staging, no transposition.
//
// Per image, the pair sums x_p . x_q are entries of the Gram matrix G = X Xt.  Neighbours are at most
// R*W + R pixels apart, so only the 32x32 tiles within D = ceil((R*W + R)/32) of the diagonal are needed, and
// by symmetry only those on or above it: a wavefront per tile, C/16 MFMAs each (config 5: 13 tiles x 12).
// Cosine reads G_pq, G_pp, G_qq; L2 is G_pp + G_qq - 2 G_pq.  The three come from the same products summed in
// the same order, so identical vectors still give exactly 0 (the cancellation that rules the Gram form out
// for f32 inputs is bounded here by the 8-bit mantissa of the inputs: see DESIGN.md).
#pragma once
#include "nfp_fast.h"

namespace nfp {

constexpr int kGramLd = 33;  // row stride of a stored tile (floats): column reads by 32 lanes hit 32 banks

// LDSX: the image is first copied into LDS (coalesced 16-byte pieces, rows padded by 16 bytes so that the 32
// fragment rows of a ds_read_b128 fall on distinct banks) and every tile reads its fragments from there; reading
// them from global memory instead costs each pixel row three times over in half-used 32-byte sectors (measured:
// 16 us against 23 us for the vector kernel, the traffic of the re-reads).  Without LDSX (image + tiles beyond
// LDS) the fragments come from global memory.
// NCHW: bf16 NCHW input, transposed into the same LDS image (thread = (pixel, 8 channels): eight 2-byte loads,
// coalesced along the pixel axis, packed into one 16-byte LDS write); needs LDSX.
template <int R, int M, bool LDSX, bool NCHW = false>
// gap / nfpm non-null (LDSX only) = the fused tail of models/NFP_Pooling.py:27-31: gap[b,c] = mean over pixels of x,
// taken from the LDS image; nfpm[b,n] = mean over pixels of the map values (before their bf16 rounding).
__global__ void __launch_bounds__(1024) fwd_gram(const KP g, const void* __restrict__ x, void* __restrict__ out,
                                                float* __restrict__ saved, int D, const unsigned char* __restrict__ ws,
                                                float* __restrict__ gap, float* __restrict__ nfpm) {
  constexpr int K = 2 * R + 1, N = K * K - 1;
  extern __shared__ __attribute__((aligned(16))) float Gt[];  // [nt * (D + 1)][32][kGramLd], then the image
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int P = g.P, C = g.C, nt = (P + 31) >> 5;
  const int wave = t >> 6, lane = t & 63, r = lane & 31, h = lane >> 5;
  const uint16_t* xb = (const uint16_t*)x + (long long)b * g.sB;  // channels-last: pixel p, channel c at xb[p * C + c]
  const int rowq = (C >> 3) + 1;  // 16-byte pieces per LDS row (one of padding)
  uint4* xl = (uint4*)(Gt + ((nt * (D + 1) * 32 * kGramLd + 3) & ~3));
  static_assert(LDSX || !NCHW, "the NCHW variant transposes through LDS");
  // output map: thread (p, n = gl, gl + G, ...).  Which pixel does tap n of pixel p read?  From the descriptor's
  // workspace table `ft` (nfp_tables.h: low 9 bits = pixel, bits 22-23 = 2 for a zero-padded tap) when the caller has
  // one — the first entry is loaded here, in flight during the tiles — else computed per thread
  const int G = max(1, T / P), gl = fdivi(t, P), pr = t - gl * P;
  const int p = gl < G ? pr : 0;  // (surplus threads idle through the barriers below)
  const uint32_t* ftt = ws != nullptr ? (const uint32_t*)(ws + ws_layout(P, R, g.mode).ft) : nullptr;
  // (round 4: ALL of the thread's table entries are requested here — up to kFte output rounds; round 3 prefetched the
  // first and loaded the others inside the output loop, one dependent L2 round trip per round: four of them at config 5)
  constexpr int kFte = 6;
  uint32_t fte[kFte];
#pragma unroll
  for (int k = 0; k < kFte; ++k) fte[k] = ftt != nullptr ? ftt[min(gl + k * G, N - 1) * P + p] : 0u;
  if (LDSX) {
    const int cq = C >> 3;  // pieces per pixel row
    if (NCHW) {
      for (int i = t; i < P * cq; i += T) {
        const int k = fdivi(i, P), pp = i - k * P;  // pixel fastest: coalesced
        const uint16_t* src = xb + (long long)(8 * k) * P + pp;
        uint16_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long long)u * P];
        uint4 w;
        w.x = v[0] | ((uint32_t)v[1] << 16);
        w.y = v[2] | ((uint32_t)v[3] << 16);
        w.z = v[4] | ((uint32_t)v[5] << 16);
        w.w = v[6] | ((uint32_t)v[7] << 16);
        xl[pp * rowq + k] = w;
      }
    } else {
      for (int i = t; i < P * cq; i += T) {
        const int pp = fdivi(i, cq), k = i - pp * cq;
        xl[pp * rowq + k] = ((const uint4*)xb)[i];
      }
    }
    __syncthreads();
    if (gap != nullptr) {
      // channel means: 32 adjacent lanes per channel octet, lane `part` sums pixels part, part + 32, ...; fixed DPP tree
      const int cq = C >> 3, part = t & 31;
      for (int k0 = 0; k0 < cq; k0 += T >> 5) {
        const int k = min(k0 + (t >> 5), cq - 1);
        float s8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] = 0.f;
        for (int pp = part; pp < P; pp += 32) {
          const uint4 w = xl[pp * rowq + k];
          const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            s8[2 * u] += __uint_as_float(wd[u] << 16);
            s8[2 * u + 1] += __uint_as_float(wd[u] & 0xFFFF0000u);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] = group_sum(s8[u], 32) * g.invP;
        if (part == 31 && k0 + (t >> 5) < cq) {
          float4* dst = (float4*)(gap + (long long)b * C + 8 * k);
          dst[0] = make_float4(s8[0], s8[1], s8[2], s8[3]);
          dst[1] = make_float4(s8[4], s8[5], s8[6], s8[7]);
        }
      }
    }
  }

  // ---- Gram tiles: tile pair (i, j = i + d), one wavefront each ---------------------------------------------
  for (int pi = wave; pi < nt * (D + 1); pi += T >> 6) {
    const int i = fdivi(pi, D + 1), j = i + (pi - i * (D + 1));
    if (j >= nt) continue;
    const int pa = min(32 * i + r, P - 1), pb = min(32 * j + r, P - 1);  // rows past P repeat the last pixel
    const uint4* A = LDSX ? xl + pa * rowq + h : (const uint4*)(xb + (long long)pa * C + 8 * h);
    const uint4* B = LDSX ? xl + pb * rowq + h : (const uint4*)(xb + (long long)pb * C + 8 * h);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int kk = 0; kk < (C >> 4); ++kk) {  // 16 channels per step = two 16-byte pieces per pixel row
      const uint4 av = A[2 * kk], bv = B[2 * kk];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc,
                                                    0, 0, 0);
    }
    float* tile = Gt + (long long)pi * 32 * kGramLd;
#pragma unroll
    for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * h) * kGramLd + r] = acc[e];  // G[32i+row][32j+r]
  }
  __syncthreads();

  // ---- outputs: thread (p, n = gl, gl + G, ...), as fwd_fast ------------------------------------------------------
  auto gram = [&](int a, int c) -> float {  // G[a][c] for pixels within the band
    const int ta = a >> 5, tc = c >> 5;
    const int lo = min(ta, tc), dd = abs(ta - tc);
    const int row = ta <= tc ? (a & 31) : (c & 31), col = ta <= tc ? (c & 31) : (a & 31);
    return Gt[((long long)(lo * (D + 1) + dd) * 32 + row) * kGramLd + col];
  };
  const int py = fdivi(p, g.W), px = p - py * g.W;
  NbrMap<R> nm;
  if (ftt == nullptr) nm.init(g, py, px);
  void* ob = (char*)out + (long long)b * N * P * 2;
  // per-pixel |x|^2 (and 1/max(|x|, eps) for cosine) once, in the LDS words behind the image / tiles' diagonal use
  float* n2t = (float*)xl;  // the image is dead: every tile is finished (barrier above)
  if (gl == 0 && t < P) {
    const float d = gram(p, p);
    n2t[p] = d;
    if (M == NFP_COSINE) n2t[P + p] = inv_norm(d, g.inv_eps);
  }
  __syncthreads();
  float* vm = n2t + 2 * P;  // [N][P] map values for the pooled means (inside the dead image)
  const float n2p = n2t[p];
  const float ip = M == NFP_COSINE ? n2t[P + p] : 0.f;
  int it = 0;
  for (int n = gl; n < (gl < G ? N : 0); n += G, ++it) {
    int q;
    if (ftt != nullptr) {
      uint32_t e = fte[0];
#pragma unroll
      for (int k = 1; k < kFte; ++k) e = it == k ? fte[k] : e;
      if (it >= kFte) e = ftt[n * P + p];
      q = (e >> 22) == 2u ? -1 : (int)(e & 511u);
    } else {
      int qy, qx;
      q = nm.get(g, n, qy, qx);
    }
    const int qc = max(q, 0);
    const float n2q = n2t[qc], dot = gram(p, qc);
    float v;
    if (M == NFP_COSINE) {
      const float s = q < 0 ? 0.f : dot * ip * n2t[P + qc];
      v = g.similarity ? s : 1.f - s;
    } else {
      float d2;
      if (g.diff)
        d2 = q < 0 ? n2p : (q == p ? 0.f : fmaxf(n2p + n2q - 2.f * dot, 0.f));
      else
        d2 = q < 0 ? 0.f : n2q;  // 'Norm' quirk (nfp.py:74 vs 85): |neighbour|
      const float dd = __builtin_amdgcn_sqrtf(d2);
      v = g.similarity ? -dd : dd;
    }
    if (nfpm == nullptr || g.pool_map) stx(ob, n * P + p, v, NFP_BF16);
    if (nfpm != nullptr) vm[n * P + p] = v;
  }
  if (M == NFP_COSINE && saved != nullptr && gl == 0) saved[(long long)b * P + p] = __builtin_amdgcn_sqrtf(n2p);
  if (nfpm != nullptr) {
    lds_barrier();   // (the map stores just issued drain under the sums — nfp_common.h)
    // wave w reduces map n = w, w + nwaves, ...: lane-strided partial sums, then a fixed shuffle tree
    for (int n = wave; n < N; n += T >> 6) {
      float sacc = 0.f;
      for (int i = lane; i < P; i += 64) sacc += vm[n * P + i];
      for (int m = 32; m >= 1; m >>= 1) sacc += __shfl_xor(sacc, m);
      if (lane == 0) nfpm[(long long)b * N + n] = sacc * g.invP;
    }
  }
}

}  // namespace nfp
