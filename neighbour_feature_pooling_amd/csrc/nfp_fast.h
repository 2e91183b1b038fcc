// nfp_fast.h — the hot-path kernels: stride 1, dilation 1, padding == R ("same" maps, every
// in-tree caller: NFP_Pooling.py:14, resnet18.py:20, texture_pooling.py:232,302), padding_mode
// zeros / reflect / replicate, cosine (nfp.py:150-159) and L2 = Norm p=2 (nfp.py:141-148),
// C % 4 == 0, H*W <= 512, NCHW or channels-last.
//
// HBM/LDS layout.  A workgroup owns one image (forward) or one image x channel block
// (backward).  x[b, c0:c0+cc] is staged ONCE into LDS as float4[cc/4][P]: four consecutive
// channels of one pixel share a 16-byte slot, so every neighbour access is one ds_read_b128
// at a fixed pixel offset from the thread's own slot.  NCHW input is staged in 4-channel x
// 4-pixel blocks (four 16-byte loads along the pixel axis, transposed in registers, four
// ds_write_b128); channels-last input is one 16-byte load per slot.  Compute thread
// (p, lane) = (t % P, t / P) owns pixel p for channel quads lane, lane+G, ...
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
#include <type_traits>
#include "nfp_tables.h"

namespace nfp {

// tuning knobs (A/B tested with scripts/ab_flags.py; the defaults are the measured best)
#ifndef NFP_RB
#define NFP_RB 3
#endif
#ifndef NFP_BWD_THREADS
#define NFP_BWD_THREADS 512
#endif
#ifndef NFP_UNROLL_B
#define NFP_UNROLL_B 1
#endif
constexpr int kRB = NFP_RB;  // NCHW staging: 4x4 blocks per thread per chunk
constexpr int kRN = 4;    // channels-last staging: slots per thread per chunk
constexpr int kBwdThreads = NFP_BWD_THREADS;  // backward keeps (2R+1)^2 weights + offsets + staged x in registers

// exact i / d for 0 <= i, quotient < 2048 (d >= 1): float multiply instead of the ~20-instruction
// integer division sequence
__device__ __forceinline__ int fast_div(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

// Image access goes through a buffer resource built from the wave-uniform image base: a 32-bit
// byte voffset per lane plus an SGPR soffset (the channel-row stride), so the four channel rows of a
// 4x4 block share ONE address VGPR and need no 64-bit vector arithmetic; stores can carry sc1
// (write-through), which streams grad_x out during the kernel instead of in the end-of-kernel flush.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
using Rsrc = __amdgpu_buffer_rsrc_t;
constexpr int kAuxSc1 = 16;  // buffer-store cache policy: sc1 = write-through
// grad_x store policy (A/B knob, scripts/ab_flags.py).  Plain write-back stores: the 196-byte NCHW rows a workgroup
// writes are adjacent in memory, and only the L2 can merge them into whole sectors — as write-through (sc1) dwords
// they left as partial sectors, 1.25x the bytes (WRITE_SIZE 8.03 MB for 6.42 MB of grad_x).  Round 1 measured sc1
// 0.45 us FASTER at the headline shape, on a buffer it rewrote in place (Infinity-Cache resident); writing a fresh
// grad_x per step, as training does, plain stores are faster at every batch: 6.2 vs 6.9 us at B = 64, 162 vs 266 us
// at B = 4096 (profiles/r02_i_store_policy_ab.txt).
#ifndef NFP_BWD_STORE_AUX
#define NFP_BWD_STORE_AUX 0
#endif
__device__ __forceinline__ Rsrc make_rsrc(const void* base, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0x7ffffff0LL ? 0x7ffffff0LL : bytes), 0x00020000);
}
template <bool BF>
__device__ __forceinline__ float4 load_px4(Rsrc r, int e, int srow) {  // 4 consecutive elements at element e (+ srow elems)
  if constexpr (!BF) {
    u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, e * 4, srow * 4, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  } else {
    u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, e * 2, srow * 2, 0);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
}
template <bool BF>
__device__ __forceinline__ float load_1(Rsrc r, int e, int srow) {
  if constexpr (!BF)
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, e * 4, srow * 4, 0));
  else
    return bf16_to_f32((uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, e * 2, srow * 2, 0));
}
template <bool BF>
__device__ __forceinline__ void store_px4(Rsrc r, int e, int srow, float4 v) {  // write-through
  if constexpr (!BF) {
    u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, e * 4, srow * 4, NFP_BWD_STORE_AUX);
  } else {
    u32x2 u = {f32_to_bf16x2(v.x, v.y),
               f32_to_bf16x2(v.z, v.w)};
    __builtin_amdgcn_raw_buffer_store_b64(u, r, e * 2, srow * 2, NFP_BWD_STORE_AUX);
  }
}
template <bool BF>
__device__ __forceinline__ void store_1(Rsrc r, int e, int srow, float v) {
  if constexpr (!BF)
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, e * 4, srow * 4, NFP_BWD_STORE_AUX);
  else
    __builtin_amdgcn_raw_buffer_store_b16((short)f32_to_bf16(v), r, e * 2, srow * 2, NFP_BWD_STORE_AUX);
}

// Staged registers of one chunk.  Every load is unconditional (indices clamped onto valid slots);
// a runtime select around a load makes hipcc serialise the loads behind vmcnt(0) waits.
template <bool NHWC>
struct Staged;
template <>
struct Staged<true> {
  float4 v[kRN];
};

// Slab row (one channel quad) stride in slots.  NCHW: P rounded up to 4.  Channels-last (round 4): the backward's lanes run
// along the channel GROUPS of a pixel (below), so 16 lanes of a ds_read / ds_write_b128 are 16 consecutive quads of one
// pixel, a row stride apart: the stride is made odd, which spreads them over the 16 bank columns.
#ifndef NFP_BWD_NHWC_LANES
#define NFP_BWD_NHWC_LANES 1   // (0 builds round 3's thread map: A/B)
#endif
__host__ __device__ inline int bwd_row_slots(int P, bool nhwc) {
  return (nhwc && NFP_BWD_NHWC_LANES) ? (((P + 3) & ~3) | 1) : ((P + 3) & ~3);
}
// channels-last: slot (cq, p) is 4 contiguous channels; thread (p, gl) takes cq = gl, gl+G, ...
template <bool BF>
__device__ __forceinline__ void stage_load(Staged<true>& s, Rsrc x, const KP& g, int c0, int ncq, int p, int gl,
                                           bool active) {
  const int g0 = active ? gl : 0;
  const int last = g0 < ncq ? g0 + ((ncq - 1 - g0) / g.G) * g.G : 0;
#pragma unroll
  for (int k = 0; k < kRN; ++k) {
    const int cq = min(g0 + k * g.G, last);
    s.v[k] = load_px4<BF>(x, p * g.C + c0 + 4 * cq, 0);
  }
}
__device__ __forceinline__ void stage_store(const Staged<true>& s, float4* slab, const KP& g, int ncq, int p, int gl,
                                            bool active) {
#pragma unroll
  for (int k = 0; k < kRN; ++k) {
    const int cq = gl + k * g.G;
    if (active && cq < ncq) slab[cq * bwd_row_slots(g.P, true) + swz(p)] = s.v[k];
  }
}

// NCHW staging without tail pixels (backward): ceil(P / 4) blocks per channel row; the last block of a row whose
// length is not a multiple of 4 starts at P - 4 instead and overlaps its predecessor (the same values are written
// twice).  No clamped tail loads: at P = 49 those were 8 of a thread's 20 load instructions, for 32 useful lanes.
struct StagedOvl {
  float4 blk[kRB][4];
};
template <bool BF>
__device__ __forceinline__ void stage_load_ovl(StagedOvl& s, Rsrc x, const KP& g, int c0, int ncq, int t, int T) {
  const int P = g.P, NQb = (P + 3) >> 2, nblk = ncq * NQb;
#pragma unroll
  for (int r = 0; r < kRB; ++r) {
    const int i = min(t + r * T, nblk - 1);
    const int cq = fdivi(i, NQb), pq = i - cq * NQb;
    const int e = (c0 + 4 * cq) * P + min(4 * pq, P - 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) s.blk[r][j] = load_px4<BF>(x, e, j * P);
  }
}
__device__ __forceinline__ void stage_store_ovl(const StagedOvl& s, float4* slab, const KP& g, int ncq, int t, int T) {
  const int P = g.P, NQb = (P + 3) >> 2, nblk = ncq * NQb, Pp = (P + 3) & ~3;
#pragma unroll
  for (int r = 0; r < kRB; ++r) {
    const int i = t + r * T;
    if (i < nblk) {
      const int cq = fdivi(i, NQb), pq = i - cq * NQb;
      const int ps = min(4 * pq, P - 4);
      float4* d = slab + cq * Pp;
      d[swz(ps)] = make_float4(s.blk[r][0].x, s.blk[r][1].x, s.blk[r][2].x, s.blk[r][3].x);
      d[swz(ps + 1)] = make_float4(s.blk[r][0].y, s.blk[r][1].y, s.blk[r][2].y, s.blk[r][3].y);
      d[swz(ps + 2)] = make_float4(s.blk[r][0].z, s.blk[r][1].z, s.blk[r][2].z, s.blk[r][3].z);
      d[swz(ps + 3)] = make_float4(s.blk[r][0].w, s.blk[r][1].w, s.blk[r][2].w, s.blk[r][3].w);
    }
  }
}

// 1 / max(sqrt(n2), eps) = min(rsqrt(n2), 1/eps): one v_rsq_f32 (1 ulp) instead of an IEEE sqrt and
// an IEEE divide (~25 instructions on the finalize critical path); rsqrt(0) = inf -> 1/eps.
__device__ __forceinline__ float inv_norm(float n2, float inv_eps) {
  return fminf(__builtin_amdgcn_rsqf(n2), inv_eps);
}

// ---- backward phase B on the matrix cores (bf16 storage) ----------------------------------------------------
// grad_x[r][c] = sum_t W[r][t] x[t][c] is a banded GEMM per image: W is the P x K2 window table `Wt` that phase A
// leaves in LDS, non-zero only for |t - r| <= R*W + R.  For a tile of 32 pixels the summed pixels t span
// 32 + 2*band (+ alignment) values: KW steps of 16.  Both operands want the SUMMED index contiguous per lane:
//   Xt [channel][pixel]  bf16  (the image block, transposed on the way in when the tensor is channels-last)
//   Wd [row pixel][t - ts] bf16, twice: hi = the top 16 bits of the f32 weight, lo = the next 16 (two MFMAs per
//      step keep 16 significand bits of every weight; the inputs themselves carry 8)
// and the same two LDS images serve both orientations: A = Xt, B = Wd gives D[channel][pixel] (lanes along
// pixels: coalesced NCHW stores), A = Wd, B = Xt gives D[pixel][channel] (channels-last stores).
// Row tiles are processed a few at a time (Wd of all of them does not fit next to Xt at config 5's shape: g.Tc).
__host__ __device__ __forceinline__ int odd_up(int v) { return v | 1; }
// A row tile's window of summed pixels starts at ts = (32 it - band) rounded DOWN to a whole 16-byte piece of Xt (8 pixels:
// lane half h reads piece 2 s + h of step s), never below 0; 32 it is a multiple of 8, so the rounding slack is the same for
// every tile: (-band) mod 8.  KW: 16-pixel steps that cover 32 rows + the band on both sides + that slack.  (Round 4: the
// start was rounded to a whole step before — 7 steps instead of 6 at config 5's 14 x 14, k = 5, and a Wd of 15 pieces per
// row instead of 13: three row tiles per round next to Xt instead of four.)
#ifndef NFP_GEMM_ALIGN8
#define NFP_GEMM_ALIGN8 1   // 0 = round 3's whole-step alignment (A/B)
#endif
__host__ __device__ __forceinline__ int gemm_kw(int band) {
  return NFP_GEMM_ALIGN8 ? (32 + 2 * band + ((-band) & 7) + 15) >> 4 : (32 + 2 * band + 30) >> 4;
}
__host__ __device__ __forceinline__ int gemm_ts(int it, int band) {
  return (32 * it - band) < 0 ? 0 : ((32 * it - band) & (NFP_GEMM_ALIGN8 ? ~7 : ~15));
}

// ---- Xt: 8 consecutive pixels of one channel per 16-byte piece; pixels past P are zero --------------------------
// The image block arrives 2-3 us after it is asked for, whatever the cache level, so each thread asks for its FIRST
// share as soon as phase A has its pair values (gemm_x_issue; every table row the gathers need was requested at entry:
// loads retire in order) and holds it in registers until the pair values under Xt are dead; gemm_stage_x then writes
// that share and fetches the rest.
// NHWC: 8 pixels x 8 channels per item: eight 16-byte loads (one per pixel), transposed in registers (v_perm_b32)
// into eight pieces (one per channel).  A wavefront takes a block of 4 channel octets x 16 pixel groups, laid on its
// lanes so that every 16 lanes write 2 octets x 8 groups: Xt rows are an odd number of pieces apart, so lanes along
// the octets — what the loads would like — land on two 16-byte LDS slots (measured: 3.7 of 21 us at config 5).
// NCHW with P % 4 == 0: a piece is two 8-byte loads, four pieces per thread.  NCHW with odd rows: 2-byte loads, not
// issued early.
__device__ __forceinline__ int gemm_xq(int P) { return odd_up(((P + 15) >> 4 << 1) + 1); }  // pieces per Xt row: P rounded up to 16 pixels, + padding, odd
constexpr int kGemmPre = 4;  // gather rounds whose table rows the matrix-core backward requests at entry
template <bool NHWC>
struct GemmX {
  uint4 v[NHWC ? 8 : 4];
};
__device__ __forceinline__ void gemm_block_of(int blk, int co4, int lane, int& k, int& gq) {
  const int bq = fdivi(blk, co4), bk = blk - bq * co4;
  k = 4 * bk + (lane & 1) + ((lane >> 3) & 2);
  gq = 16 * bq + ((lane >> 1) & 7) + ((lane >> 2) & 8);
}
__device__ __forceinline__ void gemm_fetch8(uint4 (&v)[8], const KP& g, const uint16_t* xb, int cb0, int k, int gq) {
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = *(const uint4*)(xb + (long long)min(8 * gq + u, g.P - 1) * g.C + cb0 + 8 * k);
}
__device__ __forceinline__ void gemm_fetch_row(uint4& v, const KP& g, const uint16_t* xb, int cb0, int pg, int i) {
  const int c = fdivi(i, pg), gs = i - c * pg;
  const uint16_t* src = xb + (long long)(cb0 + c) * g.P + 8 * gs;
  const uint2 lo = *(const uint2*)src;
  const uint2 hi = 8 * gs + 4 < g.P ? *(const uint2*)(src + 4) : make_uint2(0, 0);  // (P % 4 == 0: a whole half or none)
  v = make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// part 0, 1, 2 of the request (3 + 3 + 2 of the eight NHWC loads, 2 + 1 + 1 of the four NCHW pieces): one part in
// front of each gather round, so that no wavefront sits on a full load queue while it could be gathering
template <bool NHWC>
__device__ __forceinline__ void gemm_x_issue(GemmX<NHWC>& s, const KP& g, const uint16_t* xb, int cb0, int ncw, int t, int T,
                                             int part) {
  const int P = g.P, pg = (P + 7) >> 3;
  if constexpr (NHWC) {
    const int co4 = ncw >> 5, nblk = co4 * ((pg + 15) >> 4);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (wave < nblk) {  // (wavefronts past the last block, and lanes past the last pixel group, request nothing: the
      int k, gq;        // texture path takes every address it is given)
      gemm_block_of(wave, co4, t & 63, k, gq);
      if (gq < pg) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u / 3 == part) s.v[u] = *(const uint4*)(xb + (long long)min(8 * gq + u, P - 1) * g.C + cb0 + 8 * k);
      }
    }
  } else if ((P & 3) == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if ((u == 0 ? 0 : u - 1) == part) gemm_fetch_row(s.v[u], g, xb, cb0, pg, min(t + u * T, ncw * pg - 1));
  }
}
// Second form: the same request, ordered by the chunk a piece belongs to — chunk 0 in front of the first gather round, chunk 1
// in front of the second, the rest in front of the third: the first chunk's tiles start as soon as the weights are complete.
template <bool NHWC>
__device__ __forceinline__ void gemm_x_issue3(GemmX<NHWC>& s, const KP& g, const uint16_t* xb, int cb0, int ncw, int cx, int t, int T,
                                              int round) {
  const int P = g.P, pg = (P + 7) >> 3;
  if constexpr (NHWC) {
    const int co4 = ncw >> 5, nblk = co4 * ((pg + 15) >> 4);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (wave < nblk) {
      const int bk = wave - fdivi(wave, co4) * co4, c = fdivi(32 * bk, cx);
      if (min(c, 2) == round) {
        int k, gq;
        gemm_block_of(wave, co4, t & 63, k, gq);
        if (gq < pg) {
#pragma unroll
          for (int u = 0; u < 8; ++u) s.v[u] = *(const uint4*)(xb + (long long)min(8 * gq + u, P - 1) * g.C + cb0 + 8 * k);
        }
      }
    }
  } else if ((P & 3) == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(t + u * T, ncw * pg - 1), c = fdivi(fdivi(i, pg), cx);
      if (min(c, 2) == round) gemm_fetch_row(s.v[u], g, xb, cb0, pg, i);
    }
  }
}
template <bool NHWC>
__device__ __forceinline__ void gemm_stage_x(const KP& g, uint4* Xt, const uint16_t* xb, int cb0, int ncw, int t, int T,
                                             GemmX<NHWC>& pre) {
  const int P = g.P, pg = (P + 7) >> 3, xq = gemm_xq(P);
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = T >> 6;  // (wave-uniform, in an SGPR)
  if constexpr (NHWC) {
    const int co4 = ncw >> 5, nblk = co4 * ((pg + 15) >> 4);
    auto commit = [&](uint4 (&v)[8], int k, int gq) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (8 * gq + u >= P) v[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // channel 8k + j: its 16 bits of every pixel's piece
        uint32_t e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint4 &a = v[2 * u], &b = v[2 * u + 1];
          const uint32_t wa = (j >> 1) == 0 ? a.x : ((j >> 1) == 1 ? a.y : ((j >> 1) == 2 ? a.z : a.w));
          const uint32_t wb = (j >> 1) == 0 ? b.x : ((j >> 1) == 1 ? b.y : ((j >> 1) == 2 ? b.z : b.w));
          e[u] = __builtin_amdgcn_perm(wb, wa, (j & 1) ? 0x07060302u : 0x05040100u);  // {pixel 2u, pixel 2u + 1}
        }
        if (gq < pg) Xt[(long long)(8 * k + j) * xq + gq] = make_uint4(e[0], e[1], e[2], e[3]);
      }
    };
    int k, gq;
    if (wave < nblk) {  // the block requested at kernel entry
      gemm_block_of(wave, co4, lane, k, gq);
      commit(pre.v, k, min(gq, pg));
    }
    for (int blk = wave + nw; blk < nblk; blk += nw) {
      uint4 v[8];
      gemm_block_of(blk, co4, lane, k, gq);
      gq = min(gq, pg);  // (group pg: loads clamp to the last pixel, nothing is written)
      gemm_fetch8(v, g, xb, cb0, k, gq);
      commit(v, k, gq);
    }
  } else if ((P & 3) == 0) {
    const int np = ncw * pg;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // the pieces requested at kernel entry
      const int i = t + u * T, c = fdivi(i, pg);
      if (i < np) Xt[(long long)c * xq + (i - c * pg)] = pre.v[u];
    }
    for (int i0 = t + 4 * T; i0 < np; i0 += 4 * T) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) gemm_fetch_row(v[u], g, xb, cb0, pg, min(i0 + u * T, np - 1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * T, c = fdivi(i, pg);
        if (i < np) Xt[(long long)c * xq + (i - c * pg)] = v[u];
      }
    }
  } else {
    const int np = ncw * pg;
    for (int i = t; i < np; i += T) {
      const int c = fdivi(i, pg), gq = i - c * pg;  // pixel group fastest
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pp = min(8 * gq + u, P - 1);
        const uint16_t e = xb[(long long)(cb0 + c) * P + pp];
        v[u] = 8 * gq + u < P ? e : 0;
      }
      Xt[(long long)c * xq + gq] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
  }
}

// One 32 x 32 output tile on one wavefront: row tile `it` (its densified weights Wrow: hi image, lo image 32 rows further),
// channel tile ctx of the Xt image at hand = channel tile ct of the workgroup's block.
template <bool NHWC>
__device__ __forceinline__ void gemm_out_tile(const KP& g, const uint4* Wrow, const uint4* Xt, int it, int ctx, int ct, int band,
                                              int KW, int xq, int wq, uint16_t* gxb, int cb0, int lane, const float* gg) {
  const int P = g.P, C = g.C, r = lane & 31, h = lane >> 5;
  const int ts = gemm_ts(it, band);
  const int ks = min(KW, (P - ts + 15) >> 4);   // steps whose pixels exist
  const uint4* Wh = Wrow + (long long)r * wq + h;
  const uint4* Wl = Wh + 32 * wq;
  const uint4* Xr = Xt + (long long)(32 * ctx + r) * xq + (ts >> 3) + h;
  // Orientation: the accumulator holds 4 CONSECUTIVE rows per register group, so the output index that is
  // contiguous in memory goes on the rows and leaves as one 8-byte store per group: channels for
  // channels-last (A = Xt), pixels for NCHW (A = Wd; needs P % 4 == 0 for the alignment, otherwise pixels
  // stay on the lanes and leave as 2-byte stores).
  const bool rows_are_channels = NHWC || (P & 3) != 0;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) {  // element e of lane (r, h) is D[row (e & 3) + 8 (e >> 2) + 4 h][column r]
    const int ch = 32 * ct + (rows_are_channels ? (e & 3) + 8 * (e >> 2) + 4 * h : r);
    acc[e] = gg != nullptr ? gg[ch] : 0.f;
  }
  for (int s = 0; s < ks; ++s) {
    const bf16x8 wh = __builtin_bit_cast(bf16x8, Wh[2 * s]), wl = __builtin_bit_cast(bf16x8, Wl[2 * s]);
    const bf16x8 xv = __builtin_bit_cast(bf16x8, Xr[2 * s]);
    if (rows_are_channels) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, wl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xv, wh, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xv, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xv, acc, 0, 0, 0);
    }
  }
  if (NHWC) {            // D[row = channel][col = pixel]: 4 consecutive channels of pixel r per register group
    // Lane (r, h) holds channels 4h + {0-3, 8-11, 16-19, 24-27} of pixel r: four 8-byte pieces, and four store
    // instructions of 64 scattered pieces each were the longest part of a round (the texture path takes an address
    // per lane: 12 tiles x 4 x 64 per round).  v_permlane32_swap trades the packed groups between lanes r and
    // r + 32, so that each holds 8 consecutive channels twice: two 16-byte stores.
    const int pp = 32 * it + r;
    uint32_t pk[8];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      pk[2 * gq] = f32_to_bf16x2(acc[4 * gq], acc[4 * gq + 1]);
      pk[2 * gq + 1] = f32_to_bf16x2(acc[4 * gq + 2], acc[4 * gq + 3]);
    }
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2) {   // (groups gp, gp + 1): the upper lanes' gp <-> the lower lanes' gp + 1
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const auto sw = __builtin_amdgcn_permlane32_swap(pk[2 * gp + d], pk[2 * gp + 2 + d], false, false);
        pk[2 * gp + d] = sw[0];
        pk[2 * gp + 2 + d] = sw[1];
      }
    }
    if (pp < P) {
      uint16_t* dst = gxb + (long long)pp * C + cb0 + 32 * ct + 8 * h;
      *(uint4*)dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);          // channels 8h .. 8h + 7
      *(uint4*)(dst + 16) = make_uint4(pk[4], pk[5], pk[6], pk[7]);   // channels 16 + 8h .. 16 + 8h + 7
    }
  } else if ((P & 3) == 0) {  // D[row = pixel][col = channel]: 4 consecutive pixels of channel r per group
    uint16_t* dst = gxb + (long long)(cb0 + 32 * ct + r) * P + 32 * it + 4 * h;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      if (32 * it + 8 * gq + 4 * h < P) {  // (P % 4 == 0: a whole group or none)
        uint2 w;
        w.x = f32_to_bf16x2(acc[4 * gq], acc[4 * gq + 1]);
        w.y = f32_to_bf16x2(acc[4 * gq + 2], acc[4 * gq + 3]);
        *(uint2*)(dst + 8 * gq) = w;
      }
    }
  } else {               // D[row = channel][col = pixel], NCHW with odd rows: lanes along pixels, 2-byte stores
    const int pp = 32 * it + r;
    if (pp < P) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        gxb[(long long)(cb0 + 32 * ct + (e & 3) + 8 * (e >> 2) + 4 * h) * P + pp] = f32_to_bf16(acc[e]);
    }
  }
}

// gg: (fused pooling tail) this workgroup's channels of grad(GAP(x)) / P, staged in LDS by bwd_fast, or null: every
// grad_x[c][p] also gets gg[c - cb0].  (Read from global memory in the tile loop they cost 4 of 22 us at config 5.)
// Xt, Wd: the two operand images in LDS (placed by bwd_fast); pre: the share of x requested at kernel entry.
template <int R, bool NHWC>
__device__ __forceinline__ void bwd_gemm_phase(const KP& g, const float* Wt, uint4* Xt, uint4* Wd, GemmX<NHWC>& pre,
                                               const uint16_t* xb, uint16_t* gxb, int cb0, int cb1, int t, int T,
                                               const float* gg = nullptr) {
  constexpr int K = Win<R>::K, K2 = Win<R>::K2;
  const int P = g.P, band = R * g.W + R, nt = (P + 31) >> 5;
  const int KW = gemm_kw(band);                             // k-steps that cover a row tile's window after aligning its start
  const int xq = gemm_xq(P);                                // 16-byte pieces per Xt row
  const int wq = odd_up(2 * KW + 1);                        // pieces per Wd row
  const int ncw = cb1 - cb0;                                // channels of this workgroup (multiple of 32)
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = T >> 6;
  NFP_STAMP_INIT();  // (diagnostic build: stamps 7.. of the first two rounds)
  const int pg = (P + 7) >> 3;
  gemm_stage_x<NHWC>(g, Xt, xb, cb0, ncw, t, T, pre);
  for (int i = t; i < ncw * (xq - pg); i += T) {  // the rest of every row (alignment + padding) reads as zero
    const int c = fdivi(i, xq - pg), k = i - c * (xq - pg);
    Xt[(long long)c * xq + pg + k] = make_uint4(0, 0, 0, 0);
  }

  NFP_STAMP(7);
  const int nct = ncw >> 5;
  const int RT = g.Tc;  // row tiles per round: as many as fit in LDS next to Xt (launcher), at least 2
  for (int i0 = 0; i0 < nt; i0 += RT) {
    const int nrt = min(RT, nt - i0);
    __syncthreads();  // Xt staged / previous Wd consumed
    if (i0 == 0) NFP_STAMP(8);
    if (i0 == RT) NFP_STAMP(12);
    // ---- Wd for row tiles i0, i0 + 1: zero, then scatter the window slots of each row -----------------------
    for (int i = t; i < nrt * 2 * 32 * wq; i += T) Wd[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (i0 == 0) NFP_STAMP(9);
    for (int i = t; i < nrt * 32 * K2; i += T) {
      const int ri = fdivi(i, 32 * K2), rem = i - ri * 32 * K2, row = fdivi(rem, K2), j = rem - row * K2;
      const int rr = 32 * (i0 + ri) + row;
      if (rr < P) {
        const int py = fdivi(rr, g.W), px = rr - py * g.W;
        const int dy = j / K - R, dx = j % K - R;
        if (py + dy >= 0 && py + dy < g.H && px + dx >= 0 && px + dx < g.W) {
          const int ts = gemm_ts(i0 + ri, band);
          const int tt = rr + dy * g.W + dx - ts;   // 0 <= tt < 16 * KW
          const float w = Wt[rr * K2 + j];
          const uint32_t hi = __float_as_uint(w) & 0xFFFF0000u;
          const uint32_t lo = __float_as_uint(w - __uint_as_float(hi)) >> 16;
          uint16_t* base = (uint16_t*)(Wd + (long long)(ri * 2 * 32 + row) * wq);
          base[tt] = (uint16_t)(hi >> 16);
          base[(long long)32 * wq * 8 + tt] = (uint16_t)lo;   // the lo image: 32 rows of wq pieces (8 bf16 each) further
        }
      }
    }
    __syncthreads();
    if (i0 == 0) NFP_STAMP(10);
    // ---- output tiles (row tile, channel tile), one wavefront each -----------------------------------------------
    for (int ot = wave; ot < nrt * nct; ot += nw) {
      const int ri = fdivi(ot, nct), ct = ot - ri * nct;
      gemm_out_tile<NHWC>(g, Wd + (long long)ri * 2 * 32 * wq, Xt, i0 + ri, ct, ct, band, KW, xq, wq, gxb, cb0, lane, gg);
    }
    if (i0 == 0) NFP_STAMP(11);
  }
  NFP_STAMP(6);
}

// Second form's staging (bwd_gemm_phase3): the image block was REQUESTED as a whole, as in the first form (a chunk's loads
// issued one chunk ahead arrive 2-3 us later — three times the chunk's tiles), and is committed chunk by chunk as the two
// buffers come free: chunk c lives in buffer c & 1, rows = its own cx channels.  Commits the chunks [c_lo, c_hi): this
// thread's held share of them, and — loaded here — the pieces of them that no thread holds.
template <bool NHWC>
__device__ __forceinline__ void gemm_stage3(const KP& g, uint4* Xt0, const uint16_t* xb, int cb0, int ncw, int cx, int t, int T,
                                            GemmX<NHWC>& pre, int c_lo, int c_hi) {
  const int P = g.P, pg = (P + 7) >> 3, xq = gemm_xq(P);
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = T >> 6;
  auto row_of = [&](int ch, int c) { return Xt0 + ((long long)(c & 1) * cx + (ch - c * cx)) * xq; };  // channel ch (of the block) in chunk c
  if constexpr (NHWC) {
    const int co4 = ncw >> 5, nblk = co4 * ((pg + 15) >> 4);
    auto commit = [&](uint4 (&v)[8], int k, int gq, int c) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (8 * gq + u >= P) v[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // channel 8k + j: its 16 bits of every pixel's piece
        uint32_t e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint4 &a = v[2 * u], &b = v[2 * u + 1];
          const uint32_t wa = (j >> 1) == 0 ? a.x : ((j >> 1) == 1 ? a.y : ((j >> 1) == 2 ? a.z : a.w));
          const uint32_t wb = (j >> 1) == 0 ? b.x : ((j >> 1) == 1 ? b.y : ((j >> 1) == 2 ? b.z : b.w));
          e[u] = __builtin_amdgcn_perm(wb, wa, (j & 1) ? 0x07060302u : 0x05040100u);  // {pixel 2u, pixel 2u + 1}
        }
        if (gq < pg) row_of(8 * k + j, c)[gq] = make_uint4(e[0], e[1], e[2], e[3]);
      }
    };
    for (int blk = wave; blk < nblk; blk += nw) {
      const int bk = blk - fdivi(blk, co4) * co4, c = fdivi(32 * bk, cx);   // (wave-uniform: a block's 32 channels lie in one chunk)
      if (c < c_lo || c >= c_hi) continue;
      int k, gq;
      gemm_block_of(blk, co4, lane, k, gq);
      gq = min(gq, pg);  // (group pg: loads clamp to the last pixel, nothing is written)
      if (blk == wave) {
        commit(pre.v, k, gq, c);
      } else {
        uint4 v[8];
        gemm_fetch8(v, g, xb, cb0, k, gq);
        commit(v, k, gq, c);
      }
    }
  } else if ((P & 3) == 0) {
    const int np = ncw * pg;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // the pieces requested during phase A
      const int i = t + u * T, ch = fdivi(i, pg), c = fdivi(ch, cx);
      if (i < np && c >= c_lo && c < c_hi) row_of(ch, c)[i - ch * pg] = pre.v[u];
    }
    for (int i = t + 4 * T; i < np; i += T) {
      const int ch = fdivi(i, pg), c = fdivi(ch, cx);
      if (c >= c_lo && c < c_hi) {
        uint4 v;
        gemm_fetch_row(v, g, xb, cb0, pg, i);
        row_of(ch, c)[i - ch * pg] = v;
      }
    }
  } else {
    const int np = ncw * pg;
    for (int i = t; i < np; i += T) {
      const int ch = fdivi(i, pg), gq = i - ch * pg, c = fdivi(ch, cx);  // pixel group fastest
      if (c < c_lo || c >= c_hi) continue;
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pp = min(8 * gq + u, P - 1);
        const uint16_t e = xb[(long long)(cb0 + ch) * P + pp];
        v[u] = 8 * gq + u < P ? e : 0;
      }
      row_of(ch, c)[gq] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
  }
}

// Round 4, second form of the matrix-core phase B (bwd_fast<…, GEMM = 2>): the densified weights of EVERY row tile are in
// LDS when this starts — phase A wrote them there directly (no window table, no zero / scatter rounds) — and x goes
// through in chunks of g.Tc channel tiles, two buffers: the next chunk's loads fly under the current chunk's tiles.  One
// barrier per chunk.  A chunk has about as many output tiles as the workgroup has wavefronts (launcher).
template <int R, bool NHWC>
__device__ __forceinline__ void bwd_gemm_phase3(const KP& g, const uint4* Wd, uint4* Xt0, GemmX<NHWC>& pre, const uint16_t* xb,
                                                uint16_t* gxb, int cb0, int cb1, int t, int T, const float* gg = nullptr) {
  const int P = g.P, band = R * g.W + R, nt = (P + 31) >> 5, KW = gemm_kw(band);
  const int xq = gemm_xq(P), wq = odd_up(2 * KW + 1), pg = (P + 7) >> 3;
  const int ncw = cb1 - cb0, cx = 32 * g.Tc, nch = (ncw + cx - 1) / cx;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), nw = T >> 6;
  NFP_STAMP_INIT();
  gemm_stage3<NHWC>(g, Xt0, xb, cb0, ncw, cx, t, T, pre, 0, 1);   // the first chunk; the second goes in under its tiles
  for (int i = t; i < (nch > 1 ? 2 : 1) * cx * (xq - pg); i += T) {  // the rest of every row (alignment + padding) reads as zero
    const int c = fdivi(i, xq - pg), k = i - c * (xq - pg);
    Xt0[(long long)c * xq + pg + k] = make_uint4(0, 0, 0, 0);
  }
  NFP_STAMP(7);
  for (int k = 0; k < nch; ++k) {
    const int cbk = cb0 + k * cx, nctk = min(cx, cb1 - cbk) >> 5;
    const uint4* Xt = Xt0 + (long long)(k & 1) * cx * xq;
    __syncthreads();  // chunk k staged (k = 0: and every Wd row complete); chunk k - 1's tiles are done: its buffer is free
    if (k == 0) NFP_STAMP(8);
    if (k + 1 < nch) gemm_stage3<NHWC>(g, Xt0, xb, cb0, ncw, cx, t, T, pre, k + 1, k + 2);
    if (k == 1) NFP_STAMP(10);
    for (int ot = wave; ot < nt * nctk; ot += nw) {
      const int it = fdivi(ot, nctk), ctx = ot - it * nctk;
      gemm_out_tile<NHWC>(g, Wd + (long long)it * 2 * 32 * wq, Xt, it, ctx, k * g.Tc + ctx, band, KW, xq, wq, gxb, cb0, lane, gg);
    }
    if (k == 0) NFP_STAMP(9);
    if (k == 1) NFP_STAMP(11);
  }
  NFP_STAMP(6);
}

template <int R>
struct L_BRQ {
  static constexpr int v = ((Win<R>::K2 + 7) & ~7) / 8;  // 16-byte pieces of a pixel's boff row (ws_layout: BR)
};

// ---- backward -------------------------------------------------------------------------------
// POOL: grad_out is not a map but the gradients of the two pooled outputs: go[b,n,p] = gnfpm[b,n]/P
// for every p, and every grad_x[b,c,p] also receives ggap[b,c]/P (adjoint of the two means).
template <int R, int M, bool BF, bool NHWC, bool POOL = false, int GEMM = 0>   // GEMM: 0 vector phase B, 1 / 2 matrix cores
__global__ void __launch_bounds__(GEMM ? 1024 : kBwdThreads) bwd_fast(const KP g, const void* __restrict__ x,
                                                        const void* __restrict__ go, const void* __restrict__ out,
                                                        const float* __restrict__ saved, void* __restrict__ gx,
                                                        const float* __restrict__ ggap,
                                                        const float* __restrict__ gnfpm,
                                                        const unsigned char* __restrict__ ws) {
  static_assert(!GEMM || BF, "matrix-core phase B: bf16 storage only");
  constexpr int K2 = Win<R>::K2, N = Win<R>::N;
  // the diagonal is folded by a phase of its own when the weights are consumed as a table (matrix cores) or the
  // window is large; for k = 3 every compute thread folds its own pixel's nine terms while it loads its weights
  constexpr bool FOLD_PHASE = GEMM || Win<R>::RAD != 1;  // (R: radius spec of nfp_tables.h::Win)
  constexpr bool G3 = GEMM == 2;  // every row tile's densified weights written by phase A itself: bwd_gemm_phase3
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int P = g.P;
  // LDS: Wt | Dt | ipn | dfn (live to the end) | pair values | x slab.  The slab lies OVER the pair values (dead
  // once Wt is built) when both do not fit side by side (g.early == 0).
  // Matrix-core variant (round 4): Dt is dead once the diagonal is folded (A3), so it sits BEHIND the tables that live to
  // the end and the GEMM's operand images are laid over it as well: Wt | ipn | dfn | Dt | pair values.  19.6 KB more for
  // Wd at config 5's shape: three row tiles per round instead of two, three rounds of zero / scatter / multiply instead
  // of four (2.2 us each: profiles/r02_j_matrix_core_backward_ab.txt).
  // GEMM = 2: ipn | dfn | Wc (centre weights) | grad(GAP) of the block (POOL) | Wd of every row tile | pair values | Dt, the
  // x chunks (two buffers) over the last two once the diagonal is folded.
  float* Wt = (float*)lds4;        // [P][K2] gathered weights
  float* ipn = G3 ? Wt : (GEMM ? Wt + P * K2 : Wt + 2 * P * K2);   // [P] 1 / max(|x_p|, eps)
  float* dfn = ipn + P;            // [P] -1 / (|x_p| max(|x_p|, eps)), 0 where |x_p| = 0
  float* Wc = dfn + P;             // (GEMM = 2) [P] centre weights before the fold
  const int g3_band = Win<R>::RAD * g.W + Win<R>::RAD, g3_wq = odd_up(2 * gemm_kw(g3_band) + 1);
  const int g3_fixed4 = (3 * P + (POOL ? (int)(g.Cwg) : 0) + 3) >> 2;   // float4 slots in front of Wd
  uint4* g3_Wd = (uint4*)(lds4 + g3_fixed4);
  float4* g3_pv4 = (float4*)(g3_Wd + (long long)((P + 31) >> 5) * 64 * g3_wq);
  float4* pv4 = G3 ? g3_pv4 : (GEMM ? lds4 + ((P * K2 + 2 * P + 3) >> 2) + ((P * K2 + 3) >> 2) : lds4 + ((2 * P * K2 + 2 * P + 3) >> 2));
  float* Dt = G3 ? (float*)(g3_pv4 + (((M == NFP_COSINE ? 2 : 1) * (N * P + 1) + 3) >> 2))
                 : (GEMM ? (float*)(lds4 + ((P * K2 + 2 * P + 3) >> 2)) : Wt + P * K2);   // [P][K2] diagonal terms collected per slot
  float2* AD = (float2*)pv4;       // cosine: [N*P] {sg, sg*s} of pair o = n*P + p
  float* CC = (float*)pv4;         // L2:     [N*P] c = -+g/d
  float4* slab = g.early ? pv4 + (((M == NFP_COSINE ? 2 : 1) * (N * P + 1) + 3) >> 2) : pv4;  // [Cc/4][P]
  const int b = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const int cb0 = blockIdx.y * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  // matrix-core variant: Xt over Dt and the pair values (dead once Wt is built and folded), Wd behind it
  uint4* gemm_Xt = G3 ? (uint4*)g3_pv4 : (uint4*)Dt;
  uint4* gemm_Wd = G3 ? g3_Wd : gemm_Xt + (long long)(cb1 - cb0) * gemm_xq(P);
  const uint16_t* x16 = (const uint16_t*)x + (long long)b * g.sB;
  // fused pooling tail, matrix-core variant: grad(GAP(x)) / P of this workgroup's channels goes to LDS behind Wd (read
  // from global memory in the tile loop it cost 4 of 22 us at config 5).  Two values per thread are requested here.
  // (The vector kernel keeps its 16-byte global loads in the channel loop: staged the same way it measured 7.3 vs 6.7 us.)
  float* gg_s = nullptr;
  float ggv0 = 0.f, ggv1 = 0.f;
  if constexpr (POOL && GEMM) {
    const int ncw = cb1 - cb0;
    const int band = Win<R>::RAD * g.W + Win<R>::RAD, KW = gemm_kw(band);
    gg_s = G3 ? Wc + P : (float*)(gemm_Wd + (long long)g.Tc * 2 * 32 * odd_up(2 * KW + 1));
    if (g.pool_gap && t < ncw) ggv0 = ggap[(long long)b * g.C + cb0 + t];
    if (g.pool_gap && t + T < ncw) ggv1 = ggap[(long long)b * g.C + cb0 + t + T];
  }
  // Thread map.  NCHW: t = gl * P + p — lanes along the pixels of a channel row (coalesced 4-byte stores).  Channels-last
  // (round 4): t = p * G + gl — lanes along the channel groups of a pixel, so that G adjacent lanes read and write 16 G
  // contiguous bytes; with lanes along the pixels every load / store instruction touched 64 cache lines for 1 KB, and the
  // eight quads of a 128-byte line were written by eight different wavefronts (the backward ran 20-40 % behind NCHW on the
  // same bytes: [4096,512,7,7] 231 vs 163 us, profiles/r04_a_…).
  int gl, p;
  bool active;
  if constexpr (NHWC && !GEMM && NFP_BWD_NHWC_LANES) {
    const int pp = fdivi(t, g.G);
    gl = t - pp * g.G;
    active = pp < P;
    p = min(pp, P - 1);
  } else {
    gl = fast_div(t, g.invP);
    p = t - gl * P;
    active = gl < g.G;
  }
  constexpr int ES = BF ? 2 : 4;
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);  // wave-uniform
  const Rsrc gxb = make_rsrc((char*)gx + (long long)b * g.gB * ES, (long long)g.C * P * ES);
  const void* gob = (const char*)go + (long long)b * N * P * ES;
  const void* outb = (const char*)out + (long long)b * N * P * ES;
  const WsLayout L = ws_layout(P, R, g.mode);
  const int LQ = L.LW >> 3;  // 16-byte pieces per link row
  const uint4* lnk = (const uint4*)(ws + L.lnk);
  const uint16_t* tqt = (const uint16_t*)(ws + L.tq);

  NFP_STAMP_INIT();
  NFP_STAMP(0);
  // Phase A, table driven (nfp_tables.h).  Loads are issued in the order their data is needed: the geometry tables
  // and this image's grad_out / out / norms first (small, L2-resident), then the x chunk, so that the pair
  // arithmetic runs while x streams in.  One table entry / pair per thread and round, the next round's loads in
  // flight during the current round's arithmetic.
  // k = 5 (25 slots per pixel, about five entries per thread): only the centre slot and the NF slots AFTER it are
  // gathered — the pairs that link r with the pixel t under slot j are the pairs that link t with r under its slot
  // K2-1-j, so each sum is stored twice.  k = 3 has fewer entries than threads: every slot is gathered by its own
  // thread.
  constexpr bool SYM = Win<R>::RAD >= 2;
  constexpr int NJ = SYM ? Win<R>::NF + 1 : K2;
  const int NE = P * NJ, NO = N * P;
  auto entry_of = [&](int e2, int& r, int& j) {  // gathered entry e2 -> (pixel, slot); returns its table row
    r = fdivi(e2, NJ);
    j = (SYM ? K2 / 2 : 0) + e2 - r * NJ;
    return r * K2 + j;
  };
  uint4 rw0, rw1;
  uint32_t tqv;
  const uint4 kNoLinks = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  auto rows_load_to = [&](uint4& d0, uint4& d1, uint32_t& dq, int e) {
    d0 = lnk[(long long)e * LQ];
    if (LQ > 1)  // (wave-uniform; a one-piece row read twice is 16 of the 34 bytes per entry the texture path moves)
      d1 = lnk[(long long)e * LQ + LQ - 1];
    else
      d1 = kNoLinks;
    dq = tqt[e];
  };
  auto rows_load = [&](int e) { rows_load_to(rw0, rw1, tqv, e); };
  // grad_out / out of this image, 16 bytes per thread and round: VP consecutive pairs (N is a multiple of 8, so
  // N*P values are whole 16-byte pieces and every image's maps start on one)
  constexpr int VP = BF ? 8 : 4;
  uint4 gq, oq, gq0, oq0;  // (the first round's piece in registers of its own: a copy made before the x block is
                           // requested would wait for the data there)
  auto pair_load_to = [&](uint4& gd, uint4& od, int o) {  // o: first pair of the piece
    if constexpr (!POOL) gd = *(const uint4*)((const char*)gob + (long long)o * ES);
    od = *(const uint4*)((const char*)outb + (long long)o * ES);
  };
  auto pair_load = [&](int o) { pair_load_to(gq, oq, o); };
  auto pair_value = [&](const uint4& q, int k) {  // element k of a 16-byte piece
    const uint32_t wd = ((const uint32_t*)&q)[BF ? k >> 1 : k];
    return BF ? __uint_as_float(k & 1 ? wd & 0xFFFF0000u : wd << 16) : __uint_as_float(wd);
  };
  // The matrix-core variant asks for its x block at entry and loads retire in order: every table row of the first
  // PRE gather rounds is requested before it (PRE rounds cover config 5's 2548 entries on 1024 threads; the launcher
  // picks the thread count so that they cover the table: kGemmPre).
  constexpr int PRE = GEMM ? kGemmPre : 1;
  uint4 prw0[PRE], prw1[PRE];
  uint32_t ptq[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    int r_, j_;
    rows_load_to(prw0[k], prw1[k], ptq[k], entry_of(min(t + k * T, NE - 1), r_, j_));
  }
  pair_load_to(gq0, oq0, min(t * VP, NO - VP));
  // (DotProduct has no saved norms: the load reads the output map instead — in bounds, unused — rather than sit under a branch)
  // (DotProduct has no saved norms: the load reads the output map instead — in bounds, rather than sit under a branch —
  // and its bits are masked to 0: read as floats, a bf16 map holds NaN / Inf patterns, and NaN * 0 would reach ipn)
  const float nrm_raw = (M == NFP_COSINE) ? (g.unit ? (const float*)out : saved)[(long long)b * P + min(t, P - 1)] : 0.f;
  const float nrm = __int_as_float(__float_as_int(nrm_raw) & (g.unit ? 0 : -1));
  uint4 bo[L_BRQ<R>::v];  // this pixel's window offsets (phase B)
  if constexpr (!GEMM) {
    const uint4* bot = (const uint4*)(ws + L.boff) + (long long)p * L_BRQ<R>::v;
#pragma unroll
    for (int u = 0; u < L_BRQ<R>::v; ++u) bo[u] = bot[u];
  }
  __builtin_amdgcn_sched_barrier(0);  // keep these (small, needed first) loads ahead of the x chunk
  typename std::conditional<NHWC, Staged<true>, StagedOvl>::type st;
  auto x_issue = [&](int c0, int ncq) {
    if constexpr (NHWC)
      stage_load<BF>(st, xb, g, c0, ncq, p, gl, active);
    else
      stage_load_ovl<BF>(st, xb, g, c0, ncq, t, T);
  };
  auto x_commit = [&](int ncq) {
    if constexpr (NHWC)
      stage_store(st, slab, g, ncq, p, gl, active);
    else
      stage_store_ovl(st, slab, g, ncq, t, T);
  };
  GemmX<NHWC> gxr;
#ifndef NFP_VEC_X_LATE
#define NFP_VEC_X_LATE 1   // 1 = k = 5 vector kernels request their first x chunk after the pair values (their gather
                           // rounds hide it: [64,512,7,7] k = 5 10.97 -> 10.68 us, [256,192,14,14] 42.0 -> 41.3; k = 3
                           // measures the same either way and keeps the request at entry), 0 = none, 2 = all do
#endif
  constexpr bool X_LATE = !GEMM && (NFP_VEC_X_LATE == 2 || (NFP_VEC_X_LATE == 1 && Win<R>::RAD >= 2));
  if constexpr (!GEMM && !X_LATE) x_issue(cb0, min(g.Cc, cb1 - cb0) >> 2);
  // nothing that consumes a loaded value may be scheduled above this line (hipcc otherwise hoists consumers into
  // the load sequence and stalls the remaining loads behind a vmcnt wait)
  __builtin_amdgcn_sched_barrier(0);
  NFP_STAMP(1);
  if constexpr (G3) {   // Wd: whatever phase A does not write reads as zero; Dt: slots outside the image — under the first loads' latency
    const int nwd = ((P + 31) >> 5) * 64 * g3_wq;
    for (int i = t; i < nwd; i += T) g3_Wd[i] = make_uint4(0, 0, 0, 0);
    if constexpr (Win<R>::RAD >= 2)
      for (int i = t; i < P * K2; i += T) Dt[i] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
  }
  // A1: per-pair values, in the memory order of grad_out / out
  // (first round outside the loop: hipcc's wait-count pass is exact in straight-line code only, and a wait that
  // also covers the x block just requested would put its whole latency in front of phase A)
  auto pair_round = [&](int o, const uint4& gc4, const uint4& oc4) {
    // fused pooling tail: grad_out[n][p] = grad(GAP(NFP))[n] / P for every p; a thread's VP consecutive pairs lie in at
    // most two maps when P >= VP: two loads instead of one per pair
    int n0 = 0;
    float gn0 = 0.f, gn1 = 0.f;
    if constexpr (POOL) {
      n0 = fdivi(o, P);
      gn0 = gnfpm[(long long)b * N + n0] * g.invP;
      gn1 = gnfpm[(long long)b * N + min(n0 + 1, N - 1)] * g.invP;
    }
#pragma unroll
    for (int k = 0; k < VP; ++k) {
      const float oc = pair_value(oc4, k);
      const float gc = !POOL ? pair_value(gc4, k)
                       : (P >= VP ? ((o + k) >= (n0 + 1) * P ? gn1 : gn0) : gnfpm[(long long)b * N + fdivi(o + k, P)] * g.invP);
      if (M == NFP_COSINE) {
        const float s = g.osa * (oc - g.osb);   // out = osa * s + osb, osa = +-1
        const float sg = g.osa * gc;
        AD[o + k] = make_float2(sg, sg * s);
      } else {
        if constexpr (M == kSymTerm) {   // (nfp_measures.h: the measure's own coefficient — Hellinger's needs the map value)
          float c = 0.f;
          sym_switch(g.measure, [&](auto mm) { c = decltype(mm)::coef(gc, oc, 0.f, 0.f, 0.f, 0.f, g).k0; });
          CC[o + k] = c;
        } else {
          CC[o + k] = M == kNormP1 ? g.osa * gc : dist_coef(g, gc, oc);   // (p = 1: d out / d (a - b)[c] = +-g sign(a - b)[c])
        }
      }
    }
  };
  // The matrix-core variant has NO loop that loads from memory between here and the commit of its x block (the
  // launcher guarantees one pair round and at most PRE gather rounds): after such a loop the wait-count pass no
  // longer knows how many loads are in flight and waits for all of them, the x block included.
  if constexpr (!GEMM) {
    if ((t + T) * VP < NO) pair_load((t + T) * VP);
  }
  if (t * VP < NO) pair_round(t * VP, gq0, oq0);
  if constexpr (!GEMM) {
    for (int o = (t + T) * VP; o < NO; o += T * VP) {
      const uint4 gc4 = gq, oc4 = oq;
      if (o + T * VP < NO) pair_load(o + T * VP);
      pair_round(o, gc4, oc4);
    }
  }
  if (t == 0) {   // the zero behind the pair values (an empty list entry reads it: gather)
    if (M == NFP_COSINE) AD[NO] = make_float2(0.f, 0.f);
    else CC[NO] = 0.f;
  }
  if (M == NFP_COSINE && t < P) {
    const float ip = unit_or(g, __builtin_amdgcn_rcpf(fmaxf(nrm, g.eps)));   // (DotProduct: no norm factors, no diagonal)
    ipn[t] = fmaf(nrm, g.gf, ip * g.ngf);                                     // (GFC: the norm itself — nfp_common.h::cross_f)
    dfn[t] = nrm > 0.f ? -fmaf(1.f, g.gf, g.nuf * ip * g.ngf) * __builtin_amdgcn_rcpf(nrm) : 0.f;
  }
  if constexpr (G3) {   // (Wd and Dt were zeroed while the first loads were in flight)
  } else if constexpr (SYM) {  // slots before the centre whose pixel lies outside the image keep this 0
    if constexpr (GEMM) {   // (Wt and Dt apart: see the layout above)
      for (int i = t; i < P * K2; i += T) {
        Wt[i] = 0.f;
        Dt[i] = 0.f;
      }
    } else {                // (Wt and Dt are adjacent)
      for (int i = t; i < (2 * P * K2) >> 2; i += T) ((float4*)Wt)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < ((2 * P * K2) & 3)) Wt[((2 * P * K2) & ~3) + t] = 0.f;
    }
  }
  if constexpr (POOL && GEMM) {
    const int ncw = cb1 - cb0;
    if (t < ncw) gg_s[t] = ggv0 * g.invP;
    if (t + T < ncw) gg_s[t + T] = ggv1 * g.invP;
    for (int i = t + 2 * T; i < ncw; i += T) gg_s[i] = g.pool_gap ? ggap[(long long)b * g.C + cb0 + i] * g.invP : 0.f;  // (more than 2T channels)
  }
  __syncthreads();
  NFP_STAMP(2);
  if constexpr (X_LATE) {
    x_issue(cb0, min(g.Cc, cb1 - cb0) >> 2);
    __builtin_amdgcn_sched_barrier(0);
  }

  // (GEMM = 2) weight of row pixel r for the pixel under its slot j, straight into the operand image: row r of its row
  // tile, column = that pixel - the tile's first summed pixel; hi / lo halves as two bf16 images (bwd_gemm_phase's notes)
  auto wd_put_at = [&](int r, int tpx, float w) {
    const int it = r >> 5;
    uint16_t* base = (uint16_t*)(g3_Wd + (long long)(it * 64 + (r & 31)) * g3_wq);
    const uint32_t hi = __float_as_uint(w) & 0xFFFF0000u;
    const uint32_t lo = __float_as_uint(w - __uint_as_float(hi)) >> 16;
    const int tt = tpx - gemm_ts(it, g3_band);
    base[tt] = (uint16_t)(hi >> 16);
    base[(long long)32 * g3_wq * 8 + tt] = (uint16_t)lo;
  };
  // A2: window entry (r, j) = the sum of the pairs that link r with the pixel under slot j, listed by the table
  auto gather = [&](int e2, const uint4& r0, const uint4& r1, uint32_t tqc) {
    int r, j;
    const int e = entry_of(e2, r, j);
    float S = 0.f, Dj = 0.f, Dm = 0.f, wv;  // Dj: diagonal term for r, Dm: for the pixel t on the other end
    if (j != K2 / 2) {
      // One list entry.  The lists are packed (live entries first), an empty entry is 0xFFFF: its 15 index bits, clamped to
      // NO, read the ZERO the pair values end with — no predicate on the sums.  (Round 4: at config 5 the gathers were
      // VALU-bound — ~230 instructions per entry, 2.5 entries per thread, four wavefronts per SIMD — not LDS- or latency-
      // bound.)  FLAGS: the 'Norm' quirk — only the pair's NEIGHBOUR is pulled (bit 15: r is the neighbour of this pair);
      // with the difference weights every live entry counts on both diagonals (they equal S), and p = 1 has no diagonal at
      // all: the pull on x_r sits inside sign(x_r - x_t).
      auto takes = [&](auto flags_c) {
        constexpr bool FLAGS = decltype(flags_c)::value;
        auto take = [&](uint32_t ent) {
          const int o = min((int)(ent & 0x7FFFu), NO);
          if (M == NFP_COSINE) {
            const float2 v = AD[o];
            S += v.x;
            Dj += v.y;
          } else {
            const float cv = CC[o];
            S += cv;
            if constexpr (FLAGS) {
              Dj += (ent & 0x8000u) ? cv : 0.f;   // (an empty entry has the bit set and adds the zero)
              Dm += (ent & 0x8000u) ? 0.f : cv;
            }
          }
        };
        auto live = [&](uint32_t w2) { return __ballot((w2 & 0xFFFFu) != 0xFFFFu) != 0; };  // wave-uniform
        // k = 3 (one entry per thread): the first piece in one batch of LDS reads — a branch per pair of entries
        // costs an LDS round trip each; k = 5 (ten entries per thread, most of them two links long): skip the rest
        take(r0.x & 0xFFFFu);
        take(r0.x >> 16);
        if (Win<R>::RAD == 1 || live(r0.y)) {
          take(r0.y & 0xFFFFu);
          take(r0.y >> 16);
          take(r0.z & 0xFFFFu);
          take(r0.z >> 16);
          take(r0.w & 0xFFFFu);
          take(r0.w >> 16);
        }
        if (LQ > 1 && live(r1.x)) {
          take(r1.x & 0xFFFFu);
          take(r1.x >> 16);
          take(r1.y & 0xFFFFu);
          take(r1.y >> 16);
          take(r1.z & 0xFFFFu);
          take(r1.z >> 16);
          take(r1.w & 0xFFFFu);
          take(r1.w >> 16);
        }
      };
      if (M == NFP_COSINE || M == kSymTerm || g.diff) {   // (uniform)
        takes(std::false_type{});
        if (M != NFP_COSINE) Dj = Dm = ((M == kNormP1 || M == kSymTerm) ? 0.f : S);
      } else {
        takes(std::true_type{});
      }
      const int tt = tqc == 0xFFFFu ? r : (int)tqc;
      // (kSymTerm: both pairs of r with t pull on x_r through the same d term / d a (x_r, x_t): their coefficients add)
      wv = M == NFP_COSINE ? cross_f(g, ipn[r], ipn[tt]) * S : (M == kSymTerm ? S : (g.diff ? (M == kNormP1 ? S : -S) : 0.f));
      if (SYM && tqc != 0xFFFFu) {  // the same pairs, seen from t
        const int em = tt * K2 + (K2 - 1 - j);
        if constexpr (G3) wd_put_at(tt, r, wv); else Wt[em] = wv;   // (t's slot K2 - 1 - j is r: inside the image)
        Dt[em] = M == NFP_COSINE ? Dj * diag_f(g, ipn[tt], ipn[r]) : Dm;
      }
      if (M == NFP_COSINE) Dj *= diag_f(g, ipn[r], ipn[tt]);
    } else {
      // centre slot: its row carries the pixel's two tap masks instead of links (zero-padded taps, self pairs)
      uint32_t zm = r0.x, sm = r0.y;
      if (M == NFP_COSINE) {
        while (sm) {  // (p, n) with q == p: both ends of the pair are r
          const int n = __builtin_ctz(sm);
          sm &= sm - 1;
          const float2 v = AD[n * P + r];
          S += 2.f * v.x;
          Dj += 2.f * v.y;
        }
        wv = cross_f(g, ipn[r], ipn[r]) * S;
        Dj *= diag_f(g, ipn[r], ipn[r]);
      } else {
        uint32_t m = (g.diff || M == kSymTerm) ? zm : sm;  // diff: |x_p - 0| still pulls on x_p; quirk: |x_q| with q == p
        while (m) {
          const int n = __builtin_ctz(m);
          m &= m - 1;
          Dj += CC[n * P + r];
        }
        wv = 0.f;
        // diff, a pair whose both ends are r (replicate padding; reflect on tiny maps): distance 0.  L2's coefficient is
        // 0 there and nothing changes; RMSE's is +-inf (nfp.py:172-179 through torch's sqrt: no subgradient at 0), and
        // the reference's NaN on that pixel has to come out: +2c on the diagonal, -2c across = inf - inf
        if constexpr (M == kSymTerm) {
          // a pixel paired with its own padded copy: d term / d a (a, a) = 0 under every one of these terms, so nothing is
          // added — but Hellinger's coefficient there is 1 / distance = inf, and the reference's inf * 0 = NaN has to come
          // out on that pixel: the coefficients multiply an explicit zero
          float cs = 0.f;
          while (sm) {
            const int n = __builtin_ctz(sm);
            sm &= sm - 1;
            cs += CC[n * P + r];
          }
          Dj = fmaf(cs, 0.f, Dj);
        } else if (g.diff && M != kNormP1) {   // (p = 1: sign(0) = 0, nothing to add)
          while (sm) {
            const int n = __builtin_ctz(sm);
            sm &= sm - 1;
            const float c = CC[n * P + r];
            Dj += 2.f * c;
            wv -= 2.f * c;
          }
        }
      }
    }
    if constexpr (G3) {   // (tq = the pixel under the slot, 0xFFFF outside the image: nothing is summed there)
      if (j == K2 / 2) Wc[r] = wv;
      else if (tqc != 0xFFFFu) wd_put_at(r, (int)tqc, wv);
    } else {
      Wt[e] = wv;
    }
    Dt[e] = Dj;
  };
  if constexpr (GEMM) {
    // The x block is requested HERE, a part in front of each gather round: it arrives 2-3 us later whatever the
    // cache level, and the gathers need no memory (their table rows were requested at entry).  Requested at entry
    // instead, it delays the pair values: every wavefront's grad_out / out pieces then queue behind the other
    // wavefronts' x loads (measured 21.5 vs 20.8 us).
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      if (k < 3) {
        if constexpr (G3) {
          // (requested at kernel entry instead — the first chunk, or all of it through branch-free buffer loads — it delays
          // the pair values: 16.7-17.3 vs 16.2 us at config 5, profiles/r04_zb_…)
          // all of it in front of the first gather round, or behind the pair values: 15.6 / 15.8 vs 15.1 us
          gemm_x_issue3<NHWC>(gxr, g, x16, cb0, cb1 - cb0, 32 * g.Tc, t, T, k);
        } else {
          gemm_x_issue<NHWC>(gxr, g, x16, cb0, cb1 - cb0, t, T, k);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t + k * T < NE) gather(t + k * T, prw0[k], prw1[k], ptq[k]);
    }
  } else {
    if (t < NE) {
      if (t + T < NE) {  // the next round's row flies during this round's gathers
        int r_, j_;
        rows_load(entry_of(t + T, r_, j_));
      }
      gather(t, prw0[0], prw1[0], ptq[0]);
    }
    for (int e2 = t + T; e2 < NE; e2 += T) {
      const uint4 r0 = rw0, r1 = rw1;
      const uint32_t tqc = tqv;
      if (e2 + T < NE) {
        int r_, j_;
        rows_load(entry_of(e2 + T, r_, j_));
      }
      gather(e2, r0, r1, tqc);
    }
  }
  // (the pair values stay readable until the barrier: with g.early the slab does not overlap them, so the x chunk
  // is committed here, while slower wavefronts still gather)
  if constexpr (!GEMM) {
    if (g.early) x_commit(min(g.Cc, cb1 - cb0) >> 2);
  }
  __syncthreads();
  NFP_STAMP(3);
  if constexpr (FOLD_PHASE) {
    // A3: the diagonal — every pair that contains r pulls on x_r itself — folded in a fixed order
    for (int r = t; r < P; r += T) {
      float s = G3 ? Wc[r] : Wt[r * K2 + K2 / 2];
      const float fac = M == NFP_COSINE ? dfn[r] : 1.f;
#pragma unroll
      for (int j = 0; j < K2; ++j) s = fmaf(fac, Dt[r * K2 + j], s);
      if constexpr (G3) wd_put_at(r, r, s); else Wt[r * K2 + K2 / 2] = s;
    }
    __syncthreads();
  }
  NFP_STAMP(4);
  if constexpr (GEMM) {
    // (the pair values behind the tables are dead: their LDS becomes the GEMM's operand images)
    if constexpr (G3)
      bwd_gemm_phase3<Win<R>::RAD, NHWC>(g, gemm_Wd, gemm_Xt, gxr, x16, (uint16_t*)gx + (long long)b * g.gB, cb0, cb1, t, T,
                                         POOL ? gg_s : nullptr);
    else
      bwd_gemm_phase<R, NHWC>(g, Wt, gemm_Xt, gemm_Wd, gxr, x16, (uint16_t*)gx + (long long)b * g.gB, cb0, cb1, t, T,
                              POOL ? gg_s : nullptr);
    return;
  }
  float w[K2];
  int off[K2];
  {
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < K2; ++j) {
      const uint32_t wd = j & 1 ? ((const uint32_t*)bo)[j >> 1] >> 16 : ((const uint32_t*)bo)[j >> 1] & 0xFFFFu;
      off[j] = (int)(int16_t)wd;              // 0 for a slot outside the image ...
      w[j] = active ? Wt[p * K2 + j] : 0.f;   // ... whose weight phase A left at 0
      if constexpr (!FOLD_PHASE) dsum += active ? Dt[p * K2 + j] : 0.f;
    }
    if constexpr (!FOLD_PHASE) w[K2 / 2] = fmaf(M == NFP_COSINE ? dfn[active ? p : 0] : 1.f, dsum, w[K2 / 2]);
  }
  // B: one pass over the channel block; results leave straight from registers — 4 dwords per slot for NCHW, one
  // 16-byte store for channels-last (store policy: NFP_BWD_STORE_AUX above; an LDS-transposed 16-byte NCHW epilogue
  // was 1.6 us slower).
  const int Pp = bwd_row_slots(P, NHWC), sp = swz(p);
  for (int c0 = cb0; c0 < cb1; c0 += g.Cc) {
    const int ncq = min(g.Cc, cb1 - c0) >> 2;
    if (c0 > cb0) __syncthreads();  // previous chunk fully consumed
    if (c0 > cb0 || !g.early) {
      x_commit(ncq);
      __syncthreads();
    }
    // the next chunk's loads fly while this one is processed (the staging registers are free once committed)
    if (c0 + g.Cc < cb1) x_issue(c0 + g.Cc, min(g.Cc, cb1 - c0 - g.Cc) >> 2);
    NFP_STAMP(5);
    if (active) {
#pragma unroll NFP_UNROLL_B
      for (int cq = gl; cq < ncq; cq += g.G) {
        const float4* row = slab + cq * Pp + sp;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (POOL && g.pool_gap) {
          const float4 gg = *(const float4*)(ggap + (long long)b * g.C + c0 + 4 * cq);
          r4 = make_float4(gg.x * g.invP, gg.y * g.invP, gg.z * g.invP, gg.w * g.invP);
        }
        if constexpr (M == kSymTerm) {
          // grad_x[c][r] = sum_t W[r][t] d term / d a (x_r, x_t)[c]  (+ the centre weight times d term / d a (x_r, 0): zero-padded taps)
          const float4 a = row[0];
          sym_switch(g.measure, [&](auto mm) {
            using MM = decltype(mm);
            const Coef one = {1.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < K2; ++j) {
              const float4 q = j == K2 / 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : row[off[j]];
              float da, db;
              MM::grad(a.x, q.x, one, g, da, db);
              r4.x = fmaf(w[j], da, r4.x);
              MM::grad(a.y, q.y, one, g, da, db);
              r4.y = fmaf(w[j], da, r4.y);
              MM::grad(a.z, q.z, one, g, da, db);
              r4.z = fmaf(w[j], da, r4.z);
              MM::grad(a.w, q.w, one, g, da, db);
              r4.w = fmaf(w[j], da, r4.w);
            }
          });
        } else if constexpr (M == kNormP1) {
          // Norm p = 1: grad_x[c][r] = sum_t W[r][t] sign(x_r - x_t)[c]  (+ the centre weight times sign(x_r): zero-padded
          // taps with the difference weights, neighbour roles with the 'Norm' quirk)
          const float4 a = row[0];
#pragma unroll
          for (int j = 0; j < K2; ++j) {
            const float4 q = row[off[j]];
            const bool c = j == K2 / 2;
            r4.x = fmaf(w[j], sgn3(c ? a.x : a.x - q.x), r4.x);
            r4.y = fmaf(w[j], sgn3(c ? a.y : a.y - q.y), r4.y);
            r4.z = fmaf(w[j], sgn3(c ? a.z : a.z - q.z), r4.z);
            r4.w = fmaf(w[j], sgn3(c ? a.w : a.w - q.w), r4.w);
          }
        } else {
#pragma unroll
          for (int j = 0; j < K2; ++j) {
            const float4 q = row[off[j]];
            r4.x = fmaf(w[j], q.x, r4.x);
            r4.y = fmaf(w[j], q.y, r4.y);
            r4.z = fmaf(w[j], q.z, r4.z);
            r4.w = fmaf(w[j], q.w, r4.w);
          }
        }
        if constexpr (NHWC) {
          store_px4<BF>(gxb, p * g.C + c0 + 4 * cq, 0, r4);
        } else {
          const int e = (c0 + 4 * cq) * P + p;  // one address VGPR, channel rows through the SGPR offset
          store_1<BF>(gxb, e, 0, r4.x);
          store_1<BF>(gxb, e, P, r4.y);
          store_1<BF>(gxb, e, 2 * P, r4.z);
          store_1<BF>(gxb, e, 3 * P, r4.w);
        }
      }
    }
  }
  NFP_STAMP(6);
}

}  // namespace nfp
