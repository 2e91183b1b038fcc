// nfp_generic.h — the any-geometry kernels (every stride / dilation / padding /
// padding_mode / radius / layout the reference's nn.Conv2d accepts).
//
// Forward  (replaces nfp.py:152-156 / 143-145 and their ATen graph):
//   grid (B, output slices, neighbour groups of 8).  A workgroup stages a chunk of
//   channels of ONE image into LDS ([Cc][H*W], f32), thread (o, lane) walks its
//   channels and keeps term/stat sums for the centre and <= 8 neighbours of output
//   o; lanes of one output are adjacent, so the channel reduction is a wavefront
//   shuffle; lane 0 finalises and stores out[b, n, o].
// Backward (replaces autograd through cosine_similarity / linalg.norm, conv2d
//   backward and reflection_pad backward): the gather-form kernel of nfp_gather.h
//   serves every map whose index tables fit in LDS.  bwd_generic below is the
//   fallback for larger maps: grid (B, channel blocks), grad_x of the chunk is
//   accumulated in LDS with ds_add_f32 (the pad/stride adjoint is just "add at the
//   mapped index"), then written once, coalesced.  Summation order inside LDS is
//   not fixed, so this fallback is reproducible to rounding only; nfp_gather.h and
//   the fast path (nfp_fast.h) are bitwise deterministic.
#pragma once
#include "nfp_measures.h"

namespace nfp {

constexpr int kGroup = 8;  // neighbours handled per pass

// ---- LDS staging of x[b, c0:c0+cc, :, :] as f32 [cc][P] -----------------------------------
__device__ __forceinline__ void stage_chunk(float* xs, const void* x, const KP& g, int b, int c0, int cc) {
  const int t = threadIdx.x, T = blockDim.x, n = cc * g.P;
  if (g.contig) {
    const long long base = (long long)b * g.sB + (long long)c0 * g.P;
    if (g.dtype == NFP_F32) {
      const float* src = (const float*)x + base;
      if ((((uintptr_t)src) & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = t; i < n4; i += T) ((float4*)xs)[i] = ((const float4*)src)[i];
        for (int i = (n4 << 2) + t; i < n; i += T) xs[i] = src[i];
      } else {
        for (int i = t; i < n; i += T) xs[i] = src[i];
      }
    } else {
      const uint16_t* src = (const uint16_t*)x + base;
      if ((((uintptr_t)src) & 15) == 0) {
        const int n8 = n >> 3;
        for (int i = t; i < n8; i += T) {
          uint4 v = ((const uint4*)src)[i];
          float4 lo, hi;
          lo.x = __uint_as_float(v.x << 16);
          lo.y = __uint_as_float(v.x & 0xffff0000u);
          lo.z = __uint_as_float(v.y << 16);
          lo.w = __uint_as_float(v.y & 0xffff0000u);
          hi.x = __uint_as_float(v.z << 16);
          hi.y = __uint_as_float(v.z & 0xffff0000u);
          hi.z = __uint_as_float(v.w << 16);
          hi.w = __uint_as_float(v.w & 0xffff0000u);
          ((float4*)xs)[2 * i] = lo;
          ((float4*)xs)[2 * i + 1] = hi;
        }
        for (int i = (n8 << 3) + t; i < n; i += T) xs[i] = bf16_to_f32(src[i]);
      } else {
        for (int i = t; i < n; i += T) xs[i] = bf16_to_f32(src[i]);
      }
    }
  } else if (g.sC == 1) {  // channels-last: walk (pixel, channel) with channel fastest -> coalesced
    for (int i = t; i < n; i += T) {
      int pix = i / cc, c = i - pix * cc;
      int y = pix / g.W, xx = pix - y * g.W;
      xs[c * g.P + pix] = ldx(x, (long long)b * g.sB + (long long)y * g.sH + (long long)xx * g.sW + c0 + c, g.dtype);
    }
  } else {
    for (int i = t; i < n; i += T) {
      int c = i / g.P, pix = i - c * g.P;
      int y = pix / g.W, xx = pix - y * g.W;
      xs[i] = ldx(x, (long long)b * g.sB + (long long)(c0 + c) * g.sC + (long long)y * g.sH + (long long)xx * g.sW,
                  g.dtype);
    }
  }
}

// ---- LDS -> grad_x[b, c0:c0+cc, :, :] -------------------------------------------------------
__device__ __forceinline__ void unstage_chunk(const float* gs, void* gx, const KP& g, int b, int c0, int cc) {
  const int t = threadIdx.x, T = blockDim.x, n = cc * g.P;
  if (g.contig) {
    const long long base = (long long)b * g.gB + (long long)c0 * g.P;
    if (g.dtype == NFP_F32) {
      float* dst = (float*)gx + base;
      if ((((uintptr_t)dst) & 15) == 0) {
        const int n4 = n >> 2;
        for (int i = t; i < n4; i += T) ((float4*)dst)[i] = ((const float4*)gs)[i];
        for (int i = (n4 << 2) + t; i < n; i += T) dst[i] = gs[i];
      } else {
        for (int i = t; i < n; i += T) dst[i] = gs[i];
      }
    } else {
      uint16_t* dst = (uint16_t*)gx + base;
      if ((((uintptr_t)dst) & 15) == 0) {
        const int n8 = n >> 3;
        for (int i = t; i < n8; i += T) {
          float4 lo = ((const float4*)gs)[2 * i], hi = ((const float4*)gs)[2 * i + 1];
          uint4 v;
          v.x = (uint32_t)f32_to_bf16(lo.x) | ((uint32_t)f32_to_bf16(lo.y) << 16);
          v.y = (uint32_t)f32_to_bf16(lo.z) | ((uint32_t)f32_to_bf16(lo.w) << 16);
          v.z = (uint32_t)f32_to_bf16(hi.x) | ((uint32_t)f32_to_bf16(hi.y) << 16);
          v.w = (uint32_t)f32_to_bf16(hi.z) | ((uint32_t)f32_to_bf16(hi.w) << 16);
          ((uint4*)dst)[i] = v;
        }
        for (int i = (n8 << 3) + t; i < n; i += T) dst[i] = f32_to_bf16(gs[i]);
      } else {
        for (int i = t; i < n; i += T) dst[i] = f32_to_bf16(gs[i]);
      }
    }
  } else if (g.sC == 1) {
    for (int i = t; i < n; i += T) {
      int pix = i / cc, c = i - pix * cc;
      int y = pix / g.W, xx = pix - y * g.W;
      stx(gx, (long long)b * g.gB + (long long)y * g.sH + (long long)xx * g.sW + c0 + c, gs[c * g.P + pix], g.dtype);
    }
  } else {
    for (int i = t; i < n; i += T) {
      int c = i / g.P, pix = i - c * g.P;
      int y = pix / g.W, xx = pix - y * g.W;
      stx(gx, (long long)b * g.gB + (long long)(c0 + c) * g.sC + (long long)y * g.sH + (long long)xx * g.sW, gs[i],
          g.dtype);
    }
  }
}

// ---- generic forward ------------------------------------------------------------------------
template <int M>
__global__ void __launch_bounds__(1024) fwd_generic(const KP g, const void* __restrict__ x, void* __restrict__ out,
                                                    float* __restrict__ saved) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  const int b = blockIdx.x, grp = blockIdx.z;
  const int t = threadIdx.x;
  const int gl = t & (g.G - 1), ol = t / g.G;
  const int o = blockIdx.y * g.Ow + ol;
  const bool active = ol < g.Ow && o < g.O;

  int q[kGroup];
  int pc = -1;
  if (active) pc = tap_pixel(g, o, g.R, g.R);
#pragma unroll
  for (int j = 0; j < kGroup; ++j) {
    int n = grp * kGroup + j;
    q[j] = (active && n < g.N) ? nbr_pixel(g, o, n) : -1;
  }
  float acc[kGroup], sb0[kGroup], sb1[kGroup];
  float sa0 = 0.f, sa1 = 0.f;
#pragma unroll
  for (int j = 0; j < kGroup; ++j) acc[j] = sb0[j] = sb1[j] = 0.f;
  // pivots (nfp_measures.h::Pivot): channel 0 of the centre and of each neighbour; 0 for zero-padded taps
  float pa = 0.f, pb[kGroup];
#pragma unroll
  for (int j = 0; j < kGroup; ++j) pb[j] = 0.f;
  if constexpr (Pivot<M>::v) {
    auto pix0 = [&](int p) {
      const int y = p / g.W, xx = p - y * g.W;
      return ldx(x, (long long)b * g.sB + (long long)y * g.sH + (long long)xx * g.sW, g.dtype);
    };
    if (pc >= 0) pa = pix0(pc);
#pragma unroll
    for (int j = 0; j < kGroup; ++j)
      if (q[j] >= 0) pb[j] = pix0(q[j]);
  }

  for (int c0 = 0; c0 < g.C; c0 += g.Cc) {
    const int cc = min(g.Cc, g.C - c0);
    __syncthreads();
    stage_chunk(xs, x, g, b, c0, cc);
    __syncthreads();
    if (active) {
      for (int c = gl; c < cc; c += g.G) {
        const float* row = xs + c * g.P;
        float a = pc >= 0 ? row[pc] : 0.f;
        if constexpr (Pivot<M>::v) a -= pa;
        Meas<M>::stat(a, sa0, sa1);
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          float bv = q[j] >= 0 ? row[q[j]] : 0.f;
          if constexpr (Pivot<M>::v) bv -= pb[j];
          acc[j] += Meas<M>::term(a, bv, g);
          Meas<M>::stat(bv, sb0[j], sb1[j]);
        }
      }
    }
  }
  // channel reduction across the G adjacent lanes of this output
  for (int m = g.G >> 1; m >= 1; m >>= 1) {
    sa0 += __shfl_xor(sa0, m);
    if (Meas<M>::NSTAT > 1) sa1 += __shfl_xor(sa1, m);
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
      acc[j] += __shfl_xor(acc[j], m);
      if (Meas<M>::NSTAT > 0) sb0[j] += __shfl_xor(sb0[j], m);
      if (Meas<M>::NSTAT > 1) sb1[j] += __shfl_xor(sb1[j], m);
    }
  }
  if (active && gl == 0) {
#pragma unroll
    for (int j = 0; j < kGroup; ++j) {
      int n = grp * kGroup + j;
      if (n < g.N) {
        float v = Meas<M>::fin(acc[j], sa0, sa1, sb0[j], sb1[j], g);
        stx(out, ((long long)b * g.N + n) * g.O + o, v, g.odtype);
      }
    }
    if constexpr (Meas<M>::NSTAT > 0) if (saved != nullptr) {
      // per-input-pixel stats for backward, [B][NSTAT][P]; several threads may store the same pixel,
      // all with a value summed in the same order (benign duplicate stores)
      float* sv = saved + (long long)b * Meas<M>::NSTAT * g.P;
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
}

// ---- generic backward -----------------------------------------------------------------------
template <int M>
__global__ void __launch_bounds__(1024) bwd_generic(const KP g, const void* __restrict__ x,
                                                    const void* __restrict__ go, const void* __restrict__ out,
                                                    const float* __restrict__ saved, void* __restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* ga = lds + ((g.Cc * g.P + 3) & ~3);
  const int b = blockIdx.x;
  const int cb0 = blockIdx.y * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  const int t = threadIdx.x, T = blockDim.x;
  const int ol = t % g.Ow, cl = t / g.Ow;
  const int ngrp = (g.N + kGroup - 1) / kGroup;
  const float* sv = (Meas<M>::NSTAT > 0) ? saved + (long long)b * Meas<M>::NSTAT * g.P : nullptr;

  for (int c0 = cb0; c0 < cb1; c0 += g.Cc) {
    const int cc = min(g.Cc, cb1 - c0);
    __syncthreads();
    stage_chunk(xs, x, g, b, c0, cc);
    for (int i = t; i < cc * g.P; i += T) ga[i] = 0.f;
    __syncthreads();
    for (int ob = 0; ob < g.O; ob += g.Ow) {
      const int o = ob + ol;
      const bool active = o < g.O && cl < g.Tc;
      if (!active) continue;
      const int pc = tap_pixel(g, o, g.R, g.R);
      const float sp0 = (Meas<M>::NSTAT > 0 && pc >= 0) ? sv[pc] : 0.f;
      const float sp1 = (Meas<M>::NSTAT > 1 && pc >= 0) ? sv[g.P + pc] : 0.f;
      for (int grp = 0; grp < ngrp; ++grp) {
        int q[kGroup];
        Coef cf[kGroup];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          int n = grp * kGroup + j;
          q[j] = -2;  // -2: no such neighbour; -1: zero-padded tap
          if (n < g.N) {
            q[j] = nbr_pixel(g, o, n);
            long long oi = ((long long)b * g.N + n) * g.O + o;
            float sq0 = (Meas<M>::NSTAT > 0 && q[j] >= 0) ? sv[q[j]] : 0.f;
            float sq1 = (Meas<M>::NSTAT > 1 && q[j] >= 0) ? sv[g.P + q[j]] : 0.f;
            cf[j] = Meas<M>::coef(ldx(go, oi, g.godtype), ldx(out, oi, g.dtype), sp0, sp1, sq0, sq1, g);
          }
        }
        for (int c = cl; c < cc; c += g.Tc) {
          float* xrow = xs + c * g.P;
          float* grow = ga + c * g.P;
          float a = pc >= 0 ? xrow[pc] : 0.f;
          float da_sum = 0.f;
#pragma unroll
          for (int j = 0; j < kGroup; ++j) {
            if (q[j] > -2) {
              float bv = q[j] >= 0 ? xrow[q[j]] : 0.f;
              float da, db;
              Meas<M>::grad(a, bv, cf[j], g, da, db);
              da_sum += da;
              if (q[j] >= 0) atomicAdd(&grow[q[j]], db);
            }
          }
          if (pc >= 0) atomicAdd(&grow[pc], da_sum);
        }
      }
    }
    __syncthreads();
    unstage_chunk(ga, gx, g, b, c0, cc);
  }
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
