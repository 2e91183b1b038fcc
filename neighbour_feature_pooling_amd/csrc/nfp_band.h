// nfp_band.h — the hot-path forward, split into ROW BANDS of an image (replaces nfp.py:132-159 like fwd_fast).
//
// Why: at the headline shape [64,512,7,7] one workgroup per image puts 64 workgroups on 256 compute units and each
// of them ingests 100 KB and sums 512 channels alone (round 1: 3 us of a 6.2 us workgroup).  Here workgroup
// (image b, band j) OWNS output rows [j*rb, (j+1)*rb) and stages those rows plus the R rows below them; with the
// half stencil a pair (p, q = p + forward direction) is summed by the band that owns p, so no band needs rows above
// its own and nothing is combined across workgroups.  Every output (n, p) is WRITTEN by the band that owns the
// forward end of its pair (table `ft`, nfp_tables.h) — for the R rows below a band that is the band above — so each
// element of `out` has exactly one writer, no atomics, and a launch is bitwise reproducible.  (The number of bands
// follows the batch size, and with it the number of channel groups a pixel's sums are split into: results for
// different batch sizes agree to float32 rounding, not bitwise.)  At B >= 256 the launcher asks for one band = the
// whole image.
//
// LDS slab layout, staging blocks and the pixel-slot rotation are those of nfp_fast.h, with slots counted from the
// band's first staged pixel rounded down to a multiple of 8 (so that the rotation, and with it the offset table
// `foff`, is the same as for the whole image).
#pragma once
#include "nfp_fast.h"

namespace nfp {

#ifndef NFP_BAND_T
#define NFP_BAND_T 1024
#endif
constexpr int kBandT = NFP_BAND_T;  // most threads per workgroup: pixels of the band x channel groups (a power of two <= 32)
constexpr int kBandRB = 3;   // NCHW staging: 4-pixel x 4-channel blocks per thread per chunk
constexpr int kBandRN = 6;   // channels-last staging: slots per thread per chunk
#ifndef NFP_BAND_SKIP_ROUNDS
#define NFP_BAND_SKIP_ROUNDS 1   // ([256,512,7,7] forward 6.64 -> 6.48 us, [256,192,14,14] 10.56 -> 10.10; headline unchanged: profiles/r04_i_…)
#endif

// Slots per slab row (one channel quad) of a band whose staged pixels span `span` slots (a multiple of 4).  The channel
// sums read one ds_read_b128 per (quad, direction) with thread t = pixel * G + group: the hardware serves 16 lanes per
// cycle (lanes {0-3,12-15,20-27}, ...), i.e. 16/G pixels x G quads when G < 16, and those 16 slots must fall into 16
// different 16-byte bank columns.  G >= 32: the 16 lanes are 16 quads of ONE pixel -> any odd stride.  G = 16: 8 quads
// of a pixel + 8 of the next -> stride = 2 mod 4.  G = 8, 4: four quads each of four consecutive pixels -> stride = 4
// mod 8.  (Round 2 used an odd stride for every G: at [4096,512,7,7], G = 8, 52 % of the LDS-active cycles were bank
// conflicts — profiles/r03_b_fwd_band_pmc_baseline.csv; tools/lds_bank_sim_band.py reproduces 9.0 vs 4.8 cycles per read.)
__host__ __device__ inline int band_row_slots(int span, int lg) {
  if (lg >= 5 || lg <= 1) return span | 1;
  if (lg == 4) return span + 2;
  return (span & 7) == 4 ? span : span + 4;
}

template <int R>
struct FoffQ {
  static constexpr int v = ((Win<R>::NF + 7) & ~7) / 8;  // 16-byte pieces of a pixel's foff row (ws_layout: FR)
};
// (template parameter R of fwd_band: the radius SPEC of nfp_tables.h::Win — 1, 2, or 12 for radii 1 and 2 together;
// the window radius itself is g.R)

// POOL = the fused tail of models/NFP_Pooling.py:27-31: besides the neighbour maps the same pass emits
// gap[b,c] = mean over pixels of x[b,c]  (AdaptiveAvgPool2d(1), NFP_Pooling.py:27) and  nfpm[b,n] = mean over pixels of
// out[b,n]  (adaptive_avg_pool2d of the NFP maps, NFP_Pooling.py:31), both float32, summed in a fixed order, for either
// layout and storage type (the sums are taken from the float32 LDS slab).  One band per image writes the two means
// itself; with several bands per image (small batches: round 2 ran the pooled forward on one workgroup per image, 64 of
// 256 compute units at the headline shape, 8.1 us against 4.85 us for the plain forward) every band writes its share of
// the sums to row (b, band) of a scratch passed in `gap` and nfp_tile.h::pool_fold joins the bands in a fixed order.
//
// Thread maps.  Channel sums: thread t = lp * G + gl — the G channel groups of a pixel are ADJACENT LANES (G a power
// of two <= 32, g.G; g.Tc = log2 G), so the groups' partial sums are joined by a fixed DPP tree inside the wavefront:
// no LDS round trip, no barrier.  Outputs: thread t = gl' * Ps + lp' — lanes along pixels, coalesced stores.
// VAR: one of the measures that ride on these kernels through KP's run-time constants (DotProduct, GFC, RMSE:
// nfp_common.h).  Cosine and L2 themselves keep the finalize of round 2, constants folded: the three extra transcendental
// instructions of the general form sit on the critical path of a 5 us kernel (4.97 vs 4.87 us at the headline shape).
template <int R, int M, bool BF, bool NHWC, bool POOL = false, bool VAR = false>
__global__ void __launch_bounds__(1024) fwd_band(const KP g, const void* __restrict__ x, void* __restrict__ out,
                                                 float* __restrict__ saved, const unsigned char* __restrict__ ws,
                                                 int rb, float* __restrict__ gap, float* __restrict__ nfpm,
                                                 float* __restrict__ part) {
  constexpr int N = Win<R>::N, NF = Win<R>::NF;
  constexpr int ES = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  float4* slab = lds4;
  const int P = g.P, W = g.W;
  const int b = blockIdx.x, band = blockIdx.y, t = threadIdx.x, T = blockDim.x;
  // rows: owned [y0, y1), staged [y0, ye); pixels: owned [p0, po), staged [p0, pe)
  const int y0 = band * rb, y1 = min(g.H, y0 + rb), ye = min(g.H, y1 + g.R);
  const int p0 = y0 * W, po = y1 * W, pe = ye * W, Ps = pe - p0;
  const int base = p0 & ~7;                          // slot origin of the band's slab rows
  const int G = g.G, lg = g.Tc;                      // channel groups (power of two), log2
  const int Ppb = band_row_slots(((pe + 3) & ~3) - base, lg);   // slots per slab row (one channel quad)
  const int gl = t & (G - 1), lpc = t >> lg;         // channel-sum map
  const bool active = lpc < Ps;
  const int lp = min(lpc, Ps - 1), p = p0 + lp;
  const int glf = fdivi(t, Ps), lpf = t - glf * Ps, pf = p0 + lpf;   // output map
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);  // wave-uniform
  const WsLayout L = ws_layout(P, R, g.mode);
  float* Tt = (float*)(lds4 + (g.Cc >> 2) * Ppb);    // [NF+1][Ps] behind the slab: pair sums per direction, then |x|^2

  NFP_STAMP_INIT();
  NFP_STAMP(0);
  // ---- the x chunk FIRST, then the tables (round 4) ------------------------------------------------------------------
  // Rounds 2-3 requested the tables first ("small, L2"): ~170 instructions of index arithmetic and two table requests
  // stood between kernel entry and the first x request (hipcc -S of the headline instantiation), on the critical path of
  // a 5 us kernel whose long pole is the first x data.  Loads retire in order, so the tables now arrive right behind the
  // chunk — they are needed after the sums.  (-DNFP_BAND_TABLES_FIRST=1 builds the old order.)
#ifndef NFP_BAND_TABLES_FIRST
#define NFP_BAND_TABLES_FIRST 0
#endif
  const uint32_t* ftt = (const uint32_t*)(ws + L.ft);
  // this thread's outputs (n = glf, glf + Gn, ...; pf): every table entry is requested at once — round 3 prefetched the
  // first and loaded the others inside the output loop, a dependent L2 round trip per round (four at 14x14 with k = 5)
  constexpr int kFte = (N + 3) / 4 < 6 ? (N + 3) / 4 : 6;
  uint32_t fte[kFte];
  uint4 fo[FoffQ<R>::v];
  auto tables = [&]() {
    const int Gn0 = fdivi(T, Ps);
#pragma unroll
    for (int k = 0; k < kFte; ++k) fte[k] = ftt[min(glf + k * Gn0, N - 1) * P + pf];
    const uint4* fot = (const uint4*)(ws + L.foff) + (long long)p * FoffQ<R>::v;
#pragma unroll
    for (int u = 0; u < FoffQ<R>::v; ++u) fo[u] = fot[u];
  };
  if (NFP_BAND_TABLES_FIRST) {
    tables();
    __builtin_amdgcn_sched_barrier(0);
  }

  // NCHW blocks of the band: 4-pixel blocks q0 .. q1-1 of every channel; the last block of an image whose pixel
  // count is not a multiple of 4 starts at P - 4 instead (it overlaps its predecessor: same values, written twice)
  const int q0 = p0 >> 2, q1 = (pe + 3) >> 2, NQb = q1 - q0;
  float4 blk[kBandRB][4];
  float4 nv[kBandRN];
  auto issue = [&](int c0, int ncq) {
    if constexpr (NHWC) {
      const int last = gl < ncq ? gl + (((ncq - 1 - gl) >> lg) << lg) : 0;
#pragma unroll
      for (int k = 0; k < kBandRN; ++k) {
        const int cq = min(gl + k * G, last);
        nv[k] = load_px4<BF>(xb, p * g.C + c0 + 4 * cq, 0);
      }
    } else {
      const int nblk = ncq * NQb;
#pragma unroll
      for (int r = 0; r < kBandRB; ++r) {
#if NFP_BAND_SKIP_ROUNDS
        // (a round no thread of the workgroup needs — the headline shape: 768 blocks on 704 threads, two rounds of three —
        // is skipped as a whole: a wave-uniform branch around four requests, not a select around a load)
        if (r > 0 && r * T >= nblk) break;
#endif
        const int i = min(t + r * T, nblk - 1);
        const int cq = fdivi(i, NQb), pq = q0 + i - cq * NQb;
        const int e = (c0 + 4 * cq) * P + min(4 * pq, P - 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) blk[r][j] = load_px4<BF>(xb, e, j * P);
      }
    }
  };
  auto commit = [&](int ncq) {
    if constexpr (NHWC) {
#pragma unroll
      for (int k = 0; k < kBandRN; ++k) {
        const int cq = gl + k * G;
        if (active && cq < ncq) slab[cq * Ppb + swz(p) - base] = nv[k];
      }
    } else {
      const int nblk = ncq * NQb;
#pragma unroll
      for (int r = 0; r < kBandRB; ++r) {
        const int i = t + r * T;
        if (i < nblk) {
          const int cq = fdivi(i, NQb), pq = q0 + i - cq * NQb;
          const int ps = min(4 * pq, P - 4);
          float4* d = slab + cq * Ppb - base;
          // (the overlapped last block of an image with P % 4 != 0 may start below the band's slot origin: those pixels
          // belong to the band above and have no slot here.  Their values go to the slot of ps + 3 instead, which the
          // last store below then overwrites with its own value — a select on the address, not a predicated store: three
          // predicates cost 24 registers and a spill in this kernel)
          const int s3 = swz(ps + 3);
#ifdef NFP_BAND_NO_GUARD   // (A/B: round 2's unguarded stores)
          d[swz(ps)] = make_float4(blk[r][0].x, blk[r][1].x, blk[r][2].x, blk[r][3].x);
          d[swz(ps + 1)] = make_float4(blk[r][0].y, blk[r][1].y, blk[r][2].y, blk[r][3].y);
          d[swz(ps + 2)] = make_float4(blk[r][0].z, blk[r][1].z, blk[r][2].z, blk[r][3].z);
#else
          d[ps >= base ? swz(ps) : s3] = make_float4(blk[r][0].x, blk[r][1].x, blk[r][2].x, blk[r][3].x);
          d[ps + 1 >= base ? swz(ps + 1) : s3] = make_float4(blk[r][0].y, blk[r][1].y, blk[r][2].y, blk[r][3].y);
          d[ps + 2 >= base ? swz(ps + 2) : s3] = make_float4(blk[r][0].z, blk[r][1].z, blk[r][2].z, blk[r][3].z);
#endif
          d[swz(ps + 3)] = make_float4(blk[r][0].w, blk[r][1].w, blk[r][2].w, blk[r][3].w);
        }
      }
    }
  };
  issue(0, min(g.Cc, g.C) >> 2);
  __builtin_amdgcn_sched_barrier(0);
  if (!NFP_BAND_TABLES_FIRST) {
    tables();
    __builtin_amdgcn_sched_barrier(0);
  }
  NFP_STAMP(1);

  int off[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) {
    const uint32_t wd = d & 1 ? ((const uint32_t*)fo)[d >> 1] >> 16 : ((const uint32_t*)fo)[d >> 1] & 0xFFFFu;
    off[d] = p < po ? (int)(int16_t)wd : 0;   // halo pixels only need their norm: their pairs belong to the next band
  }
  float acc[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = 0.f;
  float nrm = 0.f;
  const int sp = swz(p) - base;

  for (int c0 = 0; c0 < g.C; c0 += g.Cc) {
    const int ncq = min(g.Cc, g.C - c0) >> 2;
    // Several chunks (batches beyond one workgroup per CU): the NEXT chunk's loads are issued right after this chunk's
    // commit, so that they fly under its sums (round 3 issued them after the sums and left each workgroup with nothing in
    // flight two thirds of the time: SQ_WAIT_ANY 42 % of the wave cycles at [4096,512,7,7], profiles/r03_p_…csv).  The
    // staging registers stay live through the sums; with one chunk (the headline shape) nothing changes.
    if (c0 > 0) {
      __syncthreads();  // previous chunk fully consumed
      if (!g.pf) issue(c0, ncq);
    }
    commit(ncq);
    __syncthreads();
    if (g.pf && c0 + g.Cc < g.C) {
      issue(c0 + g.Cc, min(g.Cc, g.C - c0 - g.Cc) >> 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c0 == 0) NFP_STAMP(2);
    if (POOL && g.pool_gap) {
      // channel sums of this chunk over the band's OWN pixels: four adjacent lanes per channel quad, lane `part` sums
      // pixels p0 + part, p0 + part + 4, ...; joined by a fixed xor tree.  One band per image: the mean goes straight to
      // gap[b][c]; several bands: the band's sum goes to its row of the scratch (pool_fold joins the bands in order).
      const int nbands = gridDim.y;
      float* gdst = nbands == 1 ? gap + (long long)b * g.C : part + ((long long)b * nbands + band) * (g.C + N);
      const float gscale = nbands == 1 ? g.invP : 1.f;
      for (int i0 = 0; i0 < ncq * 4; i0 += T) {
        const int i = i0 + t, cq = min(i >> 2, ncq - 1), quarter = i & 3;   // (which quarter of the pixels this lane sums)
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int pp = p0 + quarter; pp < po; pp += 4) {
          const float4 v = slab[cq * Ppb + swz(pp) - base];
          s4.x += v.x;
          s4.y += v.y;
          s4.z += v.z;
          s4.w += v.w;
        }
        s4.x = group_sum(s4.x, 4);
        s4.y = group_sum(s4.y, 4);
        s4.z = group_sum(s4.z, 4);
        s4.w = group_sum(s4.w, 4);
        if (i < ncq * 4 && quarter == 0) {
          const float4 gv = make_float4(s4.x * gscale, s4.y * gscale, s4.z * gscale, s4.w * gscale);
          if (nbands == 1)
            *(float4*)(gdst + c0 + 4 * cq) = gv;
          else   // (a scratch row another workgroup may fold: written through — nfp_common.h::pool_last_band)
            pool_store4(pool_rsrc(gdst, g.C + N), c0 + 4 * cq, gv);
        }
      }
    }
    if (active) {
      for (int cq = gl; cq < ncq; cq += G) {
        const float4* row = slab + cq * Ppb + sp;
        const float4 a = row[0];
        if constexpr (M == kSymTerm) {   // (nfp_measures.h: a symmetric per-channel term; nrm = the sums against a zero-padded tap)
          sym_switch(g.measure, [&](auto mm) {
            using MM = decltype(mm);
            nrm += (MM::term(a.x, 0.f, g) + MM::term(a.y, 0.f, g)) + (MM::term(a.z, 0.f, g) + MM::term(a.w, 0.f, g));
#pragma unroll
            for (int d = 0; d < NF; ++d) {
              const float4 q = row[off[d]];
              acc[d] += (MM::term(a.x, q.x, g) + MM::term(a.y, q.y, g)) + (MM::term(a.z, q.z, g) + MM::term(a.w, q.w, g));
            }
          });
          continue;
        }
        if (M == kNormP1)   // (Norm p = 1, nfp.py:141-148 with the class default p: sums of |.|; |x_p|_1 for the 'Norm' quirk)
          nrm += (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w));
        else
          nrm = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, fmaf(a.w, a.w, nrm))));
#pragma unroll
        for (int d = 0; d < NF; ++d) {
          const float4 q = row[off[d]];
          if (M == NFP_COSINE) {
            acc[d] = fmaf(a.x, q.x, fmaf(a.y, q.y, fmaf(a.z, q.z, fmaf(a.w, q.w, acc[d]))));
          } else if (M == kNormP1) {
            acc[d] += (fabsf(a.x - q.x) + fabsf(a.y - q.y)) + (fabsf(a.z - q.z) + fabsf(a.w - q.w));
          } else {
            const float e0 = a.x - q.x, e1 = a.y - q.y, e2 = a.z - q.z, e3 = a.w - q.w;
            acc[d] = fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, fmaf(e3, e3, acc[d]))));
          }
        }
      }
    }
  }
  // ---- channel groups joined inside the wavefront (fixed tree), one lane per pixel publishes the sums ---------
  NFP_STAMP(3);
  const int NV = (NF + 1) * Ps;
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = group_sum(acc[d], G);
  nrm = group_sum(nrm, G);
  if (active && gl == G - 1) {
#pragma unroll
    for (int d = 0; d < NF; ++d) Tt[d * Ps + lp] = acc[d];
    Tt[NF * Ps + lp] = nrm;
  }
  __syncthreads();
  NFP_STAMP(4);
  // ---- outputs (n, p) whose pair this band summed: thread (lp', n = gl', gl' + Gn, ...) ------------------------
  const float* n2 = Tt + NF * Ps;
  const int Gn = fdivi(T, Ps);
  if (glf < Gn) {
    void* ob = (char*)out + (long long)b * N * P * ES;
    const float n2p = n2[lpf];
    const float ip = VAR ? unit_or(g, inv_norm(n2p, g.inv_eps)) : inv_norm(n2p, g.inv_eps);
    int it = 0;
    for (int n = glf; n < N; n += Gn, ++it) {
      uint32_t e = fte[0];
#pragma unroll
      for (int k = 1; k < kFte; ++k) e = it == k ? fte[k] : e;
      if (it >= kFte) e = ftt[n * P + pf];
      const int kind = (int)(e >> 22), pix = (int)((e >> 9) & 511u), fi = (int)((e >> 18) & 15u);
      const int q = kind == 2 ? pf : (int)(e & 511u);
      if (pix >= p0 && pix < po) {
        const float pairv = Tt[fi * Ps + pix - p0];
        const float n2q = n2[q - p0];
        float v;
        if (M == NFP_COSINE) {
          if constexpr (VAR) {
            const float s = kind == 2 ? 0.f : prod_value(g, kind == 1 ? n2p : pairv, n2p, n2q, ip, unit_or(g, inv_norm(n2q, g.inv_eps)));
            v = fin_prod(g, s);
          } else {
            const float s = kind == 2 ? 0.f : (kind == 1 ? n2p : pairv) * ip * inv_norm(n2q, g.inv_eps);
            v = g.similarity ? s : 1.f - s;
          }
        } else {
          float d2;
          if (g.diff || M == kSymTerm)   // (kSymTerm: a pixel against its own copy sums to 0 under every one of its terms)
            d2 = kind == 2 ? n2p : (kind == 1 ? 0.f : pairv);
          else
            d2 = kind == 2 ? 0.f : n2q;  // 'Norm' quirk (nfp.py:74 vs 85): |neighbour|
          if constexpr (M == kSymTerm) {
            sym_switch(g.measure, [&](auto mm) { v = decltype(mm)::fin(d2, 0.f, 0.f, 0.f, 0.f, g); });
          } else if constexpr (M == kNormP1) {
            v = g.similarity ? -d2 : d2;   // (no root: the sum of |.| is the norm)
          } else if constexpr (VAR) {
            v = fin_dist(g, d2);
          } else {
            const float dd = __builtin_amdgcn_sqrtf(d2);
            v = g.similarity ? -dd : dd;
          }
        }
        if (!POOL || g.pool_map) stx(ob, n * P + pf, v, BF ? NFP_BF16 : NFP_F32);
        if constexpr (POOL) Tt[NV + n * Ps + lpf] = v;  // vm[n][p], behind the half-stencil table
      } else if constexpr (POOL) {
        Tt[NV + n * Ps + lpf] = 0.f;                    // (an output another band writes: not part of this band's sum)
      }
    }
    if (M == NFP_COSINE && !g.unit && saved != nullptr && glf == 0 && pf < po) saved[(long long)b * P + pf] = __builtin_amdgcn_sqrtf(n2p);
  }
  if constexpr (POOL) {
    lds_barrier();   // (not __syncthreads(): the map stores just issued drain under the sums — nfp_common.h)
    // wave w reduces map n = w, w + nwaves, ...: lane-strided partial sums over the outputs THIS band wrote, then a
    // fixed shuffle tree
    const float* vm = Tt + NV;
    const int lane = t & 63, wv = t >> 6, nw = T >> 6, nbands = gridDim.y;
    for (int n = wv; n < N; n += nw) {
      float sacc = 0.f;
      for (int i = lane; i < Ps; i += 64) sacc += vm[n * Ps + i];
      for (int m = 32; m >= 1; m >>= 1) sacc += __shfl_xor(sacc, m);
      if (lane == 0) {
        if (nbands == 1)
          nfpm[(long long)b * N + n] = sacc * g.invP;
        else
          pool_store1(pool_rsrc(part + ((long long)b * nbands + band) * (g.C + N), g.C + N), g.C + n, sacc);
      }
    }
    // several bands and a ticket counter: the band that arrives last folds all of them (nfp_common.h::pool_last_band);
    // without counters (no workspace) the caller launches pool_fold
    if (nbands > 1 && g.tickets != nullptr) {
      if (pool_last_band(g.tickets + b, nbands, (int*)lds4, t == 0))
        pool_fold_image(part + (long long)b * nbands * (g.C + N), nbands, g.C, N, g.invP,
                        g.pool_gap ? gap + (long long)b * g.C : nullptr, nfpm + (long long)b * N, t, T);
    }
  }
  NFP_STAMP(5);
}

}  // namespace nfp
