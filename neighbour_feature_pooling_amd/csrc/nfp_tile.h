// nfp_tile.h — the hot path for maps ABOVE 512 pixels: row-band kernels on a PADDED LDS slab, no tables.
//
// Who needs it: MobileNetV3_MultiStageNFP feeds NFP 112x112x16, 56x56x24 and 28x28x40 maps
// (models/texture_pooling.py:211-268), RESNET18_NFP_AT_LAYER 56x56x64 and 28x28x128 (models/resnet18.py:410-468).
// Round 2 served them with the any-geometry kernels of nfp_gather.h at 0.13-0.25 (forward) and 0.03-0.18 (backward) of
// the HBM roofline (profiles/r03_a_bigmaps_baseline_general_kernels.jsonl).  Same arithmetic as nfp_band.h /
// nfp_fast.h (nfp.py:141-159: cosine and L2 over "same" maps: stride 1, dilation 1, padding = R), different index
// scheme: the workspace tables of the small-map kernels grow with H*W and a 7x7 map is half border, a 112x112 map 3 %.
//
// A workgroup owns a BAND of rows [y0, y1) of one image.  It stages rows y0-R .. y1+R-1 of a channel chunk into LDS
// as the PADDED map the reference's F.pad would build (nfp.py:42-58: reflect / replicate / zeros): R ring columns left
// and right of every row, ring rows above / below the image, every ring slot filled from its fold source while
// staging.  In padded coordinates every tap of every pixel sits at a CONSTANT offset: the channel loops have no
// border cases, no index tables, and a zero-padded tap is a zero vector like any other.
//   forward   half stencil over padded positions (a pair {u, u+d} is summed once, by the thread of u, ring positions
//             included), then every output (n, p) of the band's rows looks its pair up; nothing is combined across
//             workgroups; an output element has one writer.
//   backward  gather form as in nfp_fast.h: per band pixel r the window weights W[r][j] (phase A), then
//             grad_x[c][r] = sum_j W[r][j] * xpad[c][r + d_j] in one pass over the slab (phase B).  Phase A needs no
//             tables either: slot j of an interior pixel links two pairs (r's tap j; the tap -j of the pixel under
//             it), and a pixel near the border also collects the pairs whose neighbour is a RING position that folds
//             onto it — enumerated by the pixel's own thread in a fixed order (no atomics, bitwise reproducible).
// The halo rows two bands share are re-read from L2: workgroup ids are mapped so that the bands of one image run on ONE
// XCD (ids are dealt round-robin over the 8 XCDs, each with its own L2).
#pragma once
#include "nfp_band.h"

namespace nfp {

// staging registers per thread and chunk: KB 4-pixel x 4-channel blocks (NCHW), KR ring slots (NCHW), KN slots (channels-last).
// Forward: one block per thread already fills the slab (T threads x 64 bytes = the 60 KB a workgroup may use), and the
// kernel must stay within 64 registers.  Backward: half as many threads per staged pixel.
constexpr int kFwdKB = 1, kFwdKR = 1, kFwdKN = 4;
constexpr int kBwdKB = 2, kBwdKR = 1, kBwdKN = 6;
constexpr int kTileKA = 3;  // backward: 16-byte pieces of grad_out / out per thread per batch (a band of >= 4 rows, k = 3: one batch)
constexpr int kXL = 4;      // left margin of a padded row: image column x sits at row position kXL + x, so that the 4-pixel
                            // blocks of the NCHW staging start on multiples of four slots (Wp is a multiple of 4)

struct TileGeo {  // by value in kernarg
  int rb, nb;     // rows per band, bands per image
  int Wp;         // row stride of the padded band in positions: kXL + W + R rounded up to 4
  int S;          // backward: channel blocks per (image, band)
};
__host__ __device__ inline int tile_row_stride(int W, int R) { return (kXL + W + R + 3) & ~3; }

// workgroup id -> (image, item of the image): ids i, i + 8, i + 16, ... share an XCD, so a group of 8 images is dealt one
// image per XCD and all `per` items (bands x channel blocks) of an image follow each other on it.  Bijective for any B
// (the last group may hold fewer than 8 images; its placement is then only partly XCD-aligned: speed, not correctness).
__device__ __forceinline__ void tile_ids(int id, int B, int per, int& b, int& item) {
  const int grp = id / (8 * per), l = id - grp * 8 * per;   // (once per kernel, wave-uniform: plain integer division)
  const int m = min(8, B - 8 * grp);
  item = l / m;
  b = 8 * grp + l - item * m;
}

// The band's padded geometry.  Two index spaces: slab POSITIONS u = yy * Wp + kXL + x (rows of stride Wp, margins
// unused) and the compact USEFUL positions v = yy * Wu + x + R, Wu = W + 2R (ring columns included) that threads and
// the per-position tables are numbered by.  A tap (dy, dx) is u + dy * Wp + dx resp. v + dy * Wu + dx.
template <int R>
struct TileBand {
  int y0, y1, rows, Wp, Wu, npos, npu, nbp;  // owned rows [y0, y1); staged rows; slab positions; useful positions; band pixels
  __device__ __forceinline__ TileBand(const KP& g, const TileGeo& tg, int band) {
    y0 = band * tg.rb;
    y1 = min(g.H, y0 + tg.rb);
    rows = y1 - y0 + 2 * R;
    Wp = tg.Wp;
    Wu = g.W + 2 * R;
    npos = rows * Wp;
    npu = rows * Wu;
    nbp = (y1 - y0) * g.W;
  }
};

// nn.Conv2d's padding_mode as ARITHMETIC: a coordinate t outside [0, n) maps to a * t + b with per-mode constants (reflect:
// -t / 2(n-1) - t; replicate: 0 / n-1; zeros: -1 = "reads 0").  nfp_common.h::map_index selects on the mode — a
// wave-uniform value, which hipcc turns into scalar BRANCHES, five per call; these kernels call it a few dozen times per
// thread in straight-line setup code, where every taken branch is an instruction-fetch stall (measured: the index work
// in front of the first load took 2.9 us of a 12.8 us workgroup).  The constants are chosen once per kernel.
struct Fold {
  int a, blo, bhiH, bhiW;
  __device__ __forceinline__ Fold(const KP& g) {
    const bool refl = g.mode == NFP_PAD_REFLECT, repl = g.mode == NFP_PAD_REPLICATE;
    a = refl ? -1 : 0;
    blo = (refl || repl) ? 0 : -1;
    bhiH = refl ? 2 * (g.H - 1) : (repl ? g.H - 1 : -1);
    bhiW = refl ? 2 * (g.W - 1) : (repl ? g.W - 1 : -1);
  }
  __device__ __forceinline__ int y(int t, int H) const { return t < 0 ? a * t + blo : (t >= H ? a * t + bhiH : t); }
  __device__ __forceinline__ int x(int t, int W) const { return t < 0 ? a * t + blo : (t >= W ? a * t + bhiW : t); }
};

// v or zeros, by component (a ternary between two float4 LVALUES selects an address and forces both into scratch memory)
__device__ __forceinline__ float4 keep_if(bool in, float x, float y, float z, float w) {
  return make_float4(in ? x : 0.f, in ? y : 0.f, in ? z : 0.f, in ? w : 0.f);
}

// Staging of one channel chunk of the padded band: float4[cq][Ppb], slot swz(u) of slab position u.
// All loads of a chunk are issued back to back into registers (indices clamped onto valid items, nothing conditional
// around a load but wave-uniform round checks) and committed to LDS later, so that arithmetic can run under their
// latency.  What the commit needs to know about an item is kept in one register (its slot | flags): the index
// arithmetic runs once.
constexpr int kTileZero = 1 << 30, kTileSkip = 1 << 29, kTileSlot = kTileSkip - 1;
template <int R, bool BF, bool NHWC, int kTileKB, int kTileKR, int kTileKN>
struct TileStage {
  float4 blk[NHWC ? 1 : kTileKB][4];
  float4 ring[NHWC ? 1 : kTileKR];
  float4 nv[NHWC ? kTileKN : 1];
  int bdst[NHWC ? 1 : kTileKB], bu0[NHWC ? 1 : kTileKB], rdst[NHWC ? 1 : kTileKR], ndst[NHWC ? kTileKN : 1];

  __device__ __forceinline__ void issue(const KP& g, const Fold& fo, const TileBand<R>& bd, Rsrc xb, int Ppb, int c0, int ncq, int t,
                                        int T) {
    const int W = g.W, H = g.H, P = g.P, tw = t & ~63;
    if constexpr (NHWC) {
      const int items = bd.npu * ncq;
#pragma unroll
      for (int k = 0; k < kTileKN; ++k) {
        ndst[k] = kTileSkip;
        if (tw + k * T < items) {  // (wave-uniform)
          const int i = min(t + k * T, items - 1);
          const int v = fdivi(i, ncq), cq = i - v * ncq;
          const int yy = fdivi(v, bd.Wu), x = v - yy * bd.Wu - R;
          const int sy = fo.y(bd.y0 - R + yy, H), sx = fo.x(x, W);
          nv[k] = load_px4<BF>(xb, (max(sy, 0) * W + max(sx, 0)) * g.C + c0 + 4 * cq, 0);
          ndst[k] = (cq * Ppb + swz(yy * bd.Wp + kXL + x)) | ((sy | sx) < 0 ? kTileZero : 0) | (t + k * T < items ? 0 : kTileSkip);
        }
      }
    } else {
      const int nbr = (W + 3) >> 2, per = bd.rows * nbr, nblk = ncq * per;
#pragma unroll
      for (int r = 0; r < kTileKB; ++r) {
        bdst[r] = kTileSkip;
        bu0[r] = 0;
        if (tw + r * T < nblk) {
          const int i = min(t + r * T, nblk - 1);
          const int cq = fdivi(i, per), rem = i - cq * per, yy = fdivi(rem, nbr), bq = rem - yy * nbr;
          const int sy = fo.y(bd.y0 - R + yy, H), xs = min(4 * bq, W - 4);
          const int e = (c0 + 4 * cq) * P + max(sy, 0) * W + xs;
#pragma unroll
          for (int j = 0; j < 4; ++j) blk[r][j] = load_px4<BF>(xb, e, j * P);
          bdst[r] = (cq * Ppb) | (sy < 0 ? kTileZero : 0) | (t + r * T < nblk ? 0 : kTileSkip);
          bu0[r] = yy * bd.Wp + kXL + xs;
        }
      }
      const int pr = bd.rows * 2 * R, nring = ncq * pr;
#pragma unroll
      for (int r = 0; r < kTileKR; ++r) {
        rdst[r] = kTileSkip;
        if (tw + r * T < nring) {
          const int i = min(t + r * T, nring - 1);
          const int cq = fdivi(i, pr), rem = i - cq * pr, yy = fdivi(rem, 2 * R), k = rem - yy * 2 * R;
          const int x = k < R ? k - R : W + k - R;   // ring column: -R .. -1, W .. W+R-1
          const int sy = fo.y(bd.y0 - R + yy, H), sx = fo.x(x, W);
          const int e = (c0 + 4 * cq) * P + max(sy, 0) * W + max(sx, 0);
          ring[r] = make_float4(load_1<BF>(xb, e, 0), load_1<BF>(xb, e, P), load_1<BF>(xb, e, 2 * P), load_1<BF>(xb, e, 3 * P));
          rdst[r] = (cq * Ppb + swz(yy * bd.Wp + kXL + x)) | ((sy | sx) < 0 ? kTileZero : 0) | (t + r * T < nring ? 0 : kTileSkip);
        }
      }
    }
  }

  __device__ __forceinline__ void commit(float4* slab) const {
    if constexpr (NHWC) {
#pragma unroll
      for (int k = 0; k < kTileKN; ++k)
        if (!(ndst[k] & kTileSkip)) slab[ndst[k] & kTileSlot] = keep_if(!(ndst[k] & kTileZero), nv[k].x, nv[k].y, nv[k].z, nv[k].w);
    } else {
#pragma unroll
      for (int r = 0; r < kTileKB; ++r) {
        if (!(bdst[r] & kTileSkip)) {
          const bool in = !(bdst[r] & kTileZero);
          float4* d = slab + (bdst[r] & kTileSlot);
          const int u0 = bu0[r];
          d[swz(u0)] = keep_if(in, blk[r][0].x, blk[r][1].x, blk[r][2].x, blk[r][3].x);
          d[swz(u0 + 1)] = keep_if(in, blk[r][0].y, blk[r][1].y, blk[r][2].y, blk[r][3].y);
          d[swz(u0 + 2)] = keep_if(in, blk[r][0].z, blk[r][1].z, blk[r][2].z, blk[r][3].z);
          d[swz(u0 + 3)] = keep_if(in, blk[r][0].w, blk[r][1].w, blk[r][2].w, blk[r][3].w);
        }
      }
#pragma unroll
      for (int r = 0; r < kTileKR; ++r)
        if (!(rdst[r] & kTileSkip)) slab[rdst[r] & kTileSlot] = keep_if(!(rdst[r] & kTileZero), ring[r].x, ring[r].y, ring[r].z, ring[r].w);
    }
  }
};

// sum over the 64 lanes of a wavefront, valid in lane 63 (fixed DPP tree: group_sum covers each half, then the row
// broadcast of lane 31 into the upper half)
__device__ __forceinline__ float wave_sum(float v) {
  v = group_sum(v, 32);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast31 into rows 2, 3
  return v;
}

// ---- forward ----------------------------------------------------------------------------------------------------------
// POOL: the fused tail of models/NFP_Pooling.py:27-31 for large maps: besides the maps this band's share of the two
// pooled sums goes to part[(b * nb + band)][C + N] (sums, not means); pool_fold joins the bands in a fixed order.
// (k = 3: two workgroups of up to 1024 threads share a compute unit — 8 wavefronts per SIMD, 64 registers; k = 5 keeps
// twelve sums and offsets per thread: its launcher caps the workgroup at 512 threads instead)
template <int R, int M, bool BF, bool NHWC, bool POOL = false>
__global__ void __launch_bounds__(1024, (R == 1 ? 8 : 4)) fwd_tile(const KP g, const TileGeo tg, const void* __restrict__ x,
                                                 void* __restrict__ out, float* __restrict__ saved,
                                                 float* __restrict__ part) {
  constexpr int N = Win<R>::N, NF = Win<R>::NF;
  constexpr int ES = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int t = threadIdx.x, T = blockDim.x;
  int b, band;
  tile_ids(blockIdx.x, g.B, tg.nb, b, band);
  const TileBand<R> bd(g, tg, band);
  const Fold fo(g);
  const int W = g.W, P = g.P, Wp = bd.Wp, Wu = bd.Wu, npu = bd.npu, nbp = bd.nbp;
  const int G = g.G, lg = g.Tc;
  const int Ppb = band_row_slots(bd.npos, lg);   // (npos is a multiple of 4)
  float4* slab = lds4;
  float* Tt = (float*)(lds4 + (g.Cc >> 2) * Ppb);  // [NF + 1][npu]: pair sums per direction, then |x|^2
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  TileStage<R, BF, NHWC, kFwdKB, kFwdKR, kFwdKN> st;
  st.issue(g, fo, bd, xb, Ppb, 0, min(g.Cc, g.C) >> 2, t, T);
  __builtin_amdgcn_sched_barrier(0);

  // channel sums: thread t = useful position * G + group (the groups of a position are adjacent lanes: joined by DPP)
  const int gl = t & (G - 1), vc = t >> lg;
  const bool active = vc < npu;
  const int v = min(vc, npu - 1), vy = fdivi(v, Wu), vx = v - vy * Wu;
  const int u = vy * Wp + kXL - R + vx, su = swz(u);
  int off[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) {
    int dy, dx;
    fdir<R>(d, dy, dx);
    const bool ok = vx + dx >= 0 && vx + dx < Wu && vy + dy < bd.rows;
    off[d] = ok ? swz(u + dy * Wp + dx) - su : 0;
  }
  float acc[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = 0.f;
  float nrm = 0.f;
  NFP_STAMP(1);

  for (int c0 = 0; c0 < g.C; c0 += g.Cc) {
    const int ncq = min(g.Cc, g.C - c0) >> 2;
    if (c0 > 0) __syncthreads();  // previous chunk fully consumed
    st.commit(slab);
    __syncthreads();
    // the next chunk's loads fly while this one is summed (the staging registers are free once committed)
    if (c0 + g.Cc < g.C) st.issue(g, fo, bd, xb, Ppb, c0 + g.Cc, min(g.Cc, g.C - c0 - g.Cc) >> 2, t, T);
    if (c0 == 0) NFP_STAMP(2);
    if constexpr (POOL) {
      // this band's share of sum over pixels of x[c]: wavefront w takes channel quads w, w + nw, ...; lanes stride over
      // the band's pixels; fixed DPP tree; one writer per channel
      const int lane = t & 63, wv = t >> 6, nw = T >> 6;
      for (int cq = wv; cq < ncq; cq += nw) {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int lp = lane; lp < nbp; lp += 64) {
          const int yl = fdivi(lp, W), xl = lp - yl * W;
          const float4 q = slab[cq * Ppb + swz((yl + R) * Wp + kXL + xl)];
          s4.x += q.x;
          s4.y += q.y;
          s4.z += q.z;
          s4.w += q.w;
        }
        s4.x = wave_sum(s4.x);
        s4.y = wave_sum(s4.y);
        s4.z = wave_sum(s4.z);
        s4.w = wave_sum(s4.w);
        if (lane == 63) *(float4*)(part + ((long long)b * tg.nb + band) * (g.C + N) + c0 + 4 * cq) = s4;
      }
    }
    if (active) {
      for (int cq = gl; cq < ncq; cq += G) {
        const float4* row = slab + cq * Ppb + su;
        const float4 a = row[0];
        nrm = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, fmaf(a.w, a.w, nrm))));
#pragma unroll
        for (int d = 0; d < NF; ++d) {
          const float4 q = row[off[d]];
          if (M == NFP_COSINE) {
            acc[d] = fmaf(a.x, q.x, fmaf(a.y, q.y, fmaf(a.z, q.z, fmaf(a.w, q.w, acc[d]))));
          } else {
            const float e0 = a.x - q.x, e1 = a.y - q.y, e2 = a.z - q.z, e3 = a.w - q.w;
            acc[d] = fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, fmaf(e3, e3, acc[d]))));
          }
        }
      }
    }
  }
  NFP_STAMP(3);
  // channel groups joined inside the wavefront; one lane per position publishes the sums
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = group_sum(acc[d], G);
  nrm = group_sum(nrm, G);
  if (active && gl == G - 1) {
#pragma unroll
    for (int d = 0; d < NF; ++d) Tt[d * npu + v] = acc[d];
    Tt[NF * npu + v] = nrm;
  }
  __syncthreads();
  NFP_STAMP(4);
  // outputs of the band's rows: thread (pixel, n = glf, glf + Gn, ...), lanes along pixels (coalesced stores)
  const float* n2 = Tt + NF * npu;
  const int glf = fdivi(t, nbp), lpf = t - glf * nbp, Gn = fdivi(T, nbp);
  float* vm = Tt + (NF + 1) * npu;  // (POOL) [N][nbp]: the band's map values, for the pooled sums
  if (glf < Gn) {
    const int yl = fdivi(lpf, W), xl = lpf - yl * W, pv = (yl + R) * Wu + xl + R;
    const int p = (bd.y0 + yl) * W + xl;
    void* ob = (char*)out + (long long)b * N * P * ES;
    const float n2p = n2[pv];
    const float ip = unit_or(g, inv_norm(n2p, g.inv_eps));
    auto one = [&](int n) {
      int dy, dx;
      tap_offset<R>(n, dy, dx);
      const bool fwd = dy > 0 || (dy == 0 && dx > 0);
      const int fi = fwd ? fidx<R>(dy, dx) : fidx<R>(-dy, -dx);
      const int qv = pv + dy * Wu + dx;
      const float pairv = Tt[fi * npu + (fwd ? pv : qv)];
      const float n2q = n2[qv];
      float val;
      if (M == NFP_COSINE) {
        const float s = prod_value(g, pairv, n2p, n2q, ip, unit_or(g, inv_norm(n2q, g.inv_eps)));
        val = fin_prod(g, s);
      } else {
        val = fin_dist(g, g.diff ? pairv : n2q);  // 'Norm' quirk (nfp.py:74 vs 85): |neighbour|
      }
      stx(ob, n * P + p, val, BF ? NFP_BF16 : NFP_F32);
      if constexpr (POOL) vm[n * nbp + lpf] = val;
    };
    if (Gn == 1) {  // (the usual case on large maps: every tap's offsets are compile-time constants)
#pragma unroll
      for (int n = 0; n < N; ++n) one(n);
    } else {
      for (int n = glf; n < N; n += Gn) one(n);
    }
    if (M == NFP_COSINE && !g.unit && saved != nullptr && glf == 0) saved[(long long)b * P + p] = __builtin_amdgcn_sqrtf(n2p);
  }
  if constexpr (POOL) {
    // this band's share of sum over pixels of out[n]: wavefront w reduces map n = w, w + nw, ... over the band's pixels
    // in a fixed order
    __syncthreads();
    const int lane = t & 63, wv = t >> 6, nw = T >> 6;
    for (int n = wv; n < N; n += nw) {
      float s = 0.f;
      for (int i = lane; i < nbp; i += 64) s += vm[n * nbp + i];
      s = wave_sum(s);
      if (lane == 63) part[((long long)b * tg.nb + band) * (g.C + N) + g.C + n] = s;
    }
  }
  NFP_STAMP(5);
}

// gap[b][c] = (sum over bands of part[b][band][c]) / P, nfpm[b][n] likewise: the bands in a fixed order
__global__ void __launch_bounds__(256) pool_fold(const float* __restrict__ part, float* __restrict__ gap,
                                                 float* __restrict__ nfpm, int B, int nb, int C, int N, float invP) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CN = C + N;
  if (i >= (long long)B * CN) return;
  const int b = (int)(i / CN), k = (int)(i - (long long)b * CN);
  float s = 0.f;
  for (int j = 0; j < nb; ++j) s += part[((long long)b * nb + j) * CN + k];
  if (k < C)
    gap[(long long)b * C + k] = s * invP;
  else
    nfpm[(long long)b * N + k - C] = s * invP;
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// POOL: grad_out is not a map: go[b,n,p] = gnfpm[b,n] / P for every p, and every grad_x[b,c,p] also gets ggap[b,c] / P.
// GFC: the general post-factors of nfp_common.h::cross_f / diag_f (a reciprocal per window slot); cosine and dot keep the
// plain product of the two per-pixel factors.
template <int R, int M, bool BF, bool NHWC, bool POOL = false, bool GFC = false>
__global__ void __launch_bounds__(512) bwd_tile(const KP g, const TileGeo tg, const void* __restrict__ x,
                                                const void* __restrict__ go, const void* __restrict__ out,
                                                const float* __restrict__ saved, void* __restrict__ gx,
                                                const float* __restrict__ ggap, const float* __restrict__ gnfpm) {
  constexpr int N = Win<R>::N, K = Win<R>::K, K2 = Win<R>::K2;
  constexpr int ES = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int t = threadIdx.x, T = blockDim.x;
  int b, item;
  tile_ids(blockIdx.x, g.B, tg.nb * tg.S, b, item);
  const int band = item / tg.S, cblk = item - band * tg.S;
  const TileBand<R> bd(g, tg, band);
  const Fold fo(g);
  const int W = g.W, H = g.H, P = g.P, Wp = bd.Wp, Wu = bd.Wu, npu = bd.npu, nbp = bd.nbp;
  const int ya = max(0, bd.y0 - R), yb = min(H, bd.y1 + R), npA = (yb - ya) * W;  // rows whose pairs touch the band
  const int cb0 = cblk * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  // LDS: ipn [npu] | dfn [nbp] | pair values [N][npA] — and the x slab over the pair values.  (The window weights never
  // go through LDS: the thread that builds a pixel's row in phase A is the thread that uses it in phase B; with several
  // channel groups per pixel every group builds the row for itself.)
  float* ipn = (float*)lds4;
  float* dfn = ipn + npu;
  float4* pv4 = lds4 + ((npu + nbp + 3) >> 2);
  float2* AD = (float2*)pv4;  // cosine: {sg, sg * s}
  float* CC = (float*)pv4;    // L2: c = -+g / d
  float4* slab = pv4;
  const int Ppb = bd.npos | 1;
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);
  const Rsrc gxb = make_rsrc((char*)gx + (long long)b * g.gB * ES, (long long)g.C * P * ES);
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  // ---- A1: per-pair values of the rows ya .. yb-1, every tap; norm factors of every useful padded position ----------
  // Every load of a batch is issued before the first value is used (16-byte pieces of grad_out / out, the saved norms),
  // then the x chunk is requested, so that the pair arithmetic runs while x streams in.
  TileStage<R, BF, NHWC, kBwdKB, kBwdKR, kBwdKN> st;
  {
    const char* gob = (const char*)go + ((long long)b * N * P + (long long)ya * W) * ES;
    const char* outb = (const char*)out + ((long long)b * N * P + (long long)ya * W) * ES;
    // (the sign convention as arithmetic: a select on a wave-uniform flag becomes a branch per value)
    const float sa = g.osa, sb = -g.osa * g.osb;   // s = osa * (out - osb): out = osa * s + osb with osa = +-1
    auto put = [&](int i, float gc, float oc) {
      if (M == NFP_COSINE) {
        const float s = fmaf(sa, oc, sb);
        const float sg = sa * gc;
        AD[i] = make_float2(sg, sg * s);
      } else {
        CC[i] = dist_coef(g, gc, oc);
      }
    };
    // saved norms of the useful padded positions (ring positions read their fold source)
    constexpr int KS = 2;   // (a band of >= 3 rows has at most 2 T useful positions; more: the loop behind)
    float nrmv[KS];
    // (every load unconditional, on an index clamped into range: a load under a branch, or one that overwrites an
    // initialised register, makes hipcc wait for it where the paths join.  Issue order = the order the data is needed in:
    // pair values, the x chunk, the norms — loads retire in order.)
    // (DotProduct has no saved norms: its loads read the output map instead — in bounds, unused — rather than sit
    // under a branch)
    const float* sv = g.unit ? (const float*)out : saved;
    auto norm_loads = [&]() {
      if (M == NFP_COSINE) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const int v = min(t + k * T, npu - 1), vy = fdivi(v, Wu), vx = v - vy * Wu;
          const int sy = fo.y(bd.y0 - R + vy, H), sx = fo.x(vx - R, W);
          nrmv[k] = sv[(long long)b * P + max(sy, 0) * W + max(sx, 0)];
        }
      }
    };
    auto norms_put = [&](int v, float nrm_raw) {
      const int vy = fdivi(v, Wu), vx = v - vy * Wu;
      const int sy = fo.y(bd.y0 - R + vy, H), sx = fo.x(vx - R, W);
      const float nrm = (sy | sx) < 0 ? 0.f : nrm_raw;
      const float ip = unit_or(g, __builtin_amdgcn_rcpf(fmaxf(nrm, g.eps)));   // (DotProduct: no norm factors, no diagonal)
      ipn[v] = GFC ? nrm : ip;                                                  // (GFC: the norm itself — nfp_common.h::cross_f)
      const int yl = vy - R, xl = vx - R;
      if (yl >= 0 && yl < bd.y1 - bd.y0 && xl >= 0 && xl < W)
        dfn[yl * W + xl] = nrm > 0.f ? -(GFC ? 1.f : g.nuf * ip) * __builtin_amdgcn_rcpf(nrm) : 0.f;
    };
    constexpr int VP = BF ? 8 : 4;  // values per 16-byte piece
    bool x_asked = false;
    if (((P | W) & (VP - 1)) == 0) {
      const int nseg = npA / VP, tot = N * nseg;
      for (int base = 0; base < tot; base += kTileKA * T) {
        uint4 gq[kTileKA], oq[kTileKA];
#pragma unroll
        for (int k = 0; k < kTileKA; ++k) {
          const int i = min(base + t + k * T, tot - 1), n = fdivi(i, nseg), q = i - n * nseg;
          const long long src = ((long long)n * P + (long long)q * VP) * ES;
          if constexpr (!POOL) gq[k] = *(const uint4*)(gob + src);
          oq[k] = *(const uint4*)(outb + src);
        }
        if (!x_asked) {
          __builtin_amdgcn_sched_barrier(0);
          st.issue(g, fo, bd, xb, Ppb, cb0, min(g.Cc, cb1 - cb0) >> 2, t, T);
          __builtin_amdgcn_sched_barrier(0);
          x_asked = true;
          norm_loads();
          __builtin_amdgcn_sched_barrier(0);
          NFP_STAMP(7);
        }
#pragma unroll
        for (int k = 0; k < kTileKA; ++k) {
          const int i = base + t + k * T;
          if (i < tot) {
            const int n = fdivi(i, nseg), q = i - n * nseg;
            const float gp = POOL ? gnfpm[(long long)b * N + n] * g.invP : 0.f;
            float gcv[VP], ocv[VP];
#pragma unroll
            for (int e = 0; e < VP; ++e) {
              const uint32_t gw = POOL ? 0u : ((const uint32_t*)&gq[k])[BF ? e >> 1 : e], ow = ((const uint32_t*)&oq[k])[BF ? e >> 1 : e];
              gcv[e] = POOL ? gp : (BF ? __uint_as_float(e & 1 ? gw & 0xFFFF0000u : gw << 16) : __uint_as_float(gw));
              ocv[e] = BF ? __uint_as_float(e & 1 ? ow & 0xFFFF0000u : ow << 16) : __uint_as_float(ow);
            }
            const int i0 = n * npA + q * VP;   // (a multiple of 4: 16-byte LDS writes, two cosine pairs / four L2 values each)
            if (M == NFP_COSINE) {
#pragma unroll
              for (int e = 0; e < VP; e += 2) {
                const float s0 = fmaf(sa, ocv[e], sb), g0 = sa * gcv[e], s1 = fmaf(sa, ocv[e + 1], sb), g1 = sa * gcv[e + 1];
                *(float4*)(AD + i0 + e) = make_float4(g0, g0 * s0, g1, g1 * s1);
              }
            } else {
#pragma unroll
              for (int e = 0; e < VP; e += 4) {
                float c4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c4[u] = dist_coef(g, gcv[e + u], ocv[e + u]);
                *(float4*)(CC + i0 + e) = make_float4(c4[0], c4[1], c4[2], c4[3]);
              }
            }
          }
        }
      }
    } else {
      st.issue(g, fo, bd, xb, Ppb, cb0, min(g.Cc, cb1 - cb0) >> 2, t, T);
      x_asked = true;
      norm_loads();
      for (int i = t; i < N * npA; i += T) {
        const int n = fdivi(i, npA), l = i - n * npA;
        const long long src = (long long)n * P + l;
        const float gc = POOL ? gnfpm[(long long)b * N + n] * g.invP : ldx(gob, src, BF ? NFP_BF16 : NFP_F32);
        put(i, gc, ldx(outb, src, BF ? NFP_BF16 : NFP_F32));
      }
    }
    NFP_STAMP(8);
    if (M == NFP_COSINE) {
#pragma unroll
      for (int k = 0; k < KS; ++k)
        if (t + k * T < npu) norms_put(t + k * T, nrmv[k]);
      for (int v = t + KS * T; v < npu; v += T) {   // (bands with more than KS * T useful positions)
        const int vy = fdivi(v, Wu), vx = v - vy * Wu;
        const int sy = fo.y(bd.y0 - R + vy, H), sx = fo.x(vx - R, W);
        norms_put(v, sv[(long long)b * P + max(sy, 0) * W + max(sx, 0)]);
      }
    }
  }
  NFP_STAMP(1);
  __syncthreads();
  NFP_STAMP(2);

  const float dneg = g.diff ? -1.f : 0.f;   // L2: cross weight = -c with the difference weights, 0 with the 'Norm' quirk
  // ---- A2: the window weights of this thread's pixel, in registers (fixed order; bitwise reproducible) -----------------
  const int gl = fdivi(t, nbp), lp = t - gl * nbp;
  const bool active = gl < g.G;
  const int yl = fdivi(lp, W), xx = lp - yl * W, y = bd.y0 + yl;
  const int pos = (yl + R) * Wp + kXL + xx, sp = swz(pos);
  const int p = y * W + xx;
  float w[K2];
  auto crossw = [&](float fr, float ft) { return GFC ? cross_f(g, fr, ft) : fr * ft; };
  auto diagw = [&](float fr, float ft) { return GFC ? diag_f(g, fr, ft) : 1.f; };
  if (active) {
    const int pv = (yl + R) * Wu + xx + R, lpA = (y - ya) * W + xx;
    const float ipr = M == NFP_COSINE ? ipn[pv] : 1.f;
    float Dsum = 0.f;
    // Pixels within R of the image border also collect the pairs whose neighbour is a RING position that folds onto them
    // (reflect / replicate).  By the pixel's own thread: a wavefront with a border pixel pays one trip of the loop below
    // per valid ring position (an edge pixel has one, a corner three) while the other wavefronts do the same for theirs;
    // as a pass of its own over the band's border pixels it kept one wavefront busy for 5.5k cycles with the rest of the
    // workgroup parked at a barrier.
    float wl[K2];   // what the ring adds to the pixel's window, summed in registers (one LDS update per slot at the end)
#pragma unroll
    for (int j = 0; j < K2; ++j) wl[j] = 0.f;
    // ring positions that fold onto this pixel: per axis the coordinate itself (bit 0) and up to 2R ring coordinates;
    // a bit mask of the valid (row, column) combinations, then one trip per VALID combination (an edge pixel has one,
    // a corner three) — not a loop over all (2R+1)^2 with a dozen scalar branches each
    constexpr int KK = 2 * R + 1;
    unsigned my = 1u, mx = 1u;
#pragma unroll
    for (int i = 1; i < KK; ++i) {
      my |= (fo.y(i <= R ? -i : H - 1 + (i - R), H) == y ? 1u : 0u) << i;
      mx |= (fo.x(i <= R ? -i : W - 1 + (i - R), W) == xx ? 1u : 0u) << i;
    }
    unsigned mask = 0u;
#pragma unroll
    for (int c = 1; c < KK * KK; ++c) mask |= (((my >> (c / KK)) & (mx >> (c % KK)) & 1u)) << c;
    while (mask != 0u) {
      const int c = __builtin_ctz(mask);
      mask &= mask - 1u;
      const int iy = fdivi(c, KK), ix = c - iy * KK;
      const int uy_ = iy == 0 ? y : (iy <= R ? -iy : H - 1 + (iy - R));
      const int ux_ = ix == 0 ? xx : (ix <= R ? -ix : W - 1 + (ix - R));
      {
        // the N taps of this ring position at once: every LDS read first, then the sums
        float2 qv[N];
        float iq[N];
        int jv[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
          int dy, dx;
          tap_offset<R>(n, dy, dx);
          const int py = uy_ - dy, px = ux_ - dx, ry = py - y, rx = px - xx;
          const bool ok = py >= 0 && py < H && px >= 0 && px < W && ry >= -R && ry <= R && rx >= -R && rx <= R && py >= ya && py < yb;
          const int idx = ok ? n * npA + (py - ya) * W + px : 0;
          jv[n] = ok ? (ry + R) * K + rx + R : -1;
          if (M == NFP_COSINE) {
            qv[n] = AD[idx];
            iq[n] = ipn[ok ? (py - bd.y0 + R) * Wu + px + R : 0];
          } else {
            qv[n] = make_float2(CC[idx], 0.f);
            iq[n] = 0.f;
          }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const bool ok = jv[n] >= 0;
          float add;
          if (M == NFP_COSINE) {
            add = ok ? crossw(ipr, iq[n]) * qv[n].x : 0.f;
            Dsum += ok ? qv[n].y * diagw(ipr, iq[n]) : 0.f;
          } else {
            add = ok ? dneg * qv[n].x : 0.f;
            Dsum += ok ? qv[n].x : 0.f;
          }
#pragma unroll
          for (int j = 0; j < K2; ++j) wl[j] += j == jv[n] ? add : 0.f;
        }
      }
    }
    // every LDS read of the pixel first, then the sums
    float2 v1[N], v2[N];
    float ipq[N];
    bool inb[N];
    bool iny[K], inx[K];
#pragma unroll
    for (int d = 0; d < K; ++d) {
      iny[d] = y + d - R >= 0 && y + d - R < H;
      inx[d] = xx + d - R >= 0 && xx + d - R < W;
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1, dy = j / K - R, dx = j % K - R, opp = N - 1 - n;
      inb[n] = iny[dy + R] && inx[dx + R];
      const int i1 = n * npA + lpA, i2 = inb[n] ? opp * npA + lpA + dy * W + dx : i1;
      if (M == NFP_COSINE) {
        v1[n] = AD[i1];
        v2[n] = AD[i2];
        ipq[n] = ipn[pv + dy * Wu + dx];
      } else {
        v1[n] = make_float2(CC[i1], 0.f);
        v2[n] = make_float2(CC[i2], 0.f);
        ipq[n] = 0.f;
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1;
      if (M == NFP_COSINE) {
        const float S = v1[n].x + (inb[n] ? v2[n].x : 0.f);
        Dsum += (v1[n].y + (inb[n] ? v2[n].y : 0.f)) * diagw(ipr, ipq[n]);
        w[j] = fmaf(crossw(ipr, ipq[n]), S, wl[j]);
      } else {
        const float c1 = v1[n].x, c2 = inb[n] ? v2[n].x : 0.f;
        w[j] = fmaf(dneg, c1 + c2, wl[j]);
        Dsum += fmaf(-dneg, c1, c2);  // 'Norm' quirk (nfp.py:74 vs 85): only the pair's NEIGHBOUR is pulled
      }
    }
    w[K2 / 2] = fmaf(M == NFP_COSINE ? dfn[lp] : 1.f, Dsum, wl[K2 / 2]);
  } else {
#pragma unroll
    for (int j = 0; j < K2; ++j) w[j] = 0.f;
  }
  NFP_STAMP(9);
  __syncthreads();  // the pair values are dead: their LDS becomes the x slab
  NFP_STAMP(3);

  // ---- B: one pass over the channel block ---------------------------------------------------------------------------
  int off[K2];
#pragma unroll
  for (int j = 0; j < K2; ++j) off[j] = swz(pos + (j / K - R) * Wp + (j % K - R)) - sp;
  for (int c0 = cb0; c0 < cb1; c0 += g.Cc) {
    const int ncq = min(g.Cc, cb1 - c0) >> 2;
    if (c0 > cb0) __syncthreads();  // previous chunk fully consumed
    st.commit(slab);
    __syncthreads();
    if (c0 == cb0) NFP_STAMP(4);
    if (c0 + g.Cc < cb1) st.issue(g, fo, bd, xb, Ppb, c0 + g.Cc, min(g.Cc, cb1 - c0 - g.Cc) >> 2, t, T);
    if (active) {
      for (int cq = gl; cq < ncq; cq += g.G) {
        const float4* row = slab + cq * Ppb + sp;
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (POOL) {
          const float4 gg = *(const float4*)(ggap + (long long)b * g.C + c0 + 4 * cq);
          r4 = make_float4(gg.x * g.invP, gg.y * g.invP, gg.z * g.invP, gg.w * g.invP);
        }
#pragma unroll
        for (int j = 0; j < K2; ++j) {
          const float4 q = row[off[j]];
          r4.x = fmaf(w[j], q.x, r4.x);
          r4.y = fmaf(w[j], q.y, r4.y);
          r4.z = fmaf(w[j], q.z, r4.z);
          r4.w = fmaf(w[j], q.w, r4.w);
        }
        if constexpr (NHWC) {
          store_px4<BF>(gxb, p * g.C + c0 + 4 * cq, 0, r4);
        } else {
          const int e = (c0 + 4 * cq) * P + p;
          store_1<BF>(gxb, e, 0, r4.x);
          store_1<BF>(gxb, e, P, r4.y);
          store_1<BF>(gxb, e, 2 * P, r4.z);
          store_1<BF>(gxb, e, 3 * P, r4.w);
        }
      }
    }
    if (c0 == cb0) NFP_STAMP(5);
  }
  NFP_STAMP(6);
}

}  // namespace nfp
