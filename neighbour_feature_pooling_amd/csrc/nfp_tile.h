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

constexpr int kTileKB = 3;  // NCHW staging: 4-pixel x 4-channel blocks per thread per chunk
constexpr int kTileKR = 2;  // NCHW staging: ring slots per thread per chunk
constexpr int kTileKN = 8;  // channels-last staging: slots per thread per chunk

struct TileGeo {  // by value in kernarg
  int rb, nb;     // rows per band, bands per image
  int Wp;         // padded row length W + 2R
  int S;          // backward: channel blocks per (image, band)
};

// workgroup id -> (image, item of the image): ids i, i + 8, i + 16, ... share an XCD, so a group of 8 images is dealt one
// image per XCD and all `per` items (bands x channel blocks) of an image follow each other on it.  Bijective for any B
// (the last group may hold fewer than 8 images; its placement is then only partly XCD-aligned: speed, not correctness).
__device__ __forceinline__ void tile_ids(int id, int B, int per, int& b, int& item) {
  const int grp = id / (8 * per), l = id - grp * 8 * per;   // (once per kernel, wave-uniform: plain integer division)
  const int m = min(8, B - 8 * grp);
  item = l / m;
  b = 8 * grp + l - item * m;
}

// The band's padded geometry.
template <int R>
struct TileBand {
  int y0, y1, rows, Wp, npos, nbp;  // owned rows [y0, y1); staged padded rows; positions = rows * Wp; band pixels
  __device__ __forceinline__ TileBand(const KP& g, const TileGeo& tg, int band) {
    y0 = band * tg.rb;
    y1 = min(g.H, y0 + tg.rb);
    rows = y1 - y0 + 2 * R;
    Wp = tg.Wp;
    npos = rows * Wp;
    nbp = (y1 - y0) * g.W;
  }
};

// v or zeros, by component (a ternary between two float4 LVALUES selects an address and forces both into scratch memory)
__device__ __forceinline__ float4 keep_if(bool in, float x, float y, float z, float w) {
  return make_float4(in ? x : 0.f, in ? y : 0.f, in ? z : 0.f, in ? w : 0.f);
}

// Staging of one channel chunk of the padded band: float4[cq][Ppb], slot swz(u) of padded position u = yy * Wp + xx.
// All loads of a chunk are issued back to back into registers (indices clamped onto valid items, nothing conditional
// around a load) and committed to LDS later, so that arithmetic can run under their latency.
template <int R, bool BF, bool NHWC>
struct TileStage {
  float4 blk[NHWC ? 1 : kTileKB][4];
  float4 ring[NHWC ? 1 : kTileKR];
  float4 nv[NHWC ? kTileKN : 1];

  __device__ __forceinline__ void issue(const KP& g, const TileBand<R>& bd, Rsrc xb, int c0, int ncq, int t, int T) {
    const int W = g.W, H = g.H, P = g.P;
    if constexpr (NHWC) {
      const int items = bd.npos * ncq;
#pragma unroll
      for (int k = 0; k < kTileKN; ++k) {
        const int i = min(t + k * T, items - 1);
        const int u = fdivi(i, ncq), cq = i - u * ncq;
        const int uy = fdivi(u, bd.Wp), ux = u - uy * bd.Wp;
        const int sy = max(map_index(bd.y0 - R + uy, H, g.mode), 0), sx = max(map_index(ux - R, W, g.mode), 0);
        nv[k] = load_px4<BF>(xb, (sy * W + sx) * g.C + c0 + 4 * cq, 0);
      }
    } else {
      const int nbr = (W + 3) >> 2, per = bd.rows * nbr, nblk = ncq * per;
#pragma unroll
      for (int r = 0; r < kTileKB; ++r) {
        const int i = min(t + r * T, nblk - 1);
        const int cq = fdivi(i, per), rem = i - cq * per, yy = fdivi(rem, nbr), bq = rem - yy * nbr;
        const int sy = max(map_index(bd.y0 - R + yy, H, g.mode), 0);
        const int e = (c0 + 4 * cq) * P + sy * W + min(4 * bq, W - 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) blk[r][j] = load_px4<BF>(xb, e, j * P);
      }
      const int pr = bd.rows * 2 * R, nring = ncq * pr;
#pragma unroll
      for (int r = 0; r < kTileKR; ++r) {
        const int i = min(t + r * T, nring - 1);
        const int cq = fdivi(i, pr), rem = i - cq * pr, yy = fdivi(rem, 2 * R), k = rem - yy * 2 * R;
        const int sy = max(map_index(bd.y0 - R + yy, H, g.mode), 0);
        const int sx = max(map_index((k < R ? k : W + k) - R, W, g.mode), 0);
        const int e = (c0 + 4 * cq) * P + sy * W + sx;
        ring[r] = make_float4(load_1<BF>(xb, e, 0), load_1<BF>(xb, e, P), load_1<BF>(xb, e, 2 * P), load_1<BF>(xb, e, 3 * P));
      }
    }
  }

  __device__ __forceinline__ void commit(const KP& g, const TileBand<R>& bd, float4* slab, int Ppb, int ncq, int t, int T) const {
    const int W = g.W, H = g.H;
    if constexpr (NHWC) {
      const int items = bd.npos * ncq;
#pragma unroll
      for (int k = 0; k < kTileKN; ++k) {
        const int i = t + k * T;
        if (i < items) {
          const int u = fdivi(i, ncq), cq = i - u * ncq;
          const int uy = fdivi(u, bd.Wp), ux = u - uy * bd.Wp;
          const bool in = map_index(bd.y0 - R + uy, H, g.mode) >= 0 && map_index(ux - R, W, g.mode) >= 0;
          slab[cq * Ppb + swz(u)] = keep_if(in, nv[k].x, nv[k].y, nv[k].z, nv[k].w);
        }
      }
    } else {
      const int nbr = (W + 3) >> 2, per = bd.rows * nbr, nblk = ncq * per;
#pragma unroll
      for (int r = 0; r < kTileKB; ++r) {
        const int i = t + r * T;
        if (i < nblk) {
          const int cq = fdivi(i, per), rem = i - cq * per, yy = fdivi(rem, nbr), bq = rem - yy * nbr;
          const bool in = map_index(bd.y0 - R + yy, H, g.mode) >= 0;
          const int u0 = yy * bd.Wp + R + min(4 * bq, W - 4);
          float4* d = slab + cq * Ppb;
          d[swz(u0)] = keep_if(in, blk[r][0].x, blk[r][1].x, blk[r][2].x, blk[r][3].x);
          d[swz(u0 + 1)] = keep_if(in, blk[r][0].y, blk[r][1].y, blk[r][2].y, blk[r][3].y);
          d[swz(u0 + 2)] = keep_if(in, blk[r][0].z, blk[r][1].z, blk[r][2].z, blk[r][3].z);
          d[swz(u0 + 3)] = keep_if(in, blk[r][0].w, blk[r][1].w, blk[r][2].w, blk[r][3].w);
        }
      }
      const int pr = bd.rows * 2 * R, nring = ncq * pr;
#pragma unroll
      for (int r = 0; r < kTileKR; ++r) {
        const int i = t + r * T;
        if (i < nring) {
          const int cq = fdivi(i, pr), rem = i - cq * pr, yy = fdivi(rem, 2 * R), k = rem - yy * 2 * R;
          const int xx = k < R ? k : W + k;
          const bool in = map_index(bd.y0 - R + yy, H, g.mode) >= 0 && map_index(xx - R, W, g.mode) >= 0;
          slab[cq * Ppb + swz(yy * bd.Wp + xx)] = keep_if(in, ring[r].x, ring[r].y, ring[r].z, ring[r].w);
        }
      }
    }
  }
};

// sum over the 64 lanes of a wavefront, in every lane's... lane 63 (fixed DPP tree: group_sum covers 32, then the row
// broadcast into the upper half)
__device__ __forceinline__ float wave_sum(float v) {
  v = group_sum(v, 32);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast31 into rows 2, 3
  return v;  // valid in lane 63
}

// ---- forward ----------------------------------------------------------------------------------------------------------
// POOL: the fused tail of models/NFP_Pooling.py:27-31 for large maps: besides the maps this band's share of the two
// pooled sums goes to part[(b * nb + band)][C + N] (sums, not means); pool_fold joins the bands in a fixed order.
template <int R, int M, bool BF, bool NHWC, bool POOL = false>
__global__ void __launch_bounds__(1024) fwd_tile(const KP g, const TileGeo tg, const void* __restrict__ x,
                                                 void* __restrict__ out, float* __restrict__ saved,
                                                 float* __restrict__ part) {
  constexpr int N = Win<R>::N, NF = Win<R>::NF;
  constexpr int ES = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  const int t = threadIdx.x, T = blockDim.x;
  int b, band;
  tile_ids(blockIdx.x, g.B, tg.nb, b, band);
  const TileBand<R> bd(g, tg, band);
  const int W = g.W, P = g.P, Wp = bd.Wp, npos = bd.npos, nbp = bd.nbp;
  const int G = g.G, lg = g.Tc;
  const int Ppb = band_row_slots((npos + 3) & ~3, lg);
  float4* slab = lds4;
  float* Tt = (float*)(lds4 + (g.Cc >> 2) * Ppb);  // [NF + 1][npos]: pair sums per direction, then |x|^2
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);

  TileStage<R, BF, NHWC> st;
  st.issue(g, bd, xb, 0, min(g.Cc, g.C) >> 2, t, T);
  __builtin_amdgcn_sched_barrier(0);

  // channel sums: thread t = position * G + group (the groups of a position are adjacent lanes: joined by DPP)
  const int gl = t & (G - 1), uc = t >> lg;
  const bool active = uc < npos;
  const int u = min(uc, npos - 1), uy = fdivi(u, Wp), ux = u - uy * Wp;
  const int su = swz(u);
  int off[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) {
    int dy, dx;
    fdir<R>(d, dy, dx);
    const bool ok = ux + dx >= 0 && ux + dx < Wp && uy + dy < bd.rows;
    off[d] = ok ? swz(u + dy * Wp + dx) - su : 0;
  }
  float acc[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = 0.f;
  float nrm = 0.f;

  for (int c0 = 0; c0 < g.C; c0 += g.Cc) {
    const int ncq = min(g.Cc, g.C - c0) >> 2;
    if (c0 > 0) {
      __syncthreads();  // previous chunk fully consumed
      st.issue(g, bd, xb, c0, ncq, t, T);
    }
    st.commit(g, bd, slab, Ppb, ncq, t, T);
    __syncthreads();
    if constexpr (POOL) {
      // this band's share of sum over pixels of x[c]: wavefront w takes channel quads w, w + nw, ...; lanes stride over
      // the band's pixels; fixed DPP tree; one writer per channel
      const int lane = t & 63, wv = t >> 6, nw = T >> 6;
      for (int cq = wv; cq < ncq; cq += nw) {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int lp = lane; lp < nbp; lp += 64) {
          const int yl = fdivi(lp, W), xl = lp - yl * W;
          const float4 v = slab[cq * Ppb + swz((yl + R) * Wp + xl + R)];
          s4.x += v.x;
          s4.y += v.y;
          s4.z += v.z;
          s4.w += v.w;
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
  // channel groups joined inside the wavefront; one lane per position publishes the sums
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = group_sum(acc[d], G);
  nrm = group_sum(nrm, G);
  if (active && gl == G - 1) {
#pragma unroll
    for (int d = 0; d < NF; ++d) Tt[d * npos + u] = acc[d];
    Tt[NF * npos + u] = nrm;
  }
  __syncthreads();
  // outputs of the band's rows: thread (pixel, n = glf, glf + Gn, ...), lanes along pixels (coalesced stores)
  const float* n2 = Tt + NF * npos;
  const int glf = fdivi(t, nbp), lpf = t - glf * nbp, Gn = fdivi(T, nbp);
  float* vm = Tt + (NF + 1) * npos;  // (POOL) [N][nbp]: the band's map values, for the pooled sums
  if (glf < Gn) {
    const int yl = fdivi(lpf, W), xl = lpf - yl * W, pos = (yl + R) * Wp + xl + R;
    const int p = (bd.y0 + yl) * W + xl;
    void* ob = (char*)out + (long long)b * N * P * ES;
    const float n2p = n2[pos];
    const float ip = inv_norm(n2p, g.inv_eps);
    for (int n = glf; n < N; n += Gn) {
      int dy, dx;
      tap_offset<R>(n, dy, dx);
      const bool fwd = dy > 0 || (dy == 0 && dx > 0);
      const int fi = fwd ? fidx<R>(dy, dx) : fidx<R>(-dy, -dx);
      const int qpos = pos + dy * Wp + dx;
      const float pairv = Tt[fi * npos + (fwd ? pos : qpos)];
      const float n2q = n2[qpos];
      float v;
      if (M == NFP_COSINE) {
        const float s = pairv * ip * inv_norm(n2q, g.inv_eps);
        v = g.similarity ? s : 1.f - s;
      } else {
        const float dd = __builtin_amdgcn_sqrtf(g.diff ? pairv : n2q);  // 'Norm' quirk (nfp.py:74 vs 85): |neighbour|
        v = g.similarity ? -dd : dd;
      }
      stx(ob, n * P + p, v, BF ? NFP_BF16 : NFP_F32);
      if constexpr (POOL) vm[n * nbp + lpf] = v;
    }
    if (M == NFP_COSINE && saved != nullptr && glf == 0) saved[(long long)b * P + p] = __builtin_amdgcn_sqrtf(n2p);
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
template <int R, int M, bool BF, bool NHWC, bool POOL = false>
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
  const int band = fdivi(item, tg.S), cblk = item - band * tg.S;
  const TileBand<R> bd(g, tg, band);
  const int W = g.W, H = g.H, P = g.P, Wp = bd.Wp, npos = bd.npos, nbp = bd.nbp;
  const int ya = max(0, bd.y0 - R), yb = min(H, bd.y1 + R), npA = (yb - ya) * W;  // rows whose pairs touch the band
  const int cb0 = cblk * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  // LDS: Wt [nbp][K2] | ipn [npos] | dfn [nbp] | pair values [N][npA] — and the x slab over the pair values
  float* Wt = (float*)lds4;
  float* ipn = Wt + nbp * K2;
  float* dfn = ipn + npos;
  float4* pv4 = lds4 + ((nbp * K2 + npos + nbp + 3) >> 2);
  float2* AD = (float2*)pv4;  // cosine: {sg, sg * s}
  float* CC = (float*)pv4;    // L2: c = -+g / d
  float4* slab = pv4;
  const int Ppb = ((npos + 3) & ~3) | 1;
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);
  const Rsrc gxb = make_rsrc((char*)gx + (long long)b * g.gB * ES, (long long)g.C * P * ES);

  TileStage<R, BF, NHWC> st;
  st.issue(g, bd, xb, cb0, min(g.Cc, cb1 - cb0) >> 2, t, T);
  __builtin_amdgcn_sched_barrier(0);

  // ---- A1: per-pair values of the rows ya .. yb-1, every tap; norm factors of every padded position ---------------
  {
    const char* gob = (const char*)go + ((long long)b * N * P + (long long)ya * W) * ES;
    const char* outb = (const char*)out + ((long long)b * N * P + (long long)ya * W) * ES;
    auto put = [&](int i, float gc, float oc) {
      if (M == NFP_COSINE) {
        const float s = g.similarity ? oc : 1.f - oc;
        const float sg = g.similarity ? gc : -gc;
        AD[i] = make_float2(sg, sg * s);
      } else {
        const float d = fabsf(oc);
        CC[i] = d == 0.f ? 0.f : (g.similarity ? -gc : gc) * __builtin_amdgcn_rcpf(d);
      }
    };
    constexpr int VP = BF ? 8 : 4;  // values per 16-byte piece
    if (((P | W) & (VP - 1)) == 0) {
      const int nseg = npA / VP, tot = N * nseg;
      for (int i0 = t; i0 < tot; i0 += T) {
        const int n = fdivi(i0, nseg), v = i0 - n * nseg;
        const long long src = ((long long)n * P + (long long)v * VP) * ES;
        uint4 gq = make_uint4(0, 0, 0, 0);
        if constexpr (!POOL) gq = *(const uint4*)(gob + src);
        const uint4 oq = *(const uint4*)(outb + src);
        const float gp = POOL ? gnfpm[(long long)b * N + n] * g.invP : 0.f;
#pragma unroll
        for (int k = 0; k < VP; ++k) {
          const uint32_t gw = ((const uint32_t*)&gq)[BF ? k >> 1 : k], ow = ((const uint32_t*)&oq)[BF ? k >> 1 : k];
          const float gc = POOL ? gp : (BF ? __uint_as_float(k & 1 ? gw & 0xFFFF0000u : gw << 16) : __uint_as_float(gw));
          const float oc = BF ? __uint_as_float(k & 1 ? ow & 0xFFFF0000u : ow << 16) : __uint_as_float(ow);
          put(n * npA + v * VP + k, gc, oc);
        }
      }
    } else {
      for (int i = t; i < N * npA; i += T) {
        const int n = fdivi(i, npA), l = i - n * npA;
        const long long src = (long long)n * P + l;
        const float gc = POOL ? gnfpm[(long long)b * N + n] * g.invP : ldx(gob, src, BF ? NFP_BF16 : NFP_F32);
        put(i, gc, ldx(outb, src, BF ? NFP_BF16 : NFP_F32));
      }
    }
    if (M == NFP_COSINE) {
      for (int u = t; u < npos; u += T) {
        const int uy = fdivi(u, Wp), ux = u - uy * Wp;
        const int sy = map_index(bd.y0 - R + uy, H, g.mode), sx = map_index(ux - R, W, g.mode);
        const float nrm = (sy < 0 || sx < 0) ? 0.f : saved[(long long)b * P + sy * W + sx];
        const float ip = __builtin_amdgcn_rcpf(fmaxf(nrm, g.eps));
        ipn[u] = ip;
        const int yl = uy - R, xl = ux - R;
        if (yl >= 0 && yl < bd.y1 - bd.y0 && xl >= 0 && xl < W)
          dfn[yl * W + xl] = nrm > 0.f ? -ip * __builtin_amdgcn_rcpf(nrm) : 0.f;
      }
    }
  }
  __syncthreads();

  // ---- A2: window weights of every band pixel, by the pixel's own thread (fixed order, one writer per row) ------------
  for (int lp = t; lp < nbp; lp += T) {
    const int yl = fdivi(lp, W), xx = lp - yl * W, y = bd.y0 + yl;
    const int pos = (yl + R) * Wp + xx + R, lpA = (y - ya) * W + xx;
    const float ipr = M == NFP_COSINE ? ipn[pos] : 1.f;
    float Dsum = 0.f, Wc = 0.f;
    float* wrow = Wt + lp * K2;
#pragma unroll
    for (int j = 0; j < K2; ++j) {
      if (j == K2 / 2) continue;
      const int dy = j / K - R, dx = j % K - R, n = j < K2 / 2 ? j : j - 1, opp = N - 1 - n;
      const bool in = y + dy >= 0 && y + dy < H && xx + dx >= 0 && xx + dx < W;
      const int i1 = n * npA + lpA, i2 = in ? opp * npA + lpA + dy * W + dx : i1;
      if (M == NFP_COSINE) {
        const float2 v1 = AD[i1], v2 = AD[i2];
        const float S = v1.x + (in ? v2.x : 0.f);
        Dsum += v1.y + (in ? v2.y : 0.f);
        wrow[j] = ipr * ipn[pos + dy * Wp + dx] * S;
      } else {
        const float c1 = CC[i1], c2 = in ? CC[i2] : 0.f;
        wrow[j] = g.diff ? -(c1 + c2) : 0.f;
        Dsum += g.diff ? c1 + c2 : c2;  // 'Norm' quirk (nfp.py:74 vs 85): only the pair's NEIGHBOUR is pulled
      }
    }
    // pairs whose neighbour is a ring position that folds onto this pixel (reflect / replicate near the border)
    if (g.mode != NFP_PAD_ZEROS) {
#pragma unroll 1
      for (int iy = 0; iy <= 2 * R; ++iy) {
        const int uy_ = iy == 0 ? y : (iy <= R ? -iy : H - 1 + (iy - R));
        if (iy != 0 && map_index(uy_, H, g.mode) != y) continue;
#pragma unroll 1
        for (int ix = 0; ix <= 2 * R; ++ix) {
          const int ux_ = ix == 0 ? xx : (ix <= R ? -ix : W - 1 + (ix - R));
          if ((iy == 0 && ix == 0) || (ix != 0 && map_index(ux_, W, g.mode) != xx)) continue;
#pragma unroll 1
          for (int n = 0; n < N; ++n) {
            int dy, dx;
            tap_offset<R>(n, dy, dx);
            const int py = uy_ - dy, px = ux_ - dx;
            if (py < 0 || py >= H || px < 0 || px >= W) continue;
            const int ry = py - y, rx = px - xx;
            if (ry < -R || ry > R || rx < -R || rx > R || py < ya || py >= yb) continue;  // (cannot happen: see header)
            const int idx = n * npA + (py - ya) * W + px, jj = (ry + R) * K + rx + R;
            float add;
            if (M == NFP_COSINE) {
              const float2 v = AD[idx];
              add = ipr * ipn[(py - bd.y0 + R) * Wp + px + R] * v.x;
              Dsum += v.y;
            } else {
              const float c = CC[idx];
              add = g.diff ? -c : 0.f;
              Dsum += c;
            }
            if (jj == K2 / 2)
              Wc += add;
            else
              wrow[jj] += add;
          }
        }
      }
    }
    wrow[K2 / 2] = fmaf(M == NFP_COSINE ? dfn[lp] : 1.f, Dsum, Wc);
  }
  __syncthreads();  // weights complete; the pair values are dead: their LDS becomes the x slab

  // ---- B: one pass over the channel block ---------------------------------------------------------------------------
  const int gl = fdivi(t, nbp), lp = t - gl * nbp;
  const bool active = gl < g.G;
  const int yl = fdivi(lp, W), xl = lp - yl * W, pos = (yl + R) * Wp + xl + R, sp = swz(pos);
  const int p = (bd.y0 + yl) * W + xl;
  float w[K2];
  int off[K2];
#pragma unroll
  for (int j = 0; j < K2; ++j) {
    off[j] = swz(pos + (j / K - R) * Wp + (j % K - R)) - sp;
    w[j] = active ? Wt[lp * K2 + j] : 0.f;
  }
  for (int c0 = cb0; c0 < cb1; c0 += g.Cc) {
    const int ncq = min(g.Cc, cb1 - c0) >> 2;
    if (c0 > cb0) __syncthreads();  // previous chunk fully consumed
    st.commit(g, bd, slab, Ppb, ncq, t, T);
    __syncthreads();
    if (c0 + g.Cc < cb1) st.issue(g, bd, xb, c0 + g.Cc, min(g.Cc, cb1 - c0 - g.Cc) >> 2, t, T);
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
  }
}

}  // namespace nfp
