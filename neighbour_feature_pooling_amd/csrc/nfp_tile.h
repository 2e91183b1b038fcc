// nfp_tile.h — the hot path for maps ABOVE 512 pixels: row-band kernels on a PADDED LDS slab, no tables.
//
// Who needs it: MobileNetV3_MultiStageNFP feeds NFP 112x112x16, 56x56x24 and 28x28x40 maps
// (models/texture_pooling.py:211-268), RESNET18_NFP_AT_LAYER 56x56x64 and 28x28x128 (models/resnet18.py:410-468).
// Round 2 served them with the any-geometry kernels of nfp_gather.h at 0.13-0.25 (forward) and 0.03-0.18 (backward) of
// the HBM roofline (profiles/r03_a_bigmaps_baseline_general_kernels.jsonl).  Same arithmetic as nfp_band.h /
// nfp_fast.h (nfp.py:141-159: cosine and L2 over "same" maps: stride 1, dilation 1, padding = R), different index
// scheme: the workspace tables of the small-map kernels grow with H*W and a 7x7 map is half border, a 112x112 map 3 %.
//
// A workgroup owns a BAND of rows [y0, y1) of one image and works on the PADDED band the reference's F.pad would build
// (nfp.py:42-58: reflect / replicate / zeros): rows y0-R .. y1+R-1, R ring columns left and right of every row, ring
// rows above / below the image.  ONE THREAD PER PADDED POSITION (times G channel groups): the workgroup is a
// (G, W + 2R, rows) block, so a thread's padded column and row are its hardware ids — no division anywhere — and the
// same thread stages its position (from the position's fold source: a ring position is a second copy of an image
// pixel), sums it, and writes its outputs.  In padded coordinates every tap of every position sits at a CONSTANT
// offset: no border cases, no index tables, and a zero-padded tap is a zero vector like any other.
//   forward   half stencil over padded positions (a pair {u, u+d} is summed once, by the thread of u, ring positions
//             included), then the thread of a band pixel looks its N pairs up; nothing is combined across workgroups;
//             an output element has one writer.
//   backward  gather form as in nfp_fast.h: per band pixel r the window weights W[r][j] (phase A), then
//             grad_x[c][r] = sum_j W[r][j] * xpad[c][r + d_j] in one pass over the slab (phase B).  Phase A is the same
//             code for every position: a position's slot j links its own pair of tap j and the pair of the opposite
//             tap of the position under it; positions that own no pairs (ring, outside the image) hold zeros.  A ring
//             position u is a copy of the image pixel r it folds onto: what its window collects belongs to r, and r's
//             thread fetches it — shifted by r - u — in a fixed order (no atomics, bitwise reproducible).
// The halo rows two bands share are re-read from L2: workgroup ids are mapped so that the bands of one image run on ONE
// XCD (ids are dealt round-robin over the 8 XCDs, each with its own L2).
//
// Why this shape (profiles/r03_u_tile_kernels_pmc_before_rewrite.csv): the first cut of these kernels (flat thread ids,
// 4-pixel staging blocks, swizzled slots) was bound by VALU issue — 446 (forward) / 1306 (backward) vector instructions
// per wavefront at [256,16,112,112], two thirds of all issue cycles, against ~150 / ~330 that the arithmetic needs: the
// rest was index arithmetic (divisions by the row length, folds, swizzles, per-tap border predicates).
#pragma once
#include <type_traits>

#include "nfp_band.h"

namespace nfp {

constexpr int kPoolSub = 1;  // fused pooling tail: a band's pixels in this many segments, each with its own row of partial sums
                             // (measured: 4 segments lose, 115 vs 100 us at [256,16,112,112] — every extra item is four DPP trees)
constexpr int kTileKQ = 4;   // channel quads a thread stages per chunk (16 registers): a chunk is 4 * G * kTileKQ channels

struct TileGeo {  // by value in kernarg
  int nb;         // bands per image: H = nb * hq + hr; band i owns hq rows, the first hr bands one more — rows
                  // [i hq + min(i, hr), (i + 1) hq + min(i + 1, hr)); every band at least R + 1 rows
  int rows;       // padded rows of the largest band = blockDim.z
  int Wu;         // padded row length W + 2R = blockDim.y
  int Ppb;        // slab slots per channel quad (>= rows * Wu; the padding spreads the channel groups over the LDS banks)
  int S;          // backward: channel blocks per (image, band)
  int hq, hr;     // H / nb, H % nb (host: exact for any H — round 3 divided band * H by nb through a float reciprocal,
                  // which is exact below 2^21 only: maps of ~5 400 rows and more got wrong band boundaries)
  int ldsw;       // 32-bit words of dynamic LDS of the launch (the -DNFP_LDS_POISON test build fills them at entry)
};

// Test build (-DNFP_LDS_POISON, tests/test_gpu_tile.py::test_row_band_kernels_with_poisoned_lds): every word of the
// workgroup's LDS is a signalling NaN before the kernel proper starts.  The row-band kernels let taps past the padded
// band read "whatever lies there" because such a value only reaches sums nobody looks up; on a fresh LDS allocation
// that is usually zeros or a previous workgroup's finite numbers, which would also hide a value that IS looked up.
__device__ __forceinline__ void lds_poison(void* lds, int words) {
#ifdef NFP_LDS_POISON
  uint32_t* w = (uint32_t*)lds;
  const int nth = blockDim.x * blockDim.y * blockDim.z, tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  for (int i = tid; i < words; i += nth) w[i] = 0x7FA00000u;
  __syncthreads();
#endif
}

// workgroup id -> (image, item of the image): ids i, i + 8, i + 16, ... share an XCD, so a group of 8 images is dealt one
// image per XCD and all `per` items (bands x channel blocks) of an image follow each other on it.  Bijective for any B
// (the last group may hold fewer than 8 images; its placement is then only partly XCD-aligned: speed, not correctness).
__device__ __forceinline__ void tile_ids(int id, int B, int per, int& b, int& item) {
  // (wave-uniform, once per workgroup; fdivi: four vector instructions where a scalar integer division takes ~40.  Exact
  // while quotient * 2^-22 stays below its 0.5 / divisor margin: the launchers keep a grid at or below kTileMaxGrid
  // workgroups and split larger batches into several launches)
  const int grp = __builtin_amdgcn_readfirstlane(fdivi(id, 8 * per)), l = id - grp * 8 * per;   // (back to scalar registers)
  const int m = min(8, B - 8 * grp);
  item = __builtin_amdgcn_readfirstlane(fdivi(l, m));
  b = 8 * grp + l - item * m;
}
constexpr int kTileMaxGrid = 1 << 18;

template <int R>
struct TileBand {
  int y0, y1, rows;  // owned rows [y0, y1); padded rows y0 - R .. y1 + R - 1
  __device__ __forceinline__ TileBand(const KP& g, const TileGeo& tg, int band) {
    y0 = __builtin_amdgcn_readfirstlane(band * tg.hq + min(band, tg.hr));
    y1 = __builtin_amdgcn_readfirstlane((band + 1) * tg.hq + min(band + 1, tg.hr));
    rows = y1 - y0 + 2 * R;
  }
};

// nn.Conv2d's padding_mode as ARITHMETIC: a coordinate t outside [0, n) maps to a * t + b with per-mode constants (reflect:
// -t / 2(n-1) - t; replicate: 0 / n-1; zeros: -1 = "reads 0").  nfp_common.h::map_index selects on the mode — a
// wave-uniform value, which hipcc turns into scalar BRANCHES, five per call.  The constants are chosen once per kernel.
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

// A buffer load whose offset lies beyond the resource's extent returns 0 and touches no memory: positions that read
// zeros (zero padding, rows past the band) OR this offset into theirs instead of a select per loaded value.  (tile_ok
// keeps every image below 0x7ffffff0 bytes.  An OR, not a ternary: hipcc turns "zero ? far : computed" into a branch
// around the computation and then no longer knows that the row stride inside it is wave-uniform — a waterfall loop per
// load.)
template <bool BF>
struct Oob {
  static constexpr int e = BF ? 0x3ffffff8 : 0x1ffffffc;   // element offset whose byte offset is 0x7ffffff0
};

// What a thread knows about its padded position.
template <int R>
struct TilePos {
  int gl, vx, vy, v;   // channel group; padded column, padded row; v = vy * Wu + vx
  int y, x;            // image coordinates (outside the image for ring positions)
  int src;             // the image pixel the position holds a copy of (clamped into the image when there is none)
  bool live;           // the row exists in this band (bands differ by a row; blockDim.z is the largest)
  bool real;           // an image pixel (not a ring position): owns pairs
  bool zero;           // reads zeros (zero padding; rows past the band)
  bool own;            // a pixel of the band's own rows: writes outputs
  int zf, zh;          // `zero` as an offset to OR in: Oob<false>::e / Oob<true>::e, or 0
  __device__ __forceinline__ TilePos(const KP& g, const TileGeo& tg, const TileBand<R>& bd, const Fold& fo) {
    gl = threadIdx.x;
    vx = threadIdx.y;
    vy = threadIdx.z;
    v = vy * tg.Wu + vx;
    y = bd.y0 - R + vy;
    x = vx - R;
    live = vy < bd.rows;
    const int sy = fo.y(y, g.H), sx = fo.x(x, g.W);
    real = live && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
    zero = !live || (sy | sx) < 0;
    src = min(max(sy, 0), g.H - 1) * g.W + max(sx, 0);
    own = real && vy >= R && vy < bd.rows - R;
    zf = zero ? Oob<false>::e : 0;
    zh = zero ? Oob<true>::e : 0;
  }
  template <bool BF>
  __device__ __forceinline__ int zoff() const { return BF ? zh : zf; }
};

// The thread's own position of a channel chunk: quads gl, gl + G, ... (the quads it will sum), every load issued back
// to back into registers, written to LDS later so that arithmetic can run under their latency.
// Channels-last, one thread per position (g.Tc >= 0): a lane reading 16 bytes of ITS pixel makes every load instruction touch
// 64 cache lines for 1 KB — four times the L1 time of the same bytes read densely ([256,64,56,56] forward 58 us against
// 37 us for NCHW).  Instead the lanes of a wavefront share the chunk of the wavefront's positions: Q = 2^Tc quads per
// chunk, Q adjacent lanes read 16 Q contiguous bytes; the pixel offset of somebody else's position comes through
// ds_bpermute.  (A workgroup's partial last wavefront keeps the one-lane-one-pixel form — a wave-uniform choice.)
// CST (bwd_tile's variant that also STORES grad_x that way, phase B) needs every lane of every wavefront in the scheme:
// CoopMap deals items i = lane + k * n over the n lanes a wavefront has.
// The slab's element — four channels of one position — in the map's STORAGE type (round 4).  Rounds 2-3 kept float4 slots
// whatever the storage: bf16 maps then took the same LDS bytes, LDS instructions and channels per chunk as float32 on half
// the HBM bytes, and ran no faster (profiles/r03_e_…: 0.26-0.36 of their roofline against 0.46-0.64).  A bf16 slot is 8
// bytes: half the LDS traffic, twice the channels per chunk at the same staging registers (KQ), products through
// v_dot2c_f32_bf16.
template <bool BF>
struct Quad {
  using T = float4;
  static constexpr int KQ = kTileKQ;        // quads a thread stages per chunk
  static __device__ __forceinline__ float4 get(const T& v) { return v; }
  static __device__ __forceinline__ T put(const float4& v) { return v; }
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <>
struct Quad<true> {
  using T = uint2;
  static constexpr int KQ = kTileKQ;   // (8 quads per chunk at the same staging registers spilled at the 64-register bound and lost 15-30 %: profiles/r04_l_…)
  static __device__ __forceinline__ float4 get(const T& v) {
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  }
  static __device__ __forceinline__ T put(const float4& v) { return make_uint2(f32_to_bf16x2(v.x, v.y), f32_to_bf16x2(v.z, v.w)); }
  static __device__ __forceinline__ T zero() { return make_uint2(0u, 0u); }
};
// four consecutive channels of one pixel (channels-last), raw
template <bool BF>
__device__ __forceinline__ typename Quad<BF>::T load_quad(Rsrc r, int e, int srow) {
  if constexpr (!BF) {
    return load_px4<false>(r, e, srow);
  } else {
    const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, e * 2, srow * 2, 0);
    return make_uint2(u.x, u.y);
  }
}
template <bool BF>
__device__ __forceinline__ void store_quad(Rsrc r, int e, int srow, const typename Quad<BF>::T& v) {
  if constexpr (!BF) {
    store_px4<false>(r, e, srow, v);
  } else {
    const u32x2 u = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(u, r, e * 2, srow * 2, NFP_BWD_STORE_AUX);
  }
}
// one position of four consecutive channel PLANES (NCHW), raw
template <bool BF>
__device__ __forceinline__ typename Quad<BF>::T load_quad_planes(Rsrc r, int e, int P) {
  if constexpr (!BF) {
    return make_float4(load_1<false>(r, e, 0), load_1<false>(r, e, P), load_1<false>(r, e, 2 * P), load_1<false>(r, e, 3 * P));
  } else {
    const uint32_t a = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, e * 2, 0, 0);
    const uint32_t b = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, e * 2, P * 2, 0);
    const uint32_t c = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, e * 2, 2 * P * 2, 0);
    const uint32_t d = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, e * 2, 3 * P * 2, 0);
    return make_uint2(a | (b << 16), c | (d << 16));
  }
}
// sum over the quad's four channels of a[c] * q[c] (+ acc): exact products, float32 sums
template <bool BF>
__device__ __forceinline__ float quad_dot(const typename Quad<BF>::T& a, const typename Quad<BF>::T& q, float acc) {
  if constexpr (!BF) {
    return fmaf(a.x, q.x, fmaf(a.y, q.y, fmaf(a.z, q.z, fmaf(a.w, q.w, acc))));
  } else {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.x), __builtin_bit_cast(bf2, q.x), acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.y), __builtin_bit_cast(bf2, q.y), acc, false);
  }
}

struct CoopMap {   // the wavefront's share of a chunk, as seen by one lane
  int lq, wb, n;   // log2 Q; the wavefront's first position; its lanes
  int lane;
  __device__ __forceinline__ CoopMap(const KP& g, int v, int npu) {
    lq = g.Tc;
    lane = __lane_id();
    wb = v - lane;
    n = min(64, npu - wb);
  }
  __device__ __forceinline__ void item(int k, int& psub, int& cq) const {
    const int i = lane + k * n;
    psub = i >> lq;
    cq = i & ((1 << lq) - 1);
  }
};
template <int R, bool BF, bool NHWC, bool CST>
struct TileStage {
  using QT = typename Quad<BF>::T;
  static constexpr int KQ = Quad<BF>::KQ;
  QT q[KQ];
  __device__ __forceinline__ void issue(const KP& g, const TilePos<R>& ps, Rsrc xb, int G, int c0, int ncq, int npu) {
    if constexpr (CST) {
      const CoopMap cm(g, ps.v, npu);
      const int eb = (ps.src * g.C) | ps.template zoff<BF>();
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        int psub, cq;
        cm.item(k, psub, cq);
        const int e = __builtin_amdgcn_ds_bpermute(psub << 2, eb) + 4 * cq;
        const bool ok = k < (1 << cm.lq) && cq < ncq;
        q[k] = load_quad<BF>(xb, ok ? e : Oob<BF>::e, c0);
      }
      return;
    }
    if constexpr (NHWC) {
      // (a workgroup's last wavefront may be partial: it keeps the one-lane-one-pixel form — a wave-uniform choice)
      if (g.Tc >= 0 && ps.v - (int)__lane_id() + 64 <= npu) {
        const int lq = g.Tc, lane = __lane_id();
        const int cq = lane & ((1 << lq) - 1), p0 = lane >> lq, step = 64 >> lq;
        const int eb = (ps.src * g.C) | ps.template zoff<BF>();
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const int psub = p0 + k * step;
          const int e = __builtin_amdgcn_ds_bpermute(psub << 2, eb) + 4 * cq;
          const bool ok = k < (1 << lq) && cq < ncq;
          q[k] = load_quad<BF>(xb, ok ? e : Oob<BF>::e, c0);
        }
        return;
      }
    }
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      if (k > 0 && k * G >= ncq) break;             // (a round no thread of the workgroup needs: wave-uniform)
      const int cq = min(ps.gl + k * G, ncq - 1);   // (clamped: nothing conditional around a load)
      if constexpr (NHWC) {
        const int e = (ps.src * g.C + c0 + 4 * cq) | ps.template zoff<BF>();
        q[k] = load_quad<BF>(xb, e, 0);
      } else {
        const int e = ((c0 + 4 * cq) * g.P + ps.src) | ps.template zoff<BF>();
        q[k] = load_quad_planes<BF>(xb, e, g.P);
      }
    }
  }
  // (a quad past the chunk's end goes to the spare slot `dump`: an address select; a predicated LDS store costs registers)
  __device__ __forceinline__ void commit(QT* slab, const KP& g, const TilePos<R>& ps, int G, int Ppb, int ncq, int dump,
                                         int npu) const {
    if constexpr (CST) {
      const CoopMap cm(g, ps.v, npu);
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        int psub, cq;
        cm.item(k, psub, cq);
        const bool ok = k < (1 << cm.lq) && cq < ncq;
        slab[ok ? cq * Ppb + cm.wb + psub : dump] = q[k];
      }
      return;
    }
    if constexpr (NHWC) {
      if (g.Tc >= 0 && ps.v - (int)__lane_id() + 64 <= npu) {
        const int lq = g.Tc, lane = __lane_id(), wb = ps.v - lane;
        const int cq = lane & ((1 << lq) - 1), p0 = lane >> lq, step = 64 >> lq;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const int psub = p0 + k * step;
          const bool ok = k < (1 << lq) && cq < ncq;
          slab[ok ? cq * Ppb + wb + psub : dump] = q[k];
        }
        return;
      }
    }
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      const int cq = ps.gl + k * G;
      slab[cq < ncq ? cq * Ppb + ps.v : dump] = q[k];
    }
  }
};

// Ring positions of a band, numbered compactly (the backward keeps a window row for each: 2R per padded row, and the 2R
// ring rows above / below the image): padded row vy, image coordinates (y, x) of the position.
template <int R>
__device__ __forceinline__ int ring_slot(int vy, int y, int x, int rows, int W, int H) {
  const bool col = x < 0 || x >= W;
  const int cslot = vy * 2 * R + (x < 0 ? x + R : x - W + R);
  const int rslot = rows * 2 * R + (y < 0 ? y + R : R + y - H) * W + x;
  return col ? cslot : rslot;
}

// sum over the 64 lanes of a wavefront, valid in lane 63 (fixed DPP tree: group_sum covers each half, then the row
// broadcast of lane 31 into the upper half)
__device__ __forceinline__ float wave_sum(float v) {
  v = group_sum(v, 32);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));  // row_bcast31 into rows 2, 3
  return v;
}

// ---- forward ----------------------------------------------------------------------------------------------------------
// POOL: the fused tail of models/NFP_Pooling.py:27-31 for large maps: besides the maps this band's share of the two
// pooled sums goes to part[(b * nb + band) * kPoolSub + segment][C + N] (sums, not means); pool_fold joins the rows in a
// fixed order.
// GFC (nfp.py:265-276) differs from cosine in how a pair sum and the two norms combine (a reciprocal per output).
// LDS: slab [Cc / 4][Ppb] float4 (POOL: later the band's maps [N][nbp]) | pair sums [NF][npu] | per-position factor [npu].
// (k = 3: two workgroups of up to 1024 threads share a compute unit — 8 wavefronts per SIMD, 64 registers)
// DMA (round 4; channels-last, one thread per position): the band is staged by LDS-DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write) into a POSITION-major slab — the 16-byte pieces of a position's channels side by side,
// as they lie in memory — in its storage type: bf16 maps keep 8 channels per piece and are summed with v_dot2c_f32_bf16
// (cosine: one instruction per channel pair and direction), half the LDS bytes of the float4 slab.  With several channel
// chunks the NEXT chunk's DMA runs under the current chunk's sums (two slabs).  See the block in the kernel.
template <int R, int M, bool BF, bool NHWC, bool POOL = false, bool GFC = false, bool DMA = false>
__global__ void __launch_bounds__(1024, (R == 1 && M != kSymTerm ? 8 : 4)) fwd_tile(const KP g, const TileGeo tg, const void* __restrict__ x,
                                                 void* __restrict__ out, float* __restrict__ saved,
                                                 float* __restrict__ part, float* __restrict__ gap, float* __restrict__ nfpm) {
  constexpr int N = Win<R>::N, NF = Win<R>::NF;
  constexpr int ES = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  lds_poison(lds4, tg.ldsw);
  int b, band;
  tile_ids(blockIdx.x, g.B, tg.nb, b, band);
  const TileBand<R> bd(g, tg, band);
  const Fold fo(g);
  const TilePos<R> ps(g, tg, bd, fo);
  const int G = blockDim.x, Wu = tg.Wu, Ppb = tg.Ppb, npu = tg.rows * Wu;
  const int W = g.W, P = g.P, v = ps.v;
  static_assert(!DMA || (NHWC && !POOL), "the LDS-DMA staging: channels-last, plain maps");
  using QT = typename Quad<BF>::T;                  // a slab slot: four channels of a position, in the storage type
  QT* slab = (QT*)lds4;
  const int dump = (g.Cc >> 2) * Ppb;               // (a spare slot behind the slab)
  // (POOL: the band's map values [N][nbp], staged for the pooled sums, lie over the slab — dead by then; should they
  // need more room than the slab has, the tables start behind them)
  const int nbpA = (tg.rows - 2 * R) * g.W;
  // DMA: g.Tc = log2 of the 16-byte pieces per position and chunk (PC); the slab(s) hold ((npu + 63) & ~63) * PC pieces
  const int lpc = DMA ? g.Tc : 0, PC = 1 << lpc, npu64 = (npu + 63) & ~63;
  const int nbuf = (DMA && g.C > g.Cc) ? 2 : 1;
  const int slabq = ((dump + 1) * (int)sizeof(QT) + 15) >> 4;   // the slab and its spare slot, in 16-byte units
  const int tt0 = DMA ? nbuf * npu64 * PC : (POOL ? max(slabq, (N * nbpA + 3) >> 2) : slabq);
  float* Tt = (float*)(lds4 + tt0);                 // [NF][npu] pair sums per direction, then the per-position factor
  float* Fq = Tt + NF * npu;
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  float acc[NF];
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = 0.f;
  float nrm = 0.f;

  if constexpr (DMA) {
    // ---- LDS-DMA staging + sums on the position-major slab -------------------------------------------------------------
    // Piece (u, k) — 16 bytes: channels 4k .. (f32) / 8k .. (bf16) of position u in this chunk — lives at slab[u * PC + (k ^ f(u))], f(u) =
    // (u >> (4 - lpc)) & (PC - 1): the XOR spreads the 16 lanes a ds_read_b128 serves per cycle (16 consecutive positions,
    // PC pieces apart) over all 16 bank columns.  A DMA instruction writes 64 consecutive 16-byte slots (M0 base + lane * 16:
    // the destination is not a per-lane scatter), so the swizzle goes on the SOURCE address: lane L of instruction i fills
    // slot i * n + L of the wavefront's n * PC slots (n = the wavefront's positions: 64, fewer in a workgroup's last one),
    // i.e. piece slot (L + i n) % PC of position (L + i n) / PC, and fetches the piece that belongs there from the position's
    // fold source (the offset of somebody else's position comes through ds_bpermute).  Positions that read zeros (zero
    // padding, rows past the band) write them with a plain ds_write instead: an exec-masked DMA lane leaves its slot alone.
    // Ordering: the DMA counts in vmcnt; __syncthreads() makes every wavefront wait for its own (vmcnt(0)) and then
    // meet the others — only then is the slab read.
    const int lane = __lane_id(), wb = v - lane, n = min(64, npu - wb);
    const long long xoff = (long long)b * g.sB * ES;
    const int srcb = ps.zero ? -1 : ps.src * g.C * ES;                       // byte offset of this position's pixel, or "zeros"
    uint4* sl = (uint4*)lds4;
    auto dma = [&](int c0, uint4* buf) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < PC) {                                                         // (wave-uniform)
          const int idx = lane + i * n, pl = idx >> lpc, ks = idx & (PC - 1);
          const int sb = __builtin_amdgcn_ds_bpermute(pl << 2, srcb);
          const int k = ks ^ (((wb + pl) >> (4 - lpc)) & (PC - 1));
          uint4* dst = buf + (long long)wb * PC + i * n;                     // wave-uniform; the lane's slot is dst + lane
          if (sb >= 0) {
            const char* gp = (const char*)x + xoff + sb + c0 * ES + k * 16;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gp,
                                             (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
          } else {
            dst[lane] = make_uint4(0u, 0u, 0u, 0u);
          }
        }
      }
    };
    // the XOR terms of this position and of its forward neighbours
    const int fmask = PC - 1, fsh = 4 - lpc;
    const int fv = (v >> fsh) & fmask;
    // slot of piece k of the position under forward direction d (recomputed per use: two arrays of NF offsets cost the
    // registers that keep two 900-thread workgroups on a compute unit)
    auto nslot = [&](int d, int k) {
      int dy, dx;
      fdir<R>(d, dy, dx);
      const int u = v + dy * Wu + dx;
      return (u << lpc) + (k ^ ((u >> fsh) & fmask));
    };
    dma(0, sl);
    NFP_STAMP(1);
    int cur = 0;
    for (int c0 = 0; c0 < g.C; c0 += g.Cc, cur ^= 1) {
      __syncthreads();   // this chunk's DMA has landed; the other slab's readers (previous chunk) are done
      if (c0 + g.Cc < g.C) dma(c0 + g.Cc, sl + (cur ^ 1) * npu64 * PC);
      if (c0 == 0) NFP_STAMP(2);
      if (ps.live) {
        const uint4* bufc = sl + cur * npu64 * PC;
        for (int k = 0; k < PC; ++k) {
          const uint4 a = bufc[(v << lpc) + (k ^ fv)];
          if constexpr (BF) {
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            const bf2 a0 = __builtin_bit_cast(bf2, a.x), a1 = __builtin_bit_cast(bf2, a.y), a2 = __builtin_bit_cast(bf2, a.z),
                      a3 = __builtin_bit_cast(bf2, a.w);
            if (M == NFP_COSINE) {
              nrm = __builtin_amdgcn_fdot2_f32_bf16(a0, a0, nrm, false);
              nrm = __builtin_amdgcn_fdot2_f32_bf16(a1, a1, nrm, false);
              nrm = __builtin_amdgcn_fdot2_f32_bf16(a2, a2, nrm, false);
              nrm = __builtin_amdgcn_fdot2_f32_bf16(a3, a3, nrm, false);
#pragma unroll
              for (int d = 0; d < NF; ++d) {
                const uint4 q = bufc[nslot(d, k)];
                float t_ = acc[d];
                t_ = __builtin_amdgcn_fdot2_f32_bf16(a0, __builtin_bit_cast(bf2, q.x), t_, false);
                t_ = __builtin_amdgcn_fdot2_f32_bf16(a1, __builtin_bit_cast(bf2, q.y), t_, false);
                t_ = __builtin_amdgcn_fdot2_f32_bf16(a2, __builtin_bit_cast(bf2, q.z), t_, false);
                t_ = __builtin_amdgcn_fdot2_f32_bf16(a3, __builtin_bit_cast(bf2, q.w), t_, false);
                acc[d] = t_;
              }
            } else {
              const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
              float af[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                af[2 * e] = __uint_as_float(aw[e] << 16);
                af[2 * e + 1] = __uint_as_float(aw[e] & 0xffff0000u);
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) nrm = M == kNormP1 ? nrm + fabsf(af[e]) : fmaf(af[e], af[e], nrm);
#pragma unroll
              for (int d = 0; d < NF; ++d) {
                const uint4 q = bufc[nslot(d, k)];
                const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
                float t_ = acc[d];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float e0 = af[2 * e] - __uint_as_float(qw[e] << 16);
                  const float e1 = af[2 * e + 1] - __uint_as_float(qw[e] & 0xffff0000u);
                  t_ = M == kNormP1 ? t_ + (fabsf(e0) + fabsf(e1)) : fmaf(e0, e0, fmaf(e1, e1, t_));
                }
                acc[d] = t_;
              }
            }
          } else {
            const float4 af = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
            if (M == kNormP1)
              nrm += (fabsf(af.x) + fabsf(af.y)) + (fabsf(af.z) + fabsf(af.w));
            else
              nrm = fmaf(af.x, af.x, fmaf(af.y, af.y, fmaf(af.z, af.z, fmaf(af.w, af.w, nrm))));
#pragma unroll
            for (int d = 0; d < NF; ++d) {
              const uint4 qu = bufc[nslot(d, k)];
              const float4 q = make_float4(__uint_as_float(qu.x), __uint_as_float(qu.y), __uint_as_float(qu.z), __uint_as_float(qu.w));
              if (M == NFP_COSINE) {
                acc[d] = fmaf(af.x, q.x, fmaf(af.y, q.y, fmaf(af.z, q.z, fmaf(af.w, q.w, acc[d]))));
              } else if (M == kNormP1) {
                acc[d] += (fabsf(af.x - q.x) + fabsf(af.y - q.y)) + (fabsf(af.z - q.z) + fabsf(af.w - q.w));
              } else {
                const float e0 = af.x - q.x, e1 = af.y - q.y, e2 = af.z - q.z, e3 = af.w - q.w;
                acc[d] = fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, fmaf(e3, e3, acc[d]))));
              }
            }
          }
        }
      }
    }
  }

  TileStage<R, BF, NHWC, false> st;
  if constexpr (!DMA) {
    st.issue(g, ps, xb, G, 0, min(g.Cc, g.C) >> 2, npu);
    __builtin_amdgcn_sched_barrier(0);
    NFP_STAMP(1);
  }

  for (int c0 = 0; c0 < (DMA ? 0 : g.C); c0 += g.Cc) {
    const int ncq = min(g.Cc, g.C - c0) >> 2;
    if (c0 > 0) __syncthreads();  // previous chunk fully consumed
    st.commit(slab, g, ps, G, Ppb, ncq, dump, npu);
    __syncthreads();
    // the next chunk's loads fly while this one is summed (the staging registers are free once committed)
    if (c0 + g.Cc < g.C) st.issue(g, ps, xb, G, c0 + g.Cc, min(g.Cc, g.C - c0 - g.Cc) >> 2, npu);
    if (c0 == 0) NFP_STAMP(2);
    if (POOL && g.pool_gap) {
      // this band's share of sum over pixels of x[c]: (channel quad, segment of the band's pixels) items dealt over the
      // FULL wavefronts (the DPP tree reads all 64 lanes; a workgroup's last wavefront may be partial); lanes stride over
      // the segment's pixels; one writer per (segment, channel).  pool_fold joins segments and bands in a fixed order.
      const int t = ps.gl + G * v, lane = t & 63, wv = t >> 6, nw = (G * npu) >> 6;
      const int nbp = (bd.y1 - bd.y0) * W, seg = (nbp + kPoolSub - 1) / kPoolSub;
      // (scratch rows another workgroup may fold: written through — nfp_common.h::pool_last_band)
      const Rsrc pb = pool_rsrc(part + ((long long)b * tg.nb + band) * kPoolSub * (g.C + N), (long long)kPoolSub * (g.C + N));
      if (wv < nw) {
        for (int it = wv; it < ncq * kPoolSub; it += nw) {
          const int cq = it / kPoolSub, sub = it - cq * kPoolSub, hi = min(nbp, (sub + 1) * seg);
          float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int lp = sub * seg + lane; lp < hi; lp += 64) {
            const int yl = fdivi(lp, W), xl = lp - yl * W;
            const float4 q = Quad<BF>::get(slab[cq * Ppb + (yl + R) * Wu + xl + R]);
            s4.x += q.x;
            s4.y += q.y;
            s4.z += q.z;
            s4.w += q.w;
          }
          s4.x = wave_sum(s4.x);
          s4.y = wave_sum(s4.y);
          s4.z = wave_sum(s4.z);
          s4.w = wave_sum(s4.w);
          if (lane == 63) pool_store4(pb, sub * (g.C + N) + c0 + 4 * cq, s4);
        }
      } else if (nw == 0 && t == 0) {  // (a workgroup below 64 threads: tiny maps, one thread adds them up)
        for (int cq = 0; cq < ncq; ++cq) {
          float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int lp = 0; lp < nbp; ++lp) {
            const int yl = lp / W, xl = lp - yl * W;
            const float4 q = Quad<BF>::get(slab[cq * Ppb + (yl + R) * Wu + xl + R]);
            s4.x += q.x;
            s4.y += q.y;
            s4.z += q.z;
            s4.w += q.w;
          }
          for (int sub = 0; sub < kPoolSub; ++sub)
            pool_store4(pb, sub * (g.C + N) + c0 + 4 * cq, sub == 0 ? s4 : make_float4(0.f, 0.f, 0.f, 0.f));
        }
      }
    }
    if (ps.live) {
      // (every tap at a constant offset; a tap past the padded band reads whatever lies there: such a pair is never
      // looked up)
      for (int cq = ps.gl; cq < ncq; cq += G) {
        const QT* r0 = slab + cq * Ppb + v;
        const QT araw = r0[0];
        if constexpr (BF && M == NFP_COSINE) {   // products of two bf16 are exact in float32: v_dot2c_f32_bf16, no unpacking
          nrm = quad_dot<BF>(araw, araw, nrm);
#pragma unroll
          for (int d = 0; d < NF; ++d) {
            int dy, dx;
            fdir<R>(d, dy, dx);
            const QT qraw = dy == 0 ? r0[dx] : (r0 + dy * Wu - R)[dx + R];
            acc[d] = quad_dot<BF>(araw, qraw, acc[d]);
          }
          continue;
        }
        const float4 a = Quad<BF>::get(araw);
        if constexpr (M == kSymTerm) {   // sums of the measure's per-channel term: one wave-uniform switch per channel quad
          sym_switch(g.measure, [&](auto mm) {
            using MM = decltype(mm);
#pragma unroll
            for (int d = 0; d < NF; ++d) {
              int dy, dx;
              fdir<R>(d, dy, dx);
              const float4 q = Quad<BF>::get(dy == 0 ? r0[dx] : (r0 + dy * Wu - R)[dx + R]);
              acc[d] += (MM::term(a.x, q.x, g) + MM::term(a.y, q.y, g)) + (MM::term(a.z, q.z, g) + MM::term(a.w, q.w, g));
            }
          });
          continue;
        }
        if (M == kNormP1)   // (Norm p = 1 — the class default, nfp.py:16,141-148 — and EMD, nfp.py:207-216: sums of |.|)
          nrm += (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w));
        else
          nrm = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, fmaf(a.w, a.w, nrm))));
#pragma unroll
        for (int d = 0; d < NF; ++d) {
          int dy, dx;
          fdir<R>(d, dy, dx);
          const float4 q = Quad<BF>::get(dy == 0 ? r0[dx] : (r0 + dy * Wu - R)[dx + R]);   // (row base + compile-time column)
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
  NFP_STAMP(3);
  // channel groups joined inside the wavefront; one lane per position publishes the sums.  The per-position factor F:
  // cosine 1 / max(|x|, eps), DotProduct 1, GFC |x| itself; L2 keeps |x|^2 (the 'Norm' quirk reads it).
#pragma unroll
  for (int d = 0; d < NF; ++d) acc[d] = group_sum(acc[d], G);
  nrm = group_sum(nrm, G);
  if (G > 16) {  // (group_sum leaves the sum in the group's last lane only: hand it to the others — they share the outputs)
    const int last = (__lane_id() | (G - 1)) << 2;
#pragma unroll
    for (int d = 0; d < NF; ++d) acc[d] = __int_as_float(__builtin_amdgcn_ds_bpermute(last, __float_as_int(acc[d])));
    nrm = __int_as_float(__builtin_amdgcn_ds_bpermute(last, __float_as_int(nrm)));
  }
  const float Fp = M == NFP_COSINE ? (GFC ? __builtin_amdgcn_sqrtf(nrm) : unit_or(g, inv_norm(nrm, g.inv_eps))) : nrm;
  if (ps.gl == 0) {
#pragma unroll
    for (int d = 0; d < NF; ++d) Tt[d * npu + v] = acc[d];
    Fq[v] = Fp;
  }
  __syncthreads();
  NFP_STAMP(4);
  // outputs of the band's rows, by the position's own threads (lanes along a row: coalesced stores); the G lanes of a
  // position share the N taps
  const int lpf = (ps.vy - R) * W + ps.x;
  float* vm = (float*)slab;  // (POOL) [N][nbpA]: the band's map values, for the pooled sums (every wavefront is past its sums)
  if (ps.own) {
    const int p = ps.y * W + ps.x;
    // (buffer stores: one offset register for all N maps of the pixel, the map through the scalar offset)
    // (POOL without a backward to follow: nobody reads the maps — an empty resource drops the stores in the memory pipeline)
    const Rsrc ob = make_rsrc((char*)out + (long long)b * N * P * ES, (!POOL || g.pool_map) ? (long long)N * P * ES : 0);
    auto one = [&](int n, auto statc) {
      constexpr bool stat = decltype(statc)::value;
      int dy, dx;
      tap_offset<R>(n, dy, dx);
      const bool fwd = dy > 0 || (dy == 0 && dx > 0);
      const int fi = fwd ? fidx<R>(dy, dx) : fidx<R>(-dy, -dx);
      const float* rq = Fq + v + dy * Wu - R;
      const float fq = rq[dx + R];
      // (one lane per position and a forward tap: the pair sum is still in this thread's registers)
      float pairv;
      if constexpr (stat)
        pairv = fwd ? acc[fi] : (Tt + fi * npu + v + dy * Wu - R)[dx + R];
      else
        pairv = Tt[fi * npu + (fwd ? v : v + dy * Wu + dx)];
      float val;
      if (M == NFP_COSINE) {
        const float s = GFC ? pairv * __builtin_amdgcn_rcpf(fmaf(Fp, fq, g.eps)) : pairv * Fp * fq;
        val = fin_prod(g, s);
      } else if (M == kNormP1) {
        val = g.osa * (g.diff ? pairv : fq);     // (no root: the sum of |.| is the norm)
      } else if (M == kSymTerm) {
        sym_switch(g.measure, [&](auto mm) { val = decltype(mm)::fin(pairv, 0.f, 0.f, 0.f, 0.f, g); });
      } else {
        val = fin_dist(g, g.diff ? pairv : fq);  // 'Norm' quirk (nfp.py:74 vs 85): |neighbour|
      }
      if constexpr (stat)
        store_1<BF>(ob, p, n * P, val);
      else
        store_1<BF>(ob, n * P + p, 0, val);
      if constexpr (POOL) vm[n * nbpA + lpf] = val;
    };
    if (G == 1) {  // (the usual case on large maps: every tap's offsets are compile-time constants)
#pragma unroll
      for (int n = 0; n < N; ++n) one(n, std::true_type{});
    } else {
      for (int n = ps.gl; n < N; n += G) one(n, std::false_type{});
    }
    if (M == NFP_COSINE && !g.unit && saved != nullptr && ps.gl == 0)
      saved[(long long)b * P + p] = GFC ? Fp : __builtin_amdgcn_sqrtf(nrm);
  }
  if constexpr (POOL) {
    // this band's share of sum over pixels of out[n]: (map, segment) items over the full wavefronts, a fixed order
    lds_barrier();   // (not __syncthreads(): the map stores just issued drain under the sums — nfp_common.h)
    const int t = ps.gl + G * v, lane = t & 63, wv = t >> 6, nw = (G * npu) >> 6;
    const int nbp = (bd.y1 - bd.y0) * W, seg = (nbp + kPoolSub - 1) / kPoolSub;
    const Rsrc pb = pool_rsrc(part + ((long long)b * tg.nb + band) * kPoolSub * (g.C + N), (long long)kPoolSub * (g.C + N));
    if (wv < nw) {
      for (int it = wv; it < N * kPoolSub; it += nw) {
        const int n = it / kPoolSub, sub = it - n * kPoolSub, hi = min(nbp, (sub + 1) * seg);
        float s = 0.f;
        for (int i = sub * seg + lane; i < hi; i += 64) s += vm[n * nbpA + i];
        s = wave_sum(s);
        if (lane == 63) pool_store1(pb, sub * (g.C + N) + g.C + n, s);
      }
    } else if (nw == 0 && t == 0) {
      for (int n = 0; n < N; ++n) {
        float s = 0.f;
        for (int i = 0; i < nbp; ++i) s += vm[n * nbpA + i];
        for (int sub = 0; sub < kPoolSub; ++sub) pool_store1(pb, sub * (g.C + N) + g.C + n, sub == 0 ? s : 0.f);
      }
    }
    // the band that arrives last folds every band's row of the image, in band order (nfp_common.h::pool_last_band);
    // without counters (a descriptor without its workspace) the caller launches pool_fold
    if (g.tickets != nullptr) {
      const int nth = blockDim.x * blockDim.y * blockDim.z;
      if (pool_last_band(g.tickets + b, tg.nb, (int*)lds4, t == 0))
        pool_fold_image(part + (long long)b * tg.nb * kPoolSub * (g.C + N), tg.nb * kPoolSub, g.C, N, g.invP,
                        g.pool_gap ? gap + (long long)b * g.C : nullptr, nfpm + (long long)b * N, t, nth);
    }
  }
  NFP_STAMP(5);
}

// gap[b][c] = (sum over bands of part[b][band][c]) / P, nfpm[b][n] likewise: the bands in a fixed order
template <int = 0>
__global__ void __launch_bounds__(256) pool_fold(const float* __restrict__ part, float* __restrict__ gap,
                                                 float* __restrict__ nfpm, int B, int nb, int C, int N, float invP) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int CN = C + N;
  // (gap == nullptr: the bands wrote no channel sums — only the N map sums of every image are folded)
  const int per = gap != nullptr ? CN : N, k0 = gap != nullptr ? 0 : C;
  if (i >= (long long)B * per) return;
  const int b = (int)(i / per), k = k0 + (int)(i - (long long)b * per);
  // (the rows in a fixed order; eight loads in flight at a time — one dependent load per row is a latency chain: 19 rows
  // took 13 us of a 35 us pooled tail)
  const float* pr = part + (long long)b * nb * CN + k;
  float s = 0.f;
  int j = 0;
  for (; j + 8 <= nb; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pr[(long long)(j + u) * CN];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; j < nb; ++j) s += pr[(long long)j * CN];
  if (k < C)
    gap[(long long)b * C + k] = s * invP;
  else
    nfpm[(long long)b * N + k - C] = s * invP;
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// POOL: grad_out is not a map: go[b,n,p] = gnfpm[b,n] / P for every p, and every grad_x[b,c,p] also gets ggap[b,c] / P.
// GFC: the general post-factors of nfp_common.h::cross_f / diag_f (a reciprocal per window slot); cosine and dot keep the
// plain product of the two per-pixel factors.
// LDS (floats): ipn [PL] | pair values [N][PL] | guard [4] (cosine: sg = +-grad_out; L2: c = -+g / d) — PL = (rows + 2R) * Wu: R
// rows of ZEROS above and below the band in every plane, so that a tap at a constant offset of ANY position reads a
// value (a column past the row's end lands in the ring columns of the next row: zeros too).  After phase A the pair
// values are dead: the x slab lies over them, and behind the slab the window rows of the ring positions
// [rows * 2R + 2R * W][K2] (ring_slot).
// (Cosine: the pair {r, r + d} has ONE similarity s, which reaches the backward twice — out[n][r] and
// out[opp n][r + d], equal up to the forward's rounding; the pull of both on |x_r| uses r's own copy, so that only the
// gradients travel through LDS.)
template <int R, int M, bool BF, bool NHWC, bool POOL = false, bool GFC = false, bool CST = false>
__global__ void __launch_bounds__(1024, (R == 1 && M != kSymTerm ? (CST ? 5 : 8) : 4)) bwd_tile(const KP g, const TileGeo tg, const void* __restrict__ x,
                                                const void* __restrict__ go, const void* __restrict__ out,
                                                const float* __restrict__ saved, void* __restrict__ gx,
                                                const float* __restrict__ ggap, const float* __restrict__ gnfpm) {
  constexpr int N = Win<R>::N, K = Win<R>::K, K2 = Win<R>::K2;
  constexpr int ES = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  lds_poison(lds4, tg.ldsw);
  int b, item;
  tile_ids(blockIdx.x, g.B, tg.nb * tg.S, b, item);
  const int band = item / tg.S, cblk = item - band * tg.S;
  const TileBand<R> bd(g, tg, band);
  const Fold fo(g);
  const TilePos<R> ps(g, tg, bd, fo);
  const int G = blockDim.x, Wu = tg.Wu, Ppb = tg.Ppb, npu = tg.rows * Wu, PL = (tg.rows + 2 * R) * Wu;
  const int W = g.W, H = g.H, P = g.P, v = ps.v;
  const int cb0 = cblk * g.Cwg, cb1 = min(g.C, cb0 + g.Cwg);
  // (a corner position's diagonal taps reach R words past a margin: the padding behind ipn — which is what lies in front
  // of plane 0 — and four words behind the last plane are zeroed too.  In front of ipn itself there is nothing: such a
  // read feeds a window slot nobody fetches, through a product with an exact 0.)
  float* ipn = (float*)lds4 + R * Wu;                       // ipn[v], margins at v < 0 and v >= npu
  float* pvb = (float*)lds4 + ((PL + 3) & ~3);
  float* PV = pvb + R * Wu;                                 // plane n at PV + n * PL
  using QT = typename Quad<BF>::T;                          // a slab slot: four channels of a position, in the storage type
  QT* slab = (QT*)pvb;
  const int dump = (g.Cc >> 2) * Ppb;                       // (a spare slot behind the slab)
  float* Wr = (float*)(slab + dump + 1) + 2 * K2;           // (2 K2 floats of slack either side: shifted reads)
  const Rsrc xb = make_rsrc((const char*)x + (long long)b * g.sB * ES, (long long)g.C * P * ES);
  const Rsrc gxb = make_rsrc((char*)gx + (long long)b * g.gB * ES, (long long)g.C * P * ES);
  const Rsrc gob = make_rsrc((const char*)go + (long long)b * N * P * ES, POOL ? 0 : (long long)N * P * ES);
  const Rsrc outb = make_rsrc((const char*)out + (long long)b * N * P * ES, (long long)N * P * ES);
  // (DotProduct has no saved norms: an empty resource, every load reads 0)
  const Rsrc svb = make_rsrc((const char*)saved + (long long)b * P * 4, (M == NFP_COSINE && !g.unit) ? (long long)P * 4 : 0);
  NFP_STAMP_INIT();
  NFP_STAMP(0);

  // ---- A1: the position's own pairs (every tap), its norm factor; all loads first, then the x chunk ----------------
  // A position that owns no pairs (ring, outside the image, rows past the band) loads zeros: its pair values come out 0.
  TileStage<R, BF, NHWC, CST> st;
  float w[K2], sv[N];
  float dfn = 0.f, ipr = 1.f;
  {
    const int ep = ps.real ? ps.y * W + ps.x : Oob<BF>::e;
    float gov[N];
    // Cosine: a RING position needs the similarity of every pair that ends on it — the output of its real neighbour's
    // opposite tap (plane N-1-n at pixel u + d_n).  Same load instruction as everybody's own outputs: the plane
    // difference and the tap offset go into the ring lanes' address (a mask, not a branch around a load).
    const bool ring = M == NFP_COSINE && ps.live && !ps.real && !ps.zero;
    const int ea = ring ? ps.y * W + ps.x : ep, rm = ring ? -1 : 0;
    bool rbad[K], cbad[K];
#pragma unroll
    for (int d = 0; d < K; ++d) {
      rbad[d] = ring && (unsigned)(ps.y + d - R) >= (unsigned)H;
      cbad[d] = ring && (unsigned)(ps.x + d - R) >= (unsigned)W;
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1, dy = j / K - R, dx = j % K - R;
      // (the plane goes into the lane's own offset: the range check of a buffer load looks at that offset alone, and a
      // ring lane's "plane difference" may be negative)
      int e = ep + n * P;
      if (M == NFP_COSINE) {
        e = ea + n * P + (((N - 1 - 2 * n) * P + dy * W + dx) & rm);
        e = (rbad[dy + R] || cbad[dx + R]) ? Oob<BF>::e : e;
      }
      sv[n] = M == kNormP1 ? 0.f : load_1<BF>(outb, e, 0);   // (p = 1: the gradient does not depend on the distance)
      gov[n] = POOL ? 0.f : load_1<BF>(gob, ep, n * P);
    }
    float nrm = 0.f;
    if (M == NFP_COSINE) nrm = load_1<false>(svb, ps.src | ps.zf, 0);
    __builtin_amdgcn_sched_barrier(0);
    st.issue(g, ps, xb, G, cb0, min(g.Cc, cb1 - cb0) >> 2, npu);
    __builtin_amdgcn_sched_barrier(0);
    NFP_STAMP(7);
    // (the sign convention as arithmetic: a select on a wave-uniform flag becomes a branch per value)
    const float sa = g.osa, sb = -g.osa * g.osb;   // s = osa * (out - osb): out = osa * s + osb with osa = +-1
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int j = n < K2 / 2 ? n : n + 1;
      float gc = gov[n];
      if constexpr (POOL) gc = ps.real ? gnfpm[(long long)b * N + n] * g.invP : 0.f;
      if (M == NFP_COSINE) {
        sv[n] = fmaf(sa, sv[n], sb);
        w[j] = sa * gc;
      } else if (M == kNormP1) {
        w[j] = ps.real ? g.osa * gc : 0.f;   // (p = 1: d out / d (a - b)[c] = +-g sign(a - b)[c]; the map itself is not needed)
      } else if (M == kSymTerm) {
        float c = 0.f;
        sym_switch(g.measure, [&](auto mm) { c = decltype(mm)::coef(gc, sv[n], 0.f, 0.f, 0.f, 0.f, g).k0; });
        w[j] = ps.real ? c : 0.f;
      } else {
        w[j] = ps.real ? dist_coef(g, gc, sv[n]) : 0.f;
      }
      PV[n * PL + v] = w[j];
    }
    if (M == NFP_COSINE) {
      const float ip = unit_or(g, __builtin_amdgcn_rcpf(fmaxf(nrm, g.eps)));   // (DotProduct: no norm factors, no diagonal)
      ipr = GFC ? nrm : ip;                                                     // (GFC: the norm itself — nfp_common.h::cross_f)
      ipn[v] = ipr;
      dfn = nrm > 0.f ? -(GFC ? 1.f : g.nuf * ip) * __builtin_amdgcn_rcpf(nrm) : 0.f;
    }
    // the zero rows above and below the band, every plane (and the guard words: the first four positions' threads)
    if (ps.vy < R || ps.vy >= tg.rows - R) {
      const int gw = min(v, 3);
      ((float*)lds4)[PL + gw] = 0.f;   // (padding, or the first words of plane 0's zero row)
      pvb[N * PL + gw] = 0.f;
      const int m = ps.vy < R ? v - R * Wu : v + R * Wu;
      if (M == NFP_COSINE) ipn[m] = 0.f;
#pragma unroll
      for (int n = 0; n < N; ++n) PV[n * PL + m] = 0.f;
    }
  }
  NFP_STAMP(1);
  __syncthreads();
  NFP_STAMP(2);

  // ---- A2: the window weights of this thread's position, in registers (the same code for every position) -------------
  float wself = 0.f;   // (kSymTerm) weights of pairs of a pixel with its own padded copy: see the merge below
  const float dneg = g.diff ? -1.f : 0.f;   // L2: cross weight = -c with the difference weights, 0 with the 'Norm' quirk
  float D = 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const int j = n < K2 / 2 ? n : n + 1, dy = j / K - R, dx = j % K - R, opp = N - 1 - n;
    const float c2 = (PV + opp * PL + v + dy * Wu - R)[dx + R];
    if (M == NFP_COSINE) {
      const float ipq = (ipn + v + dy * Wu - R)[dx + R];
      const float S = w[j] + c2;
      if (GFC) {
        const float ts = S * sv[n];   // (exactly 0 for a tap that leaves the band: whatever factor was read there stays out)
        D = ts != 0.f ? fmaf(ts, diag_f(g, ipr, ipq), D) : D;
        w[j] = cross_f(g, ipr, ipq) * S;
      } else {
        D = fmaf(S, sv[n], D);
        w[j] = ipr * ipq * S;
      }
    } else if (M == kNormP1) {
      // p = 1: grad_x[c][r] = sum_j w[j] sign(x_r - x_{r + d_j})[c] with the difference weights — a zero-padded tap is a
      // zero vector in the slab, so sign(x_r - 0) needs no case of its own, and there is no diagonal; with the 'Norm'
      // quirk a pair pulls on its NEIGHBOUR alone: D sign(x_r)[c], on the centre slot (phase B)
      const float c1 = w[j];
      D += (1.f + dneg) * c2;
      w[j] = -dneg * (c1 + c2);
    } else if (M == kSymTerm) {
      // symmetric terms: the pair (r centre, t) and the pair (t centre, r) pull on x_r through the same d term / d a (x_r, x_t)
      w[j] += c2;
    } else {
      const float c1 = w[j];
      D += fmaf(-dneg, c1, c2);  // 'Norm' quirk (nfp.py:74 vs 85): only the pair's NEIGHBOUR is pulled
      w[j] = dneg * (c1 + c2);
    }
  }
  w[K2 / 2] = (M == NFP_COSINE ? dfn : 1.f) * D;
  NFP_STAMP(9);
  __syncthreads();  // the pair values are dead: their LDS becomes the x slab
  NFP_STAMP(3);

  st.commit(slab, g, ps, G, Ppb, min(g.Cc, cb1 - cb0) >> 2, dump, npu);
  // A ring position is a copy of the image pixel it folds onto (reflect / replicate): its window row goes to LDS for
  // that pixel's thread
  if (ps.live && !ps.real && ps.gl == 0) {
    float* wrow = Wr + ring_slot<R>(ps.vy, ps.y, ps.x, tg.rows, W, H) * K2;
#pragma unroll
    for (int j = 0; j < K2; ++j) wrow[j] = w[j];
  }
  __syncthreads();
  NFP_STAMP(4);
  if (ps.own) {
    // ring positions that fold onto this pixel: per axis the coordinate itself (bit 0) and up to 2R ring coordinates;
    // a bit mask of the valid (row, column) combinations, then one trip per VALID combination (an edge pixel has one,
    // a corner three) — wavefronts without a border pixel skip the loop
    constexpr int KK = 2 * R + 1;
    unsigned my = 1u, mx = 1u;
#pragma unroll
    for (int i = 1; i < KK; ++i) {
      const int ry = i <= R ? -i : H - 1 + (i - R), rx = i <= R ? -i : W - 1 + (i - R);
      const int vyu = ry - (bd.y0 - R);
      my |= ((fo.y(ry, H) == ps.y && vyu >= 0 && vyu < bd.rows) ? 1u : 0u) << i;
      mx |= (fo.x(rx, W) == ps.x ? 1u : 0u) << i;
    }
    unsigned mask = 0u;
#pragma unroll
    for (int c = 1; c < KK * KK; ++c) mask |= (((my >> (c / KK)) & (mx >> (c % KK)) & 1u)) << c;
    while (mask != 0u) {
      const int c = __builtin_ctz(mask);
      mask &= mask - 1u;
      const int iy = fdivi(c, KK), ix = c - iy * KK;
      const int uy = iy == 0 ? ps.y : (iy <= R ? -iy : H - 1 + (iy - R));
      const int ux = ix == 0 ? ps.x : (ix <= R ? -ix : W - 1 + (ix - R));
      const int sy = ps.y - uy, sx = ps.x - ux;                       // r - u: slot j of r is slot j + (sy, sx) of u
      const float* wu = Wr + ring_slot<R>(uy - (bd.y0 - R), uy, ux, tg.rows, W, H) * K2 + sy * K + sx;
      bool oky[K], okx[K];
#pragma unroll
      for (int d = 0; d < K; ++d) {
        oky[d] = (unsigned)(d + sy) < (unsigned)K;
        okx[d] = (unsigned)(d + sx) < (unsigned)K;
      }
#pragma unroll
      for (int j = 0; j < K2; ++j) {
        // (u's own centre is not a position of r's window: the copy's pull on itself is the pixel's own — below)
        // (Norm p = 1 on the difference weights: u's slot that points at r ITSELF — a pixel and its own padded copy,
        // replicate padding; reflect with R = 2 — lands on r's centre slot, which multiplies sign(x_r) here, not x_r: the
        // pair's gradient is sign(x_u - x_r) = sign(0) = 0, so it is dropped.  The linear measures add it: it multiplies x_r.)
        const bool ok = oky[j / K] && okx[j % K] && !(j / K + sy == R && j % K + sx == R) && !((M == kNormP1 || M == kSymTerm) && j == K2 / 2);   // (kSymTerm: d term / d a (a, a) = 0 as well)
        const float val = wu[j];
        w[j] += ok ? val : 0.f;
        // (... but Hellinger's coefficient on such a pair is 1 / distance = inf, and the reference's inf * 0 = NaN has to come
        // out on that pixel: the dropped weights are kept and multiply an explicit zero in phase B)
        if (M == kSymTerm && j == K2 / 2) wself += (oky[R] && okx[R]) ? val : 0.f;
      }
      w[K2 / 2] += wu[K2 / 2 - sy * K - sx];
    }
  }

  // ---- B: one pass over the channel block ---------------------------------------------------------------------------
  // CST (channels-last, one thread per position, pixels of 96 bytes and more): a lane storing 16 bytes of ITS pixel per
  // quad leaves 64 scattered 16-byte writes per store instruction ([256,64,56,56]: the stores took 83 us of the kernel's
  // 153, against 47 of 96 us in NCHW — profiles/r03_w_tile_backward_stores_ab.txt).  Instead a wavefront turns its
  // results around in LDS — its own slots of the slab, once every wavefront has finished reading the chunk — and
  // stores as it loaded: Q adjacent lanes write 16 Q contiguous bytes of one pixel.  Four results in registers: this
  // variant is built for 96 registers (two workgroups of up to 640 threads per compute unit).
  const int p = ps.y * W + ps.x;
  for (int c0 = cb0; c0 < cb1; c0 += g.Cc) {
    const int ncq = min(g.Cc, cb1 - c0) >> 2;
    if (c0 > cb0) {
      st.commit(slab, g, ps, G, Ppb, ncq, dump, npu);
      __syncthreads();
    }
    const bool more = c0 + g.Cc < cb1;
    // (turning the results around takes 16 registers: there the next chunk is requested after the stores, not before the sums)
    if (more && !CST) st.issue(g, ps, xb, G, c0 + g.Cc, min(g.Cc, cb1 - c0 - g.Cc) >> 2, npu);
    auto one = [&](int cq) {
      const QT* rc = slab + cq * Ppb + v - R;
      float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (POOL && g.pool_gap) {
        const float4 gg = *(const float4*)(ggap + (long long)b * g.C + c0 + 4 * cq);
        r4 = make_float4(gg.x * g.invP, gg.y * g.invP, gg.z * g.invP, gg.w * g.invP);
      }
      if constexpr (M == kSymTerm) {
        const float4 a = Quad<BF>::get((rc + 0 * Wu)[R]);
        sym_switch(g.measure, [&](auto mm) {
          using MM = decltype(mm);
          const Coef one = {1.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < K2; ++j) {
            if (j == K2 / 2) continue;   // (no term of the pixel with itself; a zero-padded tap is a zero vector in the slab)
            const float4 q = Quad<BF>::get((rc + (j / K - R) * Wu)[j % K]);
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
        r4.x = fmaf(wself, 0.f, r4.x);   // (finite weights: nothing; Hellinger at distance 0: NaN, as the reference)
        r4.y = fmaf(wself, 0.f, r4.y);
        r4.z = fmaf(wself, 0.f, r4.z);
        r4.w = fmaf(wself, 0.f, r4.w);
        return r4;
      }
      if constexpr (M == kNormP1) {
        const float4 a = Quad<BF>::get((rc + 0 * Wu)[R]);
#pragma unroll
        for (int j = 0; j < K2; ++j) {
          const float4 q = Quad<BF>::get((rc + (j / K - R) * Wu)[j % K]);
          const bool c = j == K2 / 2;
          r4.x = fmaf(w[j], sgn3(c ? a.x : a.x - q.x), r4.x);
          r4.y = fmaf(w[j], sgn3(c ? a.y : a.y - q.y), r4.y);
          r4.z = fmaf(w[j], sgn3(c ? a.z : a.z - q.z), r4.z);
          r4.w = fmaf(w[j], sgn3(c ? a.w : a.w - q.w), r4.w);
        }
        return r4;
      }
#pragma unroll
      for (int j = 0; j < K2; ++j) {
        const float4 q = Quad<BF>::get((rc + (j / K - R) * Wu)[j % K]);
        r4.x = fmaf(w[j], q.x, r4.x);
        r4.y = fmaf(w[j], q.y, r4.y);
        r4.z = fmaf(w[j], q.z, r4.z);
        r4.w = fmaf(w[j], q.w, r4.w);
      }
      return r4;
    };
    if constexpr (CST) {
      constexpr int KQ = Quad<BF>::KQ;
      QT rq[KQ];                                  // (results in the storage type: what the stores take)
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        if (k < ncq) rq[k] = Quad<BF>::put(one(k));   // (wave-uniform bound; non-own positions compute something nobody stores)
        __builtin_amdgcn_sched_barrier(0);        // (one quad after the other: interleaved, their 36 LDS reads want 144 registers)
      }
      __syncthreads();  // chunk fully consumed: a position's slots now carry its results to the lanes that store them
#pragma unroll
      for (int k = 0; k < KQ; ++k)
        if (k < ncq) slab[k * Ppb + v] = rq[k];
      const CoopMap cm(g, v, npu);
      const int pc = ps.own ? p * g.C : Oob<BF>::e;
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        int psub, cq;
        cm.item(k, psub, cq);
        if (k < (1 << cm.lq)) {
          const QT r4 = slab[min(cq, ncq - 1) * Ppb + cm.wb + psub];
          const int e = __builtin_amdgcn_ds_bpermute(psub << 2, pc) + 4 * cq;
#ifdef NFP_TILE_NOSTORE   // (diagnostic build only: what the kernel costs without its grad_x stores)
          if (Quad<BF>::get(r4).x != 12345.678f) continue;
#endif
          store_quad<BF>(gxb, cq < ncq ? e : Oob<BF>::e, c0, r4);
        }
      }
      if (more) st.issue(g, ps, xb, G, c0 + g.Cc, min(g.Cc, cb1 - c0 - g.Cc) >> 2, npu);
    } else {
      if (ps.own) {
        for (int cq = ps.gl; cq < ncq; cq += G) {
          const float4 r4 = one(cq);
#ifdef NFP_TILE_NOSTORE
          if (r4.x != 12345.678f) continue;
#endif
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
      if (more) __syncthreads();  // chunk fully consumed
    }
    if (c0 == cb0) NFP_STAMP(5);
  }
  NFP_STAMP(6);
}

}  // namespace nfp
