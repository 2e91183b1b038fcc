// nfp_tables.h — index helpers of the hot-path kernels and the per-descriptor WORKSPACE tables.
//
// Everything the hot-path kernels need to know about the geometry (which taps of which pixels fold onto which
// neighbour under the padding mode, which half-stencil entry an output reads, LDS offsets of a pixel's window)
// depends on (H, W, R, padding_mode) only — not on the batch, the channels or the data.  Recomputing it per launch
// cost more than the arithmetic: at [64,512,7,7] the backward spent 1 080 instructions per thread on index work
// before its first barrier (hipcc -S), 3.9 of its 7.3 us.  The tables are built ONCE per descriptor by
// `fill_workspace` into a buffer the caller owns (include/nfp.h: nfp_workspace_bytes / nfp_workspace_init) and read
// with coalesced loads at kernel start.
//
// Sections (byte offsets from ws_layout; all 16-byte aligned):
//   lnk   [P*K2][LW] u16   backward: the (pixel, tap) pairs that LINK pixel r and the pixel t under window slot j of
//                          r — entry = n*P + p of the pair (its index into grad_out / out of one image), bit 15 set
//                          when r is that pair's NEIGHBOUR (the pair's centre is t); 0xFFFF ends the list.  The
//                          centre slot's row holds the pixel's two tap masks instead (32 bits each, see `mask`)
//   tq    [P*K2]     u16   pixel t under slot j of r, 0xFFFF outside the image
//   mask  [P][2]     u32   bit n set: tap n of pixel p reads zero padding / reads p itself (reflect on tiny maps)
//   ft    [N*P]      u32   forward: output (n, p) = q | pix << 9 | fi << 18 | kind << 22 — neighbour pixel q, the
//                          half-stencil entry Tt[fi][pix] holding its pair sum (pix = the pair's forward end, p
//                          itself for the other kinds), kind 0 pair / 1 self / 2 zero pad
//   boff  [P][BR]    i16   backward: LDS float4-slot offset of window slot j from the pixel's own slot (0 outside)
//   foff  [P][FR]    i16   forward: the same for the NF forward directions
#pragma once
#include "nfp_measures.h"

namespace nfp {

// Window / tap set of a kernel instantiation.  R = 1, 2: one radius, N = K*K - 1 taps in row-major order with the
// centre skipped (nfp.py:64-67).  R = 12: radii 1 AND 2 from one pass (the multi-radius head, nfp_heads.py:80-118,
// concatenates NFP(R=1, padding=1) and NFP(R=2, padding=2) of the same map): window of radius 2, 8 + 24 taps, the 8
// taps of the inner radius first — the order of torch.cat([nfp_R1(x), nfp_R2(x)], dim=1).
template <int R>
struct Win {
  static constexpr int RAD = R, K = 2 * R + 1, K2 = K * K, N = K2 - 1, NF = N / 2, NI = 0;
};
template <>
struct Win<12> {
  static constexpr int RAD = 2, K = 5, K2 = 25, N = 32, NF = 12, NI = 8;  // NI: taps of the inner radius (1)
};
// (dy, dx) of tap n
template <int RS>
__host__ __device__ inline void tap_offset(int n, int& dy, int& dx) {
  constexpr int NI = Win<RS>::NI;
  if (n < NI) {  // inner radius 1
    const int tp = n + (n >= 4 ? 1 : 0);
    dy = tp / 3 - 1;
    dx = tp % 3 - 1;
  } else {
    constexpr int K = Win<RS>::K, RAD = Win<RS>::RAD;
    const int m = n - NI, tp = m + (m >= (K * K) / 2 ? 1 : 0);
    dy = tp / K - RAD;
    dx = tp % K - RAD;
  }
}

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

// Per-thread neighbour maps for stride 1 / dilation 1 / pad R: my[k] / mx[k] = mapped row / column
// of kernel tap k for the thread's pixel (branch-free, computed once); any neighbour n is then two
// register selects.
template <int R>
struct NbrMap {
  static constexpr int K = 2 * R + 1;
  int my[K], mx[K];
  __device__ __forceinline__ void init(const KP& g, int py, int px) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      my[k] = map_index_bf(py + k - R, g.H, g.mode);
      mx[k] = map_index_bf(px + k - R, g.W, g.mode);
    }
  }
  __device__ __forceinline__ int get(const KP& g, int n, int& qy, int& qx) const {
    const int tp = n + (n >= (K * K) / 2 ? 1 : 0);
    const int ky = tp / K, kx = tp - ky * K;  // K is a compile-time constant
    qy = my[0];
    qx = mx[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
      qy = ky == k ? my[k] : qy;
      qx = kx == k ? mx[k] : qx;
    }
    return (qy < 0 || qx < 0) ? -1 : qy * g.W + qx;
  }
};

// LDS slot of pixel p inside a channel-quad row (row stride Pp = P rounded up to 4 slots): pixels are
// rotated inside their group of four by (p >> 3) & 3.  The NCHW staging writes a 4-pixel block per
// lane, i.e. lanes 64 B apart; unrotated that is a 4-way bank conflict on every ds_write_b128
// (measured: half of all LDS cycles); rotated, each 8-lane write group covers all 32 banks.  Reads
// of consecutive pixels stay conflict-free (a permutation inside each 64-byte group).
__device__ __forceinline__ int swz(int p) { return (p & ~3) | (((p & 3) + (p >> 3)) & 3); }


// ---- workspace layout (host and device agree through this one function) ----------------------------------------
struct WsLayout {
  int LW, BR, FR;
  long long lnk, tq, mask, ft, boff, foff, bytes;
};
__host__ __device__ inline long long ws_up16(long long v) { return (v + 15) & ~15LL; }
// Links per (pixel, slot): an in-image pair is linked by one tap of each end, plus the taps the padding folds onto
// it.  Zeros: 2.  Reflect: each axis folds at most 2 taps together, <= 4 + 4.  Replicate with R = 2 folds up to 3
// per axis at a corner and 5 on a one-pixel-wide map: <= 15.  (nfp_workspace_bytes checks the bound by enumeration.)
// Both radii together: the links of both tap sets, <= 16 for zeros / reflect (replicate would need 23: no tables).
__host__ __device__ inline int ws_link_width(int mode, int rs) {
  return (rs == 12 || (mode == NFP_PAD_REPLICATE && rs >= 2)) ? 16 : 8;
}
__host__ __device__ inline WsLayout ws_layout(int P, int rs, int mode) {  // rs: radius spec (1, 2, or 12 = 1 and 2)
  const int R = rs == 12 ? 2 : rs, K = 2 * R + 1, K2 = K * K, N = K2 - 1 + (rs == 12 ? 8 : 0), NF = (K2 - 1) / 2;
  WsLayout L;
  L.LW = ws_link_width(mode, rs);
  L.BR = (K2 + 7) & ~7;
  L.FR = (NF + 7) & ~7;
  long long o = 0;
  L.lnk = o;  o = ws_up16(o + (long long)P * K2 * L.LW * 2);
  L.tq = o;   o = ws_up16(o + (long long)P * K2 * 2);
  L.mask = o; o = ws_up16(o + (long long)P * 8);
  L.ft = o;   o = ws_up16(o + (long long)N * P * 4);
  L.boff = o; o = ws_up16(o + (long long)P * L.BR * 2);
  L.foff = o; o = ws_up16(o + (long long)P * L.FR * 2);
  L.bytes = o;
  return L;
}

// One launch fills every section (grid-stride over the largest, P*K2 items).  Stride 1, dilation 1, pad = R.
template <int RS>
__global__ void __launch_bounds__(256) fill_workspace(const KP g, unsigned char* __restrict__ ws) {
  constexpr int R = Win<RS>::RAD, K = Win<RS>::K, K2 = Win<RS>::K2, N = Win<RS>::N, NF = Win<RS>::NF;
  const WsLayout L = ws_layout(g.P, RS, g.mode);
  uint16_t* lnk = (uint16_t*)(ws + L.lnk);
  uint16_t* tq = (uint16_t*)(ws + L.tq);
  uint32_t* mask = (uint32_t*)(ws + L.mask);
  uint32_t* ft = (uint32_t*)(ws + L.ft);
  int16_t* boff = (int16_t*)(ws + L.boff);
  int16_t* foff = (int16_t*)(ws + L.foff);
  const int P = g.P, W = g.W, H = g.H;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  auto nbr = [&](int p, int n) {  // mapped neighbour pixel of tap n of pixel p, -1 = zero padding
    int dy, dx;
    tap_offset<RS>(n, dy, dx);
    const int y = map_index(p / W + dy, H, g.mode), x = map_index(p % W + dx, W, g.mode);
    return (y < 0 || x < 0) ? -1 : y * W + x;
  };
  for (int e = t0; e < P * K2; e += stride) {
    const int r = e / K2, j = e - r * K2;
    const int ty = r / W + j / K - R, tx = r % W + j % K - R;
    const bool inside = ty >= 0 && ty < H && tx >= 0 && tx < W;
    const int t = ty * W + tx;
    tq[e] = inside ? (uint16_t)t : (uint16_t)0xFFFF;
    uint16_t* row = lnk + (long long)e * L.LW;
    int cnt = 0;
    if (inside && t != r) {
      for (int n = 0; n < N && cnt < L.LW; ++n)
        if (nbr(r, n) == t) row[cnt++] = (uint16_t)(n * P + r);
      for (int n = 0; n < N && cnt < L.LW; ++n)
        if (nbr(t, n) == r) row[cnt++] = (uint16_t)((n * P + t) | 0x8000);
    }
    for (; cnt < L.LW; ++cnt) row[cnt] = 0xFFFF;
    if (t == r) {  // the centre slot has no links: its row carries the pixel's two tap masks (as in `mask`)
      uint32_t zm = 0, sm = 0;
      for (int n = 0; n < N; ++n) {
        const int q = nbr(r, n);
        zm |= (q < 0 ? 1u : 0u) << n;
        sm |= (q == r ? 1u : 0u) << n;
      }
      row[0] = (uint16_t)(zm & 0xFFFFu);
      row[1] = (uint16_t)(zm >> 16);
      row[2] = (uint16_t)(sm & 0xFFFFu);
      row[3] = (uint16_t)(sm >> 16);
    }
  }
  for (int p = t0; p < P; p += stride) {
    uint32_t zm = 0, sm = 0;
    for (int n = 0; n < N; ++n) {
      const int q = nbr(p, n);
      zm |= (q < 0 ? 1u : 0u) << n;
      sm |= (q == p ? 1u : 0u) << n;
    }
    mask[2 * p] = zm;
    mask[2 * p + 1] = sm;
    const int py = p / W, px = p % W;
    for (int j = 0; j < L.BR; ++j) {
      const int dy = j / K - R, dx = j % K - R;
      const bool ok = j < K2 && py + dy >= 0 && py + dy < H && px + dx >= 0 && px + dx < W;
      boff[p * L.BR + j] = ok ? (int16_t)(swz(p + dy * W + dx) - swz(p)) : (int16_t)0;
    }
    for (int d = 0; d < L.FR; ++d) {
      int dy = 0, dx = 0;
      if (d < NF) fdir<R>(d, dy, dx);
      const bool ok = d < NF && px + dx >= 0 && px + dx < W && py + dy < H;
      foff[p * L.FR + d] = ok ? (int16_t)(swz(p + dy * W + dx) - swz(p)) : (int16_t)0;
    }
  }
  for (int i = t0; i < N * P; i += stride) {
    const int n = i / P, p = i - n * P;
    const int q = nbr(p, n), qc = max(q, 0);
    const int dy = qc / W - p / W, dx = qc % W - p % W;
    const bool fwd = dy > 0 || (dy == 0 && dx > 0);
    const bool pair = q >= 0 && q != p;
    const int fi = pair ? (fwd ? fidx<R>(dy, dx) : fidx<R>(-dy, -dx)) : 0;
    const int kind = q < 0 ? 2 : (q == p ? 1 : 0);
    const int pix = kind == 0 ? (fwd ? p : qc) : p;  // the pixel whose band writes the output: the pair's forward end
    ft[i] = (uint32_t)qc | ((uint32_t)pix << 9) | ((uint32_t)fi << 18) | ((uint32_t)kind << 22);
  }
}

}  // namespace nfp
