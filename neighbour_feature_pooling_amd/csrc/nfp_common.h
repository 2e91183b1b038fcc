// nfp_common.h — shared device/host definitions for the gfx950 NFP kernels.
//
// Reference being replaced: models/pooling/nfp.py::NFPPooling (nfp.py:15-375).
// The reference materialises [B,C*N,H',W'] with two frozen one-hot depthwise
// convs (nfp.py:42-82) and reduces over C with ATen ops; here the neighbour
// gather is an index map into an LDS-resident slab of x and nothing of size
// [B,C,N,...] ever exists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nfp.h"

namespace nfp {

// Kernel parameter block (by value in kernarg).
struct KP {
  int B, C, H, W, P;  // P = H*W input pixels per channel
  int R, k, N, pad, stride, dil, mode;
  int rs;  // radius spec of the hot-path kernels (nfp_tables.h::Win): R, or 12 = radii 1 and 2 from one pass
  int Ho, Wo, O;  // O = Ho*Wo outputs per neighbour map
  int measure, similarity, diff, dtype;
  int godtype;  // dtype of the grad_out the backward kernel reads (f32 scratch for Attention)
  int odtype;   // dtype the general forward kernels store `out` in (f32 scratch for Attention's raw dots)
  float p, eps, q_scs;
  long long sB, sC, sH, sW;  // element strides of x; grad_x shares sC / sH / sW
  long long gB;              // batch stride of grad_x (a batch-strided view of x gets a dense gradient)
  const unsigned char* ws;   // the descriptor's constant tables (nfp_tables.h) or null
  int contig;                // x is NCHW-contiguous
  // launch shape
  int Cc;   // channels per LDS chunk
  int G;    // fwd generic: channel lanes per output (power of two <= 64)
  int Ow;   // outputs per workgroup (fwd) / per batch (bwd)
  int Cwg;  // bwd: channels per workgroup
  int Tc;   // bwd generic: channel lanes per output
  int early;  // bwd_fast: the x slab lies beside the pair values (not over them): committed during phase A
  // host-computed reciprocals (hot path: exact small-integer division by float multiply)
  float invP, invW, invNQ, invPT, inv_eps;
  // The hot-path kernels are instantiated for two per-channel forms — products (Cosine) and squared differences (L2) —
  // and serve two more measures each through these run-time constants (make_kp), with no further instantiations:
  //   out = osa * s + osb      products:   Cosine s, 1 - s (nfp.py:156-158); DotProduct (nfp.py:161-170): unit = 1 — the
  //                                         norm factors are 1 and nothing pulls on |x| (no diagonal term)
  //   out = osa * sqrt(d2 * d2s)  distances: Norm p=2 (nfp.py:141-148); RMSE (nfp.py:172-179): d2s = 1/C, and no
  //                                         subgradient at distance 0 (zero0 = 0: torch's sqrt gives inf * 0 = NaN there)
  float osa, osb, d2s;
  int unit, zero0;
  int gfc;         // GFC on the product kernels (below)
  float gf, ngf;   // (float)gfc and 1 - (float)gfc
  float uf, nuf;   // (float)unit and 1 - (float)unit: "1 if unit else v" as fmaf(v, nuf, uf) — exact, and not a branch on a
                   // wave-uniform flag in the latency-critical finalize
  // fused pooling tail (POOL kernels): which of its by-products the caller wants.  MobileNetV3_MultiStageNFP / MidNFP
  // consume adaptive_avg_pool2d(NFP(feat), 1) alone (texture_pooling.py:251-252, 320-321): no GAP(x) — pool_gap = 0
  // skips the slab sums (forward) and the adjoint of the mean (backward); without a backward to follow nobody reads
  // the maps themselves — pool_map = 0 skips their stores.
  int pool_gap, pool_map;
  int pf;   // fwd_band: several channel chunks — the next chunk's loads are issued under the current chunk's sums
  unsigned int* tickets;   // fused pooling tail, several row bands per image: one arrival counter per image (the first
                           // kTicketBytes of the descriptor's workspace, zero between launches) or null — see pool_last_band
};
constexpr int kTicketWords = 4096;              // (batches beyond it: no counters — make_kp)
constexpr int kTicketBytes = kTicketWords * 4;

// ---- single-launch combine of the row bands' pooled sums (round 4) -----------------------------------------------------
// The pooled outputs are sums over the whole image; with several workgroups (row bands) per image each band writes its
// share to a scratch row and SOMEBODY has to add the rows up.  Round 3 did that with a second launch (pool_fold: +3 us
// at the sizes where the pooled forward itself takes 6-30 us).  Here the band that ARRIVES LAST does it: every band,
// its scratch row written through to memory, draws a ticket from the image's counter; the band that draws
// nb - 1 knows every other row is complete and folds ALL rows in band order — so the result does not depend
// on which band was last: bitwise reproducible.  It then puts the counter back to 0 for the next launch.
// The counters live in the descriptor's workspace (nfp_workspace_init zeroes them): ONE pooled launch at a time per
// workspace — launches on one stream are ordered; a second stream needs a workspace of its own (include/nfp.h).
// Protocol (MI355X guide, "in-launch split-K reduction", the write-through form): the scratch rows are written with sc1
// (write-through) stores, so no release fence is needed — an agent-scope release writes back EVERY dirty line of the L2,
// and these kernels have just dirtied megabytes of maps: the first cut with fences ran 3x slower than two launches
// (profiles/r04_d_…) — every wave s_waitcnt vmcnt(0) -> barrier -> one lane: relaxed agent-scope fetch_add; the last
// arriver reads the rows with sc1 loads (no acquire fence either: pool_fold_image).
// `flag`: one LDS word nobody else uses between the two barriers below.  Returns (in every thread) whether this
// workgroup is the image's last.
__device__ __forceinline__ bool pool_last_band(unsigned int* ticket, int nb, int* flag, bool thread0) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (thread0) {
    const unsigned int old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == (unsigned int)(nb - 1);
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (all nb arrivals are in)
    *flag = last ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}
// write-through stores / cache-bypassing loads of the scratch rows (buffer instructions with sc1)
typedef unsigned int pool_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pool_rsrc(const float* base, long long floats) {
  const long long bytes = floats * 4;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0x7ffffff0LL ? 0x7ffffff0LL : bytes), 0x00020000);
}
__device__ __forceinline__ void pool_store1(__amdgpu_buffer_rsrc_t r, int i, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, i * 4, 0, 16);
}
__device__ __forceinline__ void pool_store4(__amdgpu_buffer_rsrc_t r, int i, float4 v) {
  const pool_u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, i * 4, 0, 16);
}
__device__ __forceinline__ float pool_load1(__amdgpu_buffer_rsrc_t r, int i) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, i * 4, 0, 16));
}
// gap[c] = (sum over the image's `rows` scratch rows of row[c]) / P, nfpm[n] likewise from row[C + n]: rows in order,
// eight loads in flight at a time.  gap == nullptr: the bands wrote no channel sums.
__device__ __forceinline__ void pool_fold_image(const float* __restrict__ rows0, int rows, int C, int N, float invP,
                                                float* __restrict__ gap_b, float* __restrict__ nfpm_b, int tid, int nthreads) {
  const int CN = C + N;
  const __amdgpu_buffer_rsrc_t pr = pool_rsrc(rows0, (long long)rows * CN);
  for (int k = (gap_b != nullptr ? 0 : C) + tid; k < CN; k += nthreads) {
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= rows; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pool_load1(pr, (j + u) * CN + k);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < rows; ++j) s += pool_load1(pr, j * CN + k);
    if (k < C)
      gap_b[k] = s * invP;
    else
      nfpm_b[k - C] = s * invP;
  }
}
__device__ __forceinline__ float unit_or(const KP& g, float v) { return fmaf(v, g.nuf, g.uf); }
// GFC (nfp.py:265-276): num / (|a| |b| + eps).  Its gradient has the shape of cosine's — per pair {g', g' f}, a cross
// weight and a pull on |x| — with other post-factors, so it rides on the product kernels too (round 3).  With the
// per-pixel factor F (cosine: 1 / max(|x|, eps); dot: 1; gfc: |x| itself):
//   cross weight of a pair (r, t)          cosine / dot: F_r F_t          gfc: 1 / (F_r F_t + eps)
//   what a pair adds to r's diagonal sum   cosine / dot: 1                gfc: F_t / (F_r F_t + eps)
// (then times dfn[r]: cosine -1 / (|x| max(|x|, eps)), dot 0, gfc -1 / |x|).  Blended by fmaf on gf = (float)gfc, not
// selected: a select on a wave-uniform flag is a branch.
__device__ __forceinline__ float cross_f(const KP& g, float fr, float ft) {
  return fmaf(__builtin_amdgcn_rcpf(fmaf(fr, ft, g.eps)), g.gf, fr * ft * g.ngf);
}
__device__ __forceinline__ float diag_f(const KP& g, float fr, float ft) {
  return fmaf(ft * __builtin_amdgcn_rcpf(fmaf(fr, ft, g.eps)), g.gf, g.ngf);
}
// forward: pair sum -> s.  cosine / dot: pair * ip * iq; gfc: pair / (sqrt(n2p) sqrt(n2q) + eps)
__device__ __forceinline__ float prod_value(const KP& g, float pairv, float n2p, float n2q, float ip, float iq) {
  const float gq = __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_sqrtf(n2p), __builtin_amdgcn_sqrtf(n2q), g.eps));
  return pairv * fmaf(gq, g.gf, ip * iq * g.ngf);
}
__device__ __forceinline__ float fin_prod(const KP& g, float s) { return fmaf(g.osa, s, g.osb); }
__device__ __forceinline__ float fin_dist(const KP& g, float d2) { return g.osa * __builtin_amdgcn_sqrtf(d2 * g.d2s); }
// per-pair backward scalar of a distance map: d out / d (a - b)[c] = coefficient * (a - b)[c]
__device__ __forceinline__ float dist_coef(const KP& g, float gc, float oc) {
  const float d = fabsf(oc);
  return (d == 0.f && g.zero0) ? 0.f : g.osa * g.d2s * gc * __builtin_amdgcn_rcpf(d);
}

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN kept a NaN: gfx950 converts in hardware (v_cvt_pk_bf16_f32; the integer formula with its
// NaN branch cost ~10 instructions and an exec-mask branch per element of every bf16 store)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f32_to_bf16x2(float lo, float hi) {  // one instruction: {lo, hi} packed
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float ldx(const void* x, long long i, int dtype) {
  return dtype == NFP_F32 ? ((const float*)x)[i] : bf16_to_f32(((const uint16_t*)x)[i]);
}
__device__ __forceinline__ void stx(void* x, long long i, float v, int dtype) {
  if (dtype == NFP_F32)
    ((float*)x)[i] = v;
  else
    ((uint16_t*)x)[i] = f32_to_bf16(v);
}

// nn.Conv2d padding_mode index map (reflect: -i / 2(n-1)-i; replicate: clamp; circular: wrap; zeros: -1 =
// the tap reads 0).  Branch-free: one select per mode.  make_kp guarantees pad < n (reflect) and pad <= n
// (circular), so a single fold / wrap is exact.  (Branchy index code is slow out of proportion: it runs once
// per kernel, from a cold instruction cache, and every taken branch is a fetch stall.)
__device__ __forceinline__ int map_index(int t, int n, int mode) {
  const bool lo = t < 0, hi = t >= n;
  const int refl = lo ? -t : (hi ? 2 * (n - 1) - t : t);
  const int repl = min(max(t, 0), n - 1);
  const int circ = lo ? t + n : (hi ? t - n : t);
  const int zero = (lo || hi) ? -1 : t;
  return mode == NFP_PAD_REFLECT ? refl : (mode == NFP_PAD_REPLICATE ? repl : (mode == NFP_PAD_CIRCULAR ? circ : zero));
}
__device__ __forceinline__ int map_index_bf(int t, int n, int mode) { return map_index(t, n, mode); }
// flat input pixel of kernel tap (ky,kx) for output o; -1 = zero padding
__device__ __forceinline__ int tap_pixel(const KP& g, int o, int ky, int kx) {
  int oy = o / g.Wo, ox = o - oy * g.Wo;
  int y = map_index(oy * g.stride + ky * g.dil - g.pad, g.H, g.mode);
  int x = map_index(ox * g.stride + kx * g.dil - g.pad, g.W, g.mode);
  return (y < 0 || x < 0) ? -1 : y * g.W + x;
}
// neighbour n (row-major, centre skipped: nfp.py:64-67) -> tap (ky,kx)
__device__ __forceinline__ int nbr_pixel(const KP& g, int o, int n) {
  int t = n < (g.k * g.k) / 2 ? n : n + 1;
  return tap_pixel(g, o, t / g.k, t - (t / g.k) * g.k);
}

// In-kernel phase stamps exist only in the diagnostic build (-DNFP_STAMPS, scripts/diag_stamps.py includes
// nfp_diag.h); in the product build the two macros expand to nothing.
#ifdef NFP_STAMPS
#include "nfp_diag.h"
#else
#define NFP_STAMP_INIT() do { } while (0)
#define NFP_STAMP(id) do { } while (0)
#endif

// matrix-core operand / accumulator types (v_mfma_f32_32x32x16_bf16: 8 bf16 per lane in, 16 f32 per lane out)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sgnf(float v) { return (float)((v > 0.f) - (v < 0.f)); }
// the same in three instructions (compare, select, bit-field insert of the sign) for the channel loops of Norm p = 1
__device__ __forceinline__ float sgn3(float v) { return __builtin_copysignf(v != 0.f ? 1.f : 0.f, v); }

// Per-channel arithmetic of the measures (term / grad, evaluated C*N times per output pixel): hardware
// reciprocal and square root (1 ulp) instead of the ~10-instruction IEEE sequences.  x/0, 0/0 and inf/inf
// give the same inf / NaN as a true division; every denominator they see is >= eps by construction.
__device__ __forceinline__ float frcp(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fsqrt(float v) { return __builtin_amdgcn_sqrtf(v); }

// floor(i / d) for 0 <= i, 1 <= d, i + d < 2^21: exact (the 1-ulp reciprocal and the product perturb (i + 0.5)/d
// by less than 0.5/d), in 4 instructions instead of the ~30 of an integer division.  Index arithmetic of
// once-per-kernel setup code is instruction-FETCH bound (cold I-cache), so code size there is time.
__device__ __forceinline__ int fdivi(int i, int d) {
  return (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}

// A workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also fences global memory: hipcc puts s_waitcnt
// vmcnt(0) in front of the barrier, and on gfx9 vmcnt counts STORES — a barrier right behind a kernel's output stores stalls
// every wavefront until its stores are acknowledged by memory (1-2 us).  The fused pooling tail sums the map values it
// has just stored (from an LDS copy): with this barrier its stores drain under the sums ([256,16,112,112] pooled forward
// headline 6.91 -> 6.55 us; the row-band kernels do not change: profiles/r04_o_…).  Use only where the code behind the barrier reads nothing the stores wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the G adjacent lanes of a group (G a power of two <= 32); valid in the group's LAST lane (in every lane of
// the group for G <= 16).  DPP lane exchanges: no LDS round trip, a fixed order.
#define NFP_DPP_ADD(v, ctrl, rows) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rows, 0xF, false))
__device__ __forceinline__ float group_sum(float v, int G) {
  if (G >= 2) NFP_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]: lane ^ 1
  if (G >= 4) NFP_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]: lane ^ 2
  if (G >= 8) NFP_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror: the other quad of each 8 lanes
  if (G >= 16) NFP_DPP_ADD(v, 0x140, 0xF);  // row_mirror: the other half of each row of 16
  if (G >= 32) NFP_DPP_ADD(v, 0x142, 0xA);  // row_bcast15 into rows 1 and 3: the row before
  return v;
}
#undef NFP_DPP_ADD

}  // namespace nfp
