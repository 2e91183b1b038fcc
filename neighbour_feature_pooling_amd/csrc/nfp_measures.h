// nfp_measures.h — per-measure arithmetic (f32) shared by every kernel.
//
// A measure (nfp.py:141-374) is described by
//   term(a,b)   what is summed over channels for one (centre, neighbour) pair
//   stat(a)     per-PIXEL channel sums it also needs (norms, means, ...)
//   fin(...)    the output value from those sums
//   coef(...)   per-pair scalars of the backward, from grad_out, the saved
//               output and the saved per-pixel stats
//   grad(...)   d out / d a[c], d out / d b[c] for one channel
// so that forward = one pass of sums over x and backward = one pass of x with
// per-pair scalars; the [B,C,N,H,W] tensors of the reference are never formed.
#pragma once
#include "nfp_common.h"

namespace nfp {

struct Coef {
  float k0, k1, k2, k3;
};

template <int M>
struct Meas;

// ---- Norm: -(sum |a-b|^p)^(1/p)   nfp.py:141-148 (weights nfp.py:74-80) -----------------
template <>
struct Meas<NFP_NORM> {
  static constexpr int NSTAT = 0;
  static constexpr bool kLinear = false;  // linear in x only for p == 2 (fast path checks p)
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    float v = g.diff ? a - b : b;
    if (g.p == 2.f) return v * v;
    if (g.p == 1.f) return fabsf(v);
    return powf(fabsf(v), g.p);
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    float d = g.p == 2.f ? sqrtf(acc) : (g.p == 1.f ? acc : powf(acc, 1.f / g.p));
    return g.similarity ? -d : d;
  }
  __device__ static __forceinline__ Coef coef(float go, float outv, float, float, float, float, const KP& g) {
    float d = fabsf(outv);
    float sg = g.similarity ? -go : go;
    Coef c = {0.f, 0.f, 0.f, 0.f};
    if (g.p == 1.f)
      c.k0 = sg;
    else if (g.p == 2.f)
      c.k0 = d == 0.f ? 0.f : sg / d;
    else
      c.k0 = d == 0.f ? 0.f : sg / powf(d, g.p - 1.f);
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float v = g.diff ? a - b : b;
    float t;
    if (g.p == 2.f)
      t = c.k0 * v;
    else if (g.p == 1.f)
      t = c.k0 * sgnf(v);
    else
      t = c.k0 * sgnf(v) * powf(fabsf(v), g.p - 1.f);
    da = g.diff ? t : 0.f;
    db = g.diff ? -t : t;
  }
};

// ---- Cosine: sum (a/max(|a|,eps)) (b/max(|b|,eps))   nfp.py:150-159 ----------------------
// Backward (torch 2.10 clamps the norms in place under NoGradGuard, so the
// gradient still flows through |a| with the clamped VALUE):
//   d s/d a = b^/m_a - s * a / (|a| m_a),  m = max(|a|,eps), 0 for the 2nd term when |a| = 0.
template <>
struct Meas<NFP_COSINE> {
  static constexpr int NSTAT = 1;  // saved[0] = |a|
  static constexpr bool kLinear = true;
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return a * b; }
  __device__ static __forceinline__ void stat(float a, float& s0, float&) { s0 = fmaf(a, a, s0); }
  __device__ static __forceinline__ float fin(float acc, float sa0, float, float sb0, float, const KP& g) {
    float ia = 1.f / fmaxf(sqrtf(sa0), g.eps), ib = 1.f / fmaxf(sqrtf(sb0), g.eps);
    float s = acc * ia * ib;
    return g.similarity ? s : 1.f - s;
  }
  // saved stat -> what is stored for backward
  __device__ static __forceinline__ float save0(float s0) { return sqrtf(s0); }
  __device__ static __forceinline__ Coef coef(float go, float outv, float np, float, float nq, float, const KP& g) {
    float s = g.similarity ? outv : 1.f - outv;
    float sg = g.similarity ? go : -go;
    float ip = 1.f / fmaxf(np, g.eps), iq = 1.f / fmaxf(nq, g.eps);
    float rp = np > 0.f ? 1.f / np : 0.f, rq = nq > 0.f ? 1.f / nq : 0.f;
    Coef c;
    c.k0 = sg * ip * iq;      // cross term
    c.k1 = sg * s * rp * ip;  // centre self term
    c.k2 = sg * s * rq * iq;  // neighbour self term
    c.k3 = 0.f;
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    da = c.k0 * b - c.k1 * a;
    db = c.k0 * a - c.k2 * b;
  }
};

}  // namespace nfp
