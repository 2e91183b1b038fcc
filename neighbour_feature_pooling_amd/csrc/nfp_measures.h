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
  float k0, k1, k2, k3, k4;
};

template <int M>
struct Meas;

// ---- Norm: -(sum |a-b|^p)^(1/p)   nfp.py:141-148 (weights nfp.py:74-80) -----------------
// PK = 1 / 2: the order is known at compile time (the dispatcher picks these for p == 1, the reference's
// default, and p == 2), so the per-channel code has no powf and no branch on p; PK = 0: any finite p > 0.
template <int PK>
struct MeasNorm {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ bool is1(const KP& g) { return PK == 1 || (PK == 0 && g.p == 1.f); }
  __device__ static __forceinline__ bool is2(const KP& g) { return PK == 2 || (PK == 0 && g.p == 2.f); }
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    float v = g.diff ? a - b : b;
    if (is2(g)) return v * v;
    if (is1(g)) return fabsf(v);
    return powf(fabsf(v), g.p);
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    float d = is2(g) ? sqrtf(acc) : (is1(g) ? acc : powf(acc, 1.f / g.p));
    return g.similarity ? -d : d;
  }
  __device__ static __forceinline__ Coef coef(float go, float outv, float, float, float, float, const KP& g) {
    float d = fabsf(outv);
    float sg = g.similarity ? -go : go;
    Coef c = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (is1(g))
      c.k0 = sg;
    else if (is2(g))
      c.k0 = d == 0.f ? 0.f : sg / d;
    else
      c.k0 = d == 0.f ? 0.f : sg / powf(d, g.p - 1.f);
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float v = g.diff ? a - b : b;
    float t;
    if (is2(g))
      t = c.k0 * v;
    else if (is1(g))
      t = c.k0 * sgnf(v);
    else
      t = c.k0 * sgnf(v) * powf(fabsf(v), g.p - 1.f);
    da = g.diff ? t : 0.f;
    db = g.diff ? -t : t;
  }
};
constexpr int kNormP1 = NFP_MEASURE_COUNT + 1, kNormP2 = NFP_MEASURE_COUNT + 2;  // internal dispatch ids
// The measures that are a plain sum over channels of a SYMMETRIC per-channel term and keep no per-pixel statistic —
// Geman-McClure, Canberra, Hellinger, Jeffrey, squared chord, chi-squared 1 — share one instantiation of the row-band kernels (nfp_tile.h), which
// picks term / fin / coef / grad by the descriptor's measure (a wave-uniform switch per channel quad: sym_switch, at the end
// of this file).  Like Norm p = 1: the gradient is a per-channel function of (x_r, x_t), not linear in x.
constexpr int kSymTerm = NFP_MEASURE_COUNT + 3;
template <>
struct Meas<NFP_NORM> : MeasNorm<0> {};
template <>
struct Meas<kNormP1> : MeasNorm<1> {};
template <>
struct Meas<kNormP2> : MeasNorm<2> {};

// ---- Cosine: sum (a/max(|a|,eps)) (b/max(|b|,eps))   nfp.py:150-159 ----------------------
// Backward (torch 2.10 clamps the norms in place under NoGradGuard, so the
// gradient still flows through |a| with the clamped VALUE):
//   d s/d a = b^/m_a - s * a / (|a| m_a),  m = max(|a|,eps), 0 for the 2nd term when |a| = 0.
template <>
struct Meas<NFP_COSINE> {
  static constexpr int NSTAT = 1;  // saved[0] = |a|
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return a * b; }
  __device__ static __forceinline__ void stat(float a, float& s0, float&) { s0 = fmaf(a, a, s0); }
  __device__ static __forceinline__ float fin(float acc, float sa0, float, float sb0, float, const KP& g) {
    float ia = 1.f / fmaxf(sqrtf(sa0), g.eps), ib = 1.f / fmaxf(sqrtf(sb0), g.eps);
    float s = acc * ia * ib;
    return g.similarity ? s : 1.f - s;
  }
  // accumulated stats -> what is stored per pixel for backward
  __device__ static __forceinline__ float save0(float s0, float, const KP&) { return sqrtf(s0); }
  __device__ static __forceinline__ float save1(float, float, const KP&) { return 0.f; }
  __device__ static __forceinline__ Coef coef(float go, float outv, float np, float, float nq, float, const KP& g) {
    float s = g.similarity ? outv : 1.f - outv;
    float sg = g.similarity ? go : -go;
    float ip = 1.f / fmaxf(np, g.eps), iq = 1.f / fmaxf(nq, g.eps);
    float rp = np > 0.f ? 1.f / np : 0.f, rq = nq > 0.f ? 1.f / nq : 0.f;
    Coef c;
    c.k0 = sg * ip * iq;      // cross term
    c.k1 = sg * s * rp * ip;  // centre self term
    c.k2 = sg * s * rq * iq;  // neighbour self term
    c.k3 = c.k4 = 0.f;
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    da = c.k0 * b - c.k1 * a;
    db = c.k0 * a - c.k2 * b;
  }
};


// Helper: sign convention shared by the "distance" measures (negated when similarity=True).
__device__ __forceinline__ float dist_sign(const KP& g, float go) { return g.similarity ? -go : go; }

// ---- DotProduct   nfp.py:161-170 ------------------------------------------------------------
template <>
struct Meas<NFP_DOT> {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return a * b; }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    return g.similarity ? acc : -acc;
  }
  __device__ static __forceinline__ Coef coef(float go, float, float, float, float, float, const KP& g) {
    Coef c = {g.similarity ? go : -go, 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    da = c.k0 * b;
    db = c.k0 * a;
  }
};

// ---- RMSE   nfp.py:172-179 (conv weights as Norm: nfp.py:74-80) --------------------------------
template <>
struct Meas<NFP_RMSE> {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    float v = g.diff ? a - b : b;
    return v * v;
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    float d = sqrtf(acc / (float)g.C);
    return g.similarity ? -d : d;
  }
  __device__ static __forceinline__ Coef coef(float go, float outv, float, float, float, float, const KP& g) {
    // d sqrt(m)/dm * dm/dv = v / (d*C); d == 0 gives inf here and inf*0 = NaN below, exactly as torch
    Coef c = {dist_sign(g, go) / (fabsf(outv) * (float)g.C), 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float t = c.k0 * (g.diff ? a - b : b);
    da = g.diff ? t : 0.f;
    db = g.diff ? -t : t;
  }
};

// ---- Geman-McClure   nfp.py:181-193 -----------------------------------------------------------
template <>
struct Meas<NFP_GEMAN> {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    float q = (a - b) * (a - b);
    return fdiv(q, q + g.eps);
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    float f = acc / (float)g.C;
    return g.similarity ? f : 1.f - f;
  }
  __device__ static __forceinline__ Coef coef(float go, float, float, float, float, float, const KP& g) {
    Coef c = {(g.similarity ? go : -go) / (float)g.C, 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float v = a - b, iq = frcp(v * v + g.eps);
    da = c.k0 * 2.f * v * g.eps * iq * iq;
    db = -da;
  }
};

// ---- EMD (L1)   nfp.py:207-216 ----------------------------------------------------------------
template <>
struct Meas<NFP_EMD> {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return fabsf(a - b); }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    return g.similarity ? -acc : acc;
  }
  __device__ static __forceinline__ Coef coef(float go, float, float, float, float, float, const KP& g) {
    Coef c = {dist_sign(g, go), 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    da = c.k0 * sgnf(a - b);
    db = -da;
  }
};

// ---- Canberra   nfp.py:218-227 ----------------------------------------------------------------
template <>
struct Meas<NFP_CANBERRA> {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    return fdiv(fabsf(a - b), fabsf(a) + fabsf(b) + g.eps);
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    return g.similarity ? -acc : acc;
  }
  __device__ static __forceinline__ Coef coef(float go, float, float, float, float, float, const KP& g) {
    Coef c = {dist_sign(g, go), 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float u = fabsf(a - b), iw = frcp(fabsf(a) + fabsf(b) + g.eps), s = sgnf(a - b);
    da = c.k0 * iw * (s - u * iw * sgnf(a));
    db = c.k0 * iw * (-s - u * iw * sgnf(b));
  }
};

// ---- Hellinger nfp.py:229-241 / SquaredChord nfp.py:310-324 -------------------------------------
template <int WHICH>
struct MeasChord {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    float r = fsqrt(fabsf(a) + g.eps) - fsqrt(fabsf(b) + g.eps);
    return r * r;
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    float f = WHICH == NFP_HELLINGER ? sqrtf(0.5f * acc) : acc;
    return g.similarity ? -f : f;
  }
  __device__ static __forceinline__ Coef coef(float go, float outv, float, float, float, float, const KP& g) {
    // Hellinger: d sqrt(S/2)/dS = 0.25/f (inf at f == 0 -> NaN below, as torch)
    Coef c = {dist_sign(g, go) * (WHICH == NFP_HELLINGER ? 0.25f / fabsf(outv) : 1.f), 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float ra = fsqrt(fabsf(a) + g.eps), rb = fsqrt(fabsf(b) + g.eps);
    da = c.k0 * fdiv(ra - rb, ra) * sgnf(a);
    db = c.k0 * fdiv(rb - ra, rb) * sgnf(b);
  }
};
template <>
struct Meas<NFP_HELLINGER> : MeasChord<NFP_HELLINGER> {};
template <>
struct Meas<NFP_SQUAREDCHORD> : MeasChord<NFP_SQUAREDCHORD> {};

// ---- Chi-squared 1 / 2   nfp.py:243-263 ---------------------------------------------------------
template <int WHICH>
struct MeasChi {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float wden(float a, float b, const KP& g) {
    return fabsf(a) + (WHICH == NFP_CHISQUARED1 ? fabsf(b) : 0.f) + g.eps;
  }
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    float v = a - b;
    return fdiv(v * v, wden(a, b, g));
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    return g.similarity ? -acc : acc;
  }
  __device__ static __forceinline__ Coef coef(float go, float, float, float, float, float, const KP& g) {
    Coef c = {dist_sign(g, go), 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float v = a - b, iw = frcp(wden(a, b, g)), t = v * iw;
    da = c.k0 * (2.f * t - t * t * sgnf(a));
    db = c.k0 * (-2.f * t - (WHICH == NFP_CHISQUARED1 ? t * t * sgnf(b) : 0.f));
  }
};
template <>
struct Meas<NFP_CHISQUARED1> : MeasChi<NFP_CHISQUARED1> {};
template <>
struct Meas<NFP_CHISQUARED2> : MeasChi<NFP_CHISQUARED2> {};

// ---- GFC   nfp.py:265-276: dot / (|a||b| + eps) ---------------------------------------------------
template <>
struct Meas<NFP_GFC> {
  static constexpr int NSTAT = 1;  // saved[0] = |a|
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return a * b; }
  __device__ static __forceinline__ void stat(float a, float& s0, float&) { s0 = fmaf(a, a, s0); }
  __device__ static __forceinline__ float fin(float acc, float sa0, float, float sb0, float, const KP& g) {
    float f = acc / (sqrtf(sa0) * sqrtf(sb0) + g.eps);
    return g.similarity ? f : -f;
  }
  __device__ static __forceinline__ float save0(float s0, float, const KP&) { return sqrtf(s0); }
  __device__ static __forceinline__ float save1(float, float, const KP&) { return 0.f; }
  __device__ static __forceinline__ Coef coef(float go, float outv, float np, float, float nq, float, const KP& g) {
    const float sg = g.similarity ? 1.f : -1.f;
    const float den = np * nq + g.eps, num = sg * outv * den;
    Coef c;
    c.k0 = go * sg / den;
    c.k1 = np > 0.f ? go * sg * num * nq / (den * den * np) : 0.f;
    c.k2 = nq > 0.f ? go * sg * num * np / (den * den * nq) : 0.f;
    c.k3 = c.k4 = 0.f;
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    da = c.k0 * b - c.k1 * a;
    db = c.k0 * a - c.k2 * b;
  }
};

// ---- Pearson   nfp.py:278-293 -------------------------------------------------------------------
template <>
struct Meas<NFP_PEARSON> {
  static constexpr int NSTAT = 2;  // saved[0] = mean(a), saved[1] = sum (a-mean)^2
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return a * b; }
  __device__ static __forceinline__ void stat(float a, float& s0, float& s1) {
    s0 += a;
    s1 = fmaf(a, a, s1);
  }
  __device__ static __forceinline__ float fin(float acc, float sa0, float sa1, float sb0, float sb1, const KP& g) {
    const float C = (float)g.C, ma = sa0 / C, mb = sb0 / C;
    const float sab = acc - C * ma * mb, saa = sa1 - C * ma * ma, sbb = sb1 - C * mb * mb;
    const float f = sab / sqrtf(saa * sbb + g.eps);
    return g.similarity ? f : -f;
  }
  __device__ static __forceinline__ float save0(float s0, float, const KP& g) { return s0 / (float)g.C; }
  __device__ static __forceinline__ float save1(float s0, float s1, const KP& g) { return s1 - s0 * s0 / (float)g.C; }
  __device__ static __forceinline__ Coef coef(float go, float outv, float mp, float saa, float mq, float sbb,
                                              const KP& g) {
    const float sg = g.similarity ? 1.f : -1.f;
    const float den = sqrtf(saa * sbb + g.eps), sab = sg * outv * den, d3 = den * den * den;
    Coef c;
    c.k0 = sg * go / den;
    c.k1 = sg * go * sab * sbb / d3;
    c.k2 = sg * go * sab * saa / d3;
    c.k3 = mp;
    c.k4 = mq;
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    const float ac = a - c.k3, bc = b - c.k4;
    da = c.k0 * bc - c.k1 * ac;
    db = c.k0 * ac - c.k2 * bc;
  }
};

// ---- Jeffrey   nfp.py:295-308 -------------------------------------------------------------------
template <>
struct Meas<NFP_JEFFREY> {
  static constexpr int NSTAT = 0;
  __device__ static __forceinline__ float term(float a, float b, const KP& g) {
    // ca*log(ca/cb) + cb*log(cb/ca) = (ca - cb)*log(ca/cb): one logarithm, one reciprocal
    float ca = fabsf(a) + g.eps, cb = fabsf(b) + g.eps;
    return (ca - cb) * logf(fdiv(ca, cb));
  }
  __device__ static __forceinline__ void stat(float, float&, float&) {}
  __device__ static __forceinline__ float fin(float acc, float, float, float, float, const KP& g) {
    return g.similarity ? -acc : acc;
  }
  __device__ static __forceinline__ Coef coef(float go, float, float, float, float, float, const KP& g) {
    Coef c = {dist_sign(g, go), 0.f, 0.f, 0.f, 0.f};
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP& g, float& da, float& db) {
    float ca = fabsf(a) + g.eps, cb = fabsf(b) + g.eps, r = fdiv(ca, cb), l = logf(r);
    da = c.k0 * (l + 1.f - fdiv(cb, ca)) * sgnf(a);
    db = c.k0 * (-l + 1.f - r) * sgnf(b);
  }
};

// ---- Smith   nfp.py:326-342 -----------------------------------------------------------------------
template <>
struct Meas<NFP_SMITH> {
  static constexpr int NSTAT = 1;  // saved[0] = sum |a|
  __device__ static __forceinline__ float term(float a, float b, const KP&) { return fminf(fabsf(a), fabsf(b)); }
  __device__ static __forceinline__ void stat(float a, float& s0, float&) { s0 += fabsf(a); }
  __device__ static __forceinline__ float fin(float acc, float sa0, float, float sb0, float, const KP& g) {
    float f = 1.f - acc / (fminf(sa0, sb0) + g.eps);
    return g.similarity ? f : -f;
  }
  __device__ static __forceinline__ float save0(float s0, float, const KP&) { return s0; }
  __device__ static __forceinline__ float save1(float, float, const KP&) { return 0.f; }
  __device__ static __forceinline__ Coef coef(float go, float outv, float sp, float, float sq, float, const KP& g) {
    const float sg = g.similarity ? 1.f : -1.f;
    const float mm = fminf(sp, sq) + g.eps, mn = (1.f - sg * outv) * mm;
    const float wa = sp < sq ? 1.f : (sp == sq ? 0.5f : 0.f);
    Coef c;
    c.k0 = sg * go / mm;                       // * -t (tie-split indicator of the element-wise minimum)
    c.k1 = sg * go * mn / (mm * mm) * wa;      // through min(sum|a|, sum|b|), centre side
    c.k2 = sg * go * mn / (mm * mm) * (1.f - wa);
    c.k3 = c.k4 = 0.f;
    return c;
  }
  __device__ static __forceinline__ void grad(float a, float b, const Coef& c, const KP&, float& da, float& db) {
    const float A = fabsf(a), B = fabsf(b);
    const float ta = A < B ? 1.f : (A == B ? 0.5f : 0.f);
    da = (-ta * c.k0 + c.k1) * sgnf(a);
    db = (-(1.f - ta) * c.k0 + c.k2) * sgnf(b);
  }
};

// Forward sums of a measure that subtracts channel means are taken about a per-pixel PIVOT (the pixel's
// channel-0 value, any constant works): sum a*b - C*mean(a)*mean(b) on raw values loses ~mean^2/variance
// of the float32 precision on post-ReLU features (measured 1.1e-5 relative on uniform(0.25,1.25) data,
// 1e-6 about the pivot).  save0 of such a measure is a mean: the kernel adds the pivot back.
template <int M> struct Pivot { static constexpr bool v = false; };
template <> struct Pivot<NFP_PEARSON> { static constexpr bool v = true; };

// ---- Attention   nfp.py:195-205: the dot products go through DotProduct's kernels; the softmax over
// the N neighbours and its Jacobian are separate tiny kernels (nfp_direct.h).
template <>
struct Meas<NFP_ATTENTION> : Meas<NFP_DOT> {};

// kSymTerm: the descriptor's measure -> its Meas<> (wave-uniform)
template <typename F>
__device__ __forceinline__ void sym_switch(int measure, F&& f) {
  switch (measure) {
    case NFP_GEMAN: f(Meas<NFP_GEMAN>{}); break;
    case NFP_CANBERRA: f(Meas<NFP_CANBERRA>{}); break;
    case NFP_SQUAREDCHORD: f(Meas<NFP_SQUAREDCHORD>{}); break;
    case NFP_HELLINGER: f(Meas<NFP_HELLINGER>{}); break;
    case NFP_JEFFREY: f(Meas<NFP_JEFFREY>{}); break;
    default: f(Meas<NFP_CHISQUARED1>{}); break;
  }
}

}  // namespace nfp
