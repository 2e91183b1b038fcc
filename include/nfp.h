/*
 * nfp.h — C ABI of libnfp_hip.so, the MI355X (gfx950) Neighbourhood Feature
 * Pooling forward/backward.
 *
 * The reference has no FFI layer: its operator API is the Python class
 * models/pooling/nfp.py::NFPPooling (nfp.py:15-134).  This header is the
 * boundary a binding for that class calls; every entry point names the
 * reference code it stands in for.  Plain pointers and sizes only — no torch
 * types.  All tensor pointers are DEVICE pointers; the caller owns every
 * buffer (the library never allocates, frees or retains one), kernels are
 * enqueued on the caller's hipStream_t and nothing synchronises the device.
 * Re-entrant: calls from several threads (autograd runs the backward on its own
 * thread) share only a launch counter and the lock-protected "last variant"
 * string; the error string, the dispatcher's scratch and nfp_plan's output are
 * per calling thread.  No entry point reads the environment.
 *
 * Return value of every int function: 0 = ok, <0 = NFP_E_* below (message in
 * nfp_last_error(), thread-local).  Nothing throws or aborts.
 */
#ifndef NFP_H_
#define NFP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFP_ABI_VERSION 6

/* error codes */
#define NFP_OK 0
#define NFP_E_INVALID (-1)     /* malformed descriptor / null pointer          */
#define NFP_E_UNSUPPORTED (-2) /* valid for the reference, not built here yet  */
#define NFP_E_HIP (-3)         /* a HIP runtime call failed                    */

/* nfp.py:85-120 — the `measure` string, lower-cased, dispatches to one of
 * these (same order as the if/elif chain there; 'scs' aliases 16). */
enum nfp_measure {
  NFP_NORM = 0,         /* nfp.py:141-148  LA.norm(centre-neigh, ord=p, dim=C)      */
  NFP_COSINE = 1,       /* nfp.py:150-159  F.cosine_similarity(centre, neigh, eps)  */
  NFP_DOT = 2,          /* nfp.py:161-170                                           */
  NFP_RMSE = 3,         /* nfp.py:172-179                                           */
  NFP_GEMAN = 4,        /* nfp.py:181-193                                           */
  NFP_ATTENTION = 5,    /* nfp.py:195-205  softmax over the N neighbours            */
  NFP_EMD = 6,          /* nfp.py:207-216                                           */
  NFP_CANBERRA = 7,     /* nfp.py:218-227                                           */
  NFP_HELLINGER = 8,    /* nfp.py:229-241                                           */
  NFP_CHISQUARED1 = 9,  /* nfp.py:243-252                                           */
  NFP_CHISQUARED2 = 10, /* nfp.py:254-263                                           */
  NFP_GFC = 11,         /* nfp.py:265-276                                           */
  NFP_PEARSON = 12,     /* nfp.py:278-293                                           */
  NFP_JEFFREY = 13,     /* nfp.py:295-308                                           */
  NFP_SQUAREDCHORD = 14,/* nfp.py:310-324                                           */
  NFP_SMITH = 15,       /* nfp.py:326-342                                           */
  NFP_SCS = 16,         /* nfp.py:344-374  (batch-mixing broadcast)  */
  NFP_MEASURE_COUNT = 17
};

/* nn.Conv2d padding_mode of the two frozen depthwise convs (nfp.py:42-58). */
enum nfp_pad_mode { NFP_PAD_ZEROS = 0, NFP_PAD_REFLECT = 1, NFP_PAD_REPLICATE = 2, NFP_PAD_CIRCULAR = 3 };

enum nfp_dtype { NFP_F32 = 0, NFP_BF16 = 1 }; /* storage type of x / out / grads; arithmetic is f32 */

/*
 * One NFP call.  Mirrors NFPPooling.__init__ (nfp.py:16-39) plus the input
 * geometry forward() sees (nfp.py:132-134).  Strides are in ELEMENTS, so NCHW
 * and channels-last inputs are both read in place (no transpose kernel).
 */
typedef struct nfp_desc {
  int32_t B, C, H, W;        /* input  [B,C,H,W]                                        */
  int32_t R;                 /* radius; kernel_size k = 2R+1, N = k*k-1   nfp.py:38-39  */
  int32_t pad, stride, dilation; /* conv geometry                         nfp.py:42-47  */
  int32_t pad_mode;          /* enum nfp_pad_mode                                       */
  int32_t measure;           /* enum nfp_measure                                        */
  int32_t similarity;        /* nfp.py:29 — sign / (1-x) convention of each measure     */
  int32_t diff_weights;      /* 1 when the RAW measure string was in
                                ['norm','rmse','mahalanobis'] (nfp.py:74-76): the conv
                                yields centre-neighbour; 0: pure neighbour (nfp.py:79-80) */
  int32_t dtype;             /* enum nfp_dtype                                          */
  float p;                   /* nfp.py:30 — ord of Norm, exponent of SCS                */
  float eps;                 /* nfp.py:33                                               */
  float q_scs;               /* nfp.py:34                                               */
  int64_t sxB, sxC, sxH, sxW; /* element strides of x; grad_x shares sxC / sxH / sxW    */
  int64_t sgB;               /* batch stride of grad_x in elements, 0 = sxB.  A view whose
                                images are dense but spaced apart (ViT tokens behind a class
                                token, texture_pooling.py:181-188) is read in place and still
                                gets a dense gradient                                       */
  const void* ws;            /* DEVICE pointer to the descriptor's workspace (constant tables +
                                the pooled tail's arrival counters), filled by
                                nfp_workspace_init, or NULL.  The hot-path kernels (stride 1,
                                pad = R) read their index maps from it; without it the call is
                                served by the general kernels                               */
  int32_t inner_R;           /* 0, or 1 with R = 2: ALSO produce the maps of radius 1 (padding 1)
                                from the same pass — models/nfp_heads.py:80-118 concatenates
                                NFP(R=1, padding=1) and NFP(R=2, padding=2) of one feature map.
                                out / grad_out are then [B, 8 + 24, H, W], the 8 maps of radius 1
                                first (torch.cat order).  Hot-path descriptors only (cosine / L2,
                                stride 1, padding = R, workspace set)                      */
  int32_t reserved_;         /* 0                                                           */
} nfp_desc;

int nfp_abi_version(void);
const char* nfp_last_error(void);

/*
 * Workspace of a descriptor: what the kernels keep between launches.  Two parts:
 *  - constant tables.  What the reference re-derives inside every conv call — which input pixel each kernel tap reads
 *    under the padding mode (nn.Conv2d's padding_mode, nfp.py:42-58) — depends only on (H, W, R, pad, stride, dilation,
 *    pad_mode), not on the batch, the channels or the data.  The hot-path kernels for maps of up to 512 pixels read it
 *    from a table instead of recomputing it per launch;
 *  - (ABI 6) one arrival counter per image for the fused pooling tail with several row bands per image: the band that
 *    finishes last adds up every band's share of the pooled sums inside the same launch (no second launch).  The
 *    counters are zero between launches; ONE nfp_pool_forward at a time may use a workspace — launches on one stream are
 *    ordered, a second stream needs a workspace of its own.
 *   bytes = nfp_workspace_bytes(d)        0 = this descriptor has neither (it is served without a workspace)
 *   nfp_workspace_init(d, ws, stream)     enqueue the fill of `ws` (>= bytes, 16-byte aligned, caller-owned)
 * then set d->ws = ws for nfp_forward / nfp_backward / nfp_pool_*.  One buffer serves every descriptor that
 * differs only in B, C, measure, similarity, dtype, strides, p, eps (the tables do not depend on them).  Without it
 * (d->ws = NULL) every call is still served: maps of up to 512 pixels by the general kernels, the pooled tail by one
 * band per image or a second, tiny launch.
 */
int64_t nfp_workspace_bytes(const nfp_desc* d);
int nfp_workspace_init(const nfp_desc* d, void* ws, void* hip_stream);

/* out is [B, N, Ho, Wo] contiguous; Ho/Wo as nn.Conv2d computes them
 * (nfp.py:125-130 is the square-only helper of the same formula). */
int nfp_output_shape(const nfp_desc* d, int32_t* N, int32_t* Ho, int32_t* Wo);

/* Floats of per-call state forward() hands to backward() (the autograd
 * "saved tensors" that replace the [B,C,N,H,W] graph of nfp.py:152-156).
 * 0 when the measure needs none. */
int64_t nfp_saved_floats(const nfp_desc* d);

/* NFPPooling.forward (nfp.py:132-134) for the measure in d.
 *   x      [B,C,H,W] by strides, dtype d->dtype
 *   out    [B,N,Ho,Wo] contiguous, dtype d->dtype
 *   saved  float[nfp_saved_floats(d)] or NULL when no backward will follow
 *          (Attention on bf16 maps always needs it: the raw dots live there) */
int nfp_forward(const nfp_desc* d, const void* x, void* out, float* saved, void* hip_stream);

/* The autograd backward of the same call: grad_x = d(sum(out*grad_out))/dx.
 *   grad_out [B,N,Ho,Wo] contiguous;  out / saved as written by nfp_forward
 *   grad_x   [B,C,H,W] with the strides of x; fully overwritten */
int nfp_backward(const nfp_desc* d, const void* x, const void* grad_out, const void* out,
                 const float* saved, void* grad_x, void* hip_stream);

/*
 * Fused tail of models/NFP_Pooling.py:27-31 (the wrapper every live model uses): one pass over x yields
 *   gap  [B,C] f32 = AdaptiveAvgPool2d(1)(x)                          NFP_Pooling.py:27
 *   nfpm [B,N] f32 = adaptive_avg_pool2d(NFPPooling(x), 1)            NFP_Pooling.py:29-31
 * out_map [B,N,Ho,Wo] (dtype of x) is also written: the backward needs it, callers may ignore it.
 * ABI 6: `gap` may be NULL — MobileNetV3_MultiStageNFP / MidNFP consume adaptive_avg_pool2d(NFP(feat), 1) alone
 * (texture_pooling.py:251-252, 320-321): the channel sums are then skipped; `out_map` may be NULL when no backward will
 * follow (inference): the maps are then never stored.  nfp_pool_backward takes grad_gap = NULL likewise (GAP(x) took no
 * part in the loss): no adjoint of the mean is added.
 * Served only where nfp_pool_supported(d) != 0 — cosine / dot / gfc / L2 / rmse on "same" maps (stride 1, padding = R),
 * NCHW or channels-last, float32 or bf16: maps of at most 512 pixels with the descriptor's workspace set, larger maps
 * with rows of W <= 254 (k = 3) / W <= 142 (k = 5) pixels — (W + 2R)(3R + 1) <= 1024, one thread per padded position of
 * the smallest row band — and any height the descriptor allows (H <= 32767); the answer is a dry
 * run of both launchers, so a 1 means both nfp_pool_forward and nfp_pool_backward will launch.  Otherwise compose
 * nfp_forward with ordinary pooling.
 */
int nfp_pool_supported(const nfp_desc* d);
/* Floats of `saved` the fused tail needs: the per-pixel state of nfp_saved_floats plus, for maps served by the row-band
 * kernels (above 512 pixels), every band's share of the two pooled sums (joined in a fixed order by a second, tiny
 * launch).  nfp_pool_forward requires a non-NULL `saved` of at least this size (min 1 float). */
int64_t nfp_pool_saved_floats(const nfp_desc* d);
int nfp_pool_forward(const nfp_desc* d, const void* x, float* gap, float* nfpm, void* out_map, float* saved,
                     void* hip_stream);
/* grad_x = d( sum(gap*grad_gap) + sum(nfpm*grad_nfpm) ) / dx;  out_map / saved from nfp_pool_forward. */
int nfp_pool_backward(const nfp_desc* d, const void* x, const float* grad_gap, const float* grad_nfpm,
                      const void* out_map, const float* saved, void* grad_x, void* hip_stream);

/* Telemetry: kernels enqueued by this process so far (tests use it to prove
 * the HIP path, not a fallback, produced a result). */
uint64_t nfp_launch_count(void);

/* Name of the kernel variant the last successful nfp_forward / nfp_backward /
 * nfp_pool_* of this PROCESS selected (for bench / profile bookkeeping),
 * copied into a buffer of the calling thread. */
const char* nfp_last_variant(void);

/* Telemetry: the NEXT forward / backward kernel this PROCESS launches — from whichever thread: autograd runs the
 * backward on a thread of its own, so the arming cannot be per calling thread — is bracketed by the two hipEvent_t
 * (hipExtLaunchKernel: recorded at the kernel's own start and end on the device, as a profiler's kernel trace
 * would) — per-kernel durations without a profiler and without inter-kernel gaps.  One shot; pass NULLs to
 * disarm.  SINGLE-LAUNCHER USE ONLY: while it is armed no second thread may enqueue NFP work, or the events bracket
 * that thread's kernel instead.  bench.py's *_eager_event_us figures come from here (one launch in flight at a time). */
void nfp_time_next_launch(void* start_event, void* stop_event);

/* Test hook: re-read the NFP_* A/B switches (NFP_FORCE_GENERIC, NFP_FWD_SCALAR,
 * NFP_BWD_ATOMIC, NFP_BWD_BANDS, NFP_MFMA) from the environment.  They are
 * otherwise read once, when the library is loaded. */
void nfp_reload_env(void);

/* Describe, WITHOUT touching the GPU, what nfp_forward (backward = 0) or
 * nfp_backward (backward != 0) would launch for `d` with 4 KiB-aligned buffers:
 *   "<variant> | <kernel> grid=(x,y,z) block=t lds=bytes[; <kernel> ...]"
 * written NUL-terminated into buf.  Returns what the real call would return for
 * the descriptor (NFP_E_UNSUPPORTED / NFP_E_INVALID with nfp_last_error set).
 * No reference counterpart: lets the dispatch rules (which kernel serves which
 * geometry, every launch within the device's LDS / grid limits) be tested on a
 * machine without a GPU. */
int nfp_plan(const nfp_desc* d, int32_t backward, char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* NFP_H_ */
