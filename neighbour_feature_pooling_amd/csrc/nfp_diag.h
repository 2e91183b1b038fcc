// nfp_diag.h — diagnostic build only (-DNFP_STAMPS, scripts/diag_stamps.py): thread 0 of every workgroup records
// {shader clock, 100 MHz wall clock} at phase boundaries into a buffer nothing else reads.  Never part of the
// product library: nfp_common.h includes this file only under NFP_STAMPS.
#pragma once
// (included from inside namespace nfp)
__device__ unsigned long long* nfp_stamp_buf = nullptr;
// The buffer pointer is read ONCE (a vector load + wait at kernel entry); a stamp is then one
// s_memtime/s_memrealtime pair and two stores, with no vmcnt wait, so loads in flight stay in flight.
#define NFP_STAMP_INIT() unsigned long long* nfp_sb_ = nfp_stamp_buf
#define NFP_STAMP(id)                                                                        \
  do {                                                                                       \
    if ((threadIdx.x | threadIdx.y | threadIdx.z) == 0 && nfp_sb_) {                                                       \
      unsigned long long wg = blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y;       \
      nfp_sb_[(wg * 16 + (id)) * 2] = __builtin_amdgcn_s_memtime();                         \
      nfp_sb_[(wg * 16 + (id)) * 2 + 1] = __builtin_amdgcn_s_memrealtime();                 \
    }                                                                                        \
  } while (0)
