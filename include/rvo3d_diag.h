/*
 * rvo3d_diag.h -- the extra entry point of the DIAGNOSTICS build of the library
 * (librvo3d_hip_diag.so = csrc/rvo3d_capi.hip compiled with -DRVO3D_DIAG; built and loaded
 * only by tools/diaglib.py).  Not part of the product ABI: librvo3d_hip.so exports none of
 * this, contains no phase stamps and reads no environment variable.
 *
 * The diagnostics build additionally honours, at rvo3d_create:
 *   RVO3D_ABLATE=<bits>   skip phases of the fused step (1 sweep A, 2 collision sweep,
 *                         4 rows sweep, 8 observation rows, 16 row fill, 32 stage X2,
 *                         64 stage X1) - RESULTS ARE INVALID, timing only
 *   RVO3D_LDS_PAD=<bytes> pad the dynamic LDS request (caps waves per SIMD)
 */
#ifndef RVO3D_DIAG_H
#define RVO3D_DIAG_H

#include "rvo3d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Attach a device buffer of 16 uint64 per workgroup; lane 0 of every workgroup then stores
 * s_memtime stamps at the kernel's phase boundaries (tools/stamps.py reads them).  NULL
 * detaches. */
int rvo3d_debug_stamps(rvo3d_env *h, unsigned long long *stamps);

#ifdef __cplusplus
}
#endif
#endif
