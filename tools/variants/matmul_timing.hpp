// Timing-only variants of the matrix-core mat-mul main loop (fp_matmul_mfma.cuh).  Force-included by
// tools/build_variant.sh (hipcc -include) together with ONE of the -DMF_DBG_* switches below; the
// library's own build never sees this file.  Every variant computes WRONG results by design: they
// exist to measure what the staging costs.
#pragma once
#if defined(MF_DBG_NOSTAGE)      // no DMA at all: the bare MFMA + LDS-read rate
#define MF_STAGE(J)
#elif defined(MF_DBG_SAMEK)      // every DMA re-reads k-step 0: L2-resident source
#define MF_STAGE(J) stage_part(0, (ks + 1) & 1, J)
#elif defined(MF_DBG_HALFSTAGE)  // half the DMAs
#define MF_STAGE(J) if ((J) & 1) stage_part(ks_next, (ks + 1) & 1, J)
#endif
#if defined(MF_DBG_ONEK)         // one k-step: the per-tile prologue + epilogue alone
#define MF_KSTEPS(n) 1u
#endif
#if defined(MF_DBG_NOBARRIER)
#define MF_KSTEP_BARRIER()
#endif
