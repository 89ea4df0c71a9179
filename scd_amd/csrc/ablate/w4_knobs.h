// Schedule constants and experiment hooks of gemm_w4_kernel (textually included by gemm.hip).  The values below are the shipped
// ones; every other value belongs to a measurement recorded in docs/design/gemm.md, encoder_time_budget.md or experiments_dropped.md and
// is reached by building a variant library (tools/build_variant.sh NAME gemm.hip -DW4_...=v), never at run time.  Hooks marked "results
// wrong" are timing probes: they remove work.
#pragma once
#ifndef W4_DMA_SPLIT
#define W4_DMA_SPLIT 1
#endif
#ifndef W4_LATE_BAR
#define W4_LATE_BAR 1
#endif
#ifndef W4_TN_MAJOR
#define W4_TN_MAJOR 0   // MFMA order inside a sub-step: 0 = m-tile outer (eight MFMAs share the A fragment), 1 = n-tile outer
#endif
#ifndef W4_PROBE_VALU
#define W4_PROBE_VALU 0
#endif
#ifndef W4_TNW
#define W4_TNW 8   // timing probe only (-DW4_TNW=4): the wave computes 4 of its 8 n-tiles - the main loop of a 256 x 128 block tile; results are wrong
#endif
#ifndef W4_DEFER_STORES
#define W4_DEFER_STORES 1
#endif
#ifndef W4_LATE_TM
#define W4_LATE_TM 2
#endif
#ifndef W4_WSPLIT
#define W4_WSPLIT 0     // ring-fill schedule experiment: this many of a chunk's eight W fills are issued in the ODD sub-step behind the A fills
#endif                  // (A every 4 MFMAs instead of every 6), the rest at the very start of the next even one: the chunk's last fill goes out
                        // ~16 MFMAs earlier, i.e. has ~300 more cycles to land before the barrier that waits for it
#ifndef W4_ABL_STATS
#define W4_ABL_STATS 0   // timing probe (results wrong): 1 = the LN = 2 epilogue computes / adds no row statistics
#endif
#ifndef W4_ABL_PRE
#define W4_ABL_PRE 0     // timing probe (results wrong): 1 = no bias / column-sum loads in front of the epilogue (zeros)
#endif
#ifndef W4_LN_ABL
#define W4_LN_ABL 0   // timing probes of the LayerNorm fold (results are wrong): 1 = no row-statistics loads / conversions (rstd = 1, mean = 0), 2 = no fold arithmetic either
#endif
#ifndef W4_A_MOD
#define W4_A_MOD ""      // cache-policy bits of the activation fills (" nt", " sc1", ...): experiment hook
#endif
#ifndef W4_W_MOD
#define W4_W_MOD ""
#endif
#ifndef W4_RES_NT
#define W4_RES_NT 0      // experiment: the residual rows of proj / fc2 (read once, private to the tile) fetched non-temporally
#endif
#ifndef W4_A_NT_LN1
#define W4_A_NT_LN1 0    // experiment: non-temporal activation fills in the LayerNorm-folded variants only (QKV, fc1: K = 768, A read once per n-group)
#endif
#ifndef W4_RD_LN2
#define W4_RD_LN2 2
#endif
