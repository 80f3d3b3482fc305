// Tile shapes and capacities shared by the kernels and the host driver of the recursion (ai_ncut.hip, ai_eigs.hip).
#pragma once
#ifndef AI_FINE_ROWS
#define AI_FINE_ROWS 32      // rows per block in the 16-lanes-per-row kernels (2 rows in flight per lane group; 16 / 128 measured slower, 64: 20.6 vs 17.4 us per launch on a whole 200k graph, 111 vs 121 chunks/s with the quad kernel)
#endif
#ifndef AI_COARSE_ROWS
#define AI_COARSE_ROWS 512   // rows per block in the thread-per-row kernels (256 / 1024 measured within 2 %)
#endif
#define AI_ROW_ILP (AI_FINE_ROWS / (AI_BLOCK / AI_LPR))
#define AI_SLAB_VECS 32      // Lanczos vectors per HBM slab
#define AI_ROW_PF 4         // rounds of 16 entries per row loaded together in the 16-lanes-per-row kernels
#define AI_SWEEP_VALS 40     // per-task sweep partials: cut[10], assocA[10], assocB[10], cntA[10]
#define AI_MAX_CHECKS 4096   // convergence checks per level (one counter slot each)
