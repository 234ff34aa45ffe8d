// tile_scatter.h -- the channels-first scatter-add gradients as a sorted pair stream over LDS-resident targets
// (csrc/tile_scatter.hip); called by the gradient entry points of gather_group.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace geot {

// grad_table[b, ch, j] (+)= sum over the pairs (e, t) with idx[b, e, t] == j of w[b, e, t] * grad_out[b, ch, e]
//   grad_out : (b, c, L) rows, `src_bstride` floats between batches;  idx / weight : (b, L, nt) (weight null: all 1)
//   grad_table : (b, c, m);  overwrite: every element is stored (the buffer may arrive uninitialised), else added to
// One writer per output element and one fixed summation order per call shape: bit-reproducible.
// Returns hipErrorNotSupported when the shape does not fit this path (m beyond what a CU's LDS holds, too many parts,
// or a workspace smaller than ts_ws_ints) -- the caller falls back.
hipError_t scatter_via_tiles(int b, int c, int m, int L, int nt, size_t src_bstride, const float *grad_out, const int *idx,
                             const float *weight, float *grad_table, void *workspace, long long ws_ints, hipStream_t s,
                             bool overwrite);
// 4-byte words of workspace scatter_via_tiles needs for this shape (0: the path does not apply)
long long ts_ws_ints(int b, int c, int m, long long L, int nt, bool weighted);

} // namespace geot
