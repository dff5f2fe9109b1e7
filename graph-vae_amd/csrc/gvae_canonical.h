// Canonical fp32 summation orders.
//
// The matching result must be bit-identical to the reference's CPU path, whose
// tensor expressions (graphvae/models/model.py:187-188, 267, 270) reduce through
// three fixed orders.  They are reproduced here as device functions; all files
// that include this header are compiled with -ffp-contract=off, so the only fused
// multiply-adds are the explicit __fmaf_rn calls in the norm.
//
//   cascade sum   terms are added one by one into level 0; after every 16 terms
//                 level 0 is added into level 1 and cleared (level j spills into
//                 j+1 when the running count is a multiple of 16^(j+1)); the tail
//                 left in level 0 finally receives levels 1, 2, 3 in that order.
//   ilp4 sum      four cascade sums over elements k, k+4, k+8, ... (k = 0..3) of
//                 the first 4*floor(n/4) terms; the remaining terms are added to
//                 partial 0; then partial 0 += partial 1, 2, 3.
//   column rule   in the sum over j of an (R x R) slab, output column a uses the
//                 cascade sum if a < 32*floor(R/32) (for R < 8: a < 4*floor(R/4)),
//                 otherwise the ilp4 sum.
//   row sum       (soft degree) n < 8: ilp4 over the row.  Else the row is cut into
//                 floor(n/8) groups of 8 lanes, each lane ilp4-summed over the group
//                 index; result = 0 + tail elements in order + lane 0 + ... + lane 7.
//   norm          8 lane accumulators, acc[l] = fma(x, x, acc[l]) over consecutive
//                 groups of 8; lanes folded 0+1+...+7; then the n % 8 tail elements:
//                 if there are 4 or more, the first four squares are rounded separately
//                 and added one by one, all other tail elements are fused in; sqrt.
#pragma once
#include <hip/hip_runtime.h>

namespace gvae {

// Sum of f(0), f(1), ..., f(n-1) in cascade order.  Valid for n < 2^19 (spill
// period 16); larger n would change the period and is rejected on the host.
template <class F>
__device__ __forceinline__ float cascade_sum(F f, int n) {
  float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
  int i = 0;
  while (i + 16 <= n) {
#pragma unroll
    for (int t = 0; t < 16; ++t) l0 += f(i + t);
    i += 16;
    l1 += l0;
    l0 = 0.f;
    if ((i & 0xF0) == 0) {
      l2 += l1;
      l1 = 0.f;
      if ((i & 0xF00) == 0) {
        l3 += l2;
        l2 = 0.f;
      }
    }
  }
  for (; i < n; ++i) l0 += f(i);
  l0 += l1;
  l0 += l2;
  l0 += l3;
  return l0;
}

template <class F>
__device__ __forceinline__ float ilp4_sum(F f, int n) {
  const int n4 = n >> 2;
  float p0 = cascade_sum([&](int t) { return f(4 * t + 0); }, n4);
  float p1 = cascade_sum([&](int t) { return f(4 * t + 1); }, n4);
  float p2 = cascade_sum([&](int t) { return f(4 * t + 2); }, n4);
  float p3 = cascade_sum([&](int t) { return f(4 * t + 3); }, n4);
  for (int i = n4 * 4; i < n; ++i) p0 += f(i);
  p0 += p1;
  p0 += p2;
  p0 += p3;
  return p0;
}

__device__ __forceinline__ bool column_uses_cascade(int a, int R) {
  const int grp = (R >= 8) ? 32 : 4;
  return a < (R / grp) * grp;
}

// sum over j = 0..R-1 of f(j) for output column a of an R-row slab
template <class F>
__device__ __forceinline__ float column_sum(F f, int a, int R) {
  return column_uses_cascade(a, R) ? cascade_sum(f, R) : ilp4_sum(f, R);
}

// row sum of n contiguous elements f(0..n-1) (inner reduction)
template <class F>
__device__ __forceinline__ float inner_row_sum(F f, int n) {
  if (n < 8) return ilp4_sum(f, n);
  const int cnt = n >> 3;
  float lane[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) lane[l] = ilp4_sum([&](int t) { return f(8 * t + l); }, cnt);
  float fin = 0.f;
  for (int k = cnt * 8; k < n; ++k) fin += f(k);
#pragma unroll
  for (int l = 0; l < 8; ++l) fin += lane[l];
  return fin;
}

}  // namespace gvae
