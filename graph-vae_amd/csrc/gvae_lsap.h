// Device side of the batched linear-sum-assignment: one wavefront (64 lanes) solves one graph.
//
// Reproduces scipy.optimize.linear_sum_assignment (SciPy 1.15.3, shortest
// augmenting path) on cost = -X in double precision, including its scan order and
// tie-breaking, so the permutation is identical to the reference's host round trip
// (graphvae/models/model.py:319-323):
//   - rows are augmented in order 0..n-1;
//   - the list of unscanned columns starts as n-1, n-2, ..., 0 and shrinks by
//     moving its last entry into the freed slot;
//   - a column's tentative distance is lowered only on strict '<';
//   - the next column is, among those at the minimum distance, the last one in list
//     order that is unassigned if there is one, otherwise the first one in list order.
// The column scan (the O(n) inner loop) is spread over the 64 lanes; the selection
// rule above is evaluated with wave-wide min/max reductions, which is equivalent to
// the sequential scan.  Dual variables, distances, the path and the list order live in
// registers; only the fp32 matrix itself is in LDS.
//
// Included by lsap.hip (the stand-alone kernel) and matching.hip (the S-free MPM kernel, which
// solves a graph's assignment as its epilogue, from the X it already holds in LDS).
#pragma once
#include "gvae_common.h"

namespace gvae {

// Wave-wide reductions on the data-parallel-primitive path (no LDS traffic): an xor butterfly
// inside each row of 16 lanes (quad permutes, then half-row and row mirrors), then the row
// results are folded upwards with the two row-broadcast moves and read from lane 63.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2, hi2;
  if constexpr (ROW_MASK == 0xf) {
    // every lane is written: no previous value to keep, which spares the copy of it
    lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  } else {
    lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  }
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double read_lane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// one v_min_f64 (fmin() would add two canonicalising instructions per call; no NaNs reach
// these reductions: costs are checked beforehand)
__device__ __forceinline__ double min1_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// LANES = 16, 32 or 64: how many leading lanes can hold data (the others hold the identity);
// the row-combining steps a narrower problem does not need are left out.
template <int LANES = 64>
__device__ __forceinline__ double wave_min_f64(double v) {
  v = min1_f64(v, dpp_f64<0xB1>(v));        // quad_perm [1,0,3,2]
  v = min1_f64(v, dpp_f64<0x4E>(v));        // quad_perm [2,3,0,1]
  v = min1_f64(v, dpp_f64<0x141>(v));       // row_half_mirror
  v = min1_f64(v, dpp_f64<0x140>(v));       // row_mirror: every lane of a row holds the row's minimum
  if constexpr (LANES == 16) return read_lane_f64(v, 15);
  v = min1_f64(v, dpp_f64<0x142, 0xa>(v));  // row_bcast15: rows 1 and 3 take in rows 0 and 2
  if constexpr (LANES == 32) return read_lane_f64(v, 31);
  v = min1_f64(v, dpp_f64<0x143, 0xc>(v));  // row_bcast31: rows 2 and 3 take in row 1
  return read_lane_f64(v, 63);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int v) {
  if constexpr (ROW_MASK == 0xf) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false);
  else return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
template <int LANES = 64>
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_i32<0xB1>(v));
  v = max(v, dpp_i32<0x4E>(v));
  v = max(v, dpp_i32<0x141>(v));
  v = max(v, dpp_i32<0x140>(v));
  if constexpr (LANES == 16) return __builtin_amdgcn_readlane(v, 15);
  v = max(v, dpp_i32<0x142, 0xa>(v));
  if constexpr (LANES == 32) return __builtin_amdgcn_readlane(v, 31);
  v = max(v, dpp_i32<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// value of a per-lane array element held by lane (idx & 63), chunk (idx >> 6); idx is uniform
template <int NCH>
__device__ __forceinline__ int pick_i32(const int (&a)[NCH], int idx) {
  const int l = idx & 63;
  int r = __builtin_amdgcn_readlane(a[0], l);
  if constexpr (NCH == 2) {
    if (idx >= 64) r = __builtin_amdgcn_readlane(a[1], l);
  }
  return r;
}
template <int NCH>
__device__ __forceinline__ double pick_f64(const double (&a)[NCH], int idx) {
  const int l = idx & 63;
  double r = read_lane_f64(a[0], l);
  if constexpr (NCH == 2) {
    if (idx >= 64) r = read_lane_f64(a[1], l);
  }
  return r;
}

// One wavefront per graph, the whole solver state in registers: lane l owns columns and rows
// l (and l + 64 when NCH == 2).  Column state: dual v, tentative distance spc, predecessor
// path, row4col, and pos = the column's position in the list of unscanned columns (-1 once
// scanned).  Row state: dual u, col4row, the scanned-row flag.  Only the cost matrix (= X,
// negated on read) sits in LDS.  The wave-uniform scalars (current row i, the chosen column,
// the sink) move between lanes with readlane / DPP; there is no barrier in the solver.
//
// The sequential rule "among the columns at the minimum distance, the last unassigned one in
// list order, else the first one in list order" becomes one integer maximum over the keys
//   unassigned: 0x10000 + pos      assigned: 0xFFFF - pos      (list positions are unique)
template <int NCH, int LANES = 64>
__device__ __forceinline__ int lsap_solve(const float *cost, int n, int ld, int lane,
                                          int (&col4row)[NCH]) {
  double u[NCH], v[NCH], spc[NCH];
  int path[NCH], row4col[NCH], pos[NCH], ccol[NCH];
  // scanned rows / columns of the current augmentation as wave-uniform bit masks
  unsigned long long sr[NCH], sc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u[c] = 0.0; v[c] = 0.0; spc[c] = INFINITY;
    path[c] = -1; row4col[c] = -1; col4row[c] = -1; pos[c] = -1;
    ccol[c] = min(c * 64 + lane, n - 1);     // lanes past the matrix re-read its last column
  }
  for (int cur = 0; cur < n; ++cur) {
    double minVal = 0.0;
    int i = cur, num_remaining = n, sink = -1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 64 + lane;
      pos[c] = col < n ? n - 1 - col : -1;   // the list starts as n-1, n-2, ..., 0
      sr[c] = 0ull; sc[c] = 0ull;
      spc[c] = INFINITY;
    }
    // The inner loop is a single dependent chain executed by one wavefront: it is written
    // without per-lane control flow (selects instead of branches, the cost row read by every
    // lane), which is what its latency is made of.
    while (sink == -1) {
      if (NCH == 1 || i < 64) sr[0] |= 1ull << (i & 63);
      else sr[NCH - 1] |= 1ull << (i & 63);
      const double ui = pick_f64<NCH>(u, i);
      const float *crow = cost + i * ld;
      double lowest = INFINITY;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool act = pos[c] >= 0;
        const double r = minVal + (double)(-crow[ccol[c]]) - ui - v[c];
        const bool upd = act && r < spc[c];
        spc[c] = upd ? r : spc[c];
        path[c] = upd ? i : path[c];
        lowest = min1_f64(act ? spc[c] : INFINITY, lowest);
      }
      lowest = wave_min_f64<LANES>(lowest);
      if (lowest == INFINITY) return 1;      // infeasible
      // columns at the minimum distance: usually exactly one, which the ballot names directly;
      // only a tie needs the second wave-wide reduction over the selection keys
      bool at_min[NCH];
      unsigned long long m_at[NCH];
      int n_at = 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        at_min[c] = pos[c] >= 0 && spc[c] == lowest;
        m_at[c] = __ballot(at_min[c]);
        n_at += __popcll(m_at[c]);
      }
      int j;
      if (n_at == 1) {
        j = m_at[0] ? __ffsll((long long)m_at[0]) - 1 : 64 + __ffsll((long long)m_at[NCH - 1]) - 1;
      } else {
        int key = -1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int k = ((row4col[c] == -1 ? 0x10000 + pos[c] : 0xFFFF - pos[c]) << 1) | c;
          key = max(key, at_min[c] ? k : -1);
        }
        const int best = wave_max_i32<LANES>(key);
        const int L = __ffsll((long long)__ballot(key == best)) - 1;
        j = (best & 1) * 64 + L;
      }
      minVal = lowest;
      const int rj = pick_i32<NCH>(row4col, j);
      const int pj = pick_i32<NCH>(pos, j);
      if (rj == -1) sink = j; else i = rj;
      // remove j from the list: the list's last entry moves into j's place
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        pos[c] = pos[c] == num_remaining - 1 ? pj : pos[c];
        pos[c] = c * 64 + lane == j ? -1 : pos[c];
      }
      if (NCH == 1 || j < 64) sc[0] |= 1ull << (j & 63);
      else sc[NCH - 1] |= 1ull << (j & 63);
      --num_remaining;
    }
    // dual updates
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int row = c * 64 + lane;
      // distance of the column this row is assigned to (gathered from its owner lane)
      const int cr = col4row[c] < 0 ? 0 : col4row[c];
      double s_cr = __shfl(spc[0], cr & 63, 64);
      if constexpr (NCH == 2) {
        const double s1 = __shfl(spc[1], cr & 63, 64);
        if (cr >= 64) s_cr = s1;
      }
      const bool in_sr = (sr[c] >> lane) & 1ull, in_sc = (sc[c] >> lane) & 1ull;
      if (row == cur) u[c] += minVal;
      else if (in_sr) u[c] += minVal - s_cr;
      if (in_sc) v[c] -= minVal - spc[c];
    }
    // augment along the path (wave-uniform walk)
    int j = sink;
    for (;;) {
      const int r = pick_i32<NCH>(path, j);
      const int t = pick_i32<NCH>(col4row, r);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c * 64 + lane == j) row4col[c] = r;
        if (c * 64 + lane == r) col4row[c] = j;
      }
      j = t;
      if (r == cur) break;
    }
  }
  return 0;
}

// Whole assignment of one graph by the calling wavefront: cost = -X, X as n x n floats in LDS
// (row stride ld).  Writes perm_full (reference model.py:321-323: perm[col4row[i]] = i, identity
// beyond n and on failure) and the status word (0 ok, 1 NaN / +inf in X or infeasible).
// `lane` = 0..63; every lane of the wave must call.
__device__ __forceinline__ void lsap_wave(const float *cost, int n, int ld, int lane, int64_t *pg,
                                          int n_perm, int32_t *status) {
  for (int k = lane; k < n_perm; k += 64) pg[k] = k;
  int bad = 0;
  for (int e = lane; e < n * n; e += 64) {
    const int r = e / n;
    const float x = cost[r * ld + (e - r * n)];
    bad |= (x != x) || (x == INFINITY);  // cost = -x: NaN or -inf is invalid
  }
  bad = wave_max_i32(bad);
  if (bad) {
    if (lane == 0) *status = 1;
    return;
  }
  int fail;
  if (n <= 64) {
    int col4row[1];
    if (n <= 16) fail = lsap_solve<1, 16>(cost, n, ld, lane, col4row);
    else if (n <= 32) fail = lsap_solve<1, 32>(cost, n, ld, lane, col4row);
    else fail = lsap_solve<1>(cost, n, ld, lane, col4row);
    if (!fail && lane < n) pg[col4row[0]] = lane;            // perm[col4row[i]] = i
  } else {
    int col4row[2];
    fail = lsap_solve<2>(cost, n, ld, lane, col4row);
    if (!fail) {
      pg[col4row[0]] = lane;
      if (64 + lane < n) pg[col4row[1]] = 64 + lane;
    }
  }
  if (lane == 0) *status = fail;
}

}  // namespace gvae
