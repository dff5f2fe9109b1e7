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
// the sequential scan: "among the columns at the minimum distance, the last unassigned one in
// list order, else the first one in list order" becomes one integer maximum over the keys
//   unassigned: 0x10000 + pos      assigned: 0xFFFF - pos      (list positions are unique).
// Dual variables, distances, the path and the list order live in registers, the sets of
// scanned rows / columns are wave-uniform bit masks; only the fp32 matrix itself is in LDS.
// The wave-uniform scalars (current row, the chosen column, the sink) move between lanes with
// readlane / DPP; there is no barrier in the solver.
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

// The solver for at most 64 columns (lane l owns column l and row l), written for the latency
// of one search step, which is what the kernel's run time is made of (a graph takes about n^2/2
// steps, one after the other: SciPy's solver starts every row's search from scratch and, on the
// near-rank-one matrices the MPM converges to, scans every assigned column before it reaches a
// free one).  A step costs its instructions (a lone wavefront issues one every 4-5 cycles) plus
// about 25 cycles each time a value computed by the vector unit is consumed by the scalar unit.
// Hence: the set of unscanned columns is a wave-uniform bit mask in scalar registers (it costs a
// vector lane nothing to test, and removing a column is one scalar instruction) instead of a
// per-lane test of the list position; scanned columns keep +inf as their reduction key, so the
// comparison with the minimum needs no second mask; the list position of a scanned column is
// left stale (it is never read again).
template <int LANES>
__device__ __forceinline__ int lsap_solve_1(const float *cost, int n, int ld, int lane, int &col4row) {
  double u = 0.0, v = 0.0;
  int path = -1, row4col = -1;
  col4row = -1;
  const unsigned long long valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  const float *ccol = cost + min(lane, n - 1);      // lanes past the matrix re-read its last column
  for (int cur = 0; cur < n; ++cur) {
    double minVal = 0.0, spc = INFINITY;
    int i = cur, num_remaining = n, sink = -1;
    int pos = n - 1 - lane;                          // the list starts as n-1, n-2, ..., 0
    unsigned long long act = valid, sr = 0ull, sc = 0ull;
    do {
      sr |= 1ull << i;
      const float c = ccol[i * ld];
      const double ui = read_lane_f64(u, i);
      const double r = minVal + (double)(-c) - ui - v;
      const bool a = __builtin_amdgcn_inverse_ballot_w64(act);
      const bool upd = a && r < spc;
      spc = upd ? r : spc;
      path = upd ? i : path;
      const double key = a ? spc : INFINITY;
      const double lowest = wave_min_f64<LANES>(key);
      // columns at the minimum: usually exactly one, which the ballot names directly (scanned
      // columns hold +inf, so they only match when nothing is reachable: the infeasible case)
      const unsigned long long m_at = __ballot(key == lowest);
      int j;
      if (__popcll(m_at) == 1) {
        j = __builtin_ctzll(m_at);
      } else {
        if (lowest == INFINITY) return 1;
        // tie: the last unassigned column in list order, else the first one in list order
        const int k = row4col == -1 ? 0x10000 + pos : 0xFFFF - pos;
        const int kk = key == lowest ? k : -1;
        const int best = wave_max_i32<LANES>(kk);
        j = __builtin_ctzll(__ballot(kk == best));
      }
      const int rj = __builtin_amdgcn_readlane(row4col, j);
      const int pj = __builtin_amdgcn_readlane(pos, j);
      // remove j from the list: the list's last entry moves into j's place
      --num_remaining;
      pos = pos == num_remaining ? pj : pos;
      act &= ~(1ull << j);
      sc |= 1ull << j;
      minVal = lowest;
      if (rj == -1) sink = j; else i = rj;
    } while (sink == -1);
    // dual updates: a scanned column's distance is the spc it was removed with
    {
      const int cr = col4row < 0 ? 0 : col4row;
      const double s_cr = __shfl(spc, cr, 64);
      const bool in_sr = __builtin_amdgcn_inverse_ballot_w64(sr);
      const bool in_sc = __builtin_amdgcn_inverse_ballot_w64(sc);
      if (lane == cur) u += minVal;
      else if (in_sr) u += minVal - s_cr;
      if (in_sc) v -= minVal - spc;
    }
    // augment along the path (wave-uniform walk)
    int j = sink;
    for (;;) {
      const int r = __builtin_amdgcn_readlane(path, j);
      const int t = __builtin_amdgcn_readlane(col4row, r);
      if (lane == j) row4col = r;
      if (lane == r) col4row = j;
      j = t;
      if (r == cur) break;
    }
  }
  return 0;
}

// The same for 65..128 columns: lane l owns columns (and rows) l and 64 + l; the two halves' sets
// are two scalar bit masks, a value of column / row k lives in lane k & 63 of half k >> 6 (both
// halves are read and the scalar unit selects: no branch).
__device__ __forceinline__ int pick2_i32(int a0, int a1, int k) {
  const int x0 = __builtin_amdgcn_readlane(a0, k & 63), x1 = __builtin_amdgcn_readlane(a1, k & 63);
  return k < 64 ? x0 : x1;
}
__device__ __forceinline__ double pick2_f64(double a0, double a1, int k) {
  const double x0 = read_lane_f64(a0, k & 63), x1 = read_lane_f64(a1, k & 63);
  return k < 64 ? x0 : x1;
}
__device__ __forceinline__ int lsap_solve_2(const float *cost, int n, int ld, int lane, int (&col4row)[2]) {
  double u0 = 0.0, u1 = 0.0, v0 = 0.0, v1 = 0.0;
  int path0 = -1, path1 = -1, r4c0 = -1, r4c1 = -1;
  col4row[0] = col4row[1] = -1;
  const unsigned long long valid1 = n >= 128 ? ~0ull : ((1ull << (n - 64)) - 1ull);
  const float *ccol0 = cost + lane, *ccol1 = cost + min(64 + lane, n - 1);
  for (int cur = 0; cur < n; ++cur) {
    double minVal = 0.0, spc0 = INFINITY, spc1 = INFINITY;
    int i = cur, num_remaining = n, sink = -1;
    int pos0 = n - 1 - lane, pos1 = n - 65 - lane;
    unsigned long long act0 = ~0ull, act1 = valid1, sr0 = 0ull, sr1 = 0ull, sc0 = 0ull, sc1 = 0ull;
    do {
      const unsigned long long ibit = 1ull << (i & 63);
      sr0 |= i < 64 ? ibit : 0ull;
      sr1 |= i < 64 ? 0ull : ibit;
      const float c0 = ccol0[i * ld], c1 = ccol1[i * ld];
      const double ui = pick2_f64(u0, u1, i);
      const double r0 = minVal + (double)(-c0) - ui - v0;
      const double r1 = minVal + (double)(-c1) - ui - v1;
      const bool a0 = __builtin_amdgcn_inverse_ballot_w64(act0), a1 = __builtin_amdgcn_inverse_ballot_w64(act1);
      const bool upd0 = a0 && r0 < spc0, upd1 = a1 && r1 < spc1;
      spc0 = upd0 ? r0 : spc0; path0 = upd0 ? i : path0;
      spc1 = upd1 ? r1 : spc1; path1 = upd1 ? i : path1;
      const double key0 = a0 ? spc0 : INFINITY, key1 = a1 ? spc1 : INFINITY;
      const double lowest = wave_min_f64<64>(min1_f64(key0, key1));
      const unsigned long long m0 = __ballot(key0 == lowest), m1 = __ballot(key1 == lowest);
      int j;
      if (__popcll(m0) + __popcll(m1) == 1) {
        j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
      } else {
        if (lowest == INFINITY) return 1;            // infeasible
        // tie: the last unassigned column in list order, else the first one in list order
        const int k0 = ((r4c0 == -1 ? 0x10000 + pos0 : 0xFFFF - pos0) << 1);
        const int k1 = ((r4c1 == -1 ? 0x10000 + pos1 : 0xFFFF - pos1) << 1) | 1;
        const int kk = max(key0 == lowest ? k0 : -1, key1 == lowest ? k1 : -1);
        const int best = wave_max_i32<64>(kk);
        j = (best & 1) * 64 + __builtin_ctzll(__ballot(kk == best));
      }
      const int rj = pick2_i32(r4c0, r4c1, j), pj = pick2_i32(pos0, pos1, j);
      // remove j from the list: the list's last entry moves into j's place
      --num_remaining;
      pos0 = pos0 == num_remaining ? pj : pos0;
      pos1 = pos1 == num_remaining ? pj : pos1;
      const unsigned long long jbit = 1ull << (j & 63);
      const unsigned long long jb0 = j < 64 ? jbit : 0ull, jb1 = j < 64 ? 0ull : jbit;
      act0 &= ~jb0; act1 &= ~jb1;
      sc0 |= jb0; sc1 |= jb1;
      minVal = lowest;
      if (rj == -1) sink = j; else i = rj;
    } while (sink == -1);
    // dual updates
    {
      const int cr0 = col4row[0] < 0 ? 0 : col4row[0], cr1 = col4row[1] < 0 ? 0 : col4row[1];
      const double a0 = __shfl(spc0, cr0 & 63, 64), b0 = __shfl(spc1, cr0 & 63, 64);
      const double a1 = __shfl(spc0, cr1 & 63, 64), b1 = __shfl(spc1, cr1 & 63, 64);
      const double s_cr0 = cr0 >= 64 ? b0 : a0, s_cr1 = cr1 >= 64 ? b1 : a1;
      if (lane == cur) u0 += minVal;
      else if (__builtin_amdgcn_inverse_ballot_w64(sr0)) u0 += minVal - s_cr0;
      if (64 + lane == cur) u1 += minVal;
      else if (__builtin_amdgcn_inverse_ballot_w64(sr1)) u1 += minVal - s_cr1;
      if (__builtin_amdgcn_inverse_ballot_w64(sc0)) v0 -= minVal - spc0;
      if (__builtin_amdgcn_inverse_ballot_w64(sc1)) v1 -= minVal - spc1;
    }
    // augment along the path (wave-uniform walk)
    int j = sink;
    for (;;) {
      const int r = pick2_i32(path0, path1, j);
      const int t = pick2_i32(col4row[0], col4row[1], r);
      if (lane == j) r4c0 = r;
      if (64 + lane == j) r4c1 = r;
      if (lane == r) col4row[0] = j;
      if (64 + lane == r) col4row[1] = j;
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
    int col4row;
    if (n <= 16) fail = lsap_solve_1<16>(cost, n, ld, lane, col4row);
    else if (n <= 32) fail = lsap_solve_1<32>(cost, n, ld, lane, col4row);
    else fail = lsap_solve_1<64>(cost, n, ld, lane, col4row);
    if (!fail && lane < n) pg[col4row] = lane;               // perm[col4row[i]] = i
  } else {
    int col4row[2];
    fail = lsap_solve_2(cost, n, ld, lane, col4row);
    if (!fail) {
      pg[col4row[0]] = lane;
      if (64 + lane < n) pg[col4row[1]] = 64 + lane;
    }
  }
  if (lane == 0) *status = fail;
}

}  // namespace gvae
