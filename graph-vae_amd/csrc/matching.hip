// Graph-matching kernels for gfx950: node-similarity block, materialised
// edge-similarity tensor, max-pool matching (MPM) streaming that tensor from HBM,
// and the S-free fused MPM.  Compiled with -ffp-contract=off: every product and
// sum below is a separately rounded fp32 operation, as on the reference's CPU path.
#include "gvae_canonical.h"
#include "gvae_lsap.h"
#include "gvae_common.h"

#include <mutex>
#include <type_traits>

namespace gvae {

// ---------------------------------------------------------------------------
// node similarity: NS[i,a] = 1 / (|degA[i] - degB[a]| + 1)
// one work-group per graph; A then P are staged through LDS (padded rows) so the
// per-row canonical sums read conflict-free.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void node_sim_kernel(const float *__restrict__ A,
                                                       const float *__restrict__ P,
                                                       const int32_t *__restrict__ Rarr, int ld,
                                                       int64_t stride, float *__restrict__ NS) {
  extern __shared__ float lds[];
  const int g = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int R = Rarr[g];
  if (R <= 0) return;
  const int pad = R + 1;
  float *tile = lds;                 // R * (R+1)
  float *degA = lds + R * pad;       // R
  float *degB = degA + R;            // R
  const float *Ag = A + g * stride, *Pg = P + g * stride;
  for (int pass = 0; pass < 2; ++pass) {
    const float *src = pass == 0 ? Ag : Pg;
    for (int e = tid; e < R * R; e += nt) {
      const int r = e / R, c = e - r * R;
      tile[r * pad + c] = (r == c) ? 1.0f : src[(int64_t)r * ld + c];
    }
    __syncthreads();
    for (int r = tid; r < R; r += nt) {
      const float *row = tile + r * pad;
      const float s = inner_row_sum([&](int k) { return row[k]; }, R);
      (pass == 0 ? degA : degB)[r] = s;
    }
    __syncthreads();
  }
  float *NSg = NS + g * stride;
  for (int e = tid; e < R * R; e += nt) {
    const int i = e / R, a = e - i * R;
    NSg[(int64_t)i * ld + a] = 1.0f / (fabsf(degA[i] - degB[a]) + 1.0f);
  }
}

// ---------------------------------------------------------------------------
// edge-similarity tensor, materialised.  One work-group per (graph, i): it owns the
// contiguous slab S[i,:,:,:] (R*R rows of RS floats; RS = R, or R rounded up to a
// multiple of 4 in the padded layout, where the padding repeats the row's last element
// so that it can never change a row maximum).  Each thread keeps its piece of P in
// registers and sweeps j, so every store instruction of a wave covers one contiguous
// run of the slab.  V = floats per lane per store (4 when RS % 4 == 0).
//   S[i,j,a,b] = A[i,j]*P[a,b]  (i != j, a != b);  NS[i,a] (i == j, a == b);  else 0
// ---------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ void edge_sim_body(const float *__restrict__ Pg,
                                              const float *__restrict__ NSg, int ld, int R, int RS,
                                              int i, float *__restrict__ slab, const float *arow) {
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)R * RS;        // floats per (i,j) matrix

  const int cols = RS / V;                      // vector columns per row
  const int lpr = cols < 256 ? cols : 256;      // lanes per row
  const int rows_per_pass = 256 / lpr;
  const int rslot = tid / lpr, c0 = tid - rslot * lpr;
  if (rslot >= rows_per_pass) return;
  for (int a = rslot; a < R; a += rows_per_pass) {
    for (int c = c0; c < cols; c += lpr) {
      float p[V], d[V];
      const float nsv = NSg[(int64_t)i * ld + a];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int b = min(c * V + v, R - 1);    // padding repeats the row's last element
        p[v] = (b == a) ? 0.0f : Pg[(int64_t)a * ld + b];
        d[v] = (b == a) ? nsv : 0.0f;
      }
      float *dst = slab + (int64_t)a * RS + c * V;
      for (int j = 0; j < R; ++j, dst += plane) {
        const float aij = arow[j];
        float o[V];
#pragma unroll
        for (int v = 0; v < V; ++v) o[v] = (j == i) ? d[v] : aij * p[v];
        if constexpr (V == 4) {
          *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        } else if constexpr (V == 2) {
          *reinterpret_cast<float2 *>(dst) = make_float2(o[0], o[1]);
        } else {
          dst[0] = o[0];
        }
      }
    }
  }
}

// floats reserved at the front of the dynamic LDS region of the MPM kernels for control words
constexpr int kCtlFloats = 32;

// row stride of a graph's S block: R floats (dense) or R rounded up to a multiple of 4
__host__ __device__ __forceinline__ int s_row_stride(int R, int layout) {
  return layout == GVAE_S_PADDED ? ((R + 3) & ~3) : R;
}

__global__ __launch_bounds__(256) void edge_sim_kernel(const float *__restrict__ A,
                                                       const float *__restrict__ P,
                                                       const float *__restrict__ NS,
                                                       const int32_t *__restrict__ Rarr, int ld,
                                                       int64_t stride, int layout,
                                                       float *__restrict__ S,
                                                       const int64_t *__restrict__ S_off) {
  __shared__ float arow[1024];
  const int g = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  const int R = Rarr[g];
  if (i >= R) return;
  const int RS = s_row_stride(R, layout);
  const float *Ag = A + g * stride, *Pg = P + g * stride, *NSg = NS + g * stride;
  float *slab = S + S_off[g] + (int64_t)i * R * R * RS;
  for (int j = tid; j < R; j += 256) arow[j] = (j == i) ? 1.0f : Ag[(int64_t)i * ld + j];
  __syncthreads();
  if ((RS & 3) == 0) edge_sim_body<4>(Pg, NSg, ld, R, RS, i, slab, arow);
  else if ((RS & 1) == 0) edge_sim_body<2>(Pg, NSg, ld, R, RS, i, slab, arow);
  else edge_sim_body<1>(Pg, NSg, ld, R, RS, i, slab, arow);
}

// ---------------------------------------------------------------------------
// MPM, one iteration, streaming S.  One work-group per (graph, i) reads the slab
// S[i,:,:,:] exactly once:  T[j,a] = max_b S[i,j,a,b]*X[j,b]  (0 for j == i), then
//   Xnew[i,a] = X[i,a]*S[i,i,a,a] + sum_j T[j,a]        (canonical column sum)
// Each row (j,a) of R floats is spread over `lpr` = 2^lpr_log2 adjacent lanes, V
// floats per lane per load, so a wave's load instruction covers 64/lpr consecutive
// rows = one contiguous run of the slab.  The max over b is exact in any order.
// ---------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int V>
struct VecT;
template <>
struct VecT<4> { using type = f32x4_t; };
template <>
struct VecT<2> { using type = f32x2_t; };
template <>
struct VecT<1> { using type = float; };

template <int V>
__device__ __forceinline__ void vec_to_arr(const typename VecT<V>::type &v, float *o) {
  if constexpr (V == 4) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  else if constexpr (V == 2) { o[0] = v.x; o[1] = v.y; }
  else { o[0] = v; }
}

// S is read exactly once per iteration and is far larger than the caches: stream it with the
// non-temporal policy (GVAE_STREAM_NT=0 builds the default-policy variant)
#ifndef GVAE_STREAM_NT
#define GVAE_STREAM_NT 1
#endif
// A pointer argument of a function that is not inlined is a generic one (flat instructions,
// which wait on the LDS counter as well); these say that it points to device memory.
#define GVAE_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ GVAE_GLOBAL T *as_global(T *p) {
  return (GVAE_GLOBAL T *)(uintptr_t)p;
}
// Its arguments also arrive in vector registers, as if they differed between lanes: these say that
// a value is the same in every lane (scalar registers, scalar address arithmetic, uniform loops).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ GVAE_GLOBAL T *uniform_global(T *p) {
  const uintptr_t a = (uintptr_t)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (GVAE_GLOBAL T *)(((uintptr_t)hi << 32) | lo);
}
template <int V>
__device__ __forceinline__ typename VecT<V>::type stream_load_g(const GVAE_GLOBAL float *p) {
  using vec_t = typename VecT<V>::type;
#if GVAE_STREAM_NT
  return __builtin_nontemporal_load(reinterpret_cast<const GVAE_GLOBAL vec_t *>(p));
#else
  return *reinterpret_cast<const GVAE_GLOBAL vec_t *>(p);
#endif
}
__device__ __forceinline__ f32x4_t stream_load4(const GVAE_GLOBAL float *p) {
#if GVAE_STREAM_NT
  return __builtin_nontemporal_load(reinterpret_cast<const GVAE_GLOBAL f32x4_t *>(p));
#else
  return *reinterpret_cast<const GVAE_GLOBAL f32x4_t *>(p);
#endif
}

template <int V>
__device__ __forceinline__ typename VecT<V>::type stream_load(const float *p) {
  using vec_t = typename VecT<V>::type;
#if GVAE_STREAM_NT
  return __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(p));
#else
  return *reinterpret_cast<const vec_t *>(p);
#endif
}

// max over the 2^lpr_log2 adjacent lanes that share one row; every lane of the group ends
// up with the group's maximum.  DPP moves up to 16 lanes, ds_bpermute beyond.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group_max(float m, int lpr_log2) {
  if (lpr_log2 >= 1) m = fmaxf(m, dpp_mov<0xB1>(m));   // quad_perm [1,0,3,2]
  if (lpr_log2 >= 2) m = fmaxf(m, dpp_mov<0x4E>(m));   // quad_perm [2,3,0,1]
  if (lpr_log2 >= 3) m = fmaxf(m, dpp_mov<0x141>(m));  // row_half_mirror
  if (lpr_log2 >= 4) m = fmaxf(m, dpp_mov<0x140>(m));  // row_mirror
  if (lpr_log2 >= 5) m = fmaxf(m, __shfl_xor(m, 16, 64));
  if (lpr_log2 >= 6) m = fmaxf(m, __shfl_xor(m, 32, 64));
  return m;
}

// The slab holds R*R rows (j,a) of RS floats = cols vectors of V floats (RS = R, or R rounded
// up to a multiple of 4 with the row's last element repeated).  One call handles the rows
// (j, a0 + al) for all j and al in [0, Aw): NR = R*Aw rows, numbered r' = j*Aw + al.  A row is
// shared by lpr = 2^lpr_log2 adjacent lanes; lane lr of the group takes the vectors lr,
// lr + lpr, ...  Xs holds X with the same row stride and the same repeated padding.
// T[r'] receives max_b S[i,j,a,b]*X[j,b] for every row, j == i included.
// This is the compact form for rows of 4- and 8-byte vectors (the dense layout with a node
// count that is not a multiple of 4, which only the public matching entry points produce):
// one row per group at a time, shape parameters at run time.
template <int V>
__device__ __noinline__ void mpm_iter_body(const float *__restrict__ slab_, int R, int RS, int a0,
                                           int Aw, int lpr_log2, int xs_off, int t_off) {
  using vec_t = typename VecT<V>::type;
  // (not inlined: device memory and LDS are named as such, and the arguments as wave-uniform;
  // see as_global)
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  R = uniform(R), RS = uniform(RS), a0 = uniform(a0), Aw = uniform(Aw), lpr_log2 = uniform(lpr_log2);
  const float *Xs = lds_all + uniform(xs_off);
  float *T = lds_all + uniform(t_off);
  const GVAE_GLOBAL float *slab = uniform_global(slab_);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int NR = R * Aw;
  const int cols = RS / V;
  const int lpr = 1 << lpr_log2;
  const int lr = tid & (lpr - 1);
  const int rpp = nt >> lpr_log2;       // rows per step of the whole work-group
  for (int row = tid >> lpr_log2; row < NR; row += rpp) {
    const int j = row / Aw, al = row - j * Aw;
    const GVAE_GLOBAL float *src = slab + (unsigned)((j * R + a0 + al) * RS);
    const float *xr = Xs + j * RS;
    float m = -INFINITY;
    for (int c = lr; c < cols; c += 4 * lpr) {
      vec_t sv[4];
      int cc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        cc[u] = min(c + u * lpr, cols - 1) * V;   // past the row end: its last vector again
        sv[u] = stream_load_g<V>(src + cc[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float sx[V], x[V];
        vec_to_arr<V>(sv[u], sx);
        vec_to_arr<V>(*reinterpret_cast<const vec_t *>(xr + cc[u]), x);
#pragma unroll
        for (int v = 0; v < V; ++v) m = fmaxf(m, sx[v] * x[v]);
      }
    }
    m = group_max(m, lpr_log2);
    if (lr == 0) T[row] = m;
  }
}

#ifndef GVAE_STREAM_DEPTH
#define GVAE_STREAM_DEPTH 4
#endif
#ifndef GVAE_POLL_SLEEP
#define GVAE_POLL_SLEEP 16   // units of 64 cycles between two polls of a consumer's queue slot
#endif
#ifndef GVAE_STREAM_WAVES
#define GVAE_STREAM_WAVES 5
#endif

// max with the lane a DPP pattern names, as ONE instruction (the compiler's own sequence for
// fmaxf around a DPP move is four: it re-canonicalises both operands).  s_nop 1: a DPP read of
// a register needs two wait states after the vector instruction that wrote it.
#define GVAE_DPP_MAX(NAME, PATTERN)                                                      \
  __device__ __forceinline__ float NAME(float m) {                                       \
    float r;                                                                             \
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " PATTERN " row_mask:0xf bank_mask:0xf"     \
        : "=v"(r)                                                                        \
        : "v"(m));                                                                       \
    return r;                                                                            \
  }
GVAE_DPP_MAX(dpp_max_xor1, "quad_perm:[1,0,3,2]")
GVAE_DPP_MAX(dpp_max_xor2, "quad_perm:[2,3,0,1]")
GVAE_DPP_MAX(dpp_max_half_mirror, "row_half_mirror")
GVAE_DPP_MAX(dpp_max_mirror, "row_mirror")
template <int LPR_LOG2>
__device__ __forceinline__ float group_max_ct(float m) {
  if constexpr (LPR_LOG2 >= 1) m = dpp_max_xor1(m);
  if constexpr (LPR_LOG2 >= 2) m = dpp_max_xor2(m);
  if constexpr (LPR_LOG2 >= 3) m = dpp_max_half_mirror(m);
  if constexpr (LPR_LOG2 >= 4) m = dpp_max_mirror(m);
  // 32 lanes: the other row of 16 through the LDS crossbar (swap with lane ^ 16)
  if constexpr (LPR_LOG2 >= 5)
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(m), 0x401F)));
  return m;
}

// The same rows as mpm_iter_body for 16-byte vectors, with the lanes per row and the passes
// fixed at compile time and no per-row control flow.  Every group runs the same number of
// steps: rows past the end repeat the last row, which rewrites T[NR-1] with the same value.
// The row -> (j, a) split is a multiplication by 1/Aw in fp32 (exact here: the quotient is
// below 256 and (r + 0.5)/Aw stays 0.5/Aw away from an integer), 24-bit multiplies elsewhere.
template <int LPR_LOG2, int PASSES, int RB>
struct RowStep {
  f32x4_t sv[RB][PASSES];
  int jX[RB], rw[RB];
};

template <int LPR_LOG2, int PASSES, int RB>
__device__ __forceinline__ void row_step_issue(RowStep<LPR_LOG2, PASSES, RB> &st,
                                               const GVAE_GLOBAL float *__restrict__ slab, int R, int RS, int a0,
                                               int Aw, float inv_aw, int NR, int first_row, int rpp,
                                               const int (&coff)[PASSES]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int r = min(first_row + rb * rpp, NR - 1);
    const int j = (int)(((float)r + 0.5f) * inv_aw);
    const int al = r - __mul24(j, Aw);
    const GVAE_GLOBAL float *src = slab + (unsigned)__mul24(__mul24(j, R) + a0 + al, RS);
    st.jX[rb] = __mul24(j, RS);
    st.rw[rb] = r;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) st.sv[rb][p] = stream_load4(src + coff[p]);
  }
}

template <int LPR_LOG2, int PASSES, int RB>
__device__ __forceinline__ void row_step_reduce(const RowStep<LPR_LOG2, PASSES, RB> &st,
                                                const float *Xs, float *T,
                                                const int (&coff)[PASSES]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float *xr = Xs + st.jX[rb];
    float m = -INFINITY;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const f32x4_t x = *reinterpret_cast<const f32x4_t *>(xr + coff[p]);
      const f32x4_t sx = st.sv[rb][p];
      m = fmaxf(m, sx.x * x.x);
      m = fmaxf(m, sx.y * x.y);
      m = fmaxf(m, sx.z * x.z);
      m = fmaxf(m, sx.w * x.w);
    }
    T[st.rw[rb]] = group_max_ct<LPR_LOG2>(m);
  }
}

template <int LPR_LOG2, int PASSES, int RB, bool TWO = false>
__device__ __forceinline__ void mpm_iter_rows_v4(const GVAE_GLOBAL float *__restrict__ slab, int R,
                                                 int RS, int a0, int Aw, const float *Xs, float *T) {
  constexpr int LPR = 1 << LPR_LOG2;
  int tid = threadIdx.x;
  // opaque to the optimiser: what derives from the lane number is computed here, per call,
  // instead of once per shape at kernel entry (where it would stay live across every item)
  asm volatile("" : "+v"(tid));
  const int nt = blockDim.x;
  const int NR = R * Aw, cols = RS >> 2;
  const int lr = tid & (LPR - 1), grp = tid >> LPR_LOG2;
  const int rpp = nt >> LPR_LOG2, step = rpp * RB;
  const float inv_aw = 1.0f / (float)Aw;
  int coff[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) coff[p] = min(lr + p * LPR, cols - 1) * 4;
  // One buffer of RB*PASSES loads per lane.  Measured against a second buffer with the next
  // step's loads in flight during the reduction: no gain at 32 nodes, 10 % slower at 64 (HBM
  // does not reward more than about 64 KiB outstanding per compute unit).
  // TWO: a launch whose LDS footprint leaves one work-group per compute unit has nobody else to
  // cover the round trip of a step's loads, so the next step's loads are issued before the
  // current step is reduced.
  RowStep<LPR_LOG2, PASSES, RB> A;
  if constexpr (!TWO) {
    for (int base = 0; base < NR; base += step) {
      row_step_issue(A, slab, R, RS, a0, Aw, inv_aw, NR, base + grp, rpp, coff);
      row_step_reduce(A, Xs, T, coff);
    }
  } else {
    RowStep<LPR_LOG2, PASSES, RB> B;
    row_step_issue(A, slab, R, RS, a0, Aw, inv_aw, NR, grp, rpp, coff);
    for (int base = 0; base < NR; base += 2 * step) {
      const bool hasB = base + step < NR;
      if (hasB) row_step_issue(B, slab, R, RS, a0, Aw, inv_aw, NR, base + step + grp, rpp, coff);
      row_step_reduce(A, Xs, T, coff);
      if (base + 2 * step < NR)
        row_step_issue(A, slab, R, RS, a0, Aw, inv_aw, NR, base + 2 * step + grp, rpp, coff);
      if (hasB) row_step_reduce(B, Xs, T, coff);
    }
  }
}

// The shapes as functions of their own: each gets registers of its own, whatever else the kernel
// around it holds.  X and T are offsets into the work-group's dynamic LDS.
template <int LPR_LOG2, int PASSES, int RB, bool TWO = false>
__device__ __noinline__ void mpm_iter_body_v4(const float *__restrict__ slab_, int R, int RS,
                                              int a0, int Aw, int xs_off, int t_off) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  R = uniform(R), RS = uniform(RS), a0 = uniform(a0), Aw = uniform(Aw);
  mpm_iter_rows_v4<LPR_LOG2, PASSES, RB, TWO>(uniform_global(slab_), R, RS, a0, Aw,
                                              lds_all + uniform(xs_off), lds_all + uniform(t_off));
}

// Rows of `cols` 16-byte vectors, k = 64 / cols whole consecutive rows per load instruction of a
// wavefront (lane l: row l / cols of the k, vector l % cols): one contiguous run of memory per
// instruction and at most cols - 1 idle lanes, for row lengths that no power-of-two group of
// lanes fits.  The lanes of a row fold their maxima into T[row] with LDS atomics (fp32 max is
// exact, so the order does not matter).
// Not inlined: its registers are not the other shapes'.  X and T are passed as offsets into the
// work-group's dynamic LDS (a pointer argument would be a generic one: flat instead of LDS
// instructions).
template <int RB>
__device__ __noinline__ void mpm_iter_body_flat(const float *__restrict__ slab, int R, int RS,
                                                int a0, int Aw, int xs_off, int t_off) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  R = uniform(R), RS = uniform(RS), a0 = uniform(a0), Aw = uniform(Aw);
  const float *Xs = lds_all + uniform(xs_off);
  float *T = lds_all + uniform(t_off);
  const GVAE_GLOBAL float *gslab = uniform_global(slab);
  const int tid = threadIdx.x;
  const int nw = blockDim.x >> 6, wave = tid >> 6, lane = tid & 63;
  const int NR = R * Aw, cols = RS >> 2;
  for (int e = tid; e < NR; e += blockDim.x) T[e] = -INFINITY;
  __syncthreads();
  const int k = 64 / cols;
  const int rsub = (int)(((float)lane + 0.5f) * (1.0f / (float)cols));
  const int cvec4 = (lane - rsub * cols) * 4;
  const float inv_aw = 1.0f / (float)Aw;
  if (rsub < k) {
    for (int base = 0; base < NR; base += nw * k * RB) {
      f32x4_t sv[RB];
      int jX[RB], rw[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int r = min(base + (rb * nw + wave) * k + rsub, NR - 1);
        const int j = (int)(((float)r + 0.5f) * inv_aw);
        const int al = r - __mul24(j, Aw);
        sv[rb] = stream_load4(gslab + (unsigned)(__mul24(__mul24(j, R) + a0 + al, RS) + cvec4));
        jX[rb] = __mul24(j, RS) + cvec4;
        rw[rb] = r;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const f32x4_t x = *reinterpret_cast<const f32x4_t *>(Xs + jX[rb]);
        float m = fmaxf(fmaxf(sv[rb].x * x.x, sv[rb].y * x.y), fmaxf(sv[rb].z * x.z, sv[rb].w * x.w));
        __hip_atomic_fetch_max(&T[rw[rb]], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

// picks the lanes-per-row exponent and the pass count for rows of `cols` vectors of V floats:
// 128-byte pieces per row per load instruction where the row is long enough, at most 4 passes
template <int V, bool TWO = false>
__device__ __forceinline__ void mpm_iter_dispatch(const float *__restrict__ slab, int R, int RS,
                                                  int a0, int Aw, const float *Xs, float *T,
                                                  int wide = 0) {
  const int cols = RS / V;
  int lpr_log2 = 0;
  constexpr int kBase = V == 4 ? 3 : (V == 2 ? 4 : 5);
  while ((1 << lpr_log2) < cols && lpr_log2 < kBase) ++lpr_log2;
  while ((cols + (1 << lpr_log2) - 1) >> lpr_log2 > 4) ++lpr_log2;
  // wide: rows of 9..19 vectors that are not a multiple of 128 bytes go to 16 lanes each, so
  // that one load instruction of a wavefront covers whole consecutive rows (one contiguous run
  // of memory, every cache line touched once) instead of the first 128 bytes of eight rows.
  // Steady state, TB/s with 8 / 16 lanes per row: 40 nodes 5.66 / 6.14, 44: 5.47 / 6.21, 52:
  // 5.41 / 6.19, 57: 5.20 / 5.90, 60: 5.56 / 6.25, 70: 5.39 / 5.97, 75: 5.25 / 5.43; from 21
  // vectors on (82 nodes: 5.25 / 4.56) the idle lanes of the second pass cost more.
  if (V == 4 && wide && (cols & 7) != 0 && cols > 8 && cols <= 19) lpr_log2 = 4;
  // The same from 20 vectors on with 32 lanes per row (two whole rows per instruction), TB/s with
  // 8 lanes x 3 or 4 passes / 32 lanes: 78 nodes 6.15 / 6.31, 82: 5.41 / 6.05, 91: 5.66 / 6.36,
  // 99: 5.42 / 6.44, 104: 5.74 / 6.49, 118: 5.82 / 6.40; below that the idle lanes weigh more
  // (16 lanes x 2 passes / 32 lanes: 66 nodes 5.56 / 5.19, 70: 5.96 / 5.74, 75: 5.39 / 5.13).
  if (V == 4 && wide && (cols & 7) != 0 && cols >= 20 && cols <= 32) lpr_log2 = 5;
  const int passes = (cols + (1 << lpr_log2) - 1) >> lpr_log2;
  constexpr int D = GVAE_STREAM_DEPTH;  // loads in flight per lane
  if constexpr (V == 4 && !TWO) {
    // rows of 9 or 10 vectors (33 ... 40 nodes), which leave 16 lanes per row nearly half idle:
    // seven or six whole rows per instruction.  TB/s, 16 lanes per row / whole rows: 36 nodes
    // 5.57 / 6.19, 40: 5.97 / 6.21; not where the gain in busy lanes is smaller than what the
    // LDS atomics cost (24 nodes 4.99 / 5.17, 44: 6.10 / 6.19, 52: 6.18 / 5.85, 66: 5.56 / 5.42,
    // 70: 5.96 / 5.77, 75: 5.37 / 5.51)
    if (wide && (cols == 9 || cols == 10)) {
      extern __shared__ __attribute__((aligned(16))) float lds_all[];
      return mpm_iter_body_flat<D>(slab, R, RS, a0, Aw, (int)(Xs - lds_all), (int)(T - lds_all));
    }
  }
  if constexpr (V == 4) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int xs_off = (int)(Xs - lds_all), t_off = (int)(T - lds_all);
    // every shape up to 35 vectors per row (R <= 140, the LDS limit of the kernel)
    const int key = lpr_log2 * 8 + passes;
    switch (key) {
      case 0 * 8 + 1: return mpm_iter_rows_v4<0, 1, D, TWO>(as_global(slab), R, RS, a0, Aw, Xs, T);
      case 1 * 8 + 1: return mpm_iter_rows_v4<1, 1, D, TWO>(as_global(slab), R, RS, a0, Aw, Xs, T);
      case 2 * 8 + 1: return mpm_iter_rows_v4<2, 1, D, TWO>(as_global(slab), R, RS, a0, Aw, Xs, T);
      case 3 * 8 + 1: return mpm_iter_rows_v4<3, 1, D, TWO>(as_global(slab), R, RS, a0, Aw, Xs, T);
      // (inline: rows of up to 8 vectors, whose calls are short -- 16 ... 24 nodes measured 3-5 %
      // slower as functions -- and 64-node rows, 0.856 against 0.889 of the HBM peak)
      case 3 * 8 + 2:
        return mpm_iter_rows_v4<3, 2, (D + 1) / 2, TWO>(as_global(slab), R, RS, a0, Aw, Xs, T);
      case 3 * 8 + 3: return mpm_iter_body_v4<3, 3, (D + 2) / 3, TWO>(slab, R, RS, a0, Aw, xs_off, t_off);
      case 3 * 8 + 4: return mpm_iter_body_v4<3, 4, (D + 3) / 4, TWO>(slab, R, RS, a0, Aw, xs_off, t_off);
      case 4 * 8 + 1: return mpm_iter_body_v4<4, 1, D, TWO>(slab, R, RS, a0, Aw, xs_off, t_off);
      case 4 * 8 + 2: return mpm_iter_body_v4<4, 2, (D + 1) / 2, TWO>(slab, R, RS, a0, Aw, xs_off, t_off);
      case 4 * 8 + 3: return mpm_iter_body_v4<4, 3, (D + 2) / 3, TWO>(slab, R, RS, a0, Aw, xs_off, t_off);
      case 5 * 8 + 1: return mpm_iter_body_v4<5, 1, D, TWO>(slab, R, RS, a0, Aw, xs_off, t_off);
      default: break;
    }
  }
  {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    mpm_iter_body<V>(slab, R, RS, a0, Aw, lpr_log2, (int)(Xs - lds_all), (int)(T - lds_all));
  }
}

// value of `v` in lane `lane` (a compile-time constant after unrolling) for the whole wavefront
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// maximum over the wavefront, in every lane: DPP within rows of 16 lanes, then the four rows
__device__ __forceinline__ float wave_max(float m) {
  m = group_max(m, 4);
  return fmaxf(fmaxf(lane_value(m, 0), lane_value(m, 16)), fmaxf(lane_value(m, 32), lane_value(m, 48)));
}

// chain over x[0..n) (n % 8 == 0) continuing the 8 lane accumulators; lanes 0..7 of wave 0
__device__ __forceinline__ float norm_chain(const float *x, int n, int lane, float acc) {
  // the fused chain itself is serial (one dependent fma per element of the lane); the next
  // 16 operands are read from LDS while the current 16 are consumed, which covers the LDS
  // latency with 16 dependent fma (8 would not)
  constexpr int NB = 16;
  int d = 0;
  if (n >= 8 * NB) {
    float v[NB], w[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) v[t] = x[8 * t + lane];
    for (d = 8 * NB; d + 8 * NB <= n; d += 8 * NB) {
#pragma unroll
      for (int t = 0; t < NB; ++t) w[t] = x[d + 8 * t + lane];
#pragma unroll
      for (int t = 0; t < NB; ++t) acc = __fmaf_rn(v[t], v[t], acc);
#pragma unroll
      for (int t = 0; t < NB; ++t) v[t] = w[t];
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) acc = __fmaf_rn(v[t], v[t], acc);
  }
  for (; d < n; d += 8) {
    const float v = x[d + lane];
    acc = __fmaf_rn(v, v, acc);
  }
  return acc;
}

// fold lanes 0..7 in order and fuse the tail (executed by the whole first wave)
__device__ __forceinline__ float norm_finish(float acc, const float *tail, int ntail) {
  float s = lane_value(acc, 0);
#pragma unroll
  for (int l = 1; l < 8; ++l) s = s + lane_value(acc, l);
  int k = 0;
  if (ntail >= 4)
    for (; k < 4; ++k) s = s + __fmul_rn(tail[k], tail[k]);  // rounded squares, not fused
  for (; k < ntail; ++k) s = __fmaf_rn(tail[k], tail[k], s);
  return sqrtf(s);
}

// The whole MPM of a batch in ONE launch, driven by a work queue.
//   * work item = (graph g, row i, column block q) of one iteration: stream the rows (j, a) of
//     the slab S[g][i,:,:,:] for all j and the a of block q once and produce those columns of
//     row i of Xnew[g].  A slab of up to 2 MiB is one block; larger slabs, and the slabs of
//     the last few unconverged graphs of a batch, are cut into up to 8 column blocks
//     (mpm_col_blocks), chosen anew at every iteration of a graph;
//   * the producers of a graph hand their pieces to whichever of them arrives last (per-graph
//     arrival counter: rows stored write-through and drained before the producer's ticket;
//     the last arriver reads them past its vector cache).  That work-group
//     normalises (nrm = ||Xnew||_2 in canonical order, X = Xnew / nrm) and, unless the graph
//     has converged (max|X - Xprev| < tol) or used up its iterations, pushes the graph's R
//     items of the next iteration into the queue;
//   * the queue is a ring of 8-byte {lap tag, R, item} granules with a head and a tail ticket
//     counter.  A consumer draws a ticket and polls ITS OWN slot until the granule of its
//     generation appears (or until no graph is left), then clears the slot.  At most
//     sum_g (items of g) <= capacity items are ever undrawn (a graph's next items are pushed only after
//     all of its current ones were consumed), so when a producer reaches a slot the ticket one
//     lap earlier has already been drawn; it only has to wait for that consumer's clear, which
//     never depends on another work-group.  Every spin is bounded and reports through the
//     error word instead of hanging.
// Graphs advance independently: a slow-converging graph's chain of iterations overlaps with
// everything else instead of costing one under-filled launch per iteration.
#define MPM_NULL_ITEM 0xFFFFFFFFu
// item word: graph (16 bits) | first row i (8 bits) | column block q (4 bits) | shape code (4 bits)
#define MPM_ITEM(g, i, q, nb) \
  (((unsigned)(g) << 16) | ((unsigned)(i) << 8) | ((unsigned)(q) << 4) | (unsigned)(nb))
#define MPM_MAX_COL_BLOCKS 8
#define MPM_SPIN_LIMIT (1u << 21)
// queue granule: lap tag (16 bits, never 0) | node count R of the item's graph (8 bits) | 0 (8 bits)
// | item word (32 bits).  R travels with the item so that a consumer can address X at once.
#define MPM_LAP_TAG(ticket, cap) ((unsigned)(((ticket) / (cap)) % 65535u) + 1u)
#define MPM_GRANULE(tag, R, item) \
  (((unsigned long long)(tag) << 48) | ((unsigned long long)(R) << 40) | (unsigned long long)(unsigned)(item))

// Column blocks per row of an R-node graph.  Base rule: slab bytes / 2 MiB rounded up, so that
// one large graph can occupy the whole device.  When the items in flight (graphs still
// iterating x R x blocks) are fewer than `low`, the rows are cut finer: a nearly empty device
// is bound by the latency of a graph's chain of iterations, and smaller pieces shorten the
// streaming phase of each link (measured: worth it below about a thousand items; every item
// costs a ticket, a 4 KiB X load and an arrival, each a memory round trip).  The result nb satisfies
// nb == ceil(R / ceil(R / nb)), so a consumer recovers the block width from nb alone.
__host__ __device__ __forceinline__ int mpm_col_blocks(int R, int RS, int alive, int low, int cap_kb,
                                                       int t_cap) {
  const long long slab = 4LL * R * R * RS;
  const long long cap = (long long)cap_kb << 10;
  long long t = (slab + cap - 1) / cap;
  t = t < 1 ? 1 : t;
  const long long in_flight = (long long)(alive < 1 ? 1 : alive) * R * t;
  if (in_flight < low) t *= (low + in_flight - 1) / in_flight;
  // the row maxima of a block (R rows of the block's width) must fit the LDS set aside for
  // them: t_cap floats (the launch guarantees R * ceil(R / 8) <= t_cap for its largest graph)
  const int aw_max = t_cap / R;
  const int t_min = (R + aw_max - 1) / aw_max;
  t = t < t_min ? t_min : t;
  int nb = t > MPM_MAX_COL_BLOCKS ? MPM_MAX_COL_BLOCKS : (int)t;
  nb = nb > R ? R : nb;
  for (;;) {
    const int Aw = (R + nb - 1) / nb;
    const int nb2 = (R + Aw - 1) / Aw;
    if (nb2 == nb) return nb;
    nb = nb2;
  }
}

// Shape of a graph's work items for one iteration, as a 4-bit code:
//   1..8   one row i per item, cut into that many column blocks (mpm_col_blocks);
//   9..11  2, 4 or 8 consecutive rows i per item (whole rows): with many small slabs in flight
//          the per-item costs (ticket, X reload, hand-off: about 12 ns of them serialised over
//          the whole device) weigh more than the loss of parallelism, so items are made
//          larger, up to about 512 KiB, as long as `high` items stay in flight.
__host__ __device__ __forceinline__ int mpm_item_shape(int R, int RS, int alive, int low, int high,
                                                       int cap_kb, int t_cap) {
  const long long slab = 4LL * R * R * RS;
  const long long in_flight = (long long)(alive < 1 ? 1 : alive) * R;
  int code = 0;
  // (items of up to 128 KiB are short enough that half as many in flight keep the device busy)
  for (int rpi = 8, c = 11; rpi >= 2; rpi >>= 1, --c)
    if (!code && slab * rpi <= (512LL << 10) && rpi <= R && R * R <= t_cap &&
        in_flight / rpi >= (slab * rpi <= (128LL << 10) ? high / 2 : high))
      code = c;
  return code ? code : mpm_col_blocks(R, RS, alive, low, cap_kb, t_cap);
}
__host__ __device__ __forceinline__ int mpm_shape_blocks(int code) { return code <= 8 ? code : 1; }
__host__ __device__ __forceinline__ int mpm_shape_rows(int code) { return code <= 8 ? 1 : 2 << (code - 9); }
// items of one iteration of an R-node graph in that shape
__host__ __device__ __forceinline__ int mpm_shape_items(int R, int code) {
  const int rpi = mpm_shape_rows(code);
  return rpi > 1 ? (R + rpi - 1) / rpi : R * code;
}
// item number p of the iteration -> item word
__host__ __device__ __forceinline__ unsigned mpm_shape_item(int g, int p, int code) {
  const int rpi = mpm_shape_rows(code);
  return rpi > 1 ? MPM_ITEM(g, p * rpi, 0, code) : MPM_ITEM(g, p / code, p % code, code);
}

struct MpmQueue {
  unsigned long long *slots;  // [cap]
  int32_t *words;             // [0] head  [32] tail  [64] remaining graphs  [65] their slab KiB  [96] error
  int32_t *arrive;            // [n_graphs]
  unsigned cap;
  int two_buffers;            // one work-group per compute unit: two steps of loads in flight
  int wide_rows;              // 16 or 32 lanes per row for rows that are not a multiple of 128 bytes
  int low;                    // items in flight below which rows are cut into more column blocks
  int high;                   // items in flight that must remain when rows are grouped
  int item_cap_kb;            // a row's slab above this many KiB is cut into column blocks
  // LDS of a work-group after the control words: X of the item's graph (R padded rows), then
  // t_cap floats for the row maxima of the item's column block (R x block width; a row is cut
  // into at least as many blocks as that takes) and for the pieces in which the last arriver of
  // an iteration folds the norm, then, x_floats + t_cap floats in, the diagonal factors
  int t_cap, x_floats;
  // Hand-off of the latency-bound tail (gvae_match_stream only; handoff == nullptr: never).  Once
  // the slabs of all graphs still iterating add up to less than handoff_kb KiB, one pass over
  // them takes the device less time than the chain of dependent steps of one iteration (ticket,
  // X, slab pieces, hand-off to the last arriver, normalisation, push: about 19 us), so the
  // batch is bound by that latency, not by HBM.  A graph of at most handoff_rmax nodes that
  // finishes an iteration in that state leaves the queue with handoff[g] = 1, and the S-free
  // kernel (same recurrence, same bits, no HBM traffic in its loop) runs its remaining
  // iterations after this launch.
  int32_t *handoff;           // [n_graphs] out (nullptr: no hand-off in this launch)
  int32_t *handoff_flags;     // the same words, always: cleared by every launch's set-up
  const int32_t *sfree_ok;    // [n_graphs] bit 1: the S-free form is valid for the graph
  int handoff_kb, handoff_rmax;
  unsigned long long *stats;  // [0] bytes streamed (4R^4 + 12R^2 per executed graph-iteration)  [1] graph-iterations
};
#define Q_HEAD(q) ((q).words + 0)
#define Q_TAIL(q) ((q).words + 32)
#define Q_REMAIN(q) ((q).words + 64)
#define Q_ALIVE_KB(q) ((q).words + 65)
#define Q_ERROR(q) ((q).words + 96)
// KiB of a graph's similarity tensor as the kernel streams it (one pass = one iteration)
__host__ __device__ __forceinline__ int mpm_slab_kb(int R, int RS) {
  return (int)(((long long)R * R * R * RS * 4 + 1023) >> 10);
}

// The last arriver of a graph's iteration: nrm = ||X_new||_2 in canonical order, X = X_new / nrm
// written through to memory, and the largest |X - X_prev| of each wavefront in red[wave]
// (infinity where a difference is NaN).  X_new was stored write-through by other work-groups
// and is read past the vector cache.  The norm's chains run over X_new in LDS: a graph whose
// X_new exceeds the t_cap floats at T passes through in pieces (a multiple of 8 elements each:
// the 8 lane accumulators just continue) and takes the values from memory once more for the
// division.  Not inlined: its registers are not the streaming loop's.  X and T are offsets into
// the work-group's dynamic LDS (pointer arguments would be generic ones: flat instead of LDS
// instructions); red = its first floats, red[16] carries the norm.
__device__ __noinline__ void mpm_renormalise(const float *Xn_, float *Xg_, int ldx, int xs_off,
                                             int t_off, int t_cap, int R, int RS) {
  const GVAE_GLOBAL float *Xn = uniform_global(Xn_);
  GVAE_GLOBAL float *Xg = uniform_global(Xg_);
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  ldx = uniform(ldx), t_cap = uniform(t_cap), R = uniform(R), RS = uniform(RS);
  float *red = lds_all;
  const float *Xs = lds_all + uniform(xs_off);
  float *T = lds_all + uniform(t_off);
  const int tid = threadIdx.x, nt = blockDim.x;
  const int R2 = R * R, n8 = R2 & ~7;
  const bool whole = R2 <= t_cap;
  float acc = 0.f;
  for (int c0 = 0; c0 < R2; c0 += t_cap) {
    const int len = min(t_cap, R2 - c0);
    if (c0) __syncthreads();                 // the previous piece has been consumed
    // (four loads in flight per lane: each is a round trip to L2)
    for (int e0 = tid; e0 < len; e0 += 4 * nt) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = __hip_atomic_load(&Xn[c0 + min(e0 + u * nt, len - 1)], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + u * nt < len) T[e0 + u * nt] = v[u];
    }
    __syncthreads();
    if (tid < 64) {
      if (tid < 8) acc = norm_chain(T, min(n8 - c0, len) & ~7, tid, acc);
      if (c0 + len == R2) {
        const float nrm = norm_finish(acc, T + (n8 - c0), R2 - n8);
        if (tid == 0) red[16] = nrm;
      }
    }
  }
  __syncthreads();
  const float nrm = red[16];
  float dmax = 0.f;
  bool isnan_any = false;
  auto renormalise = [&](int e, float xn) {
    const float v = xn / nrm;
    const int r = e / R, c = e - r * R;
    const float d = fabsf(v - Xs[r * RS + c]);
    isnan_any |= (d != d);
    dmax = fmaxf(dmax, d);
    // write-through: the next iteration's consumers run on other compute units
    __hip_atomic_store(&Xg[(int64_t)r * ldx + c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (whole) {
    for (int e = tid; e < R2; e += nt) renormalise(e, T[e]);
  } else {
    for (int e0 = tid; e0 < R2; e0 += 4 * nt) {
      float xn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        xn[u] = __hip_atomic_load(&Xn[min(e0 + u * nt, R2 - 1)], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + u * nt < R2) renormalise(e0 + u * nt, xn[u]);
    }
  }
  if (isnan_any) dmax = INFINITY;  // a NaN difference never satisfies `< tol`
  dmax = wave_max(dmax);
  if ((tid & 63) == 0) red[tid >> 6] = dmax;
  __syncthreads();
}

__global__ __attribute__((amdgpu_waves_per_eu(GVAE_STREAM_WAVES, GVAE_STREAM_WAVES))) void mpm_stream_kernel(const float *__restrict__ S, const int64_t *__restrict__ S_off,
                                  const int32_t *__restrict__ Rarr, int r_max,
                                  float *__restrict__ X, int ldx, int64_t strideX,
                                  float *__restrict__ Xnew, int64_t strideXnew, float tol,
                                  int early_stop, int iters, int layout, MpmQueue q,
                                  int32_t *__restrict__ iters_done) {
  // The control words live at the front of the dynamic region (kCtlFloats floats): a static
  // __shared__ object would precede the dynamic region unpadded and push its base off the
  // 16-byte alignment that the 128-bit LDS reads of X need (misaligned ones are replayed).
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float *red = lds_all;                                  // [16]
  int *ctl = reinterpret_cast<int *>(lds_all + 17);
  int &s_item = ctl[0], &s_R = ctl[1], &s_last = ctl[2], &s_cont = ctl[3], &s_base = ctl[4],
      &s_nb = ctl[5], &s_done = ctl[6], &s_alive = ctl[7], &s_alive_kb = ctl[8];
  float *lds = lds_all + kCtlFloats;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (;;) {
    // ---- draw a ticket and wait for its granule (one lane)
    if (tid == 0) {
      const unsigned ticket = (unsigned)__hip_atomic_fetch_add(Q_HEAD(q), 1, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT);
      const unsigned slot = ticket % q.cap, want = MPM_LAP_TAG(ticket, q.cap);
      int got = 0;
      unsigned item = MPM_NULL_ITEM;
      int tail_seen = -1;
      for (unsigned spins = 0;; ++spins) {
        const unsigned long long v =
            __hip_atomic_load(&q.slots[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 48) == want) {
          item = (unsigned)v;
          s_R = (int)((v >> 40) & 0xFFu);
          got = 1;
          // free the slot for the ticket one lap later
          __hip_atomic_store(&q.slots[slot], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if ((spins & 3u) == 3u) {
          if (__hip_atomic_load(Q_REMAIN(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0)
            break;                                 // every graph is finished: nothing will come
          // the bound is on waiting WITHOUT progress of the queue: a consumer whose ticket lies
          // beyond the tail idles for as long as the launch runs (seconds at hundreds of
          // iterations), which is not a fault while items are still being pushed
          const int tail_now = __hip_atomic_load(Q_TAIL(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (tail_now != tail_seen) {
            tail_seen = tail_now;
            spins = 0;
          }
        }
        if (spins > MPM_SPIN_LIMIT) {              // never expected; do not hang the device
          __hip_atomic_store(Q_ERROR(q), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(GVAE_POLL_SLEEP);
      }
      s_item = got ? (int)item : -2;
    }
    __syncthreads();
    // the item is the same for every lane: say so, which keeps what derives from it in scalar
    // registers (addresses, loop bounds)
    const int item = __builtin_amdgcn_readfirstlane(s_item);
    if (item == -2) break;
    if ((unsigned)item != MPM_NULL_ITEM) {
      const int g = (unsigned)item >> 16, i0 = (item >> 8) & 0xFF, q_blk = (item >> 4) & 0xF;
      const int shape = item & 0xF;
      const int n_blk = mpm_shape_blocks(shape);              // column blocks per row i
      const int R = __builtin_amdgcn_readfirstlane(s_R);
      const int R2 = R * R;
      const int RS = s_row_stride(R, layout);
      const int rows_here = min(mpm_shape_rows(shape), R - i0);
      // issued ahead of the X loads so that one round trip covers them all: the offset of the
      // graph's S block, and (one lane) what the last arriver of this iteration will need
      const int64_t s_off_g = S_off[g];
      int done_r = 0, alive_r = 0, alive_kb_r = 0;
      if (tid == 0) {
        done_r = __hip_atomic_load(&iters_done[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alive_r = __hip_atomic_load(Q_REMAIN(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alive_kb_r = __hip_atomic_load(Q_ALIVE_KB(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int Aw_full = (R + n_blk - 1) / n_blk;
      const int a0 = q_blk * Aw_full, Aw = min(Aw_full, R - a0);
      float *Xs = lds;            // R rows of RS floats (padding repeats the last element)
      float *T = lds + R * RS;    // R * Aw (the last arriver: pieces of up to t_cap floats)
      float *nsrow = lds + q.x_floats + q.t_cap;   // rows_here * R
      float *Xg = X + g * strideX;
      float *Xn = Xnew + g * strideXnew;
      // X was written by another work-group of this launch (write-through stores, drained
      // before the item was pushed): read it past this CU's vector cache, which is the only
      // cache another CU's stores do not reach
      // (four loads in flight per lane: each is a round trip to L2)
      for (int e0 = tid; e0 < R * RS; e0 += 4 * nt) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = min(e0 + u * nt, R * RS - 1);
          const int r = e / RS, c = e - r * RS;
          v[u] = __hip_atomic_load(&Xg[(int64_t)r * ldx + min(c, R - 1)], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e0 + u * nt < R * RS) Xs[e0 + u * nt] = v[u];
      }
      // S[i,i,a,a], the factor of X[i,a] in the update, for every row i of the item: a strided
      // gather that shares the round trip of the X loads
      const float *Sg = S + s_off_g;
      for (int e = tid; e < rows_here * Aw; e += nt) {
        const int ri = e / Aw, p = a0 + (e - ri * Aw), i = i0 + ri;
        nsrow[ri * R + p] = Sg[((int64_t)i * R2 + (int64_t)i * R + p) * RS + p];
      }
      if (tid == 0) {
        s_done = done_r;
        s_alive = alive_r;
        s_alive_kb = alive_kb_r;
      }
      __syncthreads();
      for (int ri = 0; ri < rows_here; ++ri) {
        const int i = i0 + ri;
        const float *slab = Sg + (int64_t)i * R2 * RS;
        const float *nsr = nsrow + ri * R;
        if ((RS & 3) == 0) {
          if (q.two_buffers) mpm_iter_dispatch<4, true>(slab, R, RS, a0, Aw, Xs, T);
          else mpm_iter_dispatch<4>(slab, R, RS, a0, Aw, Xs, T, q.wide_rows);
        }
        else if ((RS & 1) == 0) mpm_iter_dispatch<2>(slab, R, RS, a0, Aw, Xs, T);
        else mpm_iter_dispatch<1>(slab, R, RS, a0, Aw, Xs, T);
        __syncthreads();
        for (int p = a0 + tid; p < a0 + Aw; p += nt) {
          // the row j == i does not take part in the sum: an exact +0 at its position
          const float et = column_sum(
              [&](int jj) { return jj == i ? 0.0f : T[jj * Aw + (p - a0)]; }, p, R);
          // write-through store (no release fence needed for a write-through payload)
          __hip_atomic_store(&Xn[i * R + p], Xs[i * RS + p] * nsr[p] + et, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        if (ri + 1 < rows_here) __syncthreads();   // T is rewritten for the next row
      }
      // publish this row: every storing wave drains its stores, then one lane takes a ticket
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int before = __hip_atomic_fetch_add(&q.arrive[g], rows_here, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        s_last = (before + rows_here == R * n_blk) ? 1 : 0;
      }
      __syncthreads();
      if (s_last) {
        // ---- last arriver: normalise the graph
        // (the other producers' rows: write-through stores, read past the vector cache)
        mpm_renormalise(Xn, Xg, ldx, (int)(Xs - lds_all), (int)(T - lds_all), q.t_cap, R, RS);
        if (tid == 0) {
          float d = red[0];
          for (int w = 1; w < (nt >> 6); ++w) d = fmaxf(d, red[w]);
          // the iteration count and the number of live graphs were read when the item started
          // (the count only changes here, once per iteration; the other is a heuristic input)
          const int done = s_done + 1;
          __hip_atomic_store(&iters_done[g], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&q.arrive[g], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int cont = (!(early_stop && d < tol) && done < iters) ? 1 : 0;
          // one executed graph-iteration: S once, X read, X_new written, node block
          __hip_atomic_fetch_add(&q.stats[0], 4ull * R2 * R2 + 12ull * R2, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(&q.stats[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // latency-bound tail: the graph's remaining iterations go to the S-free kernel
          // (the number of live KiB was read when the item started: a heuristic input)
          // (a graph that has the device to itself streams its own tensor at about a quarter of
          // the device's rate, so its iteration takes the chain latency plus 4 x slab / rate:
          // the batch is latency-bound for THIS graph below handoff_kb + 4 x its slab)
          if (cont && q.handoff && R <= q.handoff_rmax &&
              (long long)s_alive_kb < (long long)q.handoff_kb + 4LL * mpm_slab_kb(R, RS) &&
              (q.sfree_ok[g] & 2)) {
            cont = 0;
            q.handoff[g] = 1;
          }
          if (!cont)
            __hip_atomic_fetch_add(Q_ALIVE_KB(q), -mpm_slab_kb(R, RS), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
          if (cont) {
            const int shape_next = mpm_item_shape(R, RS, s_alive, q.low, q.high, q.item_cap_kb, q.t_cap);
            s_nb = shape_next;
            s_base = __hip_atomic_fetch_add(Q_TAIL(q), mpm_shape_items(R, shape_next),
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          s_cont = cont;
        }
        // X, the iteration count and the reset counter are drained before anything is pushed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s_cont) {
          const unsigned base = (unsigned)s_base;
          const int shape_next = s_nb, n_next = mpm_shape_items(R, shape_next);
          for (int p = tid; p < n_next; p += nt) {
            const unsigned t = base + (unsigned)p;
            const unsigned long long gran =
                MPM_GRANULE(MPM_LAP_TAG(t, q.cap), R, mpm_shape_item(g, p, shape_next));
            unsigned long long *sl = &q.slots[t % q.cap];
            // the ticket one lap earlier has been drawn by now (at most `cap` items are ever
            // undrawn), so its consumer is polling this slot and frees it promptly
            // (on the first lap the slot is still as the host zeroed it: nothing to wait for)
            for (unsigned spins = 0;
                 t >= q.cap &&
                 __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull; ++spins) {
              if (spins > MPM_SPIN_LIMIT) {
                __hip_atomic_store(Q_ERROR(q), 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
              __builtin_amdgcn_s_sleep(4);
            }
            __hip_atomic_store(sl, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else if (tid == 0) {
          __hip_atomic_fetch_add(Q_REMAIN(q), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();  // s_item and the LDS buffers are free again
  }
}

// first generation of the ring, compact: the items of graph g start at the sum of the item
// counts of the graphs before it (one work-group: block scan over the graphs, then the fill).
// The slots beyond the total stay empty (zeroed by the host) until the tail reaches them.
__global__ __launch_bounds__(1024) void mpm_queue_init_kernel(MpmQueue q,
                                                              const int32_t *__restrict__ Rarr,
                                                              int n_graphs, int r_max, int layout,
                                                              int32_t *__restrict__ iters_done,
                                                              const int32_t *__restrict__ first_stop,
                                                              int iters) {
  // first_stop != nullptr: the first iteration (mpm_first_iter_kernel), and for small graphs the
  // second (edge_sim_iter_kernel), have been done already: iters_done holds their count, and a
  // graph that converged in them, or whose budget they were, gets no items
  __shared__ int s_part[1024];
  __shared__ int s_live[1024];
  const int tid = threadIdx.x;
  const int per = (n_graphs + 1023) / 1024;
  const int g_lo = min(tid * per, n_graphs), g_hi = min(g_lo + per, n_graphs);
  auto items_of = [&](int g, int &shape) {
    const int R = Rarr[g];
    if (R > r_max) *Q_ERROR(q) = 3;        // larger than the launch was sized for: not run, reported
    if (R <= 0 || R > r_max) { shape = 0; return 0; }
    if (first_stop && ((first_stop[g] & 1) != 0 || iters_done[g] >= iters)) { shape = 0; return 0; }
    shape = mpm_item_shape(R, s_row_stride(R, layout), n_graphs, q.low, q.high, q.item_cap_kb, q.t_cap);
    return mpm_shape_items(R, shape);
  };
  __shared__ int s_kb;
  if (tid == 0) s_kb = 0;
  __syncthreads();
  int mine = 0, live = 0, kb = 0;
  for (int g = g_lo; g < g_hi; ++g) {
    int nb;
    const int n = items_of(g, nb);
    mine += n;
    live += n > 0;
    if (n > 0) kb += mpm_slab_kb(Rarr[g], s_row_stride(Rarr[g], layout));
    q.arrive[g] = 0;
    q.handoff_flags[g] = 0;   // whether or not this launch hands graphs over: the tail reads them
    if (!first_stop) iters_done[g] = 0;
  }
  s_part[tid] = mine;
  s_live[tid] = live;
  if (kb) atomicAdd(&s_kb, kb);
  __syncthreads();
  // inclusive scan (Hillis-Steele over 1024 entries)
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = tid >= off ? s_part[tid - off] : 0;
    const int l = tid >= off ? s_live[tid - off] : 0;
    __syncthreads();
    s_part[tid] += v;
    s_live[tid] += l;
    __syncthreads();
  }
  int pos = s_part[tid] - mine;
  for (int g = g_lo; g < g_hi; ++g) {
    int nb;
    const int n = items_of(g, nb);
    for (int k = 0; k < n; ++k)
      q.slots[pos + k] = MPM_GRANULE(1u, Rarr[g], mpm_shape_item(g, k, nb));
    pos += n;
  }
  if (tid == 1023) {
    *Q_TAIL(q) = s_part[1023];
    *Q_REMAIN(q) = s_live[1023];
    *Q_ALIVE_KB(q) = s_kb;
  }
}


// ---------------------------------------------------------------------------
// First MPM iteration of the streamed matching, without reading S.  X0 is uniform
// (x0 = 1/R, reference model.py:317), so for every row (j,a) of the slab of i
//   T[j,a] = max_b fl(S[i,j,a,b] * x0) = fl((max_b S[i,j,a,b]) * x0)        (rounding is monotone)
// and the row maximum needs no tensor: for j != i the row holds fl(A[i,j]*P[a,b]) for b != a and
// a zero at b == a, hence
//   max_b S[i,j,a,b] = max(0, fl(A[i,j] * pmax[a])),   pmax[a] = max_{b != a} P[a,b]
// (pmin[a] in place of pmax[a] when A[i,j] < 0).  The rest is the iteration as the streaming
// kernel does it: Xn[i,a] = fl(x0 * NS[i,a]) + canonical column sum of T over j (j == i: +0),
// canonical norm, X = Xn / nrm, convergence test against x0.  One work-group per graph; every
// later iteration streams S.  Writes X, iters_done[g] = 1 and stop[g] (bit 0: converged; bit 1: A is
// 0/1 off the diagonal and P >= 0, i.e. the S-free form of the recurrence is valid for the graph).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mpm_first_iter_kernel(
    const float *__restrict__ A, const float *__restrict__ P, const float *__restrict__ NS,
    const int32_t *__restrict__ Rarr, int r_max, int ld, int64_t stride, float *__restrict__ X, int ldx,
    int64_t strideX, float tol, int early_stop, int a_in_lds, int32_t *__restrict__ iters_done,
    int32_t *__restrict__ stop) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float *red = lds_all;                                  // [16]
  float &s_nrm = lds_all[16];
  float *Xn = lds_all + kCtlFloats;
  const int g = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int R = Rarr[g];
  if (R <= 0 || R > r_max) {       // nothing to iterate / reported by the queue's set-up kernel
    if (tid == 0) {
      iters_done[g] = 0;
      stop[g] = 1;
    }
    return;
  }
  const int R2 = R * R, R2p = (R2 + 3) & ~3;
  float *pmax = Xn + R2p, *pmin = pmax + R;
  float *As = pmin + R;            // R*R copy of A when it fits (a_in_lds), else A is read in place
  const float *Ag = A + g * stride, *Pg = P + g * stride, *NSg = NS + g * stride;
  float *Xg = X + g * strideX;
  const float x0 = (float)(1.0 / (double)R);
  // row extrema of P without its diagonal: 8 lanes per row, folded with three shuffles
  for (int a0 = 0; a0 < R; a0 += nt >> 3) {
    const int a = a0 + (tid >> 3), l = tid & 7;
    float mx = -INFINITY, mn = INFINITY;
    if (a < R)
      for (int b = l; b < R; b += 8) {
        const float v = Pg[(int64_t)a * ld + b];
        mx = b == a ? mx : fmaxf(mx, v);
        mn = b == a ? mn : fminf(mn, v);
      }
    for (int off = 4; off > 0; off >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      mn = fminf(mn, __shfl_xor(mn, off, 64));
    }
    if (a < R && l == 0) {
      pmax[a] = R > 1 ? mx : 0.0f;
      pmin[a] = R > 1 ? mn : 0.0f;
    }
  }
  // the S-free form of the recurrence (mpm_fused_kernel) holds for a 0/1 adjacency and
  // non-negative P only: say whether this graph qualifies (the tail hand-off asks)
  int weighted = 0;
  for (int e = tid; e < R2; e += nt) {
    const int i = e / R, j = e - i * R;
    const float a = Ag[(int64_t)i * ld + j];
    if (a_in_lds) As[e] = a;
    weighted |= (i != j) && (a != 0.0f) && (a != 1.0f);
    weighted |= (i != j) && !(Pg[(int64_t)i * ld + j] >= 0.0f);
  }
  const int binary_ok = !__syncthreads_or(weighted);
  for (int e = tid; e < R2; e += nt) {
    const int i = e / R, a = e - i * R;
    const float *arow = a_in_lds ? As + i * R : Ag + (int64_t)i * ld;
    const float pa = pmax[a], pn = pmin[a];
    const float et = column_sum(
        [&](int jj) {
          if (jj == i) return 0.0f;
          const float aij = arow[jj];
          const float m = fmaxf(0.0f, aij >= 0.0f ? aij * pa : aij * pn);
          return m * x0;
        },
        a, R);
    Xn[e] = x0 * NSg[(int64_t)i * ld + a] + et;
  }
  __syncthreads();
  const int n8 = R2 & ~7;
  if (tid < 64) {
    float acc = 0.f;
    if (tid < 8) acc = norm_chain(Xn, n8, tid, acc);
    const float nrm = norm_finish(acc, Xn + n8, R2 - n8);
    if (tid == 0) s_nrm = nrm;
  }
  __syncthreads();
  const float nrm = s_nrm;
  float dmax = 0.f;
  bool isnan_any = false;
  for (int e = tid; e < R2; e += nt) {
    const float v = Xn[e] / nrm;
    const int r = e / R, c = e - r * R;
    const float d = fabsf(v - x0);
    isnan_any |= (d != d);
    dmax = fmaxf(dmax, d);
    Xg[(int64_t)r * ldx + c] = v;
  }
  if (isnan_any) dmax = INFINITY;  // a NaN difference never satisfies `< tol`
  for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = dmax;
  __syncthreads();
  if (tid == 0) {
    float d = red[0];
    for (int w = 1; w < (nt >> 6); ++w) d = fmaxf(d, red[w]);
    iters_done[g] = 1;
    stop[g] = ((early_stop && d < tol) ? 1 : 0) | (binary_ok ? 2 : 0);
  }
}

// ---------------------------------------------------------------------------
// Similarity tensor AND the MPM iteration that would read it first.  The work-group that
// produces the slab S[i,:,:,:] holds every element in a register on its way to memory, so it
// also takes row i of the iteration on the X that mpm_first_iter_kernel left:
//   T[j,a] = max_b S[i,j,a,b]*X[j,b],   Xn[i,a] = X[i,a]*NS[i,a] + canonical column sum of T
// (the arithmetic of one item of the streaming kernel: the maximum is exact in any order, the
// sum and the update are the same expressions), and the tensor is read once less.  For graphs
// that have not converged and have more than one iteration to run; the others get the plain
// producer.  The kernel is bound by its stores and needs its wavefronts, so LDS is kept small:
// up to kSimIterMaxNodes nodes X and T of the graph are resident (33 KiB at 64 nodes); above, T
// covers the few columns `a` the work-group has in flight and X is staged 16 rows at a time
// (13 KiB at 128 nodes).  mpm_second_norm_kernel finishes the iteration per graph.
// ---------------------------------------------------------------------------
constexpr int kSimIterMaxNodes = 64;
constexpr int kSimIterRowBlock = 16;
// Between the two the plain producer stays: on ragged batches (ENZYMES-shaped, up to 82 nodes)
// the launch lasts as long as its few largest slabs, which the second form writes 15-25 % slower
// (3.3 against 3.9 TB/s at 82 nodes), more than the pass over their tensors it saves (producer
// 0.14 -> 0.22 ms, streaming launch 0.95 -> 0.93 ms); dense batches of 96-128 nodes gain 6 %.
constexpr int kSimIterBigMinNodes = 89;

__device__ __forceinline__ bool sim_iter_applies(int R, int stop_word, int iters) {
  return (R <= kSimIterMaxNodes || R >= kSimIterBigMinNodes) && (stop_word & 1) == 0 && iters >= 2;
}
// floats of dynamic LDS a work-group of edge_sim_iter_kernel needs for a graph of R nodes whose
// rows of S hold RS floats (vectors of 4, 2 or 1 floats: see the kernel's dispatch)
__host__ __device__ inline size_t sim_iter_lds_floats(int R, int RS) {
  if (R <= kSimIterMaxNodes) return (size_t)R * RS + (size_t)R * R;
  const int V = (RS & 3) == 0 ? 4 : (RS & 1) == 0 ? 2 : 1;
  const int cols = RS / V;                                // lanes per row, < 256 at these sizes
  return (size_t)kSimIterRowBlock * RS + (size_t)R * (256 / cols);
}

template <int V>
__device__ __forceinline__ void edge_sim_iter_body(const float *__restrict__ Pg,
                                                   const float *__restrict__ NSg, int ld, int R, int RS,
                                                   int i, float *__restrict__ slab, const float *arow,
                                                   const float *Xs, float *T) {
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)R * RS;
  const int cols = RS / V;                      // <= 16 lanes per row at these sizes
  const int rows_per_pass = 256 / cols;
  const int rslot = tid / cols, c = tid - rslot * cols;
  // lanes of a row are adjacent: a power-of-two group folds its maxima with DPP moves, any other
  // width through an LDS maximum
  const bool pow2 = (cols & (cols - 1)) == 0;
  int lg = 0;
  while ((1 << lg) < cols) ++lg;
  if (rslot >= rows_per_pass) return;
  for (int a = rslot; a < R; a += rows_per_pass) {
    float p[V], d[V];
    const float nsv = NSg[(int64_t)i * ld + a];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int b = min(c * V + v, R - 1);      // padding repeats the row's last element
      p[v] = (b == a) ? 0.0f : Pg[(int64_t)a * ld + b];
      d[v] = (b == a) ? nsv : 0.0f;
    }
    float *dst = slab + (int64_t)a * RS + c * V;
    const float *xp = Xs + c * V;
    for (int j = 0; j < R; ++j, dst += plane, xp += RS) {
      const float aij = arow[j];
      float o[V];
#pragma unroll
      for (int v = 0; v < V; ++v) o[v] = (j == i) ? d[v] : aij * p[v];
      if constexpr (V == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      } else if constexpr (V == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(o[0], o[1]);
      } else {
        dst[0] = o[0];
      }
      float t = o[0] * xp[0];
#pragma unroll
      for (int v = 1; v < V; ++v) t = fmaxf(t, o[v] * xp[v]);
      if (pow2) {
        t = group_max(t, lg);
        if (c == 0) T[j * R + a] = t;
      } else {
        __hip_atomic_fetch_max(&T[j * R + a], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

// The same for a graph whose X and T do not fit beside many other work-groups: the columns `a`
// are taken rows_per_pass at a time (T holds those columns only), the rows of X in blocks of
// kSimIterRowBlock.  Every thread of the work-group runs every barrier.
template <int V>
__device__ __forceinline__ void edge_sim_iter_big_body(
    const float *__restrict__ Pg, const float *__restrict__ NSg, int ld, int R, int RS, int i,
    float *__restrict__ slab, const float *arow, const float *__restrict__ Xg, int ldx,
    float *__restrict__ Xn_row, float *Xb, float *T) {
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)R * RS;
  const int cols = RS / V;
  const int rpp = 256 / cols;                   // columns a in flight
  const int rslot = tid / cols, c = tid - rslot * cols;
  const bool lane_on = rslot < rpp;
  const bool pow2 = (cols & (cols - 1)) == 0;
  int lg = 0;
  while ((1 << lg) < cols) ++lg;
  for (int a0 = 0; a0 < R; a0 += rpp) {
    const int a = a0 + rslot;
    const bool act = lane_on && a < R;
    float p[V], d[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { p[v] = 0.0f; d[v] = 0.0f; }
    if (act) {
      const float nsv = NSg[(int64_t)i * ld + a];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int b = min(c * V + v, R - 1);
        p[v] = (b == a) ? 0.0f : Pg[(int64_t)a * ld + b];
        d[v] = (b == a) ? nsv : 0.0f;
      }
    }
    for (int e = tid; e < R * rpp; e += 256) T[e] = -INFINITY;
    float *dst = slab + (int64_t)a * RS + c * V;
    for (int j0 = 0; j0 < R; j0 += kSimIterRowBlock) {
      const int jn = min(kSimIterRowBlock, R - j0);
      __syncthreads();                                    // the previous block of X is used up
      for (int e = tid; e < jn * RS; e += 256) {
        const int jj = e / RS, cc = e - jj * RS;
        Xb[e] = Xg[(int64_t)(j0 + jj) * ldx + min(cc, R - 1)];
      }
      __syncthreads();
      if (act) {
        for (int jj = 0; jj < jn; ++jj) {
          const int j = j0 + jj;
          const float aij = arow[j];
          float o[V];
#pragma unroll
          for (int v = 0; v < V; ++v) o[v] = (j == i) ? d[v] : aij * p[v];
          float *dj = dst + (int64_t)j * plane;
          if constexpr (V == 4) {
            *reinterpret_cast<float4 *>(dj) = make_float4(o[0], o[1], o[2], o[3]);
          } else if constexpr (V == 2) {
            *reinterpret_cast<float2 *>(dj) = make_float2(o[0], o[1]);
          } else {
            dj[0] = o[0];
          }
          const float *xp = Xb + jj * RS + c * V;
          float t = o[0] * xp[0];
#pragma unroll
          for (int v = 1; v < V; ++v) t = fmaxf(t, o[v] * xp[v]);
          if (pow2) {
            t = group_max(t, lg);
            if (c == 0) T[j * rpp + rslot] = t;
          } else {
            __hip_atomic_fetch_max(&T[j * rpp + rslot], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    __syncthreads();
    if (tid < rpp && a0 + tid < R) {
      const int pcol = a0 + tid;
      // the row j == i does not take part in the sum: an exact +0 at its position
      const float et = column_sum([&](int jj) { return jj == i ? 0.0f : T[jj * rpp + tid]; }, pcol, R);
      Xn_row[pcol] = Xg[(int64_t)i * ldx + pcol] * NSg[(int64_t)i * ld + pcol] + et;
    }
    __syncthreads();                                      // T is reset for the next columns
  }
}

__global__ __launch_bounds__(256) void edge_sim_iter_kernel(
    const float *__restrict__ A, const float *__restrict__ P, const float *__restrict__ NS,
    const int32_t *__restrict__ Rarr, int ld, int64_t stride, int layout, float *__restrict__ S,
    const int64_t *__restrict__ S_off, const float *__restrict__ X, int ldx, int64_t strideX,
    float *__restrict__ Xnew, int64_t strideXnew, const int32_t *__restrict__ stop, int iters) {
  __shared__ float arow[1024];
  extern __shared__ __attribute__((aligned(16))) float sim_lds[];
  const int g = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  const int R = Rarr[g];
  if (i >= R) return;
  const int RS = s_row_stride(R, layout);
  const float *Ag = A + g * stride, *Pg = P + g * stride, *NSg = NS + g * stride;
  float *slab = S + S_off[g] + (int64_t)i * R * R * RS;
  for (int j = tid; j < R; j += 256) arow[j] = (j == i) ? 1.0f : Ag[(int64_t)i * ld + j];
  if (!sim_iter_applies(R, stop[g], iters)) {
    __syncthreads();
    if ((RS & 3) == 0) edge_sim_body<4>(Pg, NSg, ld, R, RS, i, slab, arow);
    else if ((RS & 1) == 0) edge_sim_body<2>(Pg, NSg, ld, R, RS, i, slab, arow);
    else edge_sim_body<1>(Pg, NSg, ld, R, RS, i, slab, arow);
    return;
  }
  const float *Xg = X + g * strideX;
  if (R > kSimIterMaxNodes) {
    float *Xb = sim_lds;                               // kSimIterRowBlock rows of RS floats
    float *Tb = sim_lds + kSimIterRowBlock * RS;       // R rows of (columns in flight) floats
    float *Xn_row = Xnew + g * strideXnew + (int64_t)i * R;
    __syncthreads();
    if ((RS & 3) == 0) edge_sim_iter_big_body<4>(Pg, NSg, ld, R, RS, i, slab, arow, Xg, ldx, Xn_row, Xb, Tb);
    else if ((RS & 1) == 0) edge_sim_iter_big_body<2>(Pg, NSg, ld, R, RS, i, slab, arow, Xg, ldx, Xn_row, Xb, Tb);
    else edge_sim_iter_big_body<1>(Pg, NSg, ld, R, RS, i, slab, arow, Xg, ldx, Xn_row, Xb, Tb);
    return;
  }
  float *Xs = sim_lds;              // R rows of RS floats, the padding repeats the last element
  float *T = sim_lds + R * RS;      // R*R
  for (int e = tid; e < R * RS; e += 256) {
    const int r = e / RS, cc = e - r * RS;
    Xs[e] = Xg[(int64_t)r * ldx + min(cc, R - 1)];
  }
  for (int e = tid; e < R * R; e += 256) T[e] = -INFINITY;
  __syncthreads();
  if ((RS & 3) == 0) edge_sim_iter_body<4>(Pg, NSg, ld, R, RS, i, slab, arow, Xs, T);
  else if ((RS & 1) == 0) edge_sim_iter_body<2>(Pg, NSg, ld, R, RS, i, slab, arow, Xs, T);
  else edge_sim_iter_body<1>(Pg, NSg, ld, R, RS, i, slab, arow, Xs, T);
  __syncthreads();
  float *Xn = Xnew + g * strideXnew;
  for (int pcol = tid; pcol < R; pcol += 256) {
    // the row j == i does not take part in the sum: an exact +0 at its position
    const float et = column_sum([&](int jj) { return jj == i ? 0.0f : T[jj * R + pcol]; }, pcol, R);
    Xn[i * R + pcol] = Xs[i * RS + pcol] * NSg[(int64_t)i * ld + pcol] + et;
  }
}

// Second half of the iteration edge_sim_iter_kernel started: canonical norm, X = Xn / norm,
// convergence test, as the last arriver of a graph does in the streaming kernel.  One
// work-group per graph; graphs the producer did not iterate are left alone.
__global__ __launch_bounds__(256) void mpm_second_norm_kernel(
    const int32_t *__restrict__ Rarr, int r_max, float *__restrict__ X, int ldx, int64_t strideX,
    const float *__restrict__ Xnew, int64_t strideXnew, float tol, int early_stop, int iters,
    int32_t *__restrict__ iters_done, int32_t *__restrict__ stop) {
  extern __shared__ __attribute__((aligned(16))) float nrm_lds[];
  __shared__ float red[4];
  __shared__ float s_nrm;
  const int g = blockIdx.x, tid = threadIdx.x, nt = 256;
  const int R = Rarr[g];
  if (R <= 0 || R > r_max || !sim_iter_applies(R, stop[g], iters)) return;
  const int R2 = R * R;
  float *T = nrm_lds;
  float *Xg = X + g * strideX;
  const float *Xn = Xnew + g * strideXnew;
  for (int e = tid; e < R2; e += nt) T[e] = Xn[e];
  __syncthreads();
  const int n8 = R2 & ~7;
  if (tid < 64) {
    float acc = 0.f;
    if (tid < 8) acc = norm_chain(T, n8, tid, acc);
    const float nrm = norm_finish(acc, T + n8, R2 - n8);
    if (tid == 0) s_nrm = nrm;
  }
  __syncthreads();
  const float nrm = s_nrm;
  float dmax = 0.f;
  bool isnan_any = false;
  for (int e = tid; e < R2; e += nt) {
    const float v = T[e] / nrm;
    const int r = e / R, c = e - r * R;
    const float d = fabsf(v - Xg[(int64_t)r * ldx + c]);
    isnan_any |= (d != d);
    dmax = fmaxf(dmax, d);
    Xg[(int64_t)r * ldx + c] = v;
  }
  if (isnan_any) dmax = INFINITY;  // a NaN difference never satisfies `< tol`
  dmax = wave_max(dmax);
  if ((tid & 63) == 0) red[tid >> 6] = dmax;
  __syncthreads();
  if (tid == 0) {
    const float d = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    iters_done[g] = 2;
    if (early_stop && d < tol) stop[g] |= 1;
  }
}

// ---------------------------------------------------------------------------
// Fused S-free MPM: one work-group per graph, Pt (P transposed, zero diagonal),
// X and M resident in LDS, A as a bit matrix.  Per iteration:
//   M[j,a]  = max_b Pt[b,a]*X[j,b]                      (>= 0; Pt[a,a] = 0)
//   Xn[i,a] = X[i,a]*NS[i,a] + sum_{j != i, A[i,j] != 0} M[j,a]   (canonical column sum,
//             absent terms contribute an exact +0 at their position)
//   X = Xn / ||Xn||_2 ; stop when max|X - Xprev| < tol
// The previous X lives in the caller's d_X buffer (global), the new one in LDS.
// ---------------------------------------------------------------------------


// Canonical column sums of NCH rows that share one column a: chain c adds, for j = 0..R-1 in the
// canonical order, M[j,a] if j is a neighbour of row c and an exact +0 otherwise.  The M values
// are loaded once per j (not once per chain) and the chains are independent, so the LDS latency
// and the add latency overlap.  Columns that the canonical order sums as four interleaved strands
// (ilp4_sum): strand s takes the terms j = 4t + s of the first 4*floor(R/4) rows, cascading after
// every 16 of ITS terms, i.e. after every 64 consecutive j; the remaining rows go to strand 0
// after the strands' levels are folded; then strand 0 receives strands 1, 2, 3.  Walking j
// upwards visits the strands round-robin, so one pass over the column feeds all 4*NCH
// accumulators.  Valid for R < 1024 (a strand never reaches its second cascade level).
//
// The neighbours come as byte masks (one byte per (i, j): 1 when j is a neighbour of i): a term is
// fma(M[j,a], mask, acc) -- M * 1 + acc rounds once, exactly as the sum does, and M * 0 + acc
// leaves acc as it is -- so a pair of chains advances with one packed fma; the sixteen masks of
// a chain and a block come with one 16-byte read.  mrow[c]: byte row of chain c, 16-byte aligned.
__device__ __forceinline__ float mask_byte(const uint32_t (&w)[2], int t) {
  return (float)((w[t >> 2] >> (8 * (t & 3))) & 0xFFu);      // v_cvt_f32_ubyteN
}

template <int NCH>
__device__ __forceinline__ void cascade_chains_b(const float *__restrict__ Mcol, int R,
                                                 const uint8_t *const (&mrow)[NCH],
                                                 float (&out)[NCH]) {
  static_assert(NCH % 2 == 0, "chains advance in pairs");
  constexpr int NP = NCH / 2;
  f32x2_t l0[NP], l1[NP], l2[NP], l3[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) l0[c] = l1[c] = l2[c] = l3[c] = f32x2_t{0.f, 0.f};
  int i = 0;
  while (i + 16 <= R) {
#pragma unroll
    for (int h = 0; h < 16; h += 8) {      // eight terms at a time keeps the live registers down
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = Mcol[(i + h + t) * R];
      uint32_t mk[NCH][2];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x2_t q = *reinterpret_cast<const u32x2_t *>(mrow[c] + i + h);
        mk[c][0] = q.x, mk[c][1] = q.y;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < NP; ++c)
          l0[c] = __builtin_elementwise_fma(f32x2_t{v[t], v[t]},
                                            f32x2_t{mask_byte(mk[2 * c], t), mask_byte(mk[2 * c + 1], t)},
                                            l0[c]);
    }
    i += 16;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      l1[c] += l0[c];
      l0[c] = f32x2_t{0.f, 0.f};
    }
    if ((i & 0xF0) == 0) {
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        l2[c] += l1[c];
        l1[c] = f32x2_t{0.f, 0.f};
      }
      if ((i & 0xF00) == 0) {
#pragma unroll
        for (int c = 0; c < NP; ++c) {
          l3[c] += l2[c];
          l2[c] = f32x2_t{0.f, 0.f};
        }
      }
    }
  }
  for (; i < R; ++i) {
    const float v = Mcol[i * R];
#pragma unroll
    for (int c = 0; c < NP; ++c)
      l0[c] = __builtin_elementwise_fma(f32x2_t{v, v},
                                        f32x2_t{(float)mrow[2 * c][i], (float)mrow[2 * c + 1][i]}, l0[c]);
  }
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    l0[c] += l1[c];
    l0[c] += l2[c];
    l0[c] += l3[c];
    out[2 * c] = l0[c].x;
    out[2 * c + 1] = l0[c].y;
  }
}

template <int NCH>
__device__ __forceinline__ void ilp4_chains_b(const float *__restrict__ Mcol, int R,
                                              const uint8_t *const (&mrow)[NCH],
                                              float (&out)[NCH]) {
  static_assert(NCH % 2 == 0, "chains advance in pairs");
  constexpr int NP = NCH / 2;
  f32x2_t l0[NP][4], l1[NP][4];
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) l0[c][q] = l1[c][q] = f32x2_t{0.f, 0.f};
  const int n44 = (R >> 2) << 2;
  int i = 0;
  for (; i + 16 <= n44; i += 16) {
#pragma unroll
    for (int h = 0; h < 16; h += 8) {
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = Mcol[(i + h + t) * R];
      uint32_t mk[NCH][2];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x2_t q = *reinterpret_cast<const u32x2_t *>(mrow[c] + i + h);
        mk[c][0] = q.x, mk[c][1] = q.y;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int c = 0; c < NP; ++c)
          l0[c][t & 3] = __builtin_elementwise_fma(
              f32x2_t{v[t], v[t]}, f32x2_t{mask_byte(mk[2 * c], t), mask_byte(mk[2 * c + 1], t)},
              l0[c][t & 3]);
    }
    if (((i + 16) & 63) == 0) {
#pragma unroll
      for (int c = 0; c < NP; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          l1[c][q] += l0[c][q];
          l0[c][q] = f32x2_t{0.f, 0.f};
        }
    }
  }
  for (; i < n44; i += 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float v = Mcol[(i + t) * R];
#pragma unroll
      for (int c = 0; c < NP; ++c)
        l0[c][t] = __builtin_elementwise_fma(
            f32x2_t{v, v}, f32x2_t{(float)mrow[2 * c][i + t], (float)mrow[2 * c + 1][i + t]}, l0[c][t]);
    }
  }
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) l0[c][q] += l1[c][q];   // the higher levels are +0: adding them changes nothing
  for (; i < R; ++i) {
    const float v = Mcol[i * R];
#pragma unroll
    for (int c = 0; c < NP; ++c)
      l0[c][0] = __builtin_elementwise_fma(
          f32x2_t{v, v}, f32x2_t{(float)mrow[2 * c][i], (float)mrow[2 * c + 1][i]}, l0[c][0]);
  }
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    l0[c][0] += l0[c][1];
    l0[c][0] += l0[c][2];
    l0[c][0] += l0[c][3];
    out[2 * c] = l0[c][0].x;
    out[2 * c + 1] = l0[c][0].y;
  }
}

// One lane's chain of the canonical norm over its cnt elements, stored contiguously (16-byte
// aligned).  The chain is serial by definition (one dependent fma per element); the next sixteen
// operands are read while the current sixteen are consumed.
__device__ __forceinline__ float norm_chain_rows(const float *__restrict__ x, int cnt) {
  float acc = 0.f;
  const int nb = cnt >> 4;
  f32x4_t A[4], B[4];
  auto load = [&](f32x4_t (&d)[4], int blk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f32x4_t *>(x + 16 * blk + 4 * q);
  };
  auto eat = [&](const f32x4_t (&d)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __fmaf_rn(d[q][e], d[q][e], acc);
  };
  if (nb > 0) load(A, 0);
  int blk = 0;
  while (blk + 2 <= nb) {
    load(B, blk + 1);
    eat(A);
    if (blk + 2 < nb) load(A, blk + 2);
    eat(B);
    blk += 2;
  }
  if (blk < nb) eat(A);
  for (int k = nb << 4; k < cnt; ++k) acc = __fmaf_rn(x[k], x[k], acc);
  return acc;
}

// LDS layout of mpm_fused_kernel, in floats after the control words.  Rows of P and X are padded
// with zeros to a multiple of four floats, so that every row is read in 16-byte pieces whatever R
// is (a zero product never changes a maximum that starts at 0); the row stride of P is 4 modulo
// 8, which spreads the 16-byte pieces of consecutive rows over all banks.
struct FusedLds {
  int PS, XS, TS, MB;     // row strides of P and X; lane stride of the norm layout; mask row bytes
  int off_X, off_M, off_mask, total;
};
__host__ __device__ inline FusedLds fused_lds_layout(int R) {
  FusedLds L;
  const int R4 = (R + 3) & ~3;
  L.PS = R4 + ((R4 & 7) == 4 ? 0 : 4);
  L.XS = R4;
  const int cnt = (R * R) >> 3;                           // elements per lane of the norm
  L.TS = ((cnt + 27) & ~31) + 4;                          // >= cnt, 4 modulo 32
  L.MB = (R + 15) & ~15;
  L.off_X = R * L.PS;
  L.off_M = L.off_X + R * L.XS;
  const int m = R * R > 8 * L.TS ? ((R * R + 3) & ~3) : 8 * L.TS;
  L.off_mask = L.off_M + m;
  L.total = L.off_mask + R * L.MB / 4;
  return L;
}
constexpr int kFusedCtl = 48;   // floats: [0,32) wavefront maxima (two sets), [32] norm, [36,44) norm tail

// NT = work-group size, KMAX = rows per lane (ceil(R / floor(NT / R)) <= KMAX <= 8 must hold).
// The previous X stays in registers; so do the node-similarity factor and the unnormalised new
// X up to KMAX = 4 (above, the factor is read from global memory at its use, where it stays in L2).
// Graphs of more than 88 nodes: mpm_fused_big_kernel.
template <int NT, int KMAX>
__global__ __launch_bounds__(NT, 4) void mpm_fused_kernel(
    const float *__restrict__ A, const float *__restrict__ P, const float *__restrict__ NS,
    const int32_t *__restrict__ Rarr, int ld, int64_t stride, float *__restrict__ X, int ldx,
    int64_t strideX, int iters, float tol, int early_stop, int32_t *__restrict__ iters_done,
    int64_t *__restrict__ perm, int n_perm, int32_t *__restrict__ status,
    const int32_t *__restrict__ resume, int r_lo, int r_hi) {
  // a launch serves the graphs of one size class (r_lo <= R <= r_hi): the work-groups of the
  // others leave at once.  resume != nullptr: only graphs with resume[g] != 0 run, continuing
  // from the X in the caller's buffer after the iters_done[g] iterations another kernel has
  // already executed
  {
    const int Rg = Rarr[blockIdx.x];
    if (Rg < r_lo || Rg > r_hi || (resume != nullptr && resume[blockIdx.x] == 0)) return;
  }
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float *red = lds_all;                                  // [2][16]
  float &s_nrm = lds_all[32];
  float *Ttail = lds_all + 36;                           // [8]
  float *lds = lds_all + kFusedCtl;
  const int g = blockIdx.x, tid = threadIdx.x;
  constexpr int nt = NT;
  static_assert(KMAX <= 8, "larger graphs: mpm_fused_big_kernel");
  constexpr int NCH = KMAX < 4 ? KMAX : 4;          // chains (rows) a lane advances together
  constexpr bool kXnRegs = KMAX <= 4;               // the unnormalised new X stays in registers
  constexpr int kChainUnroll = kXnRegs ? KMAX / NCH : 1;
  const int R = Rarr[g];
  if (R <= 0) {
    if (tid == 0) iters_done[g] = 0;
    if (perm) {
      for (int k = tid; k < n_perm; k += nt) perm[(int64_t)g * n_perm + k] = k;
      if (tid == 0) status[g] = 0;
    }
    return;
  }
  const FusedLds L = fused_lds_layout(R);
  const int R2 = R * R, PS = L.PS, XS = L.XS, TS = L.TS, MB = L.MB;
  float *Pp = lds;                 // Pp[a*PS + b] = P[a,b], 0 for a == b and for b >= R
  float *Xs = lds + L.off_X;       // Xs[j*XS + b] = X[j,b], 0 for b >= R
  float *M = lds + L.off_M;        // M[j*R + a]; after the column sums the same words hold T
  float *T = M;                    // T[l*TS + k] = Xn[8k + l] (flat element order), l = 0..7
  uint8_t *Ab = reinterpret_cast<uint8_t *>(lds + L.off_mask);    // byte masks, rows of MB bytes
  const float *Ag = A + g * stride, *Pg = P + g * stride, *NSg = NS + g * stride;
  float *Xg = X + g * strideX;

  for (int e = tid; e < R * PS; e += nt) {
    const int r = e / PS, c = e - r * PS;
    Pp[e] = (c < R && c != r) ? Pg[(int64_t)r * ld + c] : 0.0f;
  }
  for (int e = tid; e < R * XS; e += nt) {
    const int r = e / XS, c = e - r * XS;
    Xs[e] = (c < R) ? Xg[(int64_t)r * ldx + c] : 0.0f;
  }
  for (int e = tid; e < R * MB; e += nt) {
    const int r = e / MB, c = e - r * MB;
    Ab[e] = (c < R && c != r && Ag[(int64_t)r * ld + c] != 0.0f) ? 1 : 0;
  }
  if (tid < 32) red[tid] = 0.f;
  // element ownership for the column sums and the update: lane (ta, tj) owns column a = ta and
  // rows i = tj + k*TJ.  Columns 0..C-1 sum in cascade order, the others as four strands; the
  // lanes of the first kind come first and the second kind starts at a wavefront boundary when
  // the work-group has the lanes for it, so that no wavefront (or one) runs both code paths.
  const int TJ = nt / R;
  const int C = (R >= 8) ? (R & ~31) : (R & ~3);
  const int nA = C * TJ, nB = (R - C) * TJ;
  const int startB = (((nA + 63) & ~63) + nB <= nt) ? ((nA + 63) & ~63) : nA;
  bool owner = false;
  int ta = 0, tj = 0;
  if (tid < nA) {
    owner = true;
    tj = tid / C;
    ta = tid - tj * C;
  } else if (tid >= startB && tid < startB + nB) {
    owner = true;
    const int u = tid - startB;
    tj = u / (R - C);
    ta = C + (u - tj * (R - C));
  }
  // registers per lane: previous X; the unnormalised new X and the node-similarity factor
  // (KMAX <= 4; otherwise the factor is read from global memory at its use, where it stays in L2)
  constexpr bool kNsRegs = KMAX <= 4;
  constexpr int KN = kNsRegs ? KMAX : 1, KX = kXnRegs ? KMAX : 1;
  float nsv[KN], xprev[KMAX], xn[KX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = tj + k * TJ;
    if constexpr (kNsRegs) nsv[k] = (owner && i < R) ? NSg[(int64_t)i * ld + ta] : 0.f;
    xprev[k] = (owner && i < R) ? Xg[(int64_t)i * ldx + ta] : 0.f;
  }
  auto ns_of = [&](int k, int i) { return kNsRegs ? nsv[kNsRegs ? k : 0] : NSg[(int64_t)i * ld + ta]; };
  __syncthreads();

  const int n8 = R2 & ~7;
  const int JQ = (R + 3) >> 2;          // groups of 4 consecutive rows j
  const int R4 = XS;
  // phase-1 items (4 rows x 1 column); while the work-group has at least twice as many lanes,
  // the b range of an item is cut over 2 or 4 adjacent lanes
  int split_log2 = 0;
  while (split_log2 < 2 && ((R * JQ) << (split_log2 + 1)) <= nt) ++split_log2;
  int done = resume != nullptr ? iters_done[g] : 0;
  int par = 0;
  for (int it = done; it < iters; ++it) {
    // phase 1: M[j,a] = max_b Pp[a,b] * X[j,b]  (>= 0; Pp[a,a] = 0).  A lane takes one column a
    // and four consecutive rows j at a time: per 4 values of b it reads one 16-byte piece of
    // Pp's row a and four (wave-broadcast) pieces of X; products in pairs (packed multiply).
    auto phase1 = [&](auto sl_tag) {
      constexpr int SL = decltype(sl_tag)::value;
      const int part = tid & ((1 << SL) - 1);
      for (int w = tid >> SL; w < R * JQ; w += nt >> SL) {
        const int jq = w / R, a = w - jq * R;
        const int j0 = jq * 4;
        const float *pr = Pp + a * PS;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
        const float *x0 = Xs + j0 * XS;
        const float *x1 = Xs + min(j0 + 1, R - 1) * XS;
        const float *x2 = Xs + min(j0 + 2, R - 1) * XS;
        const float *x3 = Xs + min(j0 + 3, R - 1) * XS;
        for (int b = 4 * part; b < R4; b += 4 << SL) {
          const f32x4_t p = *reinterpret_cast<const f32x4_t *>(pr + b);
          const f32x4_t a0 = *reinterpret_cast<const f32x4_t *>(x0 + b);
          const f32x4_t a1 = *reinterpret_cast<const f32x4_t *>(x1 + b);
          const f32x4_t a2 = *reinterpret_cast<const f32x4_t *>(x2 + b);
          const f32x4_t a3 = *reinterpret_cast<const f32x4_t *>(x3 + b);
          f32x2_t q;
          q = p.lo * a0.lo, m0 = fmaxf(fmaxf(m0, q.x), q.y);
          q = p.hi * a0.hi, m0 = fmaxf(fmaxf(m0, q.x), q.y);
          q = p.lo * a1.lo, m1 = fmaxf(fmaxf(m1, q.x), q.y);
          q = p.hi * a1.hi, m1 = fmaxf(fmaxf(m1, q.x), q.y);
          q = p.lo * a2.lo, m2 = fmaxf(fmaxf(m2, q.x), q.y);
          q = p.hi * a2.hi, m2 = fmaxf(fmaxf(m2, q.x), q.y);
          q = p.lo * a3.lo, m3 = fmaxf(fmaxf(m3, q.x), q.y);
          q = p.hi * a3.hi, m3 = fmaxf(fmaxf(m3, q.x), q.y);
        }
        // the lanes of an item are adjacent: join their partial maxima (exact in any order)
        m0 = group_max(m0, SL), m1 = group_max(m1, SL), m2 = group_max(m2, SL), m3 = group_max(m3, SL);
        if (part == 0) {
          M[j0 * R + a] = m0;
          if (j0 + 1 < R) M[(j0 + 1) * R + a] = m1;
          if (j0 + 2 < R) M[(j0 + 2) * R + a] = m2;
          if (j0 + 3 < R) M[(j0 + 3) * R + a] = m3;
        }
      }
    };
    if (split_log2 == 0)
      phase1(std::integral_constant<int, 0>{});
    else if (split_log2 == 1)
      phase1(std::integral_constant<int, 1>{});
    else
      phase1(std::integral_constant<int, 2>{});
    __syncthreads();
    // phase 2: Xn[i,a] = Xprev[i,a]*NS[i,a] + sum_{j: A[i,j]} M[j,a]  (canonical column sums),
    // kept in registers until every lane has read M, then stored in the norm's layout over M
    // (KMAX > 4: in X's own words instead, each rewritten by its owner only)
#pragma unroll
    for (int k = 0; k < KX; ++k) xn[k] = 0.f;
    if (owner) {
      const float *Mcol = M + ta;
      const bool casc = ta < C;
#pragma unroll kChainUnroll
      for (int k0 = 0; k0 < KMAX; k0 += NCH) {
        if (tj + k0 * TJ < R) {
          float et[NCH];
          const uint8_t *br[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) br[c] = Ab + min(tj + (k0 + c) * TJ, R - 1) * MB;
          if (casc)
            cascade_chains_b<NCH>(Mcol, R, br, et);
          else
            ilp4_chains_b<NCH>(Mcol, R, br, et);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int i = tj + (k0 + c) * TJ;
            if (i < R) {
              const float val = Xs[i * XS + ta] * ns_of(k0 + c, i) + et[c];
              if constexpr (kXnRegs)
                xn[k0 + c] = val;
              else
                Xs[i * XS + ta] = val;
            }
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int i = tj + k * TJ;
      if (owner && i < R) {
        const int e = i * R + ta;
        const float val = kXnRegs ? xn[kXnRegs ? k : 0] : Xs[i * XS + ta];
        if (e < n8)
          T[(e & 7) * TS + (e >> 3)] = val;
        else
          Ttail[e - n8] = val;
      }
    }
    __syncthreads();
    // phase 3: canonical norm over the R*R elements in flat order
    if (tid < 64) {
      float acc = 0.f;
      if (tid < 8) acc = norm_chain_rows(T + tid * TS, n8 >> 3);
      const float nrm = norm_finish(acc, Ttail, R2 - n8);
      if (tid == 0) s_nrm = nrm;
    }
    __syncthreads();
    const float nrm = s_nrm;
    float dmax = 0.f;
    bool isnan_any = false;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int i = tj + k * TJ;
      if (owner && i < R) {
        const float v = (kXnRegs ? xn[kXnRegs ? k : 0] : Xs[i * XS + ta]) / nrm;
        const float prev = xprev[k];
        xprev[k] = v;
        const float d = fabsf(v - prev);
        isnan_any |= (d != d);
        dmax = fmaxf(dmax, d);
        Xs[i * XS + ta] = v;
      }
    }
    if (isnan_any) dmax = INFINITY;
    dmax = wave_max(dmax);
    if ((tid & 63) == 0) red[par * 16 + (tid >> 6)] = dmax;
    __syncthreads();
    // every lane folds the wavefronts' maxima itself (the set alternates between iterations, so
    // a lane still reading this one cannot meet the next one's writes)
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < (NT >> 6); q += 4) {
      const f32x4_t r4 = *reinterpret_cast<const f32x4_t *>(red + par * 16 + q);
      d = fmaxf(fmaxf(d, fmaxf(r4.x, r4.y)), fmaxf(r4.z, r4.w));
    }
    par ^= 1;
    ++done;
    if (early_stop && d < tol) break;
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = tj + k * TJ;
    if (owner && i < R) Xg[(int64_t)i * ldx + ta] = xprev[k];
  }
  if (tid == 0) iters_done[g] = done;
  // Epilogue (perm != nullptr): the assignment of this graph, by the first wavefront, straight
  // from the X in LDS.  A graph that converges early starts its assignment early; as a separate
  // launch the assignments start only after the slowest graph's last MPM iteration.
  if (perm != nullptr) {
    __syncthreads();               // the last iteration's Xs (iters == 0: the initial X)
    if (tid < 64) lsap_wave(Xs, R, XS, tid, perm + (int64_t)g * n_perm, n_perm, status + g);
  }
}

// LDS layout of mpm_fused_big_kernel, in floats after the control words: P as in FusedLds, ONE
// region that holds X, then the row maxima M, then the new X in the norm's layout, and byte masks.
struct FusedBigLds {
  int PS, XS, TS, MB;
  int off_XM, off_mask, total;
};
__host__ __device__ inline FusedBigLds fused_big_lds_layout(int R) {
  FusedBigLds L;
  const int R4 = (R + 3) & ~3;
  L.PS = R4 + ((R4 & 7) == 4 ? 0 : 4);
  L.XS = R4;
  const int cnt = (R * R) >> 3;
  L.TS = ((cnt + 27) & ~31) + 4;
  L.MB = (R + 15) & ~15;
  L.off_XM = R * L.PS;
  const int xm = R * L.XS > 8 * L.TS ? R * L.XS : 8 * L.TS;
  L.off_mask = L.off_XM + ((xm + 3) & ~3);
  L.total = L.off_mask + R * L.MB / 4;
  return L;
}

// S-free MPM for launches whose largest graph has 89..128 nodes: 1024 threads, one work-group per
// graph, same recurrence and roundings as mpm_fused_kernel.  P, X and the row maxima M of a
// 128-node graph do not fit in 160 KiB side by side, so M takes X's place: a lane keeps the (up
// to sixteen) maxima it produces in registers until every lane has finished reading X, then they
// are stored over it; the new X goes through registers the same way into the norm's layout over
// M, and the normalised X back over that.  The current X, which the update also needs
// element-wise (X[i,a] * NS[i,a], and the convergence test), is read back from the caller's X
// buffer, where its owner lane stored it the iteration before (it stays in L2).
constexpr int kBigNT = 1024, kBigRows = 16;
__global__ __launch_bounds__(kBigNT) void mpm_fused_big_kernel(
    const float *__restrict__ A, const float *__restrict__ P, const float *__restrict__ NS,
    const int32_t *__restrict__ Rarr, int ld, int64_t stride, float *__restrict__ X, int ldx,
    int64_t strideX, int iters, float tol, int early_stop, int32_t *__restrict__ iters_done,
    int64_t *__restrict__ perm, int n_perm, int32_t *__restrict__ status,
    const int32_t *__restrict__ resume, int r_lo, int r_hi) {
  {
    const int Rg = Rarr[blockIdx.x];
    if (Rg < r_lo || Rg > r_hi || (resume != nullptr && resume[blockIdx.x] == 0)) return;
  }
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float *red = lds_all;                                  // [2][16]
  float &s_nrm = lds_all[32];
  float *Ttail = lds_all + 36;                           // [8]
  float *lds = lds_all + kFusedCtl;
  const int g = blockIdx.x, tid = threadIdx.x;
  constexpr int nt = kBigNT;
  const int R = Rarr[g];
  if (R <= 0) {
    if (tid == 0) iters_done[g] = 0;
    if (perm) {
      for (int k = tid; k < n_perm; k += nt) perm[(int64_t)g * n_perm + k] = k;
      if (tid == 0) status[g] = 0;
    }
    return;
  }
  const FusedBigLds L = fused_big_lds_layout(R);
  const int R2 = R * R, PS = L.PS, XS = L.XS, TS = L.TS, MB = L.MB;
  float *Pp = lds;                 // Pp[a*PS + b] = P[a,b], 0 for a == b and for b >= R
  float *Xs = lds + L.off_XM;      // Xs[j*XS + b] = X[j,b], 0 for b >= R          (phase 1)
  float *M = Xs;                   // M[j*R + a]                                    (phase 2)
  float *T = Xs;                   // T[l*TS + k] = Xn[8k + l], l = 0..7            (phase 3)
  uint8_t *Ab = reinterpret_cast<uint8_t *>(lds + L.off_mask);    // byte masks, rows of MB bytes
  const float *Ag = A + g * stride, *Pg = P + g * stride, *NSg = NS + g * stride;
  float *Xg = X + g * strideX;

  for (int e = tid; e < R * PS; e += nt) {
    const int r = e / PS, c = e - r * PS;
    Pp[e] = (c < R && c != r) ? Pg[(int64_t)r * ld + c] : 0.0f;
  }
  for (int e = tid; e < R * XS; e += nt) {
    const int r = e / XS, c = e - r * XS;
    Xs[e] = (c < R) ? Xg[(int64_t)r * ldx + c] : 0.0f;
  }
  for (int e = tid; e < R * MB; e += nt) {
    const int r = e / MB, c = e - r * MB;
    Ab[e] = (c < R && c != r && Ag[(int64_t)r * ld + c] != 0.0f) ? 1 : 0;
  }
  if (tid < 32) red[tid] = 0.f;
  // element ownership as in mpm_fused_kernel: lane (ta, tj) owns column ta and rows tj + k*TJ
  const int TJ = nt / R;
  const int C = (R >= 8) ? (R & ~31) : (R & ~3);
  const int nA = C * TJ, nB = (R - C) * TJ;
  const int startB = (((nA + 63) & ~63) + nB <= nt) ? ((nA + 63) & ~63) : nA;
  bool owner = false;
  int ta = 0, tj = 0;
  if (tid < nA) {
    owner = true;
    tj = tid / C;
    ta = tid - tj * C;
  } else if (tid >= startB && tid < startB + nB) {
    owner = true;
    const int u = tid - startB;
    tj = u / (R - C);
    ta = C + (u - tj * (R - C));
  }
  __syncthreads();

  const int n8 = R2 & ~7;
  const int JQ = (R + 3) >> 2;          // groups of 4 consecutive rows j
  const int R4 = XS;
  int done = resume != nullptr ? iters_done[g] : 0;
  int par = 0;
  const int tid0 = tid, ta0 = ta, tj0 = tj;
  for (int it = done; it < iters; ++it) {
    // (the lane's coordinates pass through an opaque move once per iteration: the addresses of its
    // sixteen rows are then recomputed where they are used, a few integer operations each, instead
    // of being kept in registers across the whole loop, which made the kernel spill)
    int tid = tid0, ta = ta0, tj = tj0;
    asm volatile("" : "+v"(tid), "+v"(ta), "+v"(tj));
    // phase 1: M[j,a] = max_b Pp[a,b] * X[j,b].  A lane takes four consecutive rows j and FOUR
    // columns a = aq + k * AQ (AQ = ceil(R / 4); consecutive lanes take consecutive aq, i.e.
    // consecutive rows of Pp: the conflict-free stride): per 4 values of b it reads four 16-byte
    // pieces of Pp and four (broadcast) pieces of X for 64 products -- an eighth of a read per
    // product instead of a third with one column per lane, which at 128 nodes was 10.5 MB of LDS
    // reads per iteration, half the iteration's time.  Maxima are exact in any order.  The
    // results stay in registers until X is no longer read.
    const int AQ = (R + 3) >> 2;
    const int n_items = JQ * AQ;          // <= 32 * 32 = nt for R <= 128
    float mreg[4][4];                     // [column k][row of the quad]
#pragma unroll
    for (int k = 0; k < 4; ++k) mreg[k][0] = mreg[k][1] = mreg[k][2] = mreg[k][3] = 0.f;
    const int p1_jq = tid / AQ, p1_aq = tid - p1_jq * AQ;
    if (tid < n_items) {
      const int j0 = p1_jq * 4;
      const float *x0 = Xs + j0 * XS;
      const float *x1 = Xs + min(j0 + 1, R - 1) * XS;
      const float *x2 = Xs + min(j0 + 2, R - 1) * XS;
      const float *x3 = Xs + min(j0 + 3, R - 1) * XS;
      const float *pr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pr[k] = Pp + min(p1_aq + k * AQ, R - 1) * PS;   // (clamped: not stored)
      for (int b = 0; b < R4; b += 4) {
        const f32x4_t a0 = *reinterpret_cast<const f32x4_t *>(x0 + b);
        const f32x4_t a1 = *reinterpret_cast<const f32x4_t *>(x1 + b);
        const f32x4_t a2 = *reinterpret_cast<const f32x4_t *>(x2 + b);
        const f32x4_t a3 = *reinterpret_cast<const f32x4_t *>(x3 + b);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4_t p = *reinterpret_cast<const f32x4_t *>(pr[k] + b);
          f32x2_t t;
          t = p.lo * a0.lo, mreg[k][0] = fmaxf(fmaxf(mreg[k][0], t.x), t.y);
          t = p.hi * a0.hi, mreg[k][0] = fmaxf(fmaxf(mreg[k][0], t.x), t.y);
          t = p.lo * a1.lo, mreg[k][1] = fmaxf(fmaxf(mreg[k][1], t.x), t.y);
          t = p.hi * a1.hi, mreg[k][1] = fmaxf(fmaxf(mreg[k][1], t.x), t.y);
          t = p.lo * a2.lo, mreg[k][2] = fmaxf(fmaxf(mreg[k][2], t.x), t.y);
          t = p.hi * a2.hi, mreg[k][2] = fmaxf(fmaxf(mreg[k][2], t.x), t.y);
          t = p.lo * a3.lo, mreg[k][3] = fmaxf(fmaxf(mreg[k][3], t.x), t.y);
          t = p.hi * a3.hi, mreg[k][3] = fmaxf(fmaxf(mreg[k][3], t.x), t.y);
        }
      }
    }
    __syncthreads();               // X has been read by everyone: M takes its place
    if (tid < n_items) {
      const int j0 = p1_jq * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int a = p1_aq + k * AQ;
        if (a < R) {
          M[j0 * R + a] = mreg[k][0];
          if (j0 + 1 < R) M[(j0 + 1) * R + a] = mreg[k][1];
          if (j0 + 2 < R) M[(j0 + 2) * R + a] = mreg[k][2];
          if (j0 + 3 < R) M[(j0 + 3) * R + a] = mreg[k][3];
        }
      }
    }
    __syncthreads();
    // phase 2: Xn[i,a] = X[i,a]*NS[i,a] + sum_{j: A[i,j]} M[j,a]  (canonical column sums), four
    // rows of a lane at a time, the results in registers
    float xn[kBigRows];
#pragma unroll
    for (int k = 0; k < kBigRows; ++k) xn[k] = 0.f;
    if (owner) {
      const float *Mcol = M + ta;
      const bool casc = ta < C;
#pragma unroll
      for (int k0 = 0; k0 < kBigRows; k0 += 4) {
        if (tj + k0 * TJ < R) {
          float xv[4], nv[4], et[4];
          const uint8_t *br[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int i = min(tj + (k0 + c) * TJ, R - 1);
            br[c] = Ab + i * MB;
            xv[c] = Xg[(int64_t)i * ldx + ta];
            nv[c] = NSg[(int64_t)i * ld + ta];
          }
          if (casc)
            cascade_chains_b<4>(Mcol, R, br, et);
          else
            ilp4_chains_b<4>(Mcol, R, br, et);
#pragma unroll
          for (int c = 0; c < 4; ++c) xn[k0 + c] = xv[c] * nv[c] + et[c];
        }
      }
    }
    __syncthreads();               // M has been read by everyone: the new X takes its place
#pragma unroll
    for (int k = 0; k < kBigRows; ++k) {
      const int i = tj + k * TJ;
      if (owner && i < R) {
        const int e = i * R + ta;
        if (e < n8)
          T[(e & 7) * TS + (e >> 3)] = xn[k];
        else
          Ttail[e - n8] = xn[k];
      }
    }
    __syncthreads();
    // phase 3: canonical norm over the R*R elements in flat order
    if (tid < 64) {
      float acc = 0.f;
      if (tid < 8) acc = norm_chain_rows(T + tid * TS, n8 >> 3);
      const float nrm = norm_finish(acc, Ttail, R2 - n8);
      if (tid == 0) s_nrm = nrm;
    }
    __syncthreads();
    const float nrm = s_nrm;
    float dmax = 0.f;
    bool isnan_any = false;
#pragma unroll
    for (int k = 0; k < kBigRows; ++k) {
      const int i = tj + k * TJ;
      if (owner && i < R) {
        const float v = xn[k] / nrm;
        const float prev = Xg[(int64_t)i * ldx + ta];
        Xg[(int64_t)i * ldx + ta] = v;
        const float d = fabsf(v - prev);
        isnan_any |= (d != d);
        dmax = fmaxf(dmax, d);
        Xs[i * XS + ta] = v;
      }
    }
    if (isnan_any) dmax = INFINITY;
    dmax = wave_max(dmax);
    if ((tid & 63) == 0) red[par * 16 + (tid >> 6)] = dmax;
    __syncthreads();
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < (kBigNT >> 6); q += 4) {
      const f32x4_t r4 = *reinterpret_cast<const f32x4_t *>(red + par * 16 + q);
      d = fmaxf(fmaxf(d, fmaxf(r4.x, r4.y)), fmaxf(r4.z, r4.w));
    }
    par ^= 1;
    ++done;
    if (early_stop && d < tol) break;
  }
  if (tid == 0) iters_done[g] = done;
  if (perm != nullptr) {
    __syncthreads();               // the last iteration's Xs (no iteration: the initial X)
    if (tid < 64) lsap_wave(Xs, R, XS, tid, perm + (int64_t)g * n_perm, n_perm, status + g);
  }
}

// after the persistent kernel: a queue fault (a bounded spin ran out) turns every graph's
// iteration count into -(fault code), so that the caller can see that X is not to be trusted
__global__ void mpm_queue_check_kernel(MpmQueue q, int n_graphs, int32_t *__restrict__ iters_done) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int err = *Q_ERROR(q);
  if (t < n_graphs && err != 0) iters_done[t] = -err;
}

__global__ void zero_i32_kernel(int32_t *a, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) a[t] = 0;
}

}  // namespace gvae

using namespace gvae;

extern "C" {


int gvae_node_sim(const float *d_A, const float *d_P, const int32_t *d_R, int n_graphs, int ld,
                  int64_t stride, float *d_NS, void *stream) {
  GVAE_REQUIRE(d_A && d_P && d_R && d_NS, GVAE_EINVAL, "node_sim: null pointer");
  GVAE_REQUIRE(n_graphs >= 0 && ld > 0 && stride >= (int64_t)ld * ld, GVAE_EINVAL,
               "node_sim: bad sizes (n_graphs=%d ld=%d)", n_graphs, ld);
  GVAE_REQUIRE(ld <= 192, GVAE_ELIMIT, "node_sim: ld=%d exceeds the LDS tile (192)", ld);
  if (n_graphs == 0) return GVAE_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = sizeof(float) * ((size_t)ld * (ld + 1) + 2 * ld);
  ProfileScope ps(F_NODE_SIM, s);
  hipLaunchKernelGGL(node_sim_kernel, dim3(n_graphs), dim3(256), lds, s, d_A, d_P, d_R, ld, stride,
                     d_NS);
  return check_launch("node_sim");
}

int gvae_edge_sim(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                  int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                  const int64_t *d_S_off, void *stream) {
  GVAE_REQUIRE(d_A && d_P && d_NS && d_R && d_S && d_S_off, GVAE_EINVAL, "edge_sim: null pointer");
  GVAE_REQUIRE(n_graphs >= 0 && ld > 0 && r_max > 0 && r_max <= ld && r_max <= 1024, GVAE_EINVAL,
               "edge_sim: bad sizes (n_graphs=%d ld=%d r_max=%d)", n_graphs, ld, r_max);
  GVAE_REQUIRE(((uintptr_t)d_S & 15) == 0, GVAE_EINVAL, "edge_sim: S must be 16-byte aligned");
  GVAE_REQUIRE(s_layout == GVAE_S_DENSE || s_layout == GVAE_S_PADDED, GVAE_EINVAL,
               "edge_sim: unknown S layout %d", s_layout);
  GVAE_REQUIRE(n_graphs <= 65535, GVAE_ELIMIT, "edge_sim: more than 65535 graphs per call");
  if (n_graphs == 0) return GVAE_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfileScope ps(F_EDGE_SIM, s);
  hipLaunchKernelGGL(edge_sim_kernel, dim3(r_max, n_graphs), dim3(256), 0, s, d_A, d_P, d_NS, d_R,
                     ld, stride, s_layout, d_S, d_S_off);
  return check_launch("edge_sim");
}

static inline int64_t pad256(int64_t b) { return (b + 255) & ~255LL; }

// ring capacity: an upper bound of the items a batch can have outstanding
static inline int64_t mpm_queue_cap(int n_graphs, int r_max) {
  return (int64_t)n_graphs * r_max * MPM_MAX_COL_BLOCKS;
}

// statistics (256 B at the very front, wherever the batch size puts the rest; never cleared by a
// launch) | Xnew (n_graphs*ldx*ldx floats) | queue words (512 B) | slots (granules) | arrive |
// stop flags of the first-iteration kernel | hand-off flags
struct MpmWs {
  float *Xnew;
  int32_t *words;
  unsigned long long *slots;
  int32_t *arrive, *stop, *handoff;
  unsigned long long *stats;
  int64_t bytes;
};
static MpmWs mpm_ws_carve(void *d_ws, int n_graphs, int ldx) {
  MpmWs m;
  char *w = (char *)d_ws;
  m.stats = (unsigned long long *)w;
  w += 256;
  m.Xnew = (float *)w;
  w += pad256((int64_t)n_graphs * ldx * ldx * 4);
  m.words = (int32_t *)w;
  w += 512;
  m.slots = (unsigned long long *)w;
  w += pad256(8 * mpm_queue_cap(n_graphs, ldx));
  m.arrive = (int32_t *)w;
  w += pad256(4LL * n_graphs);
  m.stop = (int32_t *)w;
  w += pad256(4LL * n_graphs);
  m.handoff = (int32_t *)w;
  w += pad256(4LL * n_graphs);
  m.bytes = w - (char *)d_ws;
  return m;
}

int64_t gvae_mpm_stream_ws_bytes(int n_graphs, int ldx) {
  return mpm_ws_carve(nullptr, n_graphs < 0 ? 0 : n_graphs, ldx < 0 ? 0 : ldx).bytes;
}

// Base of the hand-off threshold of gvae_match_stream (0: never): a graph leaves for the S-free
// kernel once the slabs of all live graphs add up to less than this plus four of its own slabs.
// One pass over 128 MiB at the streaming rate of 6.7 TB/s takes 19 us, the latency of one
// iteration's chain of dependent steps in the queue kernel, and the queue needs about two such
// latencies of work in flight to keep HBM busy; a graph that has the device to itself streams
// its own tensor at about a quarter of the device's rate (44-57 nodes: 33-58 us per iteration).
// Measured, ms per step / kernel's share of HBM peak with a flat threshold of 64 / 128 / 256 MiB:
// 256 dense graphs of 32 nodes 3.51 / 3.31 / 3.01 ms, 78.3 / 81.0 / 81.3 %; 64 ENZYMES-shaped
// graphs 3.82 / 3.70 / 3.51 ms, 53.8 / 56.6 / 59.1 %; with the per-graph term on a base of
// 128 / 192 / 256 MiB: 3.27 / 3.14 / 3.00 ms and 3.09 / 2.88 / 2.72 ms (58.9 / 59.5 / 59.1 %).
static int64_t g_handoff_bytes = 256LL << 20;
#ifndef GVAE_HANDOFF_MAX_NODES
#define GVAE_HANDOFF_MAX_NODES GVAE_MAX_NODES_LDS
#endif

// control words + the layout of fused_lds_layout for the largest graph of the launch's size class
constexpr int kFusedByteMaskMaxNodes = 88;      // classes served by the KMAX <= 8 kernel shapes
static inline size_t mpm_fused_lds_bytes(int r_max) {
  if (r_max > kFusedByteMaskMaxNodes)
    return sizeof(float) * (size_t)(kFusedCtl + fused_big_lds_layout(r_max).total);
  return sizeof(float) * (size_t)(kFusedCtl + fused_lds_layout(r_max).total);
}
constexpr size_t kMaxFusedLds = kMaxDynLds;

static int mpm_fused_launch(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                            int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                            int64_t strideX, int iters, float tol, int early_stop,
                            int32_t *d_iters_done, int64_t *d_perm, int n_perm, int32_t *d_status,
                            void *stream, const char *who, const int32_t *d_resume = nullptr);

// first_iter_done: the first iteration was computed by mpm_first_iter_kernel (iters_done and the
// stop flags in the workspace are set); the launch streams iterations 2..iters
static int mpm_stream_launch(const float *d_S, const int64_t *d_S_off, int s_layout, const int32_t *d_R,
                             int n_graphs, int r_max, float *d_X, int ldx, int64_t strideX, int iters,
                             float tol, int early_stop, int32_t *d_iters_done, void *d_ws,
                             bool first_iter_done, void *stream, bool profiled = true,
                             int handoff_rmax = 0) {
  GVAE_REQUIRE(d_S && d_S_off && d_R && d_X && d_iters_done && d_ws, GVAE_EINVAL,
               "mpm_stream: null pointer");
  GVAE_REQUIRE(n_graphs >= 0 && r_max > 0 && r_max <= ldx && iters >= 0, GVAE_EINVAL,
               "mpm_stream: bad sizes (n_graphs=%d r_max=%d ldx=%d iters=%d)", n_graphs, r_max, ldx,
               iters);
  GVAE_REQUIRE(n_graphs <= 65535 && r_max <= 255 && mpm_queue_cap(n_graphs, r_max) < (1LL << 30),
               GVAE_ELIMIT, "mpm_stream: too many graphs (n_graphs=%d, r_max=%d)", n_graphs, r_max);
  GVAE_REQUIRE(s_layout == GVAE_S_DENSE || s_layout == GVAE_S_PADDED, GVAE_EINVAL,
               "mpm_stream: unknown S layout %d", s_layout);
  // control words | X (padded rows) | T: row maxima of an item's column block, pieces of X_new for
  // the norm | the diagonal factors of an item's rows (up to 8 rows of small graphs)
  const size_t x_floats = (size_t)r_max * ((r_max + 3) & ~3);
  const size_t ns_floats = (size_t)(r_max > 256 ? r_max : 256);
  // Two work-groups per compute unit need 79 KiB or less each.  Where T for whole rows
  // (r_max^2 floats) does not leave that, T gets what is left of the 79 KiB and the rows of the
  // large graphs are cut into as many column blocks as it takes (99 nodes: 3, 120 nodes: 3).
  // From 121 nodes on, where X alone takes most of the 79 KiB, one work-group per unit with T
  // for whole rows: uniform batches measured 6.69 against 6.58 TB/s at 126 nodes and 0.888
  // against 0.877 of the HBM peak at 128 that way, although a ragged batch with one 126-node
  // graph would stream at 5.1 instead of 4.7 TB/s with T bounded.
  constexpr size_t kHalfCuLds = 79 * 1024;
  size_t t_cap = (size_t)r_max * r_max;
  const size_t fixed = sizeof(float) * (kCtlFloats + x_floats + ns_floats);
  if (fixed + sizeof(float) * t_cap > kHalfCuLds && fixed < kHalfCuLds && r_max <= 120) {
    const size_t room = ((kHalfCuLds - fixed) / sizeof(float)) & ~(size_t)7;
    if (room >= (size_t)r_max * ceil_div(r_max, MPM_MAX_COL_BLOCKS)) t_cap = room;
  }
  t_cap = (t_cap + 7) & ~(size_t)7;
  const size_t lds = fixed + sizeof(float) * t_cap;
  GVAE_REQUIRE(lds <= kMaxDynLds, GVAE_ELIMIT, "mpm_stream: r_max=%d needs %zu bytes of LDS", r_max,
               lds);
  if (n_graphs == 0) return GVAE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (iters == 0) {
    hipLaunchKernelGGL(zero_i32_kernel, dim3(ceil_div(n_graphs, 256)), dim3(256), 0, s, d_iters_done,
                       n_graphs);
    return check_launch("mpm_stream");
  }
  const MpmWs m = mpm_ws_carve(d_ws, n_graphs, ldx);
  float *Xnew = m.Xnew;
  MpmQueue q;
  q.words = m.words;
  q.slots = m.slots;
  q.arrive = m.arrive;
  q.stats = m.stats;
  // the S-free kernel takes over the latency-bound tail (callers that hold A and P only)
  const bool handoff = handoff_rmax > 0 && first_iter_done && early_stop && g_handoff_bytes > 0;
  q.handoff = handoff ? m.handoff : nullptr;
  q.handoff_flags = m.handoff;
  q.sfree_ok = m.stop;
  q.handoff_rmax = handoff_rmax;
  q.handoff_kb = (int)(g_handoff_bytes >> 10 > INT32_MAX ? INT32_MAX : g_handoff_bytes >> 10);
  const int32_t *first_stop = first_iter_done ? m.stop : nullptr;
  const int64_t n_items = mpm_queue_cap(n_graphs, r_max);
  q.cap = (unsigned)n_items;
  const int64_t strideXnew = (int64_t)ldx * ldx;
  // (the queue words and the ring slots are adjacent in the workspace: one fill)
  GVAE_REQUIRE(reinterpret_cast<char *>(q.slots) == reinterpret_cast<char *>(q.words) + 512 &&
                   hipMemsetAsync(q.words, 0, 512 + 8 * (size_t)n_items, s) == hipSuccess,
               GVAE_ELAUNCH, "mpm_stream: memset failed");
  // Work-group size.  What counts is how many wavefronts a compute unit holds beside the one
  // that is between two items (ticket, X reload, arrival: nothing of its own overlaps these):
  // 256 threads while three or four work-groups fit, 512 threads in two work-groups beyond that,
  // 1024 in one only where X alone leaves no room for a second.  Steady state, TB/s at 256 / 1024
  // threads: 48 nodes 6.09 / 4.64, 52: 5.61 / 4.44, 57: 5.32 / 4.53, 70: 5.52 / 5.11; 1024 wins
  // at exactly 64 nodes (6.97 against 6.69).  At 256 / 512 threads: 76 nodes 5.59 / 6.06, 82:
  // 5.26 / 5.39, 88: 5.63 / 5.87, 98: 4.58 / 5.35, and ragged batches of 64 graphs whose largest
  // has 84 ... 91 nodes 4.05-4.43 / 4.61-5.20 (70 nodes: 5.14 / 4.96, 75: the same).  With T
  // bounded, 512 / 1024 threads: 99 nodes 4.72 / 4.51, 104: 5.73 / 5.23, 112: 6.11 / 5.73, 120:
  // 6.00 / 5.45, ragged batches with one graph of 99 ... 111 nodes 4.73-5.22 / 4.09-4.57.
  const int nt = (r_max == 64 || 2 * (lds + 1024) > 160 * 1024) ? 1024 : (r_max >= 76 ? 512 : 256);
  // Items in flight below which rows are cut finer, and the size above which a row's slab is cut
  // in any case: with 256-thread work-groups an item of a large graph would otherwise run for
  // 100-270 us (a work-group streams 5-20 GB/s) and a round of a small batch ends on a few such
  // items.  ENZYMES-shaped batch of 64 graphs, ms per launch at (threads, low, cap KiB):
  // (1024, 256, 2048) 3.29, (256, 1024, 2048) 3.02, (256, 1024, 1024) 2.89, (256, 1024, 512)
  // 2.67, (256, 1024, 256) 2.66, (256, 1024, 128) 2.84.  With 1024-thread work-groups only 256
  // are resident, and cutting finer than there are work-groups to take the pieces only adds
  // rounds of items (512 threads, 512 resident: 1024 KiB and 512 items measured the same as
  // 512 KiB and 1024).  Rows are grouped (high) while many small slabs are in flight: 16-node
  // graphs 3.0 -> 1.5 ms per 1024, 24-node 3.1 -> 2.4 ms per 512.
  q.low = nt == 1024 ? 256 : (nt == 512 ? 512 : 1024);
  q.high = 4096;
  q.item_cap_kb = nt == 1024 ? 2048 : (nt == 512 ? 1024 : 512);
  q.t_cap = (int)t_cap;
  q.x_floats = (int)x_floats;
  // measured on launches whose LDS leaves one work-group per CU (128-node graphs; ragged batches
  // up to 126): 0.877 against 0.89 of HBM peak, no change on the ragged batch -- off
  q.two_buffers = 0;
  q.wide_rows = 1;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)mpm_stream_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "mpm_stream: cannot reserve %zu bytes of LDS: %s",
                 lds, hipGetErrorString(e));
  }
  // at least as many work-groups as can be resident (256 CUs x 4, 2 or 1); surplus ones start
  // when others leave, which only happens once every graph is finished, and then leave at once.
  // 256-thread launches: four work-groups per CU instead of the five that fit.  The fifth adds
  // nothing to the streaming rate (measured), and the quarter of the register file it would
  // take is what lets kernels of other streams (the backward's GEMMs, 128 registers) run beside
  // this persistent kernel instead of queueing until it ends.
  const int max_blocks = nt == 256 ? 1024 : 512;
  const int blocks = n_items < max_blocks ? (int)n_items : max_blocks;
  hipLaunchKernelGGL(mpm_queue_init_kernel, dim3(1), dim3(1024), 0, s, q, d_R, n_graphs, r_max,
                     s_layout, d_iters_done, first_stop, iters);
  {
    // (the event profiler creates events on first use: not for callers that must not allocate)
    ProfileScope ps(profiled ? F_MPM_ITER : -1, s);
    hipLaunchKernelGGL(mpm_stream_kernel, dim3(blocks), dim3(nt), lds, s, d_S, d_S_off, d_R, r_max,
                       d_X, ldx, strideX, Xnew, strideXnew, tol, early_stop, iters, s_layout, q,
                       d_iters_done);
  }
  hipLaunchKernelGGL(mpm_queue_check_kernel, dim3(ceil_div(n_graphs, 256)), dim3(256), 0, s, q,
                     n_graphs, d_iters_done);
  return check_launch("mpm_stream");
}

int gvae_mpm_stream(const float *d_S, const int64_t *d_S_off, int s_layout, const int32_t *d_R,
                    int n_graphs, int r_max, float *d_X, int ldx, int64_t strideX, int iters,
                    float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream) {
  return mpm_stream_launch(d_S, d_S_off, s_layout, d_R, n_graphs, r_max, d_X, ldx, strideX, iters, tol,
                           early_stop, d_iters_done, d_ws, false, stream);
}

// ---------------------------------------------------------------------------
// One MPM iteration before its normalisation, and its gradient, for the autograd of the public
// `_mpm` (reference model.py:247-276 is differentiable; model_legacy2.py:670-684 pins finite
// gradients w.r.t. X0).  Dense S (R,R,R,R) of one graph, any values.
//   Y[i,a] = X[i,a]*S[i,i,a,a] + sum_{j != i} max_b S[i,j,a,b]*X[j,b]
// The backward replays each maximum: its gradient goes to the maximising b, in equal shares when
// several b tie (as torch's amax does).  Not on the training path (the step uses X detached).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mpm_pool_fwd_kernel(const float *__restrict__ S, int R,
                                                           const float *__restrict__ X,
                                                           float *__restrict__ Y) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= R * R) return;
  const int i = e / R, a = e - i * R;
  const int64_t R2 = (int64_t)R * R;
  float acc = X[e] * S[((int64_t)i * R + i) * R2 + (int64_t)a * R + a];
  for (int j = 0; j < R; ++j) {
    if (j == i) continue;
    const float *row = S + ((int64_t)i * R + j) * R2 + (int64_t)a * R;
    float m = -INFINITY;
    for (int b = 0; b < R; ++b) m = fmaxf(m, row[b] * X[j * R + b]);
    acc += m;
  }
  Y[e] = acc;
}

// one lane per (i, j, a); gX must be zero on entry (atomic accumulation), gS (optional) receives
// += so that the caller can sum over iterations
__global__ __launch_bounds__(256) void mpm_pool_bwd_kernel(const float *__restrict__ S, int R,
                                                           const float *__restrict__ X,
                                                           const float *__restrict__ gY,
                                                           float *__restrict__ gX,
                                                           float *__restrict__ gS) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t R2 = (int64_t)R * R;
  if (e >= R2 * R) return;
  const int i = (int)(e / R2), j = (int)((e / R) % R), a = (int)(e % R);
  const float g = gY[i * R + a];
  if (j == i) {
    const int64_t d = ((int64_t)i * R + i) * R2 + (int64_t)a * R + a;
    atomicAdd(&gX[i * R + a], g * S[d]);
    if (gS) gS[d] += g * X[i * R + a];
    return;
  }
  const int64_t base = ((int64_t)i * R + j) * R2 + (int64_t)a * R;
  float m = -INFINITY;
  for (int b = 0; b < R; ++b) m = fmaxf(m, S[base + b] * X[j * R + b]);
  int ties = 0;
  for (int b = 0; b < R; ++b) ties += (S[base + b] * X[j * R + b] == m) ? 1 : 0;
  if (ties == 0) return;                         // only NaNs in the row
  const float share = g / (float)ties;
  for (int b = 0; b < R; ++b)
    if (S[base + b] * X[j * R + b] == m) {
      atomicAdd(&gX[j * R + b], share * S[base + b]);
      if (gS) gS[base + b] += share * X[j * R + b];
    }
}

int gvae_mpm_pool_fwd(const float *d_S, int R, const float *d_X, float *d_Y, void *stream) {
  GVAE_REQUIRE(d_S && d_X && d_Y && R > 0 && R <= 255, GVAE_EINVAL, "mpm_pool_fwd: bad arguments (R=%d)", R);
  hipLaunchKernelGGL(mpm_pool_fwd_kernel, dim3((R * R + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     d_S, R, d_X, d_Y);
  return check_launch("mpm_pool_fwd");
}

int gvae_mpm_pool_bwd(const float *d_S, int R, const float *d_X, const float *d_gY, float *d_gX,
                      float *d_gS, void *stream) {
  GVAE_REQUIRE(d_S && d_X && d_gY && d_gX && R > 0 && R <= 255, GVAE_EINVAL,
               "mpm_pool_bwd: bad arguments (R=%d)", R);
  hipStream_t s = (hipStream_t)stream;
  GVAE_REQUIRE(hipMemsetAsync(d_gX, 0, sizeof(float) * (size_t)R * R, s) == hipSuccess, GVAE_ELAUNCH,
               "mpm_pool_bwd: memset failed");
  const int64_t n = (int64_t)R * R * R;
  hipLaunchKernelGGL(mpm_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_S, R, d_X,
                     d_gY, d_gX, d_gS);
  return check_launch("mpm_pool_bwd");
}

// ---------------------------------------------------------------------------
// The same matching for callers that cannot allocate device memory: the workspace of
// gvae_mpm_stream (next-X buffer, queue words, ring slots, arrival counters) lives in a
// file-scope device array instead of coming in as an argument.  Host code never passes the
// array's name to a kernel or to a memset (on the host that name is a stand-in, not a device
// address): the address is taken once per device with hipGetSymbolAddress and kept.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[GVAE_STATIC_SCRATCH_BYTES];

#define GVAE_MAX_DEVICES 64
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[GVAE_MAX_DEVICES];   // device address of `scratch` on each device

// address of this device's copy of `scratch` (every GPU holds its own, at its own address)
static int scratch_of_current_device(void **out) {
  int dev = -1;
  GVAE_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < GVAE_MAX_DEVICES, GVAE_ELAUNCH,
               "mpm_stream_static: cannot identify the current device");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    GVAE_REQUIRE(e == hipSuccess && p, GVAE_ELAUNCH,
                 "mpm_stream_static: hipGetSymbolAddress failed on device %d: %s", dev,
                 hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return GVAE_OK;
}

int gvae_mpm_stream_static(const float *d_S, const int64_t *d_S_off, int s_layout, const int32_t *d_R,
                           int n_graphs, int r_max, float *d_X, int ldx, int64_t strideX, int iters,
                           float tol, int early_stop, int32_t *d_iters_done, void *stream) {
  GVAE_REQUIRE(n_graphs >= 0 && ldx > 0, GVAE_EINVAL, "mpm_stream_static: bad sizes (n_graphs=%d ldx=%d)",
               n_graphs, ldx);
  const int64_t need = gvae_mpm_stream_ws_bytes(n_graphs, ldx);
  GVAE_REQUIRE(need <= (int64_t)GVAE_STATIC_SCRATCH_BYTES, GVAE_ELIMIT,
               "mpm_stream_static: %d graphs with ldx=%d need %lld bytes of scratch, the built-in "
               "buffer holds %d (use gvae_mpm_stream with a workspace of your own)",
               n_graphs, ldx, (long long)need, GVAE_STATIC_SCRATCH_BYTES);
  void *ws = nullptr;
  const int rc = scratch_of_current_device(&ws);
  if (rc) return rc;
  // the launch clears what must start at zero (queue words and ring slots by memset, arrival
  // counters and iteration counts in the queue's set-up kernel); the rest of the buffer is
  // written before it is read
  return mpm_stream_launch(d_S, d_S_off, s_layout, d_R, n_graphs, r_max, d_X, ldx, strideX, iters, tol,
                           early_stop, d_iters_done, ws, false, stream, /*profiled=*/false);
}

// (read when a streaming launch is enqueued; launches already enqueued keep the value they saw)
int gvae_set_stream_handoff_bytes(int64_t bytes) {
  GVAE_REQUIRE(bytes >= 0, GVAE_EINVAL, "set_stream_handoff_bytes: negative threshold");
  g_handoff_bytes = bytes;
  return GVAE_OK;
}

int gvae_mpm_stream_stats(void *d_ws, int64_t *h_bytes, int64_t *h_graph_iterations, int reset,
                          void *stream) {
  GVAE_REQUIRE(d_ws, GVAE_EINVAL, "mpm_stream_stats: null workspace");
  const MpmWs m = mpm_ws_carve(d_ws, 0, 0);
  hipStream_t s = (hipStream_t)stream;
  unsigned long long h[2] = {0, 0};
  GVAE_REQUIRE(hipMemcpyAsync(h, m.stats, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess &&
                   hipStreamSynchronize(s) == hipSuccess,
               GVAE_ELAUNCH, "mpm_stream_stats: copy failed");
  if (reset)
    GVAE_REQUIRE(hipMemsetAsync(m.stats, 0, sizeof(h), s) == hipSuccess, GVAE_ELAUNCH,
                 "mpm_stream_stats: reset failed");
  if (h_bytes) *h_bytes = (int64_t)h[0];
  if (h_graph_iterations) *h_graph_iterations = (int64_t)h[1];
  return GVAE_OK;
}

}  // extern "C"

// The iterations of the graphs the streaming kernel handed over (flags in the workspace), by the
// S-free kernel; the flags are cleared afterwards, so a second call does nothing.  May run on
// another stream than the streaming launch once that launch has finished there (the step runs it
// with the assignment, beside the next batch's similarity tensor).
int gvae::match_stream_tail(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                            int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                            int64_t strideX, int iters, float tol, int early_stop,
                            int32_t *d_iters_done, void *d_ws, void *stream) {
  // (without early stop or with a single iteration there is no streaming launch behind this call;
  // otherwise the flags are whatever that launch wrote -- its set-up clears them, its hand-off
  // decision is taken when it is enqueued -- so a change of the threshold between the launch and
  // this call cannot strand a graph)
  if (!early_stop || iters <= 1 || n_graphs <= 0) return GVAE_OK;
  GVAE_REQUIRE(d_A && d_P && d_NS && d_R && d_X && d_iters_done && d_ws, GVAE_EINVAL,
               "match_stream (tail): null pointer");
  const int r_cap = r_max < GVAE_HANDOFF_MAX_NODES ? r_max : GVAE_HANDOFF_MAX_NODES;
  const MpmWs m = mpm_ws_carve(d_ws, n_graphs, ldx);
  const int rc = mpm_fused_launch(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_cap, d_X, ldx, strideX,
                                  iters, tol, early_stop, d_iters_done, nullptr, 0, nullptr, stream,
                                  "match_stream (tail)", m.handoff);
  if (rc) return rc;
  GVAE_REQUIRE(hipMemsetAsync(m.handoff, 0, sizeof(int32_t) * (size_t)n_graphs, (hipStream_t)stream) ==
                   hipSuccess,
               GVAE_ELAUNCH, "match_stream (tail): memset failed");
  return GVAE_OK;
}

// Similarity tensor and the first MPM iteration (computed from row maxima of S without reading
// the tensor back): everything of a streamed matching that precedes the persistent launch.
int gvae::match_stream_prep(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                            int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                            const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                            float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream) {
  GVAE_REQUIRE(d_A && d_P && d_NS && d_R && d_S && d_S_off && d_X && d_iters_done && d_ws, GVAE_EINVAL,
               "match_stream: null pointer");
  GVAE_REQUIRE(n_graphs >= 0 && r_max > 0 && r_max <= ld && r_max <= ldx && iters >= 0, GVAE_EINVAL,
               "match_stream: bad sizes (n_graphs=%d r_max=%d ld=%d ldx=%d iters=%d)", n_graphs, r_max,
               ld, ldx, iters);
  if (n_graphs == 0 || iters == 0)
    return gvae_edge_sim(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, s_layout, d_S, d_S_off, stream);
  GVAE_REQUIRE(n_graphs <= 65535, GVAE_ELIMIT, "match_stream: more than 65535 graphs per call");
  GVAE_REQUIRE(d_S && ((uintptr_t)d_S & 15) == 0 && r_max <= 1024 &&
                   (s_layout == GVAE_S_DENSE || s_layout == GVAE_S_PADDED),
               GVAE_EINVAL, "match_stream: S must be 16-byte aligned, its layout dense or padded");
  const size_t r2 = (size_t)r_max * r_max;
  size_t lds = sizeof(float) * (kCtlFloats + ((r2 + 3) & ~(size_t)3) + 2 * (size_t)r_max);
  GVAE_REQUIRE(lds <= kMaxDynLds, GVAE_ELIMIT, "match_stream: r_max=%d needs %zu bytes of LDS", r_max,
               lds);
  const int a_in_lds = lds + sizeof(float) * r2 <= kMaxDynLds;    // else A is read in place (L2)
  if (a_in_lds) lds += sizeof(float) * r2;
  hipStream_t s = (hipStream_t)stream;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)mpm_first_iter_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "match_stream: cannot reserve %zu bytes of LDS: %s",
                 lds, hipGetErrorString(e));
  }
  // the stop flags sit behind the arrival counters in the workspace (see mpm_stream_launch)
  const MpmWs m = mpm_ws_carve(d_ws, n_graphs, ldx);
  hipLaunchKernelGGL(mpm_first_iter_kernel, dim3(n_graphs), dim3(r_max >= 48 ? 1024 : 256), lds, s, d_A,
                     d_P, d_NS, d_R, r_max, ld, stride, d_X, ldx, strideX, tol, early_stop, a_in_lds,
                     d_iters_done, m.stop);
  // the tensor, and with it the second iteration of the graphs small enough (X and T of the
  // largest of them in LDS: at most 33 KiB per work-group)
  // (LDS for whichever graph of at most r_max nodes needs most)
  size_t sim_floats = 0;
  for (int r = 1; r <= r_max; ++r) {
    const size_t need = sim_iter_lds_floats(r, s_row_stride(r, s_layout));
    sim_floats = need > sim_floats ? need : sim_floats;
  }
  const size_t lds_sim = sizeof(float) * sim_floats;
  const size_t lds_norm = sizeof(float) * (size_t)r_max * r_max;
  GVAE_REQUIRE(lds_sim <= kMaxDynLds && lds_norm <= kMaxDynLds, GVAE_ELIMIT,
               "match_stream: r_max=%d needs %zu bytes of LDS", r_max, lds_norm);
  if (lds_norm > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)mpm_second_norm_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_norm);
    GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "match_stream: cannot reserve %zu bytes of LDS: %s",
                 lds_norm, hipGetErrorString(e));
  }
  {
    ProfileScope ps(F_EDGE_SIM, s);
    hipLaunchKernelGGL(edge_sim_iter_kernel, dim3(r_max, n_graphs), dim3(256), lds_sim, s, d_A, d_P,
                       d_NS, d_R, ld, stride, s_layout, d_S, d_S_off, d_X, ldx, strideX, m.Xnew,
                       (int64_t)ldx * ldx, m.stop, iters);
  }
  if (iters >= 2)
    hipLaunchKernelGGL(mpm_second_norm_kernel, dim3(n_graphs), dim3(256), lds_norm,
                       s, d_R, r_max, d_X, ldx, strideX, m.Xnew, (int64_t)ldx * ldx, tol, early_stop,
                       iters, d_iters_done, m.stop);
  return check_launch("match_stream");
}

// Iterations 2..k of the graphs match_stream_prep left unconverged, by the persistent streaming
// launch; with early stop it hands the latency-bound tail of graphs the S-free kernel takes to
// match_stream_tail (flags in the workspace).
int gvae::match_stream_iterate(const int32_t *d_R, int n_graphs, int r_max, int s_layout,
                               const float *d_S, const int64_t *d_S_off, float *d_X, int ldx,
                               int64_t strideX, int iters, float tol, int early_stop,
                               int32_t *d_iters_done, void *d_ws, void *stream) {
  if (n_graphs == 0 || iters == 0)
    return mpm_stream_launch(d_S, d_S_off, s_layout, d_R, n_graphs, r_max, d_X, ldx, strideX, iters,
                             tol, early_stop, d_iters_done, d_ws, false, stream);
  if (iters == 1) return GVAE_OK;
  // Graphs small enough for the S-free kernel may be handed to it once the batch is bound by
  // the latency of an iteration rather than by HBM (see MpmQueue): same X, same iteration
  // counts, bit for bit, whichever kernel runs an iteration.
  // Every size the S-free kernel takes: an S-free iteration costs 5 us (32 nodes), 34 us (88) or
  // 78 us (128), less than the chain of one streamed iteration of a graph that has the device to
  // itself (19 us + its slab's transfer: 64 MiB - 1 GiB at a quarter of the streaming rate).
  const int r_cap = r_max < GVAE_HANDOFF_MAX_NODES ? r_max : GVAE_HANDOFF_MAX_NODES;
  const bool handoff = early_stop && g_handoff_bytes > 0 && r_cap > 0;
  return mpm_stream_launch(d_S, d_S_off, s_layout, d_R, n_graphs, r_max, d_X, ldx, strideX, iters, tol,
                           early_stop, d_iters_done, d_ws, true, stream, true, handoff ? r_cap : 0);
}

int gvae::match_stream_run(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                     int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                     const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                     float tol, int early_stop, int32_t *d_iters_done, void *d_ws, bool run_tail,
                     void *stream) {
  int rc = match_stream_prep(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, s_layout, d_S, d_S_off, d_X,
                             ldx, strideX, iters, tol, early_stop, d_iters_done, d_ws, stream);
  if (rc) return rc;
  rc = match_stream_iterate(d_R, n_graphs, r_max, s_layout, d_S, d_S_off, d_X, ldx, strideX, iters, tol,
                            early_stop, d_iters_done, d_ws, stream);
  if (rc || !run_tail) return rc;
  return match_stream_tail(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, d_X, ldx, strideX, iters, tol,
                           early_stop, d_iters_done, d_ws, stream);
}

extern "C" {

int gvae_match_stream(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                      int n_graphs, int ld, int64_t stride, int r_max, int s_layout, float *d_S,
                      const int64_t *d_S_off, float *d_X, int ldx, int64_t strideX, int iters,
                      float tol, int early_stop, int32_t *d_iters_done, void *d_ws, void *stream) {
  return match_stream_run(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, s_layout, d_S, d_S_off, d_X,
                          ldx, strideX, iters, tol, early_stop, d_iters_done, d_ws, true, stream);
}

static int mpm_fused_launch(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                            int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                            int64_t strideX, int iters, float tol, int early_stop,
                            int32_t *d_iters_done, int64_t *d_perm, int n_perm, int32_t *d_status,
                            void *stream, const char *who, const int32_t *d_resume) {
  GVAE_REQUIRE(d_A && d_P && d_NS && d_R && d_X && d_iters_done, GVAE_EINVAL, "%s: null pointer", who);
  GVAE_REQUIRE(n_graphs >= 0 && r_max > 0 && r_max <= ld && r_max <= ldx && iters >= 0, GVAE_EINVAL,
               "%s: bad sizes (n_graphs=%d r_max=%d ld=%d ldx=%d)", who, n_graphs, r_max, ld, ldx);
  size_t lds = mpm_fused_lds_bytes(r_max);
  int r_lo = INT32_MIN, r_hi = r_max;
  GVAE_REQUIRE(r_max <= GVAE_MAX_NODES_LDS && lds <= kMaxFusedLds, GVAE_ELIMIT,
               "%s: r_max=%d needs %zu bytes of LDS (limit 160 KiB)", who, r_max, lds);
  if (n_graphs == 0) return GVAE_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfileScope ps(F_MPM_FUSED, s);
  auto launch = [&](auto kernel, int nt) -> int {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      GVAE_REQUIRE(e == hipSuccess, GVAE_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", who, lds,
                   hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3(n_graphs), dim3(nt), lds, s, d_A, d_P, d_NS, d_R, ld, stride, d_X,
                       ldx, strideX, iters, tol, early_stop, d_iters_done, d_perm, n_perm, d_status,
                       d_resume, r_lo, r_hi);
    return GVAE_OK;
  };
  auto pick = [&](int rm) -> int {
    // rows per lane = ceil(R / floor(NT / R)) must not exceed KMAX
    if (rm <= 16) return launch(mpm_fused_kernel<256, 4>, 256);
    if (rm <= 32) return launch(mpm_fused_kernel<512, 2>, 512);
    if (rm <= 44) return launch(mpm_fused_kernel<512, 4>, 512);
    if (rm <= 64) return launch(mpm_fused_kernel<1024, 4>, 1024);
    if (rm <= kFusedByteMaskMaxNodes) return launch(mpm_fused_kernel<1024, 8>, 1024);
    return launch(mpm_fused_big_kernel, kBigNT);
  };
  // Resumed graphs (the tail of the streamed matching; at most kFusedByteMaskMaxNodes nodes) all
  // run in ONE launch, in the kernel shape of the largest: the tail takes as long as its slowest
  // graph, which is one of the large ones, whatever shape the small ones run in.  One launch per
  // size class on streams of the library's own, forked from and joined to the caller's stream, was
  // faster when the streams found hardware queues of their own (ENZYMES-shaped batches of up to
  // 126 nodes: 2.09 against 2.13 ms per step) and slower when they shared one with the caller's
  // streams, a process has four (2.29 ms; batches of up to 40 nodes: 0.43-0.57 against 0.41 ms).
  const int rc1 = pick(r_max);
  return rc1 ? rc1 : check_launch(who);
}

int gvae_mpm_fused(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                   int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                   int64_t strideX, int iters, float tol, int early_stop, int32_t *d_iters_done,
                   void *stream) {
  return mpm_fused_launch(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, d_X, ldx, strideX, iters,
                          tol, early_stop, d_iters_done, nullptr, 0, nullptr, stream, "mpm_fused");
}

int gvae_mpm_fused_assign(const float *d_A, const float *d_P, const float *d_NS, const int32_t *d_R,
                          int n_graphs, int ld, int64_t stride, int r_max, float *d_X, int ldx,
                          int64_t strideX, int iters, float tol, int early_stop,
                          int32_t *d_iters_done, int64_t *d_perm, int n_perm, int32_t *d_status,
                          void *stream) {
  GVAE_REQUIRE(d_perm && d_status && n_perm >= r_max, GVAE_EINVAL,
               "mpm_fused_assign: perm / status missing or n_perm=%d < r_max=%d", n_perm, r_max);
  // The epilogue pays while every graph's work-group is resident at once (1024 / 512 / 256
  // work-groups of 256 / 512 / 1024 threads on 256 CUs): a graph is then assigned as soon as it
  // has converged.  With more graphs than that, an assignment that holds a whole work-group's
  // registers and LDS delays the next graph's start, and the stand-alone kernel (one wavefront
  // and 4 n^2 bytes of LDS per graph, all graphs resident) is faster: 1024 graphs of 64 nodes,
  // 50 iterations each: 3.9 ms per step against 4.8.
  const int resident = r_max <= 32 ? 1024 : r_max <= 64 ? 512 : 256;
  if (n_graphs <= resident)
    return mpm_fused_launch(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, d_X, ldx, strideX,
                            iters, tol, early_stop, d_iters_done, d_perm, n_perm, d_status, stream,
                            "mpm_fused_assign");
  int rc = mpm_fused_launch(d_A, d_P, d_NS, d_R, n_graphs, ld, stride, r_max, d_X, ldx, strideX, iters,
                            tol, early_stop, d_iters_done, nullptr, 0, nullptr, stream,
                            "mpm_fused_assign");
  if (rc) return rc;
  return gvae_lsap(d_X, ldx, strideX, d_R, n_graphs, r_max, d_perm, n_perm, d_status, stream);
}

}  // extern "C"
