// Embedding-sized linears (M = batch rows, a few dozen): time / condition MLPs and the GlobalContext squeeze-excite MLP
// (reference: unet.py:356-367, residual.py:20-26,33-37).  These are weight-streaming problems -- 2*M*N*K FLOPs against N*K*4 B of
// fp32 master weights read ONCE -- that went through the conv GEMM path at 12 launches per linear and direction (casts, pads,
// two weight packs, 128-row tiles for 32 rows, separate bias / activation / column-sum kernels).  Here: fp32 in, fp32 out, the
// fp32 master weight read in place, input SiLU / output sigmoid / bias and their derivatives fused, three kernels in total:
//   skinny_fwd : y = out_act( in_act(x) W^T + b )                   one workgroup per 32 output columns, K split over 4 waves
//   skinny_dx  : dx = ( (dy * out_act'(y)) W ) * in_act'(x)          one workgroup per 32 input columns, N split over 4 waves
//   skinny_dw  : dW += (dy * out_act')^T in_act(x) ;  db += column sums      one wave per 32x32 tile of dW
// Arithmetic follows the compute mode like every other linear: MODE 1 rounds both operands to bf16 and uses
// v_mfma_f32_32x32x16_bf16 (what torch autocast does to nn.Linear / 1x1 Conv1d); MODE 0 uses v_mfma_f32_32x32x2f32 (exact f32).
#include "common.hpp"

#define SK_ACT_NONE 0
#define SK_ACT_SILU 1
#define SK_ACT_SIGMOID 2

// 8 consecutive floats row[k .. k+8) with zero fill past K (and for a null row)
__device__ __forceinline__ void sk_load8(const float* row, int k, int K, float (&v)[8]) {
  if (row != nullptr && k + 8 <= K && ((reinterpret_cast<uintptr_t>(row + k) & 15) == 0)) {
    load8(row + k, v);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (row != nullptr && k + j < K) ? row[k + j] : 0.f;
  }
}

template <int MODE>
__device__ __forceinline__ void sk_mma(const float (&a)[8], const float (&b)[8], f32x16& acc) {
  if constexpr (MODE == 1) {
    bf16x8 fa, fb;
#pragma unroll
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)a[j]; fb[j] = (__bf16)b[j]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  }
}

// cross-wave sum of the four partial 32x32 accumulators: red[wave][row*32 + col]; returns after the barrier
__device__ __forceinline__ void sk_reduce_store(float* red, int wave, int lane, const f32x16& acc) {
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[r];
  __syncthreads();
}

// Loads of all three kernels are BRANCH-FREE in the vector (VEC) instantiations: hipcc waits out every global load that sits under
// an exec-masked or scalar branch before it issues the next one, and the first version of these kernels (loads inside `if (m < M)`,
// an aligned / unaligned choice per load) ran one memory round trip per k-step -- 17-18 us for a 32x256x128 weight gradient, 30 us
// for the 2048-wide FiLM projections, whatever the size.  VEC (every row 16-byte aligned, K and N multiples of 8; the host checks):
// indices are clamped into range, the loads of four k-steps are issued together, out-of-range values are zeroed by selects.

// 8 consecutive floats of a row that is known to be readable at [k, k+8) -- no branch
__device__ __forceinline__ void sk_zero8(float (&v)[8], bool keep) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = keep ? v[j] : 0.f;
}

// y[m][n] = out_act( sum_k in_act(x[m][k]) W[n][k] + b[n] )
template <int MODE, bool VEC>
__device__ __forceinline__ void sk_fwd_tile(const float* __restrict__ x, long ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                            float* __restrict__ y, long ldy, int M, int N, int K, int in_act, int out_act, int n0, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  for (int m0 = 0; m0 < M; m0 += 32) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (VEC) {
      const float* wrow = W + (long)min(n0 + lr, N - 1) * K;            // clamped rows: their products land in accumulator rows /
      const float* xrow = x + (long)min(m0 + lr, M - 1) * ldx;          // columns that are never stored
      for (int kb = 16 * wave + 8 * lh; kb < K + 8 * lh; kb += 256) {   // (+ 8 * lh: both lane halves run the same trip count)
        float a[4][8], b[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kc = min(kb + 64 * u, K - 8);
          load8(xrow + kc, a[u]);
          load8(wrow + kc, b[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = kb + 64 * u < K;
          sk_zero8(a[u], ok);
          sk_zero8(b[u], ok);
          if (in_act == SK_ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[u][j] = silu_f(a[u][j]);
          }
          sk_mma<MODE>(a[u], b[u], acc);
        }
      }
    } else {
      const float* wrow = (n0 + lr < N) ? W + (long)(n0 + lr) * K : nullptr;
      const float* xrow = (m0 + lr < M) ? x + (long)(m0 + lr) * ldx : nullptr;
      for (int kb = 16 * wave; kb < K; kb += 64) {
        float a[8], b[8];
        sk_load8(xrow, kb + 8 * lh, K, a);
        sk_load8(wrow, kb + 8 * lh, K, b);
        if (in_act == SK_ACT_SILU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = silu_f(a[j]);
        }
        sk_mma<MODE>(a, b, acc);
      }
    }
    sk_reduce_store(red, wave, lane, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 5, col = idx & 31;
      const int m = m0 + row, n = n0 + col;
      if (m < M && n < N) {
        float v = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx] + (bias ? bias[n] : 0.f);
        if (out_act == SK_ACT_SIGMOID) v = sigmoid_f(v);
        else if (out_act == SK_ACT_SILU) v = silu_f(v);
        y[(long)m * ldy + n] = v;
      }
    }
    __syncthreads();
  }
}

template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ y, long ldy, int M, int N, int K, int in_act, int out_act) {
  __shared__ float red[4 * 1024];
  sk_fwd_tile<MODE, VEC>(x, ldx, W, bias, y, ldy, M, N, K, in_act, out_act, blockIdx.x * 32, red);
}

// last descriptor whose block0 <= bid (block0 ascending from 0): which linear of a group a workgroup serves
__device__ __forceinline__ int sk_find_desc(const osuf_linear_desc* __restrict__ descs, int n, int bid) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block0 <= bid) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Several linears that share ONE input (the FiLM projections: every ResidualBlock applies Sequential(SiLU, Linear) to the same
// (B, 2048) embedding, residual.py:104-111) in one launch: 35 launches of 16-64 workgroups each become one of ~1,200.
template <int MODE>
__global__ __launch_bounds__(256) void skinny_fwd_group_kernel(const float* __restrict__ x, long ldx, const osuf_linear_desc* __restrict__ descs, int n,
                                                               int M, int K, int in_act) {
  __shared__ float red[4 * 1024];
  const osuf_linear_desc d = descs[sk_find_desc(descs, n, blockIdx.x)];
  const int n0 = ((int)blockIdx.x - d.block0) * 32;
  if (n0 >= d.N) return;
  sk_fwd_tile<MODE, true>(x, ldx, d.W, d.bias, d.y, d.ldy, M, d.N, K, in_act, SK_ACT_NONE, n0, red);
}

__device__ __forceinline__ float sk_dact_from_out(int out_act, float yv) {      // derivative expressed through the OUTPUT
  return out_act == SK_ACT_SIGMOID ? yv * (1.f - yv) : 1.f;
}

// dx[m][k] += in_act'(x[m][k]) * sum_{n in this block's slice} dz[m][n] W[n][k],   dz = dy * out_act'(y)
// grid (K/32, nsplit): a 32-column strip of W is a strided read (128 B per row), so the N range is cut into `nsplit` slices to
// put >= 256 workgroups on the chip; slices combine by fp32 atomics into a zeroed dx (the in_act' factor distributes over them).
template <int MODE, bool VEC>
__device__ __forceinline__ void sk_dx_tile(const float* __restrict__ dy, long lddy, const float* __restrict__ y, long ldy,
                                           const float* __restrict__ W, const float* __restrict__ x, long ldx, float* __restrict__ dx,
                                           long lddx, int M, int N, int K, int in_act, int out_act, int k0, int n_begin, int n_end, bool atomic,
                                           float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  for (int m0 = 0; m0 < M; m0 += 32) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (VEC) {
      const long mrow = min(m0 + lr, M - 1);
      const float* dyrow = dy + mrow * lddy;
      const float* yrow = y + mrow * ldy;                                // only dereferenced with an output activation
      const float* wcol = W + min(k0 + lr, K - 1);
      for (int nb = n_begin + 16 * wave + 8 * lh; nb < n_end + 8 * lh; nb += 128) {
        float a[2][8], yv[2][8], b[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int nc = min(nb + 64 * u, N - 8);
          load8(dyrow + nc, a[u]);
          if (out_act != SK_ACT_NONE) load8(yrow + nc, yv[u]);           // (uniform condition: no exec mask)
#pragma unroll
          for (int j = 0; j < 8; ++j) b[u][j] = wcol[(long)(nc + j) * K];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool ok = nb + 64 * u < n_end;                           // n_end is a multiple of 8 here: whole 8-row groups
          if (out_act != SK_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[u][j] *= sk_dact_from_out(out_act, yv[u][j]);
          }
          sk_zero8(a[u], ok);
          sk_zero8(b[u], ok);
          sk_mma<MODE>(a[u], b[u], acc);
        }
      }
    } else {
      const bool mok = m0 + lr < M;
      const float* dyrow = mok ? dy + (long)(m0 + lr) * lddy : nullptr;
      const float* yrow = (mok && out_act != SK_ACT_NONE) ? y + (long)(m0 + lr) * ldy : nullptr;
#pragma unroll 2
      for (int nb = n_begin + 16 * wave; nb < n_end; nb += 64) {
        float a[8], b[8];
        sk_load8(dyrow, nb + 8 * lh, n_end, a);
        if (out_act != SK_ACT_NONE) {
          float yv[8];
          sk_load8(yrow, nb + 8 * lh, n_end, yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] *= sk_dact_from_out(out_act, yv[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = nb + 8 * lh + j;
          b[j] = (n < n_end && k0 + lr < K) ? W[(long)n * K + k0 + lr] : 0.f;
        }
        sk_mma<MODE>(a, b, acc);
      }
    }
    sk_reduce_store(red, wave, lane, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 5, col = idx & 31;
      const int m = m0 + row, k = k0 + col;
      if (m < M && k < K) {
        float v = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx];
        if (in_act == SK_ACT_SILU) v *= silu_grad_f(x[(long)m * ldx + k]);
        if (atomic) atomic_add_f32(dx + (long)m * lddx + k, v);
        else dx[(long)m * lddx + k] = v;
      }
    }
    __syncthreads();
  }
}

template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void skinny_dx_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ y, long ldy,
                                                        const float* __restrict__ W, const float* __restrict__ x, long ldx, float* __restrict__ dx,
                                                        long lddx, int M, int N, int K, int in_act, int out_act, int n_per_split) {
  __shared__ float red[4 * 1024];
  const int n_begin = blockIdx.y * n_per_split;
  sk_dx_tile<MODE, VEC>(dy, lddy, y, ldy, W, x, ldx, dx, lddx, M, N, K, in_act, out_act, blockIdx.x * 32, n_begin, min(N, n_begin + n_per_split),
                        gridDim.y > 1, red);
}

// dx += in_act'(x) * sum_i dy_i W_i over a group of linears with one shared input: grid (K/32, sum_i ceil(N_i / 512)); every slice
// adds into the zeroed dx with atomics (descs[i].block0 = running sum of the slice counts)
template <int MODE>
__global__ __launch_bounds__(256) void skinny_dx_group_kernel(const osuf_linear_desc* __restrict__ descs, int n, const float* __restrict__ x, long ldx,
                                                              float* __restrict__ dx, long lddx, int M, int K, int in_act) {
  __shared__ float red[4 * 1024];
  const osuf_linear_desc d = descs[sk_find_desc(descs, n, blockIdx.y)];
  const int n_begin = ((int)blockIdx.y - d.block0) * 512;
  if (n_begin >= d.N) return;
  sk_dx_tile<MODE, true>(d.dy, d.lddy, nullptr, 0, d.W, x, ldx, dx, lddx, M, d.N, K, in_act, SK_ACT_NONE, blockIdx.x * 32, n_begin,
                         min(d.N, n_begin + 512), true, red);
}

// dW[n][k] (+)= sum_m dz[m][n] in_act(x[m][k]) ;  db[n] += sum_m dz[m][n]      one wave per 32x32 tile, 4 tiles per workgroup
// (branch-free for every shape: row / column indices are clamped, out-of-range values zeroed by selects)
template <int MODE>
__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ y, long ldy,
                                                        const float* __restrict__ x, long ldx, float* __restrict__ dW, float* __restrict__ db,
                                                        int M, int N, int K, int in_act, int out_act, int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
  const int ktiles = (K + 31) >> 5, ntiles = (N + 31) >> 5;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ktiles * ntiles) return;
  const int n0 = (tile / ktiles) * 32, k0 = (tile % ktiles) * 32;
  const int n = n0 + lr, k = k0 + lr;
  const int nc = min(n, N - 1), kc = min(k, K - 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int mb = 0; mb < M; mb += 32) {
    float a[2][8], yv[2][8], b[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long mc = min(mb + 16 * u + 8 * lh + j, M - 1);
        a[u][j] = dy[mc * lddy + nc];
        if (out_act != SK_ACT_NONE) yv[u][j] = y[mc * ldy + nc];         // (uniform condition)
        b[u][j] = x[mc * ldx + kc];
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool mok = mb + 16 * u + 8 * lh + j < M;
        float dz = a[u][j];
        if (out_act != SK_ACT_NONE) dz *= sk_dact_from_out(out_act, yv[u][j]);
        float xv = in_act == SK_ACT_SILU ? silu_f(b[u][j]) : b[u][j];
        dz = (mok && n < N) ? dz : 0.f;
        xv = (mok && k < K) ? xv : 0.f;
        a[u][j] = dz; b[u][j] = xv;
        bsum += dz;
      }
      sk_mma<MODE>(a[u], b[u], acc);
    }
  }
  if (db != nullptr && k0 == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0 && n < N) db[n] += bsum;                     // the k0 == 0 tile is the only writer of db[n0 .. n0+32)
  }
  if (k < K) {
    if (accumulate) {                                        // read-modify-write: the 16 old values are fetched together first
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = dW[(long)min(n0 + (r & 3) + 8 * (r >> 2) + 4 * lh, N - 1) * K + k];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += old[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (nn < N) dW[(long)nn * K + k] = acc[r];
    }
  }
}

static bool sk_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool sk_bad(int M, int N, int K, int in_act, int out_act) {
  return M <= 0 || N <= 0 || K <= 0 || (in_act != SK_ACT_NONE && in_act != SK_ACT_SILU) ||
         (out_act != SK_ACT_NONE && out_act != SK_ACT_SIGMOID);
}

extern "C" int osuf_skinny_fwd(int mode, const float* x, long ldx, const float* W, const float* bias, float* y, long ldy, int M, int N, int K,
                               int in_act, int out_act, hipStream_t stream) {
  if (!x || !W || !y || sk_bad(M, N, K, in_act, out_act)) return OSUF_EINVAL;
  const dim3 grid((N + 31) / 32);
  const bool vec = K % 8 == 0 && ldx % 4 == 0 && sk_al16(x) && sk_al16(W);
  void (*kern)(const float*, long, const float*, const float*, float*, long, int, int, int, int, int) =
      mode == OSUF_DT_BF16 ? (vec ? skinny_fwd_kernel<1, true> : skinny_fwd_kernel<1, false>) : (vec ? skinny_fwd_kernel<0, true> : skinny_fwd_kernel<0, false>);
  hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, x, ldx, W, bias, y, ldy, M, N, K, in_act, out_act);
  return osuf_launch_status();
}

/* dx may be NULL (input needs no gradient); dW / db may be NULL (frozen parameters); accumulate: dW += instead of = (db is always +=) */
extern "C" int osuf_skinny_bwd(int mode, const float* dy, long lddy, const float* y, long ldy, const float* x, long ldx, const float* W,
                               float* dx, long lddx, float* dW, float* db, int M, int N, int K, int in_act, int out_act, int accumulate,
                               hipStream_t stream) {
  if (!dy || !x || !W || sk_bad(M, N, K, in_act, out_act) || (out_act != SK_ACT_NONE && !y)) return OSUF_EINVAL;
  if (dx) {
    const int ktiles = (K + 31) / 32;
    int nsplit = (256 + ktiles - 1) / ktiles;                 // about one workgroup per CU
    const int max_split = (N + 127) / 128;                    // but at least 128 rows of W per slice
    if (nsplit > max_split) nsplit = max_split;
    if ((long)N * K <= 64 * 1024) nsplit = 1;                 // <= 256 KB of weights (the GlobalContext MLPs): one slice per strip -- plain
    if (nsplit < 1) nsplit = 1;                               // stores, no memset launch, and the walk over N is 4 steps at most
    int nps = (N + nsplit - 1) / nsplit;
    nps = (nps + 63) / 64 * 64;                                // whole 64-row steps (4 waves x 16) and 8-float alignment of the slices
    nsplit = (N + nps - 1) / nps;
    if (nsplit > 1) (void)hipMemsetAsync(dx, 0, (size_t)M * lddx * sizeof(float), stream);
    const dim3 grid(ktiles, nsplit);
    const bool vec = N % 8 == 0 && lddy % 4 == 0 && sk_al16(dy) && (out_act == SK_ACT_NONE || (ldy % 4 == 0 && sk_al16(y)));
    void (*kern)(const float*, long, const float*, long, const float*, const float*, long, float*, long, int, int, int, int, int, int) =
        mode == OSUF_DT_BF16 ? (vec ? skinny_dx_kernel<1, true> : skinny_dx_kernel<1, false>) : (vec ? skinny_dx_kernel<0, true> : skinny_dx_kernel<0, false>);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream, dy, lddy, y, ldy, W, x, ldx, dx, lddx, M, N, K, in_act, out_act, nps);
  }
  if (dW) {
    const int tiles = ((K + 31) / 32) * ((N + 31) / 32);
    const dim3 grid((tiles + 3) / 4);
    if (mode == OSUF_DT_BF16)
      hipLaunchKernelGGL(skinny_dw_kernel<1>, grid, dim3(256), 0, stream, dy, lddy, y, ldy, x, ldx, dW, db, M, N, K, in_act, out_act, accumulate);
    else
      hipLaunchKernelGGL(skinny_dw_kernel<0>, grid, dim3(256), 0, stream, dy, lddy, y, ldy, x, ldx, dW, db, M, N, K, in_act, out_act, accumulate);
  }
  return osuf_launch_status();
}

/* Group forms for linears that share one input x (M, K): descs is a DEVICE array of n osuf_linear_desc.
 *   osuf_skinny_fwd_group: y_i = in_act(x) W_i^T + b_i for every i; block0 = running sum of ceil(N_i / 32), total_blocks = its total.
 *   osuf_skinny_dx_group : dx = in_act'(x) * sum_i dy_i W_i (dx is overwritten); block0 = running sum of ceil(N_i / 512).
 * Vector path only: K and every N_i multiples of 8, 16-byte aligned rows (EINVAL otherwise -- use the per-linear entry points). */
static bool sk_group_bad(const void* x, long ldx, int M, int K, int in_act) {
  return !x || M <= 0 || K <= 0 || K % 8 || ldx % 4 || !sk_al16(x) || (in_act != SK_ACT_NONE && in_act != SK_ACT_SILU);
}
extern "C" int osuf_skinny_fwd_group(int mode, const float* x, long ldx, const osuf_linear_desc* descs, int n, int total_blocks, int M, int K,
                                     int in_act, hipStream_t stream) {
  if (sk_group_bad(x, ldx, M, K, in_act) || !descs || n <= 0 || total_blocks <= 0) return OSUF_EINVAL;
  if (mode == OSUF_DT_BF16) hipLaunchKernelGGL(skinny_fwd_group_kernel<1>, dim3(total_blocks), dim3(256), 0, stream, x, ldx, descs, n, M, K, in_act);
  else hipLaunchKernelGGL(skinny_fwd_group_kernel<0>, dim3(total_blocks), dim3(256), 0, stream, x, ldx, descs, n, M, K, in_act);
  return osuf_launch_status();
}
extern "C" int osuf_skinny_dx_group(int mode, const osuf_linear_desc* descs, int n, int total_slices, const float* x, long ldx, float* dx, long lddx,
                                    int M, int K, int in_act, hipStream_t stream) {
  if (sk_group_bad(x, ldx, M, K, in_act) || !descs || n <= 0 || total_slices <= 0 || !dx || lddx < K) return OSUF_EINVAL;
  hipError_t e = hipMemsetAsync(dx, 0, (size_t)M * lddx * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((K + 31) / 32, total_slices);
  if (mode == OSUF_DT_BF16) hipLaunchKernelGGL(skinny_dx_group_kernel<1>, grid, dim3(256), 0, stream, descs, n, x, ldx, dx, lddx, M, K, in_act);
  else hipLaunchKernelGGL(skinny_dx_group_kernel<0>, grid, dim3(256), 0, stream, descs, n, x, ldx, dx, lddx, M, K, in_act);
  return osuf_launch_status();
}
