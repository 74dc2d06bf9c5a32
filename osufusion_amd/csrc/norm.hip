// HBM-bound normalisation / gating kernels (channels-last [B*L][C] activations), gfx950.
//   GroupNorm(1,C) + FiLM + SiLU     residual.py:62-88   (stats come from the conv GEMM epilogue)
//   LayerNorm(C)                      unet.py:117,127
//   GlobalContext pooling + gate      residual.py:14-37,135
// Two skeletons: "column reduce" (a workgroup owns a row chunk of ONE sample, threads own 8-channel
// chunks, partials go out as fp32 atomics to [B][C]) and "row-wise" (G lanes of a wave own one row,
// wave64 xor-shuffle reductions).  All loads/stores are 16 B per lane.
#include "common.hpp"

struct ColGeom {
  int cp, rp;          // chunk lanes, row lanes (cp * rp <= 256)
};
__device__ __forceinline__ ColGeom col_geom(int chunks) {
  ColGeom g;
  g.cp = chunks;
  g.rp = 256 / chunks;
  return g;
}

static constexpr float kEps = 1e-5f;

// ------------------------------------------------------------------------------------------------
// GroupNorm(1, C)
// ------------------------------------------------------------------------------------------------
__global__ void gn_finalize_kernel(const double* stats, float* mr, int B, double inv_count) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double mean = stats[2 * b] * inv_count;
  double var = stats[2 * b + 1] * inv_count - mean * mean;      // biased variance (torch semantics)
  if (var < 0.0) var = 0.0;
  mr[2 * b] = (float)mean;
  mr[2 * b + 1] = (float)(1.0 / sqrt(var + (double)kEps));
}

// Statistics without atomics (osuf_gn_stats; the sampler's bit-reproducible path): stage 1, grid (row chunks, samples) -- every
// thread sums its elements in a fixed order, lanes / waves are combined by shuffles and a fixed-order LDS pass, and the chunk's
// (sum, sum of squares) goes to part[b][chunk][2]; stage 2 adds a sample's chunks in index order.  Same inputs, same bits.
__device__ __forceinline__ double group_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* y, long ldy, double* part, int C, int L, int rows_per_block, int nchunk) {
  const int chunks = C >> 3;
  const int b = blockIdx.y;
  const ColGeom cg = col_geom(chunks);
  const int ch = threadIdx.x % cg.cp, rl = threadIdx.x / cg.cp;
  const int n_begin = blockIdx.x * rows_per_block, n_end = min(L, n_begin + rows_per_block);
  float s1 = 0.f, s2 = 0.f;
  if (rl < cg.rp) {
    for (int n = n_begin + rl; n < n_end; n += cg.rp) {
      float v[8];
      load8(y + ((long)b * L + n) * ldy + ch * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
    }
  }
  const double d1 = group_sum_f64((double)s1), d2 = group_sum_f64((double)s2);
  __shared__ double red[4][2];
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = d1; red[threadIdx.x >> 6][1] = d2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int w = threadIdx.x;
    part[((long)b * nchunk + blockIdx.x) * 2 + w] = ((red[0][w] + red[1][w]) + red[2][w]) + red[3][w];
  }
}
__global__ void gn_finalize_parts_kernel(const double* part, float* mr, int B, int nchunk, double inv_count) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < nchunk; ++i) { s1 += part[((long)b * nchunk + i) * 2]; s2 += part[((long)b * nchunk + i) * 2 + 1]; }
  double mean = s1 * inv_count;
  double var = s2 * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  mr[2 * b] = (float)mean;
  mr[2 * b + 1] = (float)(1.0 / sqrt(var + (double)kEps));
}

// h = silu( ((y-mean)*rstd*gamma + beta) * (1+scale) + shift );  ss = [B][2C] (scale | shift) or null
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const T* y, long ldy, T* h, long ldh, const float* mr,
                                                           const float* gamma, const float* beta, const float* ss,
                                                           int M, int C, int L, int rows_per_block,
                                                           const double* stats, double inv_count, float* mr_out, int nparts) {
  // grid (row blocks of one sample, sample): a thread keeps ONE 8-channel chunk and walks rows, so gamma / beta / scale / shift /
  // mean / rstd are loaded once per thread and no index division runs per element (the flat idx / chunks, m / L form spent more
  // VALU on 64-bit divisions and operand reloads than on the normalisation itself)
  const int chunks = C >> 3;
  const int b = blockIdx.y;
  const ColGeom cg = col_geom(chunks);
  const int ch = threadIdx.x % cg.cp, rl = threadIdx.x / cg.cp;
  float mean, rstd;
  if (stats) {                                             // (sum, sum of squares) of the sample, finalised here (what osuf_gn_finalize computes):
    double t1, t2;
    if (nparts > 0) {
      // osuf_gn_apply_fwd_parts (the sampler's bit-reproducible path): stats = osuf_gn_stats_parts' [B][nparts][2] partial sums.  Every wave adds
      // them in the same fixed order (lane-strided, then the xor butterfly), so all waves of all workgroups hold the same bits -- this replaces
      // gn_finalize_parts_kernel's one-thread-per-sample serial loop (11 us per launch, 115 launches per DDIM step).  Before the early return
      // below: the butterfly needs whole waves.
      const int lane = threadIdx.x & 63;
      double a1 = 0.0, a2 = 0.0;
      for (int i = lane; i < nparts; i += 64) { a1 += stats[((long)b * nparts + i) * 2]; a2 += stats[((long)b * nparts + i) * 2 + 1]; }
      t1 = group_sum_f64(a1); t2 = group_sum_f64(a2);
    } else {                                               // ... as the producing GEMM's epilogue left them (102 tiny launches per step less)
      t1 = stats[2 * b]; t2 = stats[2 * b + 1];
    }
    const double m1 = t1 * inv_count;
    double var = t2 * inv_count - m1 * m1;
    if (var < 0.0) var = 0.0;
    mean = (float)m1;
    rstd = (float)(1.0 / sqrt(var + (double)kEps));
    if (blockIdx.x == 0 && threadIdx.x == 0) { mr_out[2 * b] = mean; mr_out[2 * b + 1] = rstd; }   // kept for the backward
  } else {
    mean = mr[2 * b]; rstd = mr[2 * b + 1];
  }
  if (rl >= cg.rp) return;
  const int c = ch * 8;
  float g[8], bt[8], k[8], sh[8];
  load8(gamma + c, g);
  load8(beta + c, bt);
#pragma unroll
  for (int e = 0; e < 8; ++e) { k[e] = 1.f; sh[e] = 0.f; }
  if (ss) {
    float sc[8];
    load8(ss + (long)b * 2 * C + c, sc);
    load8(ss + (long)b * 2 * C + C + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) k[e] = 1.f + sc[e];
  }
  // out = silu(((v - mean) * rstd * g + bt) * k + sh) = silu(v * A + Bc)
  float A[8], Bc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { A[e] = rstd * g[e] * k[e]; Bc[e] = (bt[e] - mean * rstd * g[e]) * k[e] + sh[e]; }
  const int n_end = min(L, (int)(blockIdx.x + 1) * rows_per_block);
  for (int n = blockIdx.x * rows_per_block + rl; n < n_end; n += cg.rp) {
    const long m = (long)b * L + n;
    float v[8];
    load8(y + m * ldy + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(fmaf(v[e], A[e], Bc[e]));
    store8(h + m * ldh + c, v);
  }
}

// column-reduce geometry shared by several kernels

// T1[b][c] = sum_n du * xhat, T2[b][c] = sum_n du   with du = dh * silu'(u)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* dh, long lddh, const T* y, long ldy, const float* mr,
                                                            const float* gamma, const float* beta, const float* ss,
                                                            float* T12, int C, int L, int rows_per_block) {
  const int chunks = C >> 3;
  const int b = blockIdx.y;
  const ColGeom cg = col_geom(chunks);
  const int ch = threadIdx.x % cg.cp, rl = threadIdx.x / cg.cp;
  const int c = ch * 8;
  const int n_begin = blockIdx.x * rows_per_block, n_end = min(L, n_begin + rows_per_block);
  float t1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < cg.rp) {
    const float mean = mr[2 * b], rstd = mr[2 * b + 1];
    float g[8], bt[8], sc[8], sh[8];
    load8(gamma + c, g);
    load8(beta + c, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
    if (ss) { load8(ss + (long)b * 2 * C + c, sc); load8(ss + (long)b * 2 * C + C + c, sh); }
    // four row steps per trip, their eight 16-byte loads issued together (rows clamped, the surplus masked out below): one step
    // per trip kept 8 KB in flight per CU at the deep levels (C = 1024: 2 rows per step, 256 workgroups) and ran at 1.5 TB/s
    for (int n0 = n_begin + rl; n0 < n_end; n0 += 4 * cg.rp) {
      float v[4][8], d[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long m = (long)b * L + min(n0 + q * cg.rp, n_end - 1);
        load8(y + m * ldy + c, v[q]);
        load8(dh + m * lddh + c, d[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = n0 + q * cg.rp < n_end;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xh = (v[q][e] - mean) * rstd;
          float u = (xh * g[e] + bt[e]) * (1.f + sc[e]) + sh[e];
          float du = ok ? d[q][e] * silu_grad_f(u) : 0.f;
          xh = ok ? xh : 0.f;
          t1[e] += du * xh;
          t2[e] += du;
          t3[e] += xh;
          t4[e] += xh * xh;
        }
      }
    }
  }
  // reduce over row lanes through LDS, then one atomic per (b, c)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);            // [4][rp][C]
  if (rl < cg.rp) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * cg.rp + rl) * C + c + e] = t1[e]; red[(1 * cg.rp + rl) * C + c + e] = t2[e]; red[(2 * cg.rp + rl) * C + c + e] = t3[e];
      red[(3 * cg.rp + rl) * C + c + e] = t4[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C; i += blockDim.x) {
    const int w = i / C, cc = i - w * C;
    float s = 0.f;
    for (int r = 0; r < cg.rp; ++r) s += red[(w * cg.rp + r) * C + cc];
    atomic_add_f32(T12 + ((long)b * 4 + w) * C + cc, s);
  }
}

// per sample: S1 = sum_c gamma*(1+scale)*T2, S2 = sum_c gamma*(1+scale)*T1 ; dscale, dshift ; dgamma, dbeta (atomics over b);
// and, when dbias is given, the gradient of the bias of the convolution that produced y -- the column sums of dy -- in closed
// form from the per-(b, c) sums (T3 = sum_l xhat), instead of a pass over the dy tensor the apply kernel is about to write:
//   sum_l dy[b,l,c] = rstd_b * ( gamma_c k_bc T2[b,c] - L * S1_b/cnt - (S2_b/cnt) * T3[b,c] )
// and (dyy, for the DoRA magnitude gradient, lora_layers.py:86-90) sum_l dy*y with y = xhat/rstd + mean, T4 = sum_l xhat^2:
//   sum_l dy[b,l,c] y[b,l,c] = ( gamma_c k_bc T1 - a1 T3 - a2 T4 ) + mean_b * sum_l dy[b,l,c]
// Round 5: no longer a launch of its own (102 six-microsecond launches per train step) -- EVERY workgroup of gn_bwd_apply_kernel evaluates the two sums of
// its sample (<= 4 x 1,024 floats from L2, the same order as before: bit-identical S1 / S2) and the first workgroup of each sample (`side`) writes the
// side outputs.  Returns (S1, S2) to every thread; contains the block's only barrier.
__device__ __forceinline__ void gn_bwd_sums(const float* T12, const float* gamma, const float* beta, const float* ss, float* S, float* dss, float* dgamma,
                                            float* dbeta, float* dbias, float* dyy, const float* mr, int b, int C, int L, float inv_count, bool side,
                                            float& S1_out, float& S2_out) {
  float s1 = 0.f, s2 = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float t1 = T12[((long)b * 4 + 0) * C + c], t2 = T12[((long)b * 4 + 1) * C + c];
    const float g = gamma[c], bt = beta[c];
    const float k = ss ? 1.f + ss[(long)b * 2 * C + c] : 1.f;
    s1 += g * k * t2;
    s2 += g * k * t1;
    if (side) {
      if (dss) { dss[(long)b * 2 * C + c] = g * t1 + bt * t2; dss[(long)b * 2 * C + C + c] = t2; }
      if (dgamma) {
        atomic_add_f32(dgamma + c, k * t1);
        atomic_add_f32(dbeta + c, k * t2);
      }
    }
  }
  __shared__ float r1[4], r2[4];
  s1 = group_sum<64>(s1);
  s2 = group_sum<64>(s2);
  if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
  __syncthreads();
  // dgamma == NULL: y was not normalised at all (Block(norm=False), residual.py:71; the caller passes mean 0, rstd 1, gamma 1, beta 0):
  // no statistics, hence none of their gradient terms
  const float S1 = dgamma ? r1[0] + r1[1] + r1[2] + r1[3] : 0.f, S2 = dgamma ? r2[0] + r2[1] + r2[2] + r2[3] : 0.f;
  S1_out = S1; S2_out = S2;
  if (!side) return;
  if (threadIdx.x == 0) { S[2 * b] = S1; S[2 * b + 1] = S2; }
  if (dbias) {
    const float mean = mr[2 * b], rstd = mr[2 * b + 1];
    const float a1 = S1 * inv_count, a2 = S2 * inv_count;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float t1 = T12[((long)b * 4 + 0) * C + c], t2 = T12[((long)b * 4 + 1) * C + c], t3 = T12[((long)b * 4 + 2) * C + c];
      const float gk = gamma[c] * (ss ? 1.f + ss[(long)b * 2 * C + c] : 1.f);
      const float sdy = rstd * (gk * t2 - (float)L * a1 - a2 * t3);
      atomic_add_f32(dbias + c, sdy);
      if (dyy) atomic_add_f32(dyy + c, (gk * t1 - a1 * t3 - a2 * T12[((long)b * 4 + 3) * C + c]) + mean * sdy);
    }
  }
}

// dy = rstd * ( gamma*(1+scale)*du - S1/cnt - xhat*S2/cnt )        same (row block, sample) grid as the forward apply
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* dh, long lddh, const T* y, long ldy, T* dy, long lddy,
                                                           const float* mr, const float* gamma, const float* beta, const float* ss,
                                                           const float* T12, float* S, float* dss, float* dgamma, float* dbeta, float* dbias, float* dyy,
                                                           int M, int C, int L, float inv_count, int rows_per_block) {
  const int chunks = C >> 3;
  const int b = blockIdx.y;
  const ColGeom cg = col_geom(chunks);
  const int ch = threadIdx.x % cg.cp, rl = threadIdx.x / cg.cp;
  float S1, S2;
  gn_bwd_sums(T12, gamma, beta, ss, S, dss, dgamma, dbeta, dbias, dyy, mr, b, C, L, inv_count, blockIdx.x == 0, S1, S2);
  if (rl >= cg.rp) return;
  const int c = ch * 8;
  const float mean = mr[2 * b], rstd = mr[2 * b + 1];
  const float a1 = S1 * inv_count, a2 = S2 * inv_count;
  float g[8], bt[8], k[8], sh[8];
  load8(gamma + c, g);
  load8(beta + c, bt);
#pragma unroll
  for (int e = 0; e < 8; ++e) { k[e] = 1.f; sh[e] = 0.f; }
  if (ss) {
    float sc[8];
    load8(ss + (long)b * 2 * C + c, sc);
    load8(ss + (long)b * 2 * C + C + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) k[e] = 1.f + sc[e];
  }
  const int n_end = min(L, (int)(blockIdx.x + 1) * rows_per_block);
  for (int n = blockIdx.x * rows_per_block + rl; n < n_end; n += cg.rp) {
    const long m = (long)b * L + n;
    float v[8], d[8];
    load8(y + m * ldy + c, v);
    load8(dh + m * lddh + c, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - mean) * rstd;
      const float u = (xh * g[e] + bt[e]) * k[e] + sh[e];
      const float du = d[e] * silu_grad_f(u);
      v[e] = rstd * (g[e] * k[e] * du - a1 - xh * a2);
    }
    store8(dy + m * lddy + c, v);
  }
}

// ------------------------------------------------------------------------------------------------
// row-wise skeleton: G lanes per row (G power of two <= 64), up to 4 chunks of 8 channels per lane
// ------------------------------------------------------------------------------------------------
static constexpr int kMaxCh = 4;

template <typename T, int NCH>                                // NCH: chunks of 8 channels per lane, as in ln_bwd_kernel (86 -> 8 waves per SIMD at NCH = 1)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, long ldx, T* out, long ldo, float* mr, const float* gamma,
                                                     const float* beta, int M, int C, int G) {
  const int chunks = C >> 3;
  const int rows_per_wave = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane % G, gr = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const float invC = 1.f / (float)C;
  for (long m0 = (long)wave_global * rows_per_wave; m0 < M; m0 += (long)waves_total * rows_per_wave) {
    const long m = m0 + gr;
    const bool rok = m < M;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = gl + j * G;
      if (rok && ch < chunks) {
        load8(x + m * ldx + ch * 8, v[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[j][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
      }
    }
    s = group_sum_dyn(s, G);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = gl + j * G;
      if (ch < chunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = v[j][e] - mean; q += d * d; }
      }
    }
    q = group_sum_dyn(q, G);
    const float rstd = rsqrtf(q * invC + kEps);
    if (rok) {
      if (gl == 0 && mr) { mr[2 * m] = mean; mr[2 * m + 1] = rstd; }
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int ch = gl + j * G;
        if (ch < chunks) {
          float g[8], bt[8], o[8];
          load8(gamma + ch * 8, g);
          load8(beta + ch * 8, bt);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + bt[e];
          store8(out + m * ldo + ch * 8, o);
        }
      }
    }
  }
}

// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat));  dgamma += sum_m dy*xhat; dbeta += sum_m dy
// NCH = chunks of 8 channels per lane (1, 2 or 4): sized by the template, not by NCH -- with the arrays of four chunks the C = 256 LayerNorms of the
// UNet (one chunk per lane) ran at 176 VGPRs = two waves per SIMD with one pair of rows in flight per wave, i.e. at the latency of its loads (57.7 us for
// 201 MB); at NCH = 1 the kernel fits eight waves per SIMD
// NT = threads per block: the per-block dgamma / dbeta atomics all land on the same 2 C addresses and cost ~18 ns per block (measured: 512 / 1024 / 2048 /
// 4096 blocks of 256 threads 64 / 57 / 75 / 113 us at M = 131,072, C = 256), so the waves that hide the load latency come from FEWER, larger blocks
template <typename T, int NCH, int NT = 256>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const T* dy, long lddy, const T* x, long ldx, T* dx, long lddx, const float* mr,
                                                     const float* gamma, float* dgamma, float* dbeta, int M, int C, int G) {
  const int chunks = C >> 3;
  const int rows_per_wave = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane % G, gr = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const float invC = 1.f / (float)C;
  float ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; }
  for (long m0 = (long)wave_global * rows_per_wave; m0 < M; m0 += (long)waves_total * rows_per_wave) {
    const long m = m0 + gr;
    const bool rok = m < M;
    const float mean = rok ? mr[2 * m] : 0.f, rstd = rok ? mr[2 * m + 1] : 0.f;
    float xh[NCH][8], gd[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = gl + j * G;
      if (rok && ch < chunks) {
        float xv[8], dv[8], g[8];
        load8(x + m * ldx + ch * 8, xv);
        load8(dy + m * lddy + ch * 8, dv);
        load8(gamma + ch * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[j][e] = (xv[e] - mean) * rstd;
          gd[j][e] = g[e] * dv[e];
          s1 += gd[j][e];
          s2 += gd[j][e] * xh[j][e];
          ag[j][e] += dv[e] * xh[j][e];
          ab[j][e] += dv[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[j][e] = 0.f; gd[j][e] = 0.f; }
      }
    }
    s1 = group_sum_dyn(s1, G) * invC;
    s2 = group_sum_dyn(s2, G) * invC;
    if (rok) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int ch = gl + j * G;
        if (ch < chunks) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rstd * (gd[j][e] - s1 - xh[j][e] * s2);
          store8(dx + m * lddx + ch * 8, o);
        }
      }
    }
  }
  // reduce dgamma/dbeta partials: lanes with equal gl across the wave (xor over the row-group bits), then across the
  // block's 4 waves through LDS, then ONE atomic per channel per block (the grid is capped at 256 blocks by the host)
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) { ag[j][e] += __shfl_xor(ag[j][e], o, 64); ab[j][e] += __shfl_xor(ab[j][e], o, 64); }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);                 // [NT / 64 waves][2][C]
  const int wave = threadIdx.x >> 6;
  if (gr == 0) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = gl + j * G;
      if (ch < chunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[(wave * 2 + 0) * C + ch * 8 + e] = ag[j][e]; red[(wave * 2 + 1) * C + ch * 8 + e] = ab[j][e]; }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    const int w = i / C, cc = i - w * C;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) s += red[(k * 2 + w) * C + cc];
    atomic_add_f32((w == 0 ? dgamma : dbeta) + cc, s);
  }
}

// ------------------------------------------------------------------------------------------------
// GlobalContext
// ------------------------------------------------------------------------------------------------
// rowdot[m] = h[m,:] . w[b or 0][:] + bias        (w_stride = 0: shared vector (to_k); = C: per-sample vector)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* h, long ldh, const float* w, long w_stride, const float* bias,
                                                     float* out, int M, int C, int L, int G) {
  const int chunks = C >> 3;
  const int rows_per_wave = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane % G, gr = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  const float bv = bias ? bias[0] : 0.f;
  for (long m0 = (long)wave_global * rows_per_wave; m0 < M; m0 += (long)waves_total * rows_per_wave) {
    const long m = m0 + gr;
    const bool rok = m < M;
    const long b = rok ? m / L : 0;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxCh; ++j) {
      const int ch = gl + j * G;
      if (rok && ch < chunks) {
        float v[8], wv[8];
        load8(h + m * ldh + ch * 8, v);
        load8(w + b * w_stride + ch * 8, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e] * wv[e];
      }
    }
    s = group_sum_dyn(s, G);
    if (rok && gl == 0) out[m] = s + bv;
  }
}

// GlobalContext pooling in ONE pass over h (round 5; residual.py:29-31: logits = to_k(h), softmax over the sequence, pooled = sum_n p[n] h[n]):
// the row's logit, a running softmax (per lane: running maximum m, sum l, the lane's weighted column sums relative to m) and the block's partial
// (acc[C], m, l) in part[b][blk][C + 2]; the raw logits are kept in logit[B*L] for the second stage's probabilities.  No atomics: the second
// stage adds the blocks in order, so the result is bit-reproducible (what the sampler needs) -- and h is read ONCE where rowdot + wcolsum read it twice.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void gca_pool_kernel(const T* h, long ldh, const float* wk, const float* bk, float* part, float* logit, int C, int L,
                                                       int G, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);                 // [4 waves][C + 2]
  const int chunks = C >> 3;
  const int rows_per_wave = 64 / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane % G, gr = lane / G;
  const int b = blockIdx.y;
  const int n_begin = blockIdx.x * rows_per_block, n_end = min(L, n_begin + rows_per_block);
  const float bv = bk ? bk[0] : 0.f;
  float w[NCH][8], acc[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int ch = gl + j * G;
#pragma unroll
    for (int e = 0; e < 8; ++e) { w[j][e] = 0.f; acc[j][e] = 0.f; }
    if (ch < chunks) load8(wk + ch * 8, w[j]);
  }
  float m = -INFINITY, l = 0.f;
  for (int n0 = n_begin + wave * rows_per_wave; n0 < n_end; n0 += 4 * rows_per_wave) {
    const int n = n0 + gr;
    const bool rok = n < n_end;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = gl + j * G;
      if (rok && ch < chunks) {
        load8(h + ((long)b * L + n) * ldh + ch * 8, v[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[j][e] * w[j][e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
      }
    }
    s = group_sum_dyn(s, G) + bv;                             // every lane of the row's group holds the logit
    if (rok) {
      if (gl == 0) logit[(long)b * L + n] = s;
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn), pr = __expf(s - mn);   // first row: m = -inf, alpha = 0
      l = l * alpha + pr;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = acc[j][e] * alpha + pr * v[j][e];
      m = mn;
    }
  }
  // the row groups of a wave (same gl, other gr): bring them to a common maximum and add
  for (int o = G; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o, 64), lo_ = __shfl_xor(l, o, 64);
    const float M = fmaxf(m, mo);
    const float fa = m == -INFINITY ? 0.f : __expf(m - M), fb = mo == -INFINITY ? 0.f : __expf(mo - M);
    l = l * fa + lo_ * fb;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float ao = __shfl_xor(acc[j][e], o, 64); acc[j][e] = acc[j][e] * fa + ao * fb; }
    m = M;
  }
  // the four waves through LDS, in wave order
  if (gr == 0) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = gl + j * G;
      if (ch < chunks) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * (C + 2) + ch * 8 + e] = acc[j][e];
      }
    }
    if (gl == 0) { red[wave * (C + 2) + C] = m; red[wave * (C + 2) + C + 1] = l; }
  }
  __syncthreads();
  float mw[4], f[4];
  float M = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) { mw[k] = red[k * (C + 2) + C]; M = fmaxf(M, mw[k]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = mw[k] == -INFINITY ? 0.f : __expf(mw[k] - M);
  float* dst = part + ((long)b * gridDim.x + blockIdx.x) * (C + 2);
  for (int i = threadIdx.x; i < C; i += 256)
    dst[i] = ((red[i] * f[0] + red[(C + 2) + i] * f[1]) + red[2 * (C + 2) + i] * f[2]) + red[3 * (C + 2) + i] * f[3];
  if (threadIdx.x == 0) {
    dst[C] = M;
    dst[C + 1] = ((red[C + 1] * f[0] + red[(C + 2) + C + 1] * f[1]) + red[2 * (C + 2) + C + 1] * f[2]) + red[3 * (C + 2) + C + 1] * f[3];
  }
}

// second stage, one workgroup per sample: the blocks' partials in block order -> pooled[b][C]; the probabilities p[b][n] = exp(logit - M) / Z in place
// of the logits (the backward's operand)
__global__ __launch_bounds__(256) void gca_pool_finish_kernel(const float* part, float* logit_p, float* pooled, int C, int L, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* fb = reinterpret_cast<float*>(smem);                  // [nblk] block factors exp(m_blk - M)
  __shared__ float redm[4], redz[4];
  const int b = blockIdx.x;
  const float* pb = part + (long)b * nblk * (C + 2);
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < nblk; k += 256) mx = fmaxf(mx, pb[(long)k * (C + 2) + C]);
  mx = group_max<64>(mx);
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
  __syncthreads();
  const float M = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  for (int k = threadIdx.x; k < nblk; k += 256) fb[k] = __expf(pb[(long)k * (C + 2) + C] - M);
  __syncthreads();
  // Z: four threads add a quarter of the blocks each, in block order, then the four sums in order: a fixed order for any nblk
  float z = 0.f;
  if (threadIdx.x < 4) {
    const int per = (nblk + 3) / 4;
    for (int k = threadIdx.x * per; k < min(nblk, (threadIdx.x + 1) * per); ++k) z += pb[(long)k * (C + 2) + C + 1] * fb[k];
    redz[threadIdx.x] = z;
  }
  __syncthreads();
  const float Z = ((redz[0] + redz[1]) + redz[2]) + redz[3];
  const float inv = 1.f / Z;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int k = 0; k < nblk; ++k) a += pb[(long)k * (C + 2) + c] * fb[k];
    pooled[(long)b * C + c] = a * inv;
  }
  float* row = logit_p + (long)b * L;
  for (int n = threadIdx.x; n < L; n += 256) row[n] = __expf(row[n] - M) * inv;
}

// in-place softmax over the L logits of each sample (one workgroup per sample)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* p, int L) {
  float* row = p + (long)blockIdx.x * L;
  __shared__ float red[4];
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < L; i += 256) mx = fmaxf(mx, row[i]);
  mx = group_max<64>(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) { float e = __expf(row[i] - mx); row[i] = e; s += e; }
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = threadIdx.x; i < L; i += 256) row[i] *= inv;
}

// out[b][c] += sum_n w[b*L+n] * a[n][c] * (bmul ? bmul[n][c] : 1)      (w may be null -> 1)
//   pooled   = wsum(p, h)          dgate = wsum(null, dout*h)          dwk_b = wsum(dlogit, h)
template <typename T>
__global__ __launch_bounds__(256) void wcolsum_kernel(const T* a, long lda, const T* bmul, long ldb, const float* w, float* out,
                                                      int C, int L, int rows_per_block, float* part) {
  const int chunks = C >> 3;
  const int b = blockIdx.y;
  const ColGeom cg = col_geom(chunks);
  const int ch = threadIdx.x % cg.cp, rl = threadIdx.x / cg.cp;
  const int c = ch * 8;
  const int n_begin = blockIdx.x * rows_per_block, n_end = min(L, n_begin + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < cg.rp) {
    for (int n0 = n_begin + rl; n0 < n_end; n0 += 4 * cg.rp) {       // four row steps per trip, loads first (see gn_bwd_reduce_kernel)
      float v[4][8], u[4][8], wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long m = (long)b * L + min(n0 + q * cg.rp, n_end - 1);
        load8(a + m * lda + c, v[q]);
        if (bmul) load8(bmul + m * ldb + c, u[q]);                     // (uniform conditions: no exec mask)
        wv[q] = w ? w[m] : 1.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float wq = n0 + q * cg.rp < n_end ? wv[q] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wq * (bmul ? v[q][e] * u[q][e] : v[q][e]);
      }
    }
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);            // [rp][C]
  if (rl < cg.rp) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl * C + c + e] = acc[e];
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < C; cc += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < cg.rp; ++r) s += red[r * C + cc];
    if (part) part[((long)b * gridDim.x + blockIdx.x) * C + cc] = s;      // reproducible path: summed in chunk order below
    else atomic_add_f32(out + (long)b * C + cc, s);
  }
}
// out[b][c] = sum over a sample's row chunks, in chunk order, of the wcolsum partials (no atomics)
__global__ __launch_bounds__(256) void colsum_parts_kernel(const float* part, float* out, int C, int nchunk) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < nchunk; ++i) s += part[((long)b * nchunk + i) * C + c];
  out[(long)b * C + c] = s;
}

// out = h * gate[b][:] + res        (res may be null: out = h * gate)
template <typename T>
__global__ __launch_bounds__(256) void gate_residual_kernel(const T* h, long ldh, const float* gate, const T* res, long ldr,
                                                            T* out, long ldo, int M, int C, int L) {
  const int chunks = C >> 3;
  const long total = (long)M * chunks;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / chunks), c = (int)(idx - (long)m * chunks) * 8;
    const int b = m / L;
    float v[8], g[8], r[8];
    load8(h + (long)m * ldh + c, v);
    load8(gate + (long)b * C + c, g);
    if (res) {
      load8(res + (long)m * ldr + c, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * g[e] + r[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= g[e];
    }
    store8(out + (long)m * ldo + c, v);
  }
}

// GlobalContext backward, row-wise:
//   dp = h[m,:].dpooled[b,:] ; dlogit = p[m] * (dp - sdot[b]) ; dh = dout*gate[b] + p[m]*dpooled[b] + dlogit*wk
// dwk / dbk (optional, accumulated into): the gradients of the pooling logits' weight and bias, sum_m dlogit[m] * h[m][:] and
// sum_m dlogit[m] -- taken from the h row this kernel already holds in registers (round 1 re-read h in a wcolsum pass and
// summed dlogit with two torch reductions: 3 launches and one pass over h per block)
// NCH = channel chunks (8 elements) per lane = ceil(C/8 / G), a template parameter so that the row loop has no branch at all: the
// h and dout rows of an iteration are fetched together up front (row and chunk indices clamped, surplus lanes masked by selects).
// The first version loaded h under `if (row ok && chunk ok)`, reduced, then loaded dout under a second such branch: two exposed
// memory round trips per pair of rows, and with the dwk accumulators (fewer waves per CU) 51 us per launch for 200 MB.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void gca_bwd_apply_kernel(const T* dout, long lddo, const T* h, long ldh, T* dh, long lddh,
                                                            const float* p, const float* gate, const float* dpooled,
                                                            const float* sdot, const float* wk, float* dlogit,
                                                            int M, int C, int L, int G, float* dwk, float* dbk, float* part) {
  const int chunks = C >> 3;
  const int rows_per_wave = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane % G, gr = lane / G;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int waves_total = (gridDim.x * blockDim.x) >> 6;
  float aw[NCH][8], wv8[NCH][8], ab = 0.f;
  int cofs[NCH];
  bool cok[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    cok[j] = gl + j * G < chunks;
    cofs[j] = min(gl + j * G, chunks - 1) * 8;
    load8(wk + cofs[j], wv8[j]);
#pragma unroll
    for (int e = 0; e < 8; ++e) aw[j][e] = 0.f;
  }
  for (long m0 = (long)wave_global * rows_per_wave; m0 < M; m0 += (long)waves_total * rows_per_wave) {
    const long m = m0 + gr;
    const bool rok = m < M;
    const long mc = rok ? m : M - 1;
    const long b = mc / L;
    float hv[NCH][8], dv[NCH][8], dp[NCH][8], g[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      load8(h + mc * ldh + cofs[j], hv[j]);
      load8(dout + mc * lddo + cofs[j], dv[j]);
      load8(dpooled + b * C + cofs[j], dp[j]);
      load8(gate + b * C + cofs[j], g[j]);
    }
    const float pm = p[mc], sd = sdot[b];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float sj = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sj += hv[j][e] * dp[j][e];
      s += cok[j] ? sj : 0.f;
    }
    s = group_sum_dyn(s, G);
    const float dl = pm * (s - sd);
    if (rok && gl == 0) dlogit[m] = dl;
    ab += (rok && gl == 0) ? dl : 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool ok = rok && cok[j];
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        aw[j][e] += ok ? dl * hv[j][e] : 0.f;
        o[e] = dv[j][e] * g[j][e] + pm * dp[j][e] + dl * wv8[j][e];
      }
      if (ok) store8(dh + m * lddh + cofs[j], o);
    }
  }
  if (dwk) {                                               // uniform: rows of a wave (lanes with equal gl) -> waves of the block -> atomics
    extern __shared__ __attribute__((aligned(16))) char smem_gca[];
    float* red = reinterpret_cast<float*>(smem_gca);       // [4][C] | [4]
    for (int off = G; off < 64; off <<= 1) {
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) aw[j][e] += __shfl_xor(aw[j][e], off, 64);
      ab += __shfl_xor(ab, off, 64);
    }
    const int wv = threadIdx.x >> 6;
    if (gr == 0) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        if (cok[j]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) red[wv * C + cofs[j] + e] = aw[j][e];
        }
      }
      if (gl == 0) red[4 * C + wv] = ab;
    }
    __syncthreads();
    // every workgroup holds a partial of the SAME C (+1) sums: added straight into dwk, a few thousand adders queue on each address
    // (~12 ns per atomic and address: +25-40 us per launch, measured) -- so the partials go to a [blocks][C + 1] slab (plain stores)
    // and gca_dwk_reduce_kernel sums them; the atomic form remains for callers without a workspace
    for (int cc = threadIdx.x; cc < C; cc += blockDim.x) {
      const float v = (red[cc] + red[C + cc]) + (red[2 * C + cc] + red[3 * C + cc]);
      if (part) part[(long)blockIdx.x * (C + 1) + cc] = v; else atomic_add_f32(dwk + cc, v);
    }
    if (threadIdx.x == 0) {
      const float v = (red[4 * C] + red[4 * C + 1]) + (red[4 * C + 2] + red[4 * C + 3]);
      if (part) part[(long)blockIdx.x * (C + 1) + C] = v; else if (dbk) atomic_add_f32(dbk, v);
    }
  }
}

// dwk[c] += sum_i part[i][c], dbk += sum_i part[i][C]: grid (ceil((C+1)/64), 16); a workgroup = 64 columns x 4 row lanes over its
// sixteenth of the rows, eight loads in flight per lane; the 16 row slices meet by atomics (16 adders per address)
__global__ __launch_bounds__(256) void gca_dwk_reduce_kernel(const float* __restrict__ part, int nrows, int C, float* dwk, float* dbk) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int per = (nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  const int cc = min(col, C);
  float s = 0.f;
  for (int r = r0 + rl; r < r1; r += 32) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = part[(long)min(r + 4 * q, r1 - 1) * (C + 1) + cc];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += r + 4 * q < r1 ? v[q] : 0.f;
  }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col <= C) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (col < C) atomic_add_f32(dwk + col, t); else if (dbk) atomic_add_f32(dbk, t);
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static inline int pick_group(int chunks) {
  int g = 1;
  while (g < chunks && g < 64) g <<= 1;
  return g;
}
static inline int ew_grid(long total_threads) {
  long blocks = (total_threads + 255) / 256;
  if (blocks > 2048) blocks = 2048;                   // 256 CUs x 8, grid-stride the rest
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
static inline int row_grid(long M, int G) {
  long rows_per_block = 4L * (64 / G);
  long blocks = (M + rows_per_block - 1) / rows_per_block;
  if (blocks > 4096) blocks = 4096;
  return (int)blocks;
}
#define DISPATCH_T(dtype, ...)                                  \
  if ((dtype) == OSUF_DT_BF16) { using T = bf16_t; __VA_ARGS__; } \
  else if ((dtype) == OSUF_DT_F32) { using T = float; __VA_ARGS__; } \
  else return OSUF_EUNSUPPORTED;

static inline bool bad_c(int C) { return C <= 0 || (C & 7) || C > 8 * 256; }

extern "C" int osuf_gn_finalize(const double* stats, float* mr, int B, long count, hipStream_t stream) {
  if (B <= 0 || count <= 0) return OSUF_EINVAL;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, stats, mr, B, 1.0 / (double)count);
  return osuf_launch_status();
}

extern "C" long osuf_gn_stats_workspace_bytes(int M, int C, int L) {
  if (M <= 0 || L <= 0 || M % L || C <= 0 || C % 8 || C > 2048) return 0;
  const int rpb = (256 / (C / 8)) * 8;
  return (long)(M / L) * ((L + rpb - 1) / rpb) * 2 * (long)sizeof(double);
}
// mean / rstd of GroupNorm(1, C) over each sample of y, by two fixed-order reduction stages (bit-reproducible; one extra read of y)
extern "C" int osuf_gn_stats(int dtype, const void* y, long ldy, double* partial, float* mr, int M, int C, int L, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || ldy % 8 || !partial) return OSUF_EINVAL;
  const int rpb = (256 / (C / 8)) * 8, nchunk = (L + rpb - 1) / rpb, B = M / L;
  DISPATCH_T(dtype, hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(nchunk, B), dim3(256), 0, stream, (const T*)y, ldy, partial, C, L, rpb, nchunk));
  hipLaunchKernelGGL(gn_finalize_parts_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, partial, mr, B, nchunk, 1.0 / ((double)L * C));
  return osuf_launch_status();
}

// stage 1 of osuf_gn_stats alone: per-(sample, row chunk) partial sums into partial[B][nparts][2], nparts = osuf_gn_stats_workspace_bytes / (16 B);
// osuf_gn_apply_fwd_parts finishes them inside the apply kernel
extern "C" int osuf_gn_stats_parts(int dtype, const void* y, long ldy, double* partial, int M, int C, int L, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || ldy % 8 || !partial) return OSUF_EINVAL;
  const int rpb = (256 / (C / 8)) * 8, nchunk = (L + rpb - 1) / rpb, B = M / L;
  DISPATCH_T(dtype, hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(nchunk, B), dim3(256), 0, stream, (const T*)y, ldy, partial, C, L, rpb, nchunk));
  return osuf_launch_status();
}

static int gn_apply_fwd_launch(int dtype, const void* y, long ldy, void* h, long ldh, const float* mr, const float* gamma, const float* beta,
                               const float* ss, int M, int C, int L, const double* stats, long count, float* mr_out, hipStream_t stream, int nparts = 0) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || ldy % 8 || ldh % 8) return OSUF_EINVAL;
  {
    const int rp = 256 / (C / 8);
    const int rpb = rp * 8;                                // 8 rows per thread
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_apply_fwd_kernel<T>, dim3((L + rpb - 1) / rpb, M / L), dim3(256), 0, stream,
                                         (const T*)y, ldy, (T*)h, ldh, mr, gamma, beta, ss, M, C, L, rpb, stats,
                                         stats ? 1.0 / (double)count : 0.0, mr_out, nparts));
  }
  return osuf_launch_status();
}
extern "C" int osuf_gn_apply_fwd(int dtype, const void* y, long ldy, void* h, long ldh, const float* mr, const float* gamma,
                                 const float* beta, const float* ss, int M, int C, int L, hipStream_t stream) {
  if (!mr) return OSUF_EINVAL;
  return gn_apply_fwd_launch(dtype, y, ldy, h, ldh, mr, gamma, beta, ss, M, C, L, nullptr, 0, nullptr, stream);
}
/* The same with the statistics still raw: stats[b] = (sum, sum of squares) of sample b over `count` = L * C elements (the GEMM
 * epilogue's output); osuf_gn_finalize's arithmetic runs in the kernel and (mean, rstd) is also written to mr_out[B][2]. */
extern "C" int osuf_gn_apply_fwd_stats(int dtype, const void* y, long ldy, void* h, long ldh, const double* stats, long count, float* mr_out,
                                       const float* gamma, const float* beta, const float* ss, int M, int C, int L, hipStream_t stream) {
  if (!stats || !mr_out || count <= 0) return OSUF_EINVAL;
  return gn_apply_fwd_launch(dtype, y, ldy, h, ldh, nullptr, gamma, beta, ss, M, C, L, stats, count, mr_out, stream);
}
/* The same from osuf_gn_stats_parts' partial sums (fixed summation order: bit-reproducible); (mean, rstd) also to mr_out[B][2]. */
extern "C" int osuf_gn_apply_fwd_parts(int dtype, const void* y, long ldy, void* h, long ldh, const double* partial, float* mr_out,
                                       const float* gamma, const float* beta, const float* ss, int M, int C, int L, hipStream_t stream) {
  if (!partial || !mr_out || bad_c(C) || L <= 0) return OSUF_EINVAL;
  const int rpb = (256 / (C / 8)) * 8, nchunk = (L + rpb - 1) / rpb;
  return gn_apply_fwd_launch(dtype, y, ldy, h, ldh, nullptr, gamma, beta, ss, M, C, L, partial, (long)L * C, mr_out, stream, nchunk);
}

// T1234: [B][4][C] fp32, must be zero on entry.  dss may be null (no FiLM).  dgamma / dbeta / dbias (the latter optional: gradient
// of the bias of the conv feeding this norm = column sums of dy) / dyy (optional, needs dbias: column sums of dy*y) are accumulated into.
extern "C" int osuf_gn_bwd(int dtype, const void* dh, long lddh, const void* y, long ldy, void* dy, long lddy, const float* mr,
                           const float* gamma, const float* beta, const float* ss, float* T123, float* S, float* dss,
                           float* dgamma, float* dbeta, float* dbias, float* dyy, int M, int C, int L, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || ldy % 8 || lddh % 8 || lddy % 8) return OSUF_EINVAL;
  const int B = M / L;
  const int chunks = C / 8;
  const int rp = 256 / chunks;
  const int rows_per_block = 64;                           // (16 / 32 / 128 measured: within +-3 us of 64 at every level shape, 16 much slower)
  if ((dyy && !dbias) || ((dgamma == nullptr) != (dbeta == nullptr))) return OSUF_EINVAL;
  const size_t lds = (size_t)4 * rp * C * sizeof(float);
  const float inv_count = 1.0f / ((float)L * (float)C);
  DISPATCH_T(dtype, hipLaunchKernelGGL(gn_bwd_reduce_kernel<T>, dim3((L + rows_per_block - 1) / rows_per_block, B), dim3(256), lds,
                                       stream, (const T*)dh, lddh, (const T*)y, ldy, mr, gamma, beta, ss, T123, C, L, rows_per_block));
  {
    const int rpb = rp * 8;                                // 8 rows per thread; the per-sample sums + side outputs (the former finalize launch) ride along
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3((L + rpb - 1) / rpb, B), dim3(256), 0, stream, (const T*)dh,
                                         lddh, (const T*)y, ldy, (T*)dy, lddy, mr, gamma, beta, ss, T123, S, dss, dgamma, dbeta, dbias, dyy,
                                         M, C, L, inv_count, rpb));
  }
  return osuf_launch_status();
}

extern "C" int osuf_ln_fwd(int dtype, const void* x, long ldx, void* out, long ldo, float* mr, const float* gamma, const float* beta,
                           int M, int C, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || ldx % 8 || ldo % 8) return OSUF_EINVAL;
  const int G = pick_group(C / 8);
  const int nch = (C / 8 + G - 1) / G;
#define LN_FWD_LAUNCH(NCH_) hipLaunchKernelGGL((ln_fwd_kernel<T, NCH_>), dim3(row_grid(M, G)), dim3(256), 0, stream, (const T*)x, ldx, (T*)out, ldo, mr, gamma, beta, M, C, G)
  DISPATCH_T(dtype, if (nch == 1) LN_FWD_LAUNCH(1); else if (nch == 2) LN_FWD_LAUNCH(2); else LN_FWD_LAUNCH(4));
#undef LN_FWD_LAUNCH
  return osuf_launch_status();
}

extern "C" int osuf_ln_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, void* dx, long lddx, const float* mr,
                           const float* gamma, float* dgamma, float* dbeta, int M, int C, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || ldx % 8 || lddy % 8 || lddx % 8) return OSUF_EINVAL;
  const int G = pick_group(C / 8);
  const int nch = (C / 8 + G - 1) / G;                // chunks per lane: 1 (C <= 512), 2 (<= 1024), 3 or 4 (<= 2048)
  // one chunk per lane: 1,024 threads; two: 512 (the LDS reduction holds NT / 64 x 2 C floats <= 64 KiB); OSUF_LN_BWD_SMALLBLOCKS=1 = 256 everywhere (A/B)
  const int nt = getenv("OSUF_LN_BWD_SMALLBLOCKS") != nullptr ? 256 : nch == 1 ? 1024 : nch == 2 ? 512 : 256;
  long blocks = (M + (nt / 64) * (64 / G) - 1) / ((nt / 64) * (64 / G));
  const char* cap_env = getenv("OSUF_LN_BWD_BLOCKS");
  const long cap = cap_env ? atol(cap_env) : (nt == 256 || (nt == 512 && M >= 32768)) ? 512 : 256;   // bounds the dgamma / dbeta atomics (one per channel and block)
  if (blocks > cap) blocks = cap;
  const size_t lds = (size_t)(nt / 64) * 2 * C * sizeof(float);
#define LN_BWD_LAUNCH(NCH_, NT_) hipLaunchKernelGGL((ln_bwd_kernel<T, NCH_, NT_>), dim3((int)blocks), dim3(NT_), lds, stream, (const T*)dy, lddy, \
                                                    (const T*)x, ldx, (T*)dx, lddx, mr, gamma, dgamma, dbeta, M, C, G)
  DISPATCH_T(dtype, if (nt == 1024) LN_BWD_LAUNCH(1, 1024); else if (nt == 512) LN_BWD_LAUNCH(2, 512); else if (nch == 1) LN_BWD_LAUNCH(1, 256);
                    else if (nch == 2) LN_BWD_LAUNCH(2, 256); else LN_BWD_LAUNCH(4, 256));
#undef LN_BWD_LAUNCH
  return osuf_launch_status();
}

extern "C" int osuf_rowdot(int dtype, const void* h, long ldh, const float* w, long w_stride, const float* bias, float* out,
                           int M, int C, int L, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || ldh % 8 || w_stride % 8) return OSUF_EINVAL;
  const int G = pick_group(C / 8);
  DISPATCH_T(dtype, hipLaunchKernelGGL(rowdot_kernel<T>, dim3(row_grid(M, G)), dim3(256), 0, stream, (const T*)h, ldh, w, w_stride,
                                       bias, out, M, C, L, G));
  return osuf_launch_status();
}

// rows per workgroup: ~1,024 workgroups over the batch, 32 ... 128 rows each (tools/bench_gca.py, B = 32: L = 4096 / 8192 best at 128, 2048 at 64, <= 1024 at 32;
// more rows amortise the end-of-block reduction, fewer keep the chip full); OSUF_GCA_RPB overrides (A/B)
static int gca_pool_rows_per_block(int M) {
  const char* e = getenv("OSUF_GCA_RPB");
  if (e) return atoi(e) > 0 ? atoi(e) : 64;
  int r = 32;
  while (r < 128 && (long)M / (2 * r) >= 1024) r <<= 1;
  return r;
}
extern "C" long osuf_gca_pool_workspace_bytes(int M, int C, int L) {
  if (M <= 0 || L <= 0 || M % L || C <= 0 || C % 8 || C > 2048) return 0;
  const int rpb = 32;                                    // the smallest block the launch may pick: an upper bound
  return (long)(M / L) * ((L + rpb - 1) / rpb) * (C + 2) * (long)sizeof(float);
}
/* GlobalContext pooling in one pass over h: p[B*L] receives softmax_n(h . wk + bk), pooled[B][C] = sum_n p[n] h[n]; `part` =
 * osuf_gca_pool_workspace_bytes(M, C, L) bytes of scratch.  Fixed summation order (bit-reproducible). */
extern "C" int osuf_gca_pool(int dtype, const void* h, long ldh, const float* wk, const float* bk, float* part, float* p, float* pooled,
                             int M, int C, int L, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || ldh % 8 || !part || !p || !pooled || !wk) return OSUF_EINVAL;
  const int B = M / L, G = pick_group(C / 8), nch = (C / 8 + G - 1) / G;
  const int rpb = std::max(32, gca_pool_rows_per_block(M)), nblk = (L + rpb - 1) / rpb;
  const size_t lds = (size_t)4 * (C + 2) * sizeof(float);
  if ((size_t)nblk * sizeof(float) > 60000) return OSUF_EUNSUPPORTED;
#define GCA_POOL_LAUNCH(NCH_) hipLaunchKernelGGL((gca_pool_kernel<T, NCH_>), dim3(nblk, B), dim3(256), lds, stream, (const T*)h, ldh, wk, bk, part, p, C, L, G, rpb)
  DISPATCH_T(dtype, if (nch == 1) GCA_POOL_LAUNCH(1); else if (nch == 2) GCA_POOL_LAUNCH(2); else GCA_POOL_LAUNCH(4));
#undef GCA_POOL_LAUNCH
  hipLaunchKernelGGL(gca_pool_finish_kernel, dim3(B), dim3(256), (size_t)nblk * sizeof(float), stream, part, p, pooled, C, L, nblk);
  return osuf_launch_status();
}

extern "C" int osuf_softmax_rows(float* p, int B, int L, hipStream_t stream) {
  if (B <= 0 || L <= 0) return OSUF_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(B), dim3(256), 0, stream, p, L);
  return osuf_launch_status();
}

// out [B][C] fp32 is accumulated into (zero it first)
// partial (NULL, or B * ceil(L / 64) * C floats): with it the row chunks' sums are stored and added in chunk order by a second
// kernel -- `out` is then overwritten (no zero-init needed) and bit-reproducible; without it they meet by fp32 atomics in `out`
extern "C" int osuf_wcolsum(int dtype, const void* a, long lda, const void* bmul, long ldb, const float* w, float* out,
                            int B, int C, int L, float* partial, hipStream_t stream) {
  if (bad_c(C) || B <= 0 || L <= 0 || lda % 8 || (bmul && ldb % 8)) return OSUF_EINVAL;
  const int rp = 256 / (C / 8);
  const int rows_per_block = 64;
  const int nchunk = (L + rows_per_block - 1) / rows_per_block;
  const size_t lds = (size_t)rp * C * sizeof(float);
  DISPATCH_T(dtype, hipLaunchKernelGGL(wcolsum_kernel<T>, dim3(nchunk, B), dim3(256), lds, stream,
                                       (const T*)a, lda, (const T*)bmul, ldb, w, out, C, L, rows_per_block, partial));
  if (partial) hipLaunchKernelGGL(colsum_parts_kernel, dim3((C + 255) / 256, B), dim3(256), 0, stream, partial, out, C, nchunk);
  return osuf_launch_status();
}

extern "C" int osuf_gate_residual(int dtype, const void* h, long ldh, const float* gate, const void* res, long ldr, void* out, long ldo,
                                  int M, int C, int L, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || ldh % 8 || (res && ldr % 8) || ldo % 8) return OSUF_EINVAL;
  DISPATCH_T(dtype, hipLaunchKernelGGL(gate_residual_kernel<T>, dim3(ew_grid((long)M * (C / 8))), dim3(256), 0, stream, (const T*)h, ldh,
                                       gate, (const T*)res, ldr, (T*)out, ldo, M, C, L));
  return osuf_launch_status();
}

static inline int gca_blocks(int M, int C, bool dwk) {
  int blocks = row_grid(M, pick_group(C / 8));
  if (dwk && blocks > 2048) blocks = 2048;                 // bounds the dwk partials (C + 1 per block)
  return blocks;
}
/* bytes of workspace that let osuf_gca_bwd_apply sum its dwk / dbk partials through a slab instead of same-address atomics */
extern "C" long osuf_gca_bwd_apply_workspace_bytes(int M, int C) {
  if (M <= 0 || bad_c(C)) return 0;
  return (long)gca_blocks(M, C, true) * (C + 1) * (long)sizeof(float);
}

extern "C" int osuf_gca_bwd_apply(int dtype, const void* dout, long lddo, const void* h, long ldh, void* dh, long lddh, const float* p,
                                  const float* gate, const float* dpooled, const float* sdot, const float* wk, float* dlogit,
                                  int M, int C, int L, float* dwk, float* dbk, float* workspace, long workspace_bytes, hipStream_t stream) {
  if (bad_c(C) || M <= 0 || L <= 0 || M % L || lddo % 8 || ldh % 8 || lddh % 8 || (dbk && !dwk)) return OSUF_EINVAL;
  const int G = pick_group(C / 8);
  const size_t lds = dwk ? (size_t)(4 * C + 4) * sizeof(float) : 0;
  const int blocks = gca_blocks(M, C, dwk != nullptr);
  float* part = (dwk && workspace && workspace_bytes >= (long)blocks * (C + 1) * (long)sizeof(float)) ? workspace : nullptr;
  const int nch = (C / 8 + G - 1) / G;                     // 1..4 (C <= 2048, G = 64 from 512 channels on)
#define GCA_LAUNCH(NCH) DISPATCH_T(dtype, hipLaunchKernelGGL((gca_bwd_apply_kernel<T, NCH>), dim3(blocks), dim3(256), lds, stream, \
    (const T*)dout, lddo, (const T*)h, ldh, (T*)dh, lddh, p, gate, dpooled, sdot, wk, dlogit, M, C, L, G, dwk, dbk, part))
  if (nch == 1) { GCA_LAUNCH(1); } else if (nch == 2) { GCA_LAUNCH(2); } else if (nch == 3) { GCA_LAUNCH(3); } else { GCA_LAUNCH(4); }
#undef GCA_LAUNCH
  if (part) hipLaunchKernelGGL(gca_dwk_reduce_kernel, dim3((C + 1 + 63) / 64, 16), dim3(256), 0, stream, part, blocks, C, dwk, dbk);
  return osuf_launch_status();
}
