// MQA attention for head dims other than 64 (attn_dim_head is a free constructor argument of the reference: unet.py:105-123,
// diffusion.py:16-30; SURVEY 8c's tiny configuration uses 16).  The UNet the benchmarks run has 64-wide heads and goes through the
// tuned kernels of attn.hip; these are the same algorithms -- forward, query-stationary dQ, key-stationary dK/dV, as the plain kernels
// there -- written once for a padded head dim DP in {32, 64, 128} (a 16-wide head runs as DP = 32 with zero columns), without the
// software pipelines.  Tiles are [rows][DP] bf16 images, 16-byte chunks xor-swizzled by the row; both row reads (ds_read_b128) and
// column reads (ds_read_b64_tr_b16) recompute the writer's address function, so any DP works by construction.
// Included by attn.hip (it uses AttnArgs, acc_to_frag, fast_exp2, store4, kLog2e).
#pragma once

template <int DP>
struct GenTile {
  static constexpr int RB = 2 * DP, CH = DP / 8, KS = DP / 16, DT = DP / 32;
  __device__ static __forceinline__ int off(int row, int colbyte) {
    const int x = (row >> 1) & 7;
    const int f = (((x & 1) << 2) | (x >> 1)) & (CH - 1);
    return row * RB + ((((colbyte >> 4) ^ f) << 4) | (colbyte & 15));
  }
  // MFMA operand fragment of tile rows: lane (r = lane & 31, h = lane >> 5) holds tile[32 rb + r][16 ks + 8 h .. +7]
  __device__ static __forceinline__ bf16x8 row_frag(const char* tile, int lane, int ks, int rb) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + off(32 * rb + (lane & 31), (2 * ks + (lane >> 5)) * 16)));
  }
  // transposed fragment: element e of lane (r, h) = tile[rowbase + 8 (e >> 2) + 4 h + (e & 3)][32 dt + r] (the k order of an accumulator
  // handed on as the B operand, see acc_to_frag)
  __device__ static __forceinline__ bf16x8 tr_frag(const char* tile, int lane, int rowbase, int dt) {
    const int lh = lane >> 5, cb = ((lane >> 4) & 1) * 16, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
    const int col2 = (dt * 32 + cb + 4 * tp) * 2;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off(rowbase + 4 * lh + tq, col2)));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + off(rowbase + 8 + 4 * lh + tq, col2)));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
  // cooperative fill of a [rows][DP] image from `rows` global rows (row stride ld elements, hd real columns; rows >= limit and the
  // padding columns are zero)
  __device__ static __forceinline__ void fill(char* tile, const bf16_t* src, long ld, int rows, int row0, int limit, int hd, int tid, int nt) {
    for (int cid = tid; cid < rows * CH; cid += nt) {
      const int row = cid / CH, chunk = cid - row * CH;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (row0 + row < limit && chunk * 8 < hd) z = *reinterpret_cast<const u32x4*>(src + (long)(row0 + row) * ld + chunk * 8);
      *reinterpret_cast<u32x4*>(tile + off(row, chunk * 16)) = z;
    }
  }
};

// one query row's fragments straight from global memory (B operand of S^T = K Q^T): lane (r, h) holds q[row][16 ks + 8 h .. +7]
template <int DP>
__device__ __forceinline__ void gen_load_row_frags(bf16x8 (&f)[DP / 16], const bf16_t* row, bool ok, int hd, int lh) {
#pragma unroll
  for (int ks = 0; ks < DP / 16; ++ks) {
    u32x4 z = {0u, 0u, 0u, 0u};
    if (ok && 16 * ks + 8 * lh < hd) z = *reinterpret_cast<const u32x4*>(row + 16 * ks + 8 * lh);
    f[ks] = __builtin_bit_cast(bf16x8, z);
  }
}

// accumulators [DT][16] of one row (lane (r, h): d = 32 dt + 8 g + 4 h + e) -> global, x mul, fp32 or bf16, real columns only
template <int DP>
__device__ __forceinline__ void gen_store_row(void* base, long elem_off, int is_bf16, const f32x16 (&acc)[DP / 32], float mul, int hd, int lh) {
#pragma unroll
  for (int dt = 0; dt < DP / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = dt * 32 + 8 * g + 4 * lh;
      if (d0 >= hd) continue;
      const float v4[4] = {acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul, acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul};
      if (is_bf16) store4(reinterpret_cast<bf16_t*>(base) + elem_off + d0, v4);
      else store4(reinterpret_cast<float*>(base) + elem_off + d0, v4);
    }
}

// ---- forward (attention.py:94-99 under unet.py:125-141): 4 waves = 4 (head, 32-query block) pairs share each 64-key K / V tile.
// MASKED: Attend's attn_mask (attention.py:77-99) -- the reference casts it to bf16 and hands it to SDPA as an ADDITIVE bias of the
// scaled scores (a bool mask therefore adds 1.0 / 0.0; that is the reference's behaviour and is kept), broadcast over (B, H, N, N).
template <int DP, bool MASKED = false>
__global__ __launch_bounds__(256) void mqa_gen_fwd_kernel(AttnArgs a, int hd) {
  using T = GenTile<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // K image 64 x RB | V image 64 x RB
  constexpr int TILE = 64 * T::RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, nqb = (a.N + 31) >> 5;
  const int vb = blockIdx.x * 4 + wave;
  const bool active = vb < nqb * a.H;
  const int h = active ? vb % a.H : 0, pb = active ? vb / a.H : 0;
  const int qrow = pb * 32 + lr;
  const bool qok = active && qrow < a.N;
  const float c = a.cexp;
  bf16x8 qf[T::KS];
  gen_load_row_frags<DP>(qf, a.q + ((long)b * a.N + qrow) * a.ldq + h * hd, qok, hd, lh);
  f32x16 o[T::DT];
#pragma unroll
  for (int i = 0; i < T::DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (a.N + 63) >> 6;
  for (int j = 0; j < ntiles; ++j) {
    __syncthreads();                                                   // the previous tile has been consumed
    T::fill(smem, a.k + (long)b * a.N * a.ldk, a.ldk, 64, j * 64, a.N, hd, tid, 256);
    T::fill(smem + TILE, a.v + (long)b * a.N * a.ldv, a.ldv, 64, j * 64, a.N, hd, tid, 256);
    __syncthreads();
    f32x16 s[2];
    if constexpr (MASKED) {                                            // the bias starts the accumulator, in units of the raw dot product
      // (loads issued ahead of the MFMAs; indices clamped instead of branched on: rows / keys past N are discarded below)
      const bf16_t* mrow = a.mask + (long)b * a.mask_b + (long)h * a.mask_h + (long)(qok ? qrow : 0) * a.mask_q;
      const float inv_scale = 1.f / a.scale;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          s[kt][r] = bf16_to_f32(mrow[(long)(key < a.N ? key : a.N - 1) * a.mask_k]) * inv_scale;
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < T::KS; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::row_frag(smem, lane, ks, kt), qf[ks], s[kt], 0, 0, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= a.N) s[kt][r] = -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
    float m_new = fmaxf(m_run, mx);                                   // (unmasked: a tile always holds >= 1 valid key, m_new is finite)
    if constexpr (MASKED) m_new = m_new == -INFINITY ? -3.0e38f : m_new;   // a row whose keys so far are all masked with -inf: p = 0, no NaN
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int i = 0; i < T::DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float p = fast_exp2(fmaf(s[kt][r], c, -m_run)); s[kt][r] = p; ps += p; }
    l_run += ps;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const bf16x8 pf = acc_to_frag(s[s4 >> 1], s4 & 1);
#pragma unroll
      for (int dt = 0; dt < T::DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::tr_frag(smem + TILE, lane, (s4 >> 1) * 32 + (s4 & 1) * 16, dt), pf, o[dt], 0, 0, 0);
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qok) {
    const float inv = 1.f / l_tot;
    if (lh == 0) a.lse2[((long)b * a.H + h) * a.N + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
    const long m = (long)b * a.N + qrow;
#pragma unroll
    for (int dt = 0; dt < T::DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = dt * 32 + 8 * g + 4 * lh;
        if (d0 >= hd) continue;
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = round_bf16(o[dt][4 * g + e] * inv);
        if (a.o_is_f32) store4(reinterpret_cast<float*>(a.o) + m * a.ldo + h * hd + d0, v4);
        else store4(reinterpret_cast<bf16_t*>(a.o) + m * a.ldo + h * hd + d0, v4);
      }
  }
}

// ---- backward, dQ (query-stationary): dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]; gradient of the ROTATED q (x scale), no RoPE transpose
template <int DP>
__global__ __launch_bounds__(256) void mqa_gen_bwd_dq_kernel(AttnArgs a, int hd) {
  using T = GenTile<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = 64 * T::RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y, nqb = (a.N + 31) >> 5;
  const int vb = blockIdx.x * 4 + wave;
  const bool active = vb < nqb * a.H;
  const int h = active ? vb % a.H : 0, pb = active ? vb / a.H : 0;
  const int qrow = pb * 32 + lr;
  const bool qok = active && qrow < a.N;
  const float c = a.cexp;
  bf16x8 qf[T::KS], dof[T::KS];
  gen_load_row_frags<DP>(qf, a.q + ((long)b * a.N + qrow) * a.ldq + h * hd, qok, hd, lh);
  gen_load_row_frags<DP>(dof, a.dout + ((long)b * a.N + qrow) * a.lddo + h * hd, qok, hd, lh);
  const long sidx = ((long)b * a.H + h) * a.N + qrow;
  const float L2 = qok ? a.lse2[sidx] : INFINITY;
  const float dl = qok ? a.delta[sidx] : 0.f;
  f32x16 acc[T::DT];
#pragma unroll
  for (int i = 0; i < T::DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int ntiles = (a.N + 63) >> 6;
  for (int j = 0; j < ntiles; ++j) {
    __syncthreads();
    T::fill(smem, a.k + (long)b * a.N * a.ldk, a.ldk, 64, j * 64, a.N, hd, tid, 256);
    T::fill(smem + TILE, a.v + (long)b * a.N * a.ldv, a.ldv, 64, j * 64, a.N, hd, tid, 256);
    __syncthreads();
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < T::KS; ++ks) {
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::row_frag(smem, lane, ks, kt), qf[ks], s[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::row_frag(smem + TILE, lane, ks, kt), dof[ks], dp[kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(s[kt][r], c, -L2));            // keys past N: zero K rows -> their dS meets zero K rows below
        s[kt][r] = p * (dp[kt][r] - dl);
      }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const bf16x8 df = acc_to_frag(s[s4 >> 1], s4 & 1);
#pragma unroll
      for (int dt = 0; dt < T::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::tr_frag(smem, lane, (s4 >> 1) * 32 + (s4 & 1) * 16, dt), df, acc[dt], 0, 0, 0);
    }
  }
  if (qok) gen_store_row<DP>(a.dq, ((long)b * a.N + qrow) * a.lddq + h * hd, a.g_bf16, acc, a.scale, hd, lh);
}

// ---- backward, dK / dV (key-stationary): 4 waves = 128 keys sweep every (head, 32-query block) pair; gradients of the ROTATED k (x scale) and v
template <int DP>
__global__ __launch_bounds__(256) void mqa_gen_bwd_dkv_kernel(AttnArgs a, int hd) {
  using T = GenTile<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // Q image 32 x RB | dO image 32 x RB | lse 128 | delta 128
  constexpr int TILE = 32 * T::RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int nkb = (a.N + 127) / 128;
  const int b = blockIdx.x / nkb, kb = blockIdx.x - b * nkb;
  const int key = kb * 128 + wave * 32 + lr;
  const bool kok = key < a.N;
  const float c = a.cexp;
  const int nqb = (a.N + 31) >> 5;
  bf16x8 kf[T::KS], vf[T::KS];
  gen_load_row_frags<DP>(kf, a.k + ((long)b * a.N + key) * a.ldk, kok, hd, lh);
  gen_load_row_frags<DP>(vf, a.v + ((long)b * a.N + key) * a.ldv, kok, hd, lh);
  f32x16 dk[T::DT], dv[T::DT];
#pragma unroll
  for (int i = 0; i < T::DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }
  float* ls = reinterpret_cast<float*>(smem + 2 * TILE);
  for (int pb = 0; pb < nqb; ++pb)
    for (int h = 0; h < a.H; ++h) {
      __syncthreads();
      T::fill(smem, a.q + (long)b * a.N * a.ldq + h * hd, a.ldq, 32, pb * 32, a.N, hd, tid, 256);
      T::fill(smem + TILE, a.dout + (long)b * a.N * a.lddo + h * hd, a.lddo, 32, pb * 32, a.N, hd, tid, 256);
      if (tid < 64) {
        const int qr = pb * 32 + (tid & 31);
        const long sidx = ((long)b * a.H + h) * a.N + qr;
        ls[tid] = tid < 32 ? (qr < a.N ? a.lse2[sidx] : INFINITY) : (qr < a.N ? a.delta[sidx] : 0.f);      // padded query rows: P = dS = 0
      }
      __syncthreads();
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < T::KS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::row_frag(smem, lane, ks, 0), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::row_frag(smem + TILE, lane, ks, 0), vf[ks], dp, 0, 0, 0);
      }
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p = fast_exp2(fmaf(s[r], c, -ls[qi]));
        s[r] = p;
        ds[r] = p * (dp[r] - ls[32 + qi]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_to_frag(s, s2);
        const bf16x8 df = acc_to_frag(ds, s2);
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::tr_frag(smem + TILE, lane, s2 * 16, dt), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T::tr_frag(smem, lane, s2 * 16, dt), df, dk[dt], 0, 0, 0);
        }
      }
    }
  if (kok) {
    gen_store_row<DP>(a.dk, ((long)b * a.N + key) * a.lddk, a.g_bf16, dk, a.scale, hd, lh);
    gen_store_row<DP>(a.dv, ((long)b * a.N + key) * a.lddk, a.g_bf16, dv, 1.f, hd, lh);
  }
}

// ---- elementwise companions for any head dim that is a multiple of 16 -----------------------------------------------------------
// delta[b][h][n] = sum_d dO * O: one thread per (row, head)
template <typename TO>
__global__ __launch_bounds__(256) void attn_delta_gen_kernel(const bf16_t* dout, long lddo, const TO* o, long ldo, float* delta, int B, int H, int N, int hd) {
  const long total = (long)B * N * H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / H;
    const int h = (int)(idx - m * H);
    float s = 0.f;
    for (int d = 0; d < hd; d += 8) {
      float x[8], y[8];
      load8(dout + m * lddo + h * hd + d, x);
      load8(o + m * ldo + h * hd + d, y);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[e] * y[e];
    }
    const long bb = m / N, n = m - bb * N;
    delta[(bb * H + h) * N + n] = s;
  }
}

// RoPE (attention.py:52-58, half-split) + cast, and its transpose on gradients; tables [N][hd / 2].  One thread: 8 columns d0..d0+7 of the
// first half of one head and their partners d0 + hd/2 ..  DIR = +1: y1 = x1 c - x2 s, y2 = x2 c + x1 s; DIR = -1: the transpose.
template <typename TI, typename TOUT, int DIR>
__global__ __launch_bounds__(256) void rope_gen_kernel(const TI* in, long ld_in, TOUT* out, long ld_out, const float* cosb, const float* sinb,
                                                       int M, int N, int n_rot_heads, int n_heads_total, int hd) {
  const int half = hd / 2, per = half / 8;
  const long total = (long)M * n_heads_total * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int sub = (int)(idx % per);
    const long rh = idx / per;
    const long m = rh / n_heads_total;
    const int h = (int)(rh - m * n_heads_total);
    const int n = (int)(m % N);
    float x1[8], x2[8];
    load8(in + m * ld_in + h * hd + sub * 8, x1);
    load8(in + m * ld_in + h * hd + half + sub * 8, x2);
    if (h < n_rot_heads) {
      float cs[8], sn[8];
      load8(cosb + (long)n * half + sub * 8, cs);
      load8(sinb + (long)n * half + sub * 8, sn);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = DIR > 0 ? sn[e] : -sn[e];
        const float a1 = x1[e] * cs[e] - x2[e] * sg;
        const float a2 = x2[e] * cs[e] + x1[e] * sg;
        x1[e] = a1; x2[e] = a2;
      }
    }
    store8(out + m * ld_out + h * hd + sub * 8, x1);
    store8(out + m * ld_out + h * hd + half + sub * 8, x2);
  }
}
