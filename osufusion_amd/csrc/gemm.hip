// Tap-aware GEMMs for the conv1d / linear layers of the denoiser, gfx950 only.
//
//   gemm_nt : C[m][n] = epi( sum_t sum_k A[rowmap(m,t)][k] * W[t][n][k] )       (fwd conv / linear / dgrad)
//   gemm_tn : dW[t][n1][n2] += sum_m dY[m][n1] * X[rowmap(m,t)][n2]             (wgrad, split over m, fp32 atomics)
//
// Activations are channels-last ([B*L][C], C contiguous) so a k-tap conv1d is k accumulating GEMMs whose
// A rows are shifted by the tap; both MFMA operands are K-contiguous (weights are pre-packed [tap][N][K]).
// bf16 storage -> v_mfma_f32_32x32x16_bf16; f32 storage -> v_mfma_f32_32x32x2_f32 (exact f32 fmaf chain).
// Block tile 128x128, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA tiles; K-step = 128 bytes of K per row.
// Global -> registers -> LDS double buffering (one barrier per K-step), XOR-swizzled 16-B chunks so the
// ds_read_b128 fragment reads are conflict-free; epilogue goes through LDS so HBM stores are full rows.
#include "common.hpp"
#include <stdlib.h>

struct RowMap {
  int Lin, Lout, stride, pad, mode;
};
// mode 0: src = i*stride + t - pad, zero outside [0, Lin)
// mode 1: as 0, but src == Lin reflects to Lin-2            (Downsample: F.pad(x,(0,1),"reflect"), unet.py:84-87)
// mode 2: u = i*stride + t - pad over the nearest-x2 upsampled input, zero outside [0, 2*Lin), src = u>>1
//                                                           (Upsample fwd, unet.py:66-69; with stride=2 its dgrad)
// mode 3: dgrad of mode 1 (stride 2, k3): t<3: e = i - t, valid iff e even and 0 <= e/2 < Lin, src = e/2;
//         t==3: valid iff i == Lout-2, src = Lin-1          (the reflected column's contribution)
__device__ __forceinline__ int map_row(const RowMap& rm, int i, int t) {
  if (rm.mode == 3) {
    if (t == 3) return (i == rm.Lout - 2) ? rm.Lin - 1 : -1;
    int e = i - t;
    if (e < 0 || (e & 1)) return -1;
    e >>= 1;
    return e < rm.Lin ? e : -1;
  }
  int s = i * rm.stride + t - rm.pad;
  if (rm.mode == 2) {
    if (s < 0 || s >= 2 * rm.Lin) return -1;
    return s >> 1;
  }
  if (rm.mode == 1 && s == rm.Lin) s = rm.Lin - 2;
  return (s >= 0 && s < rm.Lin) ? s : -1;
}

struct GemmArgs {
  const void* A; const void* W; void* C; void* C2; const void* R; const void* U;
  const float* bias; const float* rscale; double* stats;
  float* delta; int heads;         // osuf_gemm_nt_rowdot: R is not added but dotted, per 64-column head, with the bf16-rounded output
  long lda, ldw, tapstride, ldc, ldc2, ldr, ldu;
  int M, N, K, taps;
  RowMap rm;
  int act;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> { static constexpr int BK = 64; };
template <> struct Mma<float> { static constexpr int BK = 32; };

static constexpr int kTile = 128;
static constexpr int kStageBytes = 2 * kTile * 128;   // A tile + B tile, 128 B of K per row

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// LOADS = false: instantiation for launches without residual / dact operand -- no vector-memory load inside the store loop, hence no
// `s_waitcnt vmcnt(0)` (which also waits for the previous pass's stores) in front of every pass; see gemm_big_epilogue_impl.
template <typename T, bool LOADS>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& g, f32x16 (&acc)[2][2], char* smem, int m0, int n0, int tid, int lane, int wr, int wc) {
  const int lr = lane & 31, lh = lane >> 5;
  // ---- epilogue: accumulators -> LDS (fp32 [128][128]) -> coalesced row stores -------------------------
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        cs[row * kTile + wc * 64 + j * 32 + lr] = acc[i][j][r];
      }
  // per-sample (sum, sumsq) slots of this tile live in the 512 B behind the C tile (all LDS is dynamic: G17)
  float* sstat = reinterpret_cast<float*>(smem + 2 * kStageBytes);
  int sb0 = 0;
  if (g.stats) {
    sb0 = m0 / g.rm.Lout;
    if (tid < 2 * 34) sstat[tid] = 0.f;
  }
  __syncthreads();

  T* C = reinterpret_cast<T*>(g.C);
  T* C2 = reinterpret_cast<T*>(g.C2);
  const T* R = LOADS ? reinterpret_cast<const T*>(g.R) : nullptr;
  const T* U = LOADS ? reinterpret_cast<const T*>(g.U) : nullptr;
  const int col4 = (tid & 31) * 4;
  const int n = n0 + col4;
  const bool nok = n < g.N;        // N % 4 == 0 is required by the host wrapper
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias && nok) { f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n); bias4[0] = b[0]; bias4[1] = b[1]; bias4[2] = b[2]; bias4[3] = b[3]; }
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + (tid >> 5);
    const int m = m0 + row;
    float s1 = 0.f, s2 = 0.f;
    int bidx = 0;
    if (m < g.M && nok) {
      f32x4 a4 = *reinterpret_cast<const f32x4*>(cs + row * kTile + col4);
      float v[4] = {a4[0] + bias4[0], a4[1] + bias4[1], a4[2] + bias4[2], a4[3] + bias4[3]};
      if (C2) store4(C2 + (long)m * g.ldc2 + n, v);
      if (g.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
      } else if (g.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sigmoid_f(v[e]);
      }
      if (U) {
        float u[4];
        load4(U + (long)m * g.ldu + n, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= silu_grad_f(u[e]);
      }
      bidx = m / g.rm.Lout;
      if (R) {
        float rr[4];
        load4(R + (long)m * g.ldr + n, rr);
        if (g.delta) {                  // delta[b][head][pos] = sum over the head's 64 columns (16 lanes) of bf16(C) * R
          float part = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) part += round_bf16(v[e]) * rr[e];
          part = group_sum<16>(part);
          if ((tid & 15) == 0) g.delta[((long)bidx * g.heads + (n >> 6)) * g.rm.Lout + (m - bidx * g.rm.Lout)] = part;
        } else if (g.rscale) {
          f32x4 sc = *reinterpret_cast<const f32x4*>(g.rscale + (long)bidx * g.N + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[e] * sc[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += rr[e];
        }
      }
      store4(C + (long)m * g.ldc + n, v);
      if (g.stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float q = ElemTraits<T>::rnd(v[e]); s1 += q; s2 += q * q; }
      }
    }
    if (g.stats) {                       // uniform branch; one row per 32-lane half
      s1 = group_sum<32>(s1);
      s2 = group_sum<32>(s2);
      if ((tid & 31) == 0 && m < g.M) {
        int slot = bidx - sb0;
        if (slot < 34) { atomicAdd(&sstat[2 * slot], s1); atomicAdd(&sstat[2 * slot + 1], s2); }
        else { atomic_add_f64(g.stats + 2 * (long)bidx, (double)s1); atomic_add_f64(g.stats + 2 * (long)bidx + 1, (double)s2); }
      }
    }
  }
  if (g.stats) {
    __syncthreads();
    if (tid < 2 * 34) {
      int slot = tid >> 1;
      long b = sb0 + slot;
      float v = sstat[tid];
      if (v != 0.f && b * g.rm.Lout < g.M) atomic_add_f64(g.stats + 2 * b + (tid & 1), (double)v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// fp32 GEMMs as three bf16 MFMAs on split operands ("f32x3", dtype OSUF_DT_F32X3: fp32 storage, a = a_hi + a_lo with a_hi = bf16(a),
// a_lo = bf16(a - a_hi); a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate).  The dropped a_lo b_lo term and the rounding of
// the lo parts are 2^-17 relative -- fp32 inputs kept to ~17 bits -- against 2^-9 for plain bf16 operands, at 3/16 of the cost of the
// exact v_mfma_f32_32x32x2_f32 path (which runs at 1/16 of the bf16 MFMA rate).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
  typedef __attribute__((ext_vector_type(8))) float f32x8;
  f32x8 v, r;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x[i];
  hi = __builtin_convertvector(v, bf16x8);
  const u32x4 hb = __builtin_bit_cast(u32x4, hi);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[2 * i] = x[2 * i] - __uint_as_float(hb[i] << 16);
    r[2 * i + 1] = x[2 * i + 1] - __uint_as_float(hb[i] & 0xFFFF0000u);
  }
  lo = __builtin_convertvector(r, bf16x8);
}
__device__ __forceinline__ void mfma_x3(f32x16& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);      // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], char* smem, int m0, int n0, int tid, int lane, int wr, int wc) {
  if (g.R == nullptr && g.U == nullptr) gemm_epilogue_impl<T, false>(g, acc, smem, m0, n0, tid, lane, wr, wc);
  else gemm_epilogue_impl<T, true>(g, acc, smem, m0, n0, tid, lane, wr, wc);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = Mma<T>::BK;
  constexpr int EPC = ElemTraits<T>::kPer16B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (g.N + kTile - 1) / kTile;
  const int m0 = (blockIdx.x / tiles_n) * kTile, n0 = (blockIdx.x % tiles_n) * kTile;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);

  // staging assignment: chunk column c (16 B of K), rows r0 + 32*i
  const int c = tid & 7, r0 = tid >> 3;
  int a_base[4], a_pos[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < g.M) { int b = m / g.rm.Lout; a_base[i] = b * g.rm.Lin; a_pos[i] = m - b * g.rm.Lout; }
    else { a_base[i] = 0; a_pos[i] = -1; }
    b_ok[i] = (n0 + r0 + 32 * i) < g.N;
  }
  const int ksteps = (g.K + BK - 1) / BK;
  const int nsteps = g.taps * ksteps;

  // Two statically named register sets: the tile of step s+2 is issued at the top of step s and written to LDS at the
  // bottom of step s+1, so every global load has ~2 K-steps (>= 1000 MFMA cycles at 2 waves/SIMD) to land.
  u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
  auto load_regs = [&](int step, u32x4 (&ra)[4], u32x4 (&rb)[4]) {
    const int t = step / ksteps, kb = step - t * ksteps;
    const int k = kb * BK + c * EPC;
    const bool kok = k < g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 z = {0u, 0u, 0u, 0u};
      ra[i] = z; rb[i] = z;
      if (kok && a_pos[i] >= 0) {
        int s = map_row(g.rm, a_pos[i], t);
        if (s >= 0) ra[i] = *reinterpret_cast<const u32x4*>(A + (long)(a_base[i] + s) * g.lda + k);
      }
      if (kok && b_ok[i])
        rb[i] = *reinterpret_cast<const u32x4*>(W + (long)t * g.tapstride + (long)(n0 + r0 + 32 * i) * g.ldw + k);
    }
  };
  auto store_lds = [&](int buf, const u32x4 (&ra)[4], const u32x4 (&rb)[4]) {
    char* sa = smem + buf * kStageBytes;
    char* sb = sa + kTile * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int off = swz_off(r0 + 32 * i, c);
      *reinterpret_cast<u32x4*>(sa + off) = ra[i];
      *reinterpret_cast<u32x4*>(sb + off) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) {
    const char* sa = smem + buf * kStageBytes;
    const char* sb = sa + kTile * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const u32x4*>(sa + swz_off(wr * 64 + i * 32 + lr, 2 * ks + lh));
        fb[i] = *reinterpret_cast<const u32x4*>(sb + swz_off(wc * 64 + i * 32 + lr, 2 * ks + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                 __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
            f32x4 va = __builtin_bit_cast(f32x4, fa[i]), vb = __builtin_bit_cast(f32x4, fb[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], vb[e], acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  // prologue: step 0 -> LDS[0]; step 1 in flight in set 1
  load_regs(0, ra0, rb0);
  if (nsteps > 1) load_regs(1, ra1, rb1);
  store_lds(0, ra0, rb0);
  __syncthreads();
  for (int step = 0; step < nsteps; step += 2) {
    // even step: compute LDS[0]; set 1 (step+1) -> LDS[1]; issue step+2 into set 0
    if (step + 2 < nsteps) load_regs(step + 2, ra0, rb0);
    compute(0);
    if (step + 1 < nsteps) store_lds(1, ra1, rb1);
    __syncthreads();
    if (step + 1 >= nsteps) break;
    // odd step: compute LDS[1]; set 0 (step+2) -> LDS[0]; issue step+3 into set 1
    if (step + 3 < nsteps) load_regs(step + 3, ra1, rb1);
    compute(1);
    if (step + 2 < nsteps) store_lds(0, ra0, rb0);
    __syncthreads();
  }

  gemm_epilogue<T>(g, acc, smem, m0, n0, tid, lane, wr, wc);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant of gemm_nt: tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write: the
// ds_write_b128 path moves only ~79 B/clk/CU and was the bottleneck of the register-staged loop).  One wave-instruction
// fills 1 KiB = 8 tile rows; the LDS image is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE chunk
// (cdna_hip_programming.md rule 21) and undone by swz_off() on the read side.  Rows outside the tensor (conv padding,
// M / N / K tails) read from a 64-byte device zero page.
// ---------------------------------------------------------------------------------------------------------
__device__ uint4 g_zero_page[4];
typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = Mma<T>::BK;
  constexpr int EPC = ElemTraits<T>::kPer16B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = (g.N + kTile - 1) / kTile;
  const int m0 = (blockIdx.x / tiles_n) * kTile, n0 = (blockIdx.x % tiles_n) * kTile;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // DMA assignment: instruction i of this wave fills tile rows (wave*4+i)*8 + (lane>>3), LDS chunk position lane&7
  int a_base[4], a_pos[4], chunk[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    chunk[i] = (lane & 7) ^ ((row >> 1) & 7);            // logical K chunk stored at this lane's LDS position
    const int m = m0 + row;
    if (m < g.M) { int b = m / g.rm.Lout; a_base[i] = b * g.rm.Lin; a_pos[i] = m - b * g.rm.Lout; }
    else { a_base[i] = 0; a_pos[i] = -1; }
    b_ok[i] = (n0 + row) < g.N;
  }
  const int ksteps = (g.K + BK - 1) / BK;
  const int nsteps = g.taps * ksteps;

  auto issue = [&](int step, int buf) {
    const int t = step / ksteps, kb = step - t * ksteps;
    char* sa = smem + buf * kStageBytes + wave * 4096;
    char* sb = sa + kTile * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb * BK + chunk[i] * EPC;
      const bool kok = k < g.K;
      const char* pa = zero;
      const char* pb = zero;
      if (kok && a_pos[i] >= 0) {
        int s = map_row(g.rm, a_pos[i], t);
        if (s >= 0) pa = reinterpret_cast<const char*>(A + (long)(a_base[i] + s) * g.lda + k);
      }
      if (kok && b_ok[i]) pb = reinterpret_cast<const char*>(W + (long)t * g.tapstride + (long)(n0 + (wave * 4 + i) * 8 + (lane >> 3)) * g.ldw + k);
      __builtin_amdgcn_global_load_lds((gas_ptr)pa, (las_ptr)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gas_ptr)pb, (las_ptr)(sb + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  issue(0, 0);
  __syncthreads();                                         // hipcc drains the DMA (vmcnt(0)) in front of the barrier
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) issue(step + 1, buf ^ 1);
    const char* sa = smem + buf * kStageBytes;
    const char* sb = sa + kTile * 128;
    if constexpr (SPLIT) {
      // two 16-deep bf16 k-steps per 32-float stage: lane (r, h) takes the two 16-byte chunks 4 kk + h and 4 kk + 2 + h of its row (any
      // k order works as long as A and B use the same one)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + swz_off(wr * 64 + i * 32 + lr, 4 * kk + lh));
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + swz_off(wr * 64 + i * 32 + lr, 4 * kk + 2 + lh));
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb + swz_off(wc * 64 + i * 32 + lr, 4 * kk + lh));
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(sb + swz_off(wc * 64 + i * 32 + lr, 4 * kk + 2 + lh));
          const float xa[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          const float xb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          split_bf16x8(xa, ah[i], al[i]);
          split_bf16x8(xb, bh[i], bl[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mfma_x3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
      }
      __syncthreads();
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const u32x4*>(sa + swz_off(wr * 64 + i * 32 + lr, 2 * ks + lh));
        fb[i] = *reinterpret_cast<const u32x4*>(sb + swz_off(wc * 64 + i * 32 + lr, 2 * ks + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                 __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
            f32x4 va = __builtin_bit_cast(f32x4, fa[i]), vb = __builtin_bit_cast(f32x4, fb[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], vb[e], acc[i][j], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }
  gemm_epilogue<T>(g, acc, smem, m0, n0, tid, lane, wr, wc);
}

// ---------------------------------------------------------------------------------------------------------
// 256x256 block tile, 8 waves (2x4), wave tile 128x64 = 4x2 MFMA tiles, LDS-DMA staging, 2 x 64 KiB ring.
// Why: at full MFMA rate a 128x128x64 tile needs 64 B/clk/CU of L2->LDS traffic and its 64x64 wave tiles need 256 B/clk of
// ds_read_b128 -- both AT the CU's limits.  This geometry needs 32 B/clk and 192 B/clk.  One workgroup per CU.
// Epilogue: each wave transposes its own 64x64 sub-tiles through a private 16 KiB LDS slice (no block barrier).
// ---------------------------------------------------------------------------------------------------------
// The 256^2 kernel's output rows leave with the non-temporal hint: a launch writes 67-300 MB that nothing re-reads before the launch
// ends, and as plain stores those lines displaced the A / W panels the other workgroups of the XCD were still reading from L2
// (tools/bench_gemm.py, same box: q|kv 256 -> 1152 168 -> 150 us, to_out 94 -> 86, ff2 53 -> 47, k3 convs +1-4 %).
__device__ __forceinline__ void store8_nt(bf16_t* p, const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
  __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ void store8_nt(float* p, const float (&v)[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p));
  __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4));
}

static constexpr int kBig = 256;
static constexpr int kBigStage = 2 * kBig * 128;        // 64 KiB: A 256 rows + B 256 rows, 128 B of K each

// LOADS = false is the instantiation for launches without a residual (R) or an activation-derivative operand (U): its store loop
// contains no vector-memory LOAD.  With a load anywhere in the loop body hipcc places `s_waitcnt vmcnt(0)` in front of every pass
// (a pending load's destination registers are reused), and since stores count in vmcnt too, every pass then waited for the
// previous pass's global stores to be acknowledged: ~0.3 us x 32 passes = the 10 us per-tile "fixed cost" of the round-1 triage
// (found with per-phase triage builds of this function: no stores / no LDS transpose / no epilogue).  Preloading R / U per half
// into registers and fully unrolling the passes was also tried: slower than this plain split (code size).
template <typename T, int MODE>          // 0: no R / U;  1: R only, 2: U only (operand preloaded per half);  3: both (loads inside the passes)
__device__ __forceinline__ void gemm_big_epilogue_impl(const GemmArgs& g, f32x16 (&acc)[4][2], char* smem, int m0, int n0, int tid, int lane,
                                                       int wave, int wr, int wc) {
  const int lr = lane & 31, lh = lane >> 5;
  // ---- epilogue: per wave, two halves of a 64x64 fp32 sub-tile through its private 16 KiB LDS slice; read back as 8 columns per
  //      lane (8 lanes per row, 8 rows per pass, 8 passes per half) so that a bf16 row segment leaves as one 16-byte store.
  //      LDS image: row-major [64][64] floats with the 16-B chunk index XORed by (row >> 1) & 1 -- that keeps both the
  //      ds_write_b32 of the accumulators and the two ds_read_b128 per lane free of bank conflicts (lane groups of b128 mix 4 rows).
  float* cs = reinterpret_cast<float*>(smem + wave * 16384);
  float* sstat = reinterpret_cast<float*>(smem + 2 * kBigStage);
  int sb0 = 0;
  if (g.stats) {
    sb0 = m0 / g.rm.Lout;
    for (int i = tid; i < 2 * 258; i += 512) sstat[i] = 0.f;      // a 256-row tile spans at most 257 samples
    __syncthreads();
  }
  T* C = reinterpret_cast<T*>(g.C);
  T* C2 = reinterpret_cast<T*>(g.C2);
  const T* R = (MODE & 1) ? reinterpret_cast<const T*>(g.R) : nullptr;
  const T* U = (MODE & 2) ? reinterpret_cast<const T*>(g.U) : nullptr;
  const int col8 = (lane & 7) * 8;
  const int prow = lane >> 3;                              // row of this lane inside a pass
  const int n = n0 + wc * 64 + col8;
  const bool nok = n < g.N;                                // N % 8 == 0 (launcher)
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (g.bias && nok) load8(g.bias + n, bias8);
  const int rd0 = ((2 * (lane & 7)) ^ ((prow >> 1) & 1)) << 2;           // float offsets of this lane's two 16-B chunks in its row
  const int rd1 = ((2 * (lane & 7) + 1) ^ ((prow >> 1) & 1)) << 2;
  constexpr int RAWN = sizeof(T) == 2 ? 4 : 8;             // 8 elements of T as 32-bit words
  typedef __attribute__((ext_vector_type(RAWN))) uint32_t raw8_t;
  auto unpack = [&](const raw8_t& rw, float (&f)[8]) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(rw[i] << 16); f[2 * i + 1] = __uint_as_float(rw[i] & 0xFFFF0000u); }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = __uint_as_float(rw[i]);
    }
  };
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // MODE 1 / 2: this lane's 8 x 8 residual (or dact) values of the half, fetched before the passes so that the pass loop holds
    // no vector-memory load (kept as raw bits: converting inside the guarded block would put a wait behind every single load)
    raw8_t pre[8];
    if constexpr (MODE == 1 || MODE == 2) {
      const T* P = MODE == 1 ? R : U;
      const long ldp = MODE == 1 ? g.ldr : g.ldu;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = m0 + wr * 128 + half * 64 + it * 8 + prow;
        pre[it] = raw8_t{};
        if (m < g.M && nok) pre[it] = *reinterpret_cast<const raw8_t*>(P + (long)m * ldp + n);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = j * 32 + lr;
          cs[row * 64 + ((((col >> 2) ^ ((row >> 1) & 1)) << 2) | (col & 3))] = acc[half * 2 + i][j][r];
        }
    __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    // sample index of this lane's row, carried across the passes (+8 rows each) instead of one integer division per pass
    int bidx = 0, bpos = 0;
    if (g.stats != nullptr || g.rscale != nullptr || g.delta != nullptr) {
      const int mfirst = m0 + wr * 128 + half * 64 + prow;
      bidx = mfirst / g.rm.Lout;
      bpos = mfirst - bidx * g.rm.Lout;
    }
    // GroupNorm statistics: per-lane running sums, reduced over the 8 lanes of a row and added to the tile's LDS slot only when
    // the row's sample changes (and once at the end of the half) -- with L >= 64 that is one reduction per half instead of 8
    float s1 = 0.f, s2 = 0.f;
    int sslot = bidx - sb0;
    auto flush_stats = [&]() {
      const float t1 = group_sum<8>(s1), t2 = group_sum<8>(s2);
      if ((lane & 7) == 0 && (t1 != 0.f || t2 != 0.f)) { atomicAdd(&sstat[2 * sslot], t1); atomicAdd(&sstat[2 * sslot + 1], t2); }
      s1 = 0.f; s2 = 0.f;
    };
#pragma unroll(MODE == 1 || MODE == 2 ? 8 : 4)
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + prow;
      const int m = m0 + wr * 128 + half * 64 + row;
      if (g.stats && bidx - sb0 != sslot) { flush_stats(); sslot = bidx - sb0; }
      if (m < g.M && nok) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cs + row * 64 + rd0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(cs + row * 64 + rd1);
        float v[8] = {a0[0] + bias8[0], a0[1] + bias8[1], a0[2] + bias8[2], a0[3] + bias8[3],
                      a1[0] + bias8[4], a1[1] + bias8[5], a1[2] + bias8[6], a1[3] + bias8[7]};
        if (C2) store8(C2 + (long)m * g.ldc2 + n, v);
        if (g.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        } else if (g.act == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = sigmoid_f(v[e]);
        }
        if (U) {
          float u[8];
          if constexpr (MODE == 2) unpack(pre[it], u);
          else load8(U + (long)m * g.ldu + n, u);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= silu_grad_f(u[e]);
        }
        if (R) {
          float rr[8];
          if constexpr (MODE == 1) unpack(pre[it], rr);
          else load8(R + (long)m * g.ldr + n, rr);
          if (g.delta) {                // as in gemm_epilogue_impl; a head's 64 columns are this row's 8 lanes
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) part += round_bf16(v[e]) * rr[e];
            part = group_sum<8>(part);
            if ((lane & 7) == 0) g.delta[((long)bidx * g.heads + (n >> 6)) * g.rm.Lout + bpos] = part;
          } else if (g.rscale) {
            float sc[8];
            load8(g.rscale + (long)bidx * g.N + n, sc);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e] * sc[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
          }
        }
        store8_nt(C + (long)m * g.ldc + n, v);   // streaming: see store8_nt
        if (g.stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { float q = ElemTraits<T>::rnd(v[e]); s1 += q; s2 += q * q; }
        }
      }
      bpos += 8;
      while (bpos >= g.rm.Lout) { bpos -= g.rm.Lout; ++bidx; }
    }
    if (g.stats) flush_stats();                             // slot < 258 by construction: no global atomic near the store loop
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }
  if (g.stats) {
    __syncthreads();
    for (int i = tid; i < 2 * 258; i += 512) {
      const long b = sb0 + (i >> 1);
      const float v = sstat[i];
      if (v != 0.f && b * g.rm.Lout < g.M) atomic_add_f64(g.stats + 2 * b + (i & 1), (double)v);
    }
  }
}

template <typename T>
__device__ __forceinline__ void gemm_big_epilogue(const GemmArgs& g, f32x16 (&acc)[4][2], char* smem, int m0, int n0, int tid, int lane, int wave,
                                                  int wr, int wc) {
  if (g.R == nullptr && g.U == nullptr) gemm_big_epilogue_impl<T, 0>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
  else if (g.U == nullptr) gemm_big_epilogue_impl<T, 1>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
  else if (g.R == nullptr) gemm_big_epilogue_impl<T, 2>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
  else gemm_big_epilogue_impl<T, 3>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
}

// DBG (bottleneck triage builds, selected by OSUF_GEMM_DBG; results are garbage unless 0): 1 = no MFMA, 2 = no LDS fragment
// reads, 3 = no global->LDS DMA, 4 = DMA only, 5 = DMA only from one hot 1-KiB region (memory-side vs LDS-side cost).
// SPLIT (T = float, OSUF_DT_F32X3): fp32 stages (32 k per row), every fragment split in registers into bf16 hi + lo, three bf16
// MFMAs per product (mfma_x3) -- two 16-deep k-steps per stage, 48 MFMAs per wave and stage against 32 of the bf16 kernel.
template <typename T, int DBG = 0, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_big_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BK = Mma<T>::BK;
  constexpr int EPC = ElemTraits<T>::kPer16B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.N + kBig - 1) / kBig;
  // XCD-aware tile order (speed only): block ids are dealt round-robin to the 8 XCDs, so the N-tiles that share one
  // M-tile's A rows get ids 8 apart -> same XCD (same L2), dispatched back to back.
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int mt = (qid / tiles_n) * 8 + xcd;
  const int m0 = mt * kBig, n0 = (qid % tiles_n) * kBig;
  if (m0 >= g.M) return;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // Per-lane DMA sources.  The row maps (taps, reflect / nearest-x2 / dgrad geometries) are evaluated once per TAP; inside a
  // tap every K-step is `pointer += 128 B`.  (Round-1 triage: re-deriving the addresses every K-step cost more issue slots
  // than the 32 MFMAs of the step -- removing it took the K=2048 linear from 726 to the number in DESIGN.md.)
  int a_base[4], a_pos[4], koff[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    koff[i] = ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
    const int m = m0 + row;
    if (m < g.M) { int b = m / g.rm.Lout; a_base[i] = b * g.rm.Lin; a_pos[i] = m - b * g.rm.Lout; }
    else { a_base[i] = 0; a_pos[i] = -1; }
    b_ok[i] = (n0 + row) < g.N;
  }
  const int ksteps = (g.K + BK - 1) / BK;
  const int nsteps = g.taps * ksteps;
  const bool ktail = (g.K % BK) != 0;                      // only then can a 16-B chunk of the last K-step lie beyond K

  const char* pa[4];
  const char* pb[4];
  int ia[4], ib[4];
  auto set_tap = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pa[i] = zero; ia[i] = 0;
      if (a_pos[i] >= 0) {
        int s = map_row(g.rm, a_pos[i], t);
        if (s >= 0) { pa[i] = reinterpret_cast<const char*>(A + (long)(a_base[i] + s) * g.lda + koff[i]); ia[i] = BK * (int)sizeof(T); }
      }
      pb[i] = zero; ib[i] = 0;
      if (b_ok[i]) {
        pb[i] = reinterpret_cast<const char*>(W + (long)t * g.tapstride + (long)(n0 + (wave * 4 + i) * 8 + (lane >> 3)) * g.ldw + koff[i]);
        ib[i] = BK * (int)sizeof(T);
      }
    }
  };
  int itap = 0, ikb = 0;                                   // (tap, K-step) of the NEXT stage to be issued
  set_tap(0);
  // One quarter (one A + one B DMA instruction per wave) of the next stage; the four quarters are spread over the four
  // k16 sub-steps of the current stage so the memory pipe accepts them while the MFMAs run (issuing all eight up front
  // stalled every wave of the barrier-synchronised workgroup on the address queue with the matrix cores idle).
  auto issue_part = [&](int buf, int i) {
    char* sa = smem + buf * kBigStage + wave * 4096;
    char* sb = sa + kBig * 128;
    const char* qa = pa[i];
    const char* qb = pb[i];
    if (ktail && ikb == ksteps - 1 && ikb * BK + koff[i] >= g.K) { qa = zero; qb = zero; }
    if (DBG == 5) { qa = reinterpret_cast<const char*>(A) + lane * 16; qb = reinterpret_cast<const char*>(W) + lane * 16; }
    __builtin_amdgcn_global_load_lds((gas_ptr)qa, (las_ptr)(sa + i * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gas_ptr)qb, (las_ptr)(sb + i * 1024), 16, 0, 0);
    pa[i] += ia[i];
    pb[i] += ib[i];
  };
  auto issue_done = [&]() {
    if (++ikb == ksteps) {
      ikb = 0;
      if (++itap < g.taps) set_tap(itap);
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_part(buf, i);
    issue_done();
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  if (DBG != 3) issue(0);
  __syncthreads();
  u32x4 dfa[4], dfb[2];
  if (DBG == 2) {
    for (int i = 0; i < 4; ++i) dfa[i] = u32x4{(uint32_t)lane, 1u, 2u, 3u};
    for (int j = 0; j < 2; ++j) dfb[j] = u32x4{(uint32_t)lane, 5u, 6u, 7u};
  }
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    const bool more = DBG != 3 && step + 1 < nsteps;
    const char* sa = smem + buf * kBigStage;
    const char* sb = sa + kBig * 128;
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (more) { issue_part(buf ^ 1, 2 * ks); issue_part(buf ^ 1, 2 * ks + 1); }
        bf16x8 ah[4], al[4], bh[2], bl[2];
        auto frag = [&](const char* base, int row, bf16x8& hi, bf16x8& lo) {      // k = 16 ks + 8 lh + 0..7 of this lane's row
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + swz_off(row, 4 * ks + 2 * lh));
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + swz_off(row, 4 * ks + 2 * lh + 1));
          const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          split_bf16x8(x, hi, lo);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) frag(sa, wr * 128 + i * 32 + lr, ah[i], al[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) frag(sb, wc * 64 + j * 32 + lr, bh[j], bl[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mfma_x3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 fa[4], fb[2];
      if (more) issue_part(buf ^ 1, ks);
      if (DBG >= 4) continue;
      if (DBG == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(dfa[i])); fa[i] = dfa[i]; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(dfb[j])); fb[j] = dfb[j]; }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sa + swz_off(wr * 128 + i * 32 + lr, 2 * ks + lh));
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sb + swz_off(wc * 64 + j * 32 + lr, 2 * ks + lh));
      }
      if (DBG == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                 __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
            f32x4 va = __builtin_bit_cast(f32x4, fa[i]), vb = __builtin_bit_cast(f32x4, fb[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], vb[e], acc[i][j], 0, 0, 0);
          }
        }
    }
    }
    if (more) issue_done();
    __syncthreads();
  }

  gemm_big_epilogue<T>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
}

// ---------------------------------------------------------------------------------------------------------
// The same 256 x 256 bf16 tile on an 8-phase schedule (cdna_hip_programming.md section 5, "The 256^2 8-phase template"): the loop above
// drains its LDS-DMA at one __syncthreads() per 64-deep K-step, so DMA latency, fragment reads and MFMAs of a step run one after the other
// (triage: DMA 1.48 us + MFMA/LDS 1.09 us -> 2.4 us per step).  Here
//   * a K-tile (64 deep) is four HALF-TILES of 16 KiB -- B0 / A0 / B1 / A1 = the 32 B columns / 64 A rows every wave needs for its first /
//     second pair of 32-row (32-column) MFMA tiles -- and four PHASES, one 64 x 32 quadrant of every wave's 128 x 64 output per phase:
//       P1 reads B0, A0 (12 ds_read_b128) -> acc[0..1][0]   P2 reads B1 (4) -> acc[0..1][1]   P3 reads A1 (8) -> acc[2..3][1]   P4 -> acc[2..3][0]
//     (per accumulator the k order is the loop's above: results are bit-identical);
//   * every phase stages ONE half-tile (2 global_load_lds_dwordx4 per lane), six half-tiles ahead of the one it reads first: P1 stages
//     A1(t+1), P2 B0(t+2), P3 A0(t+2), P4 B1(t+2).  The only vector-memory wait of a K-tile is a counted `s_waitcnt vmcnt(6)` in P4 (three
//     half-tiles stay in flight; K-tile t+1 has landed), never vmcnt(0); the barriers are raw s_barrier (a __syncthreads() would drain);
//   * a staged half-tile is read one phase after the wait that retires it, and a slot is re-staged two phases after its last read -- one
//     phase for B0, whose reads an `s_waitcnt lgkmcnt(8)` retires before P1's first barrier;
//   * waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its 8-MFMA cluster (s_setprio 1) while its partner
//     reads fragments and issues DMA;
//   * the fragment reads are inline asm: for a C++ LDS load hipcc waits vmcnt(0) whenever an LDS-DMA is in flight.
// LDS map: A buf 0 | A buf 1 | B buf 0 | B buf 1, 32 KiB each (fragment reads reach both buffers and every tile row through the 16-bit
// immediate), then the GroupNorm slots; row / swizzle image of a buffer and the epilogue as in gemm_nt_big_kernel.
// ---------------------------------------------------------------------------------------------------------
template <int OFF> __device__ __forceinline__ void lds_read_b128(u32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int V> struct IntC { static constexpr int value = V; };

// DBG (timing-only triage builds, OSUF_GEMM_DBG with OSUF_GEMM_8P; results are garbage unless 0): 1 = no MFMA, 2 = no fragment reads,
// 3 = no DMA, 4 = DMA only, 5 = DMA only with B from one hot KiB (A's feed alone), 6 = DMA only with A from one hot KiB (B's feed alone)
__device__ uint4 g_zero_row[1024];                           // 16 KiB of zeros: the source "row" of padded / out-of-range A rows (K <= 8192 bf16)
static constexpr int kP8Tbl = 2 * kBigStage + 2112;          // LDS: per-tap source rows of the tile's 256 A rows, [taps][256] ints, behind the stat slots
static constexpr int kP8MaxTaps = 16;

template <int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_big8_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int BK = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.N + kBig - 1) / kBig;
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;       // tile order as in gemm_nt_big_kernel
  const int mt = (qid / tiles_n) * 8 + xcd;
  const int m0 = mt * kBig, n0 = (qid % tiles_n) * kBig;
  if (m0 >= g.M) return;
  const char* A = reinterpret_cast<const char*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_row);
  const int ksteps = (g.K + BK - 1) / BK;
  const int n = g.taps * ksteps;                             // K-tiles (K % 64 == 0: launcher)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;

  // The row maps (taps, reflect / nearest-x2 / dgrad geometries) are evaluated ONCE, into an LDS table of source rows (-1: zeros), before the
  // first DMA is in flight (plain LDS stores + __syncthreads()); a tap switch inside the pipeline is then four ds_read_b32 per lane.
  {
    int* tbl = reinterpret_cast<int*>(smem + kP8Tbl);
    if (tid < kBig) {
      const int m = m0 + tid;
      const int b = m / g.rm.Lout, pos = m - b * g.rm.Lout;
      for (int t = 0; t < g.taps; ++t) {
        const int src = m < g.M ? map_row(g.rm, pos, t) : -1;
        tbl[t * kBig + tid] = src >= 0 ? b * g.rm.Lin + src : -1;
      }
    }
    __syncthreads();
  }

  // DMA roles.  Half-tile s of A = tile rows wr' * 128 + s * 64 + 0..63 (wr' = 0, 1); of B = tile rows wc' * 64 + s * 32 + 0..31
  // (wc' = 0..3).  A wave moves two 8-row pieces (j) of every half-tile: A rows (wave >> 2) * 128 + s * 64 + (wave & 3) * 16 + 8 j + lane / 8,
  // B rows (wave >> 1) * 64 + s * 32 + (wave & 1) * 16 + 8 j + lane / 8.  Every K-step is `pointer += 128 B` (zero rows walk the zero row);
  // W rows beyond N are clamped to N - 1 (their output columns are never stored), so a tap switch on the B side is one uniform delta.
  const char* pa[2][2];
  const char* pb[2][2];
  auto a_row = [&](int s, int j) { return (wave >> 2) * 128 + s * 64 + (wave & 3) * 16 + j * 8 + (lane >> 3); };
  auto b_row = [&](int s, int j) { return (wave >> 1) * 64 + s * 32 + (wave & 1) * 16 + j * 8 + (lane >> 3); };
  const long lda_b = g.lda * (long)sizeof(T);
  auto set_tap_a = [&](int t) {
    int src[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        asm volatile("ds_read_b32 %0, %1" : "=v"(src[s][j]) : "v"(lds0 + (uint32_t)(kP8Tbl + (t * kBig + a_row(s, j)) * 4)));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(src[0][0]), "+v"(src[0][1]), "+v"(src[1][0]), "+v"(src[1][1]));
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int chunk = ((lane & 7) ^ ((a_row(s, j) >> 1) & 7)) * 16;
        pa[s][j] = (src[s][j] >= 0 ? A + (long)src[s][j] * lda_b : zero) + chunk;
      }
  };
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rb = b_row(s, j);
      pb[s][j] = reinterpret_cast<const char*>(W + (long)min(n0 + rb, g.N - 1) * g.ldw + ((lane & 7) ^ ((rb >> 1) & 7)) * 8);
    }
  const long tap_delta_b = (g.tapstride - (long)ksteps * BK) * (long)sizeof(T);      // from the end of tap t's K range to the start of tap t + 1's
  const int aw = ((wave >> 2) * 128 + (wave & 3) * 16) * 128;  // LDS byte offsets of this wave's first piece in a half-tile 0
  const int bw = 65536 + ((wave >> 1) * 64 + (wave & 1) * 16) * 128;
  int st_kb = 0, st_tap = 0;                                 // (tap, K-step) of the K-tile whose half-tiles are being staged
  // half-tile H (0 = B0, 1 = A0, 2 = B1, 3 = A1) of the K-tile being staged, into buffer BUF
  auto stage = [&](auto hc, auto bufc) {
    constexpr int H = decltype(hc)::value, BUF = decltype(bufc)::value, S = H >> 1;
    constexpr bool IS_A = (H & 1) != 0;
    if (DBG != 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* q = IS_A ? pa[S][j] : pb[S][j];
        if ((DBG == 5 && !IS_A) || (DBG == 6 && IS_A)) q = (IS_A ? A : reinterpret_cast<const char*>(W)) + lane * 16;   // that operand from one hot KiB
        char* dst = smem + BUF * 32768 + (IS_A ? aw + S * 8192 : bw + S * 4096) + j * 1024;
        __builtin_amdgcn_global_load_lds((gas_ptr)q, (las_ptr)dst, 16, 0, 0);
        if (IS_A) pa[S][j] += 128; else pb[S][j] += 128;
      }
    }
  };
  // behind the last half-tile (A1) of a K-tile: on to the next K-tile, and at the end of a tap's K range to the next tap
  auto next_ktile = [&]() {
    if (++st_kb == ksteps) {
      st_kb = 0;
      if (++st_tap < g.taps) {
        set_tap_a(st_tap);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) pb[s][j] += tap_delta_b;
      }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row lr of the wave's first A / B MFMA tile, chunk (2 ks + lh) ^ swizzle; tile i / j, buffer: immediates
  const int lr = lane & 31, lh = lane >> 5;
  uint32_t ka[4], kb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint32_t ch = (uint32_t)(((2 * ks + lh) ^ ((lr >> 1) & 7)) << 4);
    ka[ks] = lds0 + (uint32_t)((wr * 128 + lr) * 128) + ch;
    kb[ks] = lds0 + 65536u + (uint32_t)((wc * 64 + lr) * 128) + ch;
  }
  u32x4 fa[2][4], fb0[4], fb1[4];
  if (DBG == 2) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[0][ks] = u32x4{(uint32_t)lane, 1u, 2u, 3u}; fa[1][ks] = fa[0][ks]; fb0[ks] = u32x4{(uint32_t)lane, 5u, 6u, 7u}; fb1[ks] = fb0[ks];
    }
  }
#define P8_WAIT_A(cnt)                                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), \
               "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]));
#define P8_WAIT_B(cnt, fb) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
#define P8_MFMA(i0, j, fbv)                                                                                                   \
  if (DBG != 1 && DBG < 4) {                                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                        \
      acc[i0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[0][ks]), __builtin_bit_cast(bf16x8, fbv[ks]), acc[i0][j], 0, 0, 0); \
      acc[i0 + 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[1][ks]), __builtin_bit_cast(bf16x8, fbv[ks]), acc[i0 + 1][j], 0, 0, 0); \
    }                                                                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                                                            \
  } else {                                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) { asm volatile("" ::"v"(fa[0][ks]), "v"(fa[1][ks]), "v"(fbv[ks])); }         \
  }
  constexpr bool RD = DBG != 2 && DBG < 4;
  // one K-tile (four phases) out of buffer BUF; t = its index
  auto tile = [&](auto bufc, int t) {
    constexpr int BUF = decltype(bufc)::value, O = BUF * 32768;
    // ---- P1: B0, A0 -> acc[0..1][0]; stage A1(t + 1)
    if (RD) {
      lds_read_b128<O>(fb0[0], kb[0]); lds_read_b128<O>(fb0[1], kb[1]); lds_read_b128<O>(fb0[2], kb[2]); lds_read_b128<O>(fb0[3], kb[3]);
      __builtin_amdgcn_sched_barrier(0);
      lds_read_b128<O>(fa[0][0], ka[0]); lds_read_b128<O>(fa[0][1], ka[1]); lds_read_b128<O>(fa[0][2], ka[2]); lds_read_b128<O>(fa[0][3], ka[3]);
      lds_read_b128<O + 4096>(fa[1][0], ka[0]); lds_read_b128<O + 4096>(fa[1][1], ka[1]); lds_read_b128<O + 4096>(fa[1][2], ka[2]);
      lds_read_b128<O + 4096>(fa[1][3], ka[3]);
    }
    if (t + 1 < n) { stage(IntC<3>{}, IntC<BUF ^ 1>{}); next_ktile(); }
    if (RD) { P8_WAIT_B(8, fb0) }                               // B0's reads are done before anyone passes the barrier: P2 re-stages B0's slot
    __builtin_amdgcn_s_barrier();
    if (RD) { P8_WAIT_A(0) }
    __builtin_amdgcn_sched_barrier(0);
    P8_MFMA(0, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P2: B1 -> acc[0..1][1]; stage B0(t + 2)
    if (RD) { lds_read_b128<O + 4096>(fb1[0], kb[0]); lds_read_b128<O + 4096>(fb1[1], kb[1]); lds_read_b128<O + 4096>(fb1[2], kb[2]); lds_read_b128<O + 4096>(fb1[3], kb[3]); }
    if (t + 2 < n) stage(IntC<0>{}, IntC<BUF>{});
    __builtin_amdgcn_s_barrier();
    if (RD) { P8_WAIT_B(0, fb1) }
    __builtin_amdgcn_sched_barrier(0);
    P8_MFMA(0, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P3: A1 -> acc[2..3][1]; stage A0(t + 2)
    if (RD) {
      lds_read_b128<O + 8192>(fa[0][0], ka[0]); lds_read_b128<O + 8192>(fa[0][1], ka[1]); lds_read_b128<O + 8192>(fa[0][2], ka[2]);
      lds_read_b128<O + 8192>(fa[0][3], ka[3]);
      lds_read_b128<O + 12288>(fa[1][0], ka[0]); lds_read_b128<O + 12288>(fa[1][1], ka[1]); lds_read_b128<O + 12288>(fa[1][2], ka[2]);
      lds_read_b128<O + 12288>(fa[1][3], ka[3]);
    }
    if (t + 2 < n) stage(IntC<1>{}, IntC<BUF>{});
    __builtin_amdgcn_s_barrier();
    if (RD) { P8_WAIT_A(0) }
    __builtin_amdgcn_sched_barrier(0);
    P8_MFMA(2, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P4: (B0 still in registers) -> acc[2..3][0]; stage B1(t + 2); K-tile t + 1 has landed behind the counted wait
    if (t + 2 < n) {
      stage(IntC<2>{}, IntC<BUF>{});
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    P8_MFMA(2, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // prologue: K-tile 0 and three half-tiles of K-tile 1
  set_tap_a(0);
  stage(IntC<0>{}, IntC<0>{}); stage(IntC<1>{}, IntC<0>{}); stage(IntC<2>{}, IntC<0>{}); stage(IntC<3>{}, IntC<0>{});
  next_ktile();
  if (n > 1) {
    stage(IntC<0>{}, IntC<1>{}); stage(IntC<1>{}, IntC<1>{}); stage(IntC<2>{}, IntC<1>{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                   // waves 4-7 run one barrier behind
  int t = 0;
  for (; t + 1 < n; t += 2) {
    tile(IntC<0>{}, t);
    tile(IntC<1>{}, t + 1);
  }
  if (t < n) tile(IntC<0>{}, t);
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef P8_WAIT_A
#undef P8_WAIT_B
#undef P8_MFMA
  gemm_big_epilogue<T>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
}

// ---------------------------------------------------------------------------------------------------------
// k = 3 "same" convolutions (residual.py:70 and their input gradients; mode 0, stride 1, pad 1, L % 256 == 0, K % 64 == 0) on the same
// 256 x 256 tile with the activation panel SHARED by the three taps.  The plain kernel streams an A tile per (tap, K-step) although
// tap t's rows are the same rows shifted by t - 1; its loop is bound by the L2 -> LDS feed (touching the next A lines
// early made it slower, more bytes through the same pipe), so bytes per MFMA are what count.  Here a K-step loads ONE panel of 258 rows
// (tile rows -1 .. 256; the two halo rows come from the zero page at a sample edge -- a tile never straddles samples) and three B
// tiles: 129 KiB per three steps instead of 192.  K-steps are the outer loop, taps the inner one (fp32 summation order differs from
// the plain kernel's tap-major order).  LDS: 2 panels x 264 rows x 128 B + 2 x 32 KiB of B = 130 KiB; epilogue shared.
// ---------------------------------------------------------------------------------------------------------
static constexpr int kPanelRows = 264;                       // 33 DMA groups of 8 rows
static constexpr int kPanelBytes = kPanelRows * 128;

// (T = float: the fp32 mode's split-bf16 form -- fp32 panels and B tiles of 32 k per row, fragments split in registers, mfma_x3)
template <typename T>
__global__ __launch_bounds__(512, 2) void gemm_nt_big_halo3_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool X3 = sizeof(T) == 4;
  constexpr int EPC = 16 / (int)sizeof(T);                     // elements per 16-B chunk; a row of a stage is 8 chunks = 128 B of k
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.N + kBig - 1) / kBig;
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;       // tile order as in gemm_nt_big_kernel
  const int mt = (qid / tiles_n) * 8 + xcd;
  const int m0 = mt * kBig, n0 = (qid % tiles_n) * kBig;
  if (m0 >= g.M) return;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int L = g.rm.Lout;
  const int pos0 = m0 % L;                                   // position of the tile's first row inside its sample
  const int ksteps = g.K / (8 * EPC);

  // A panel: group gi (8 rows) of wave w is 4 w + i, i < 4; wave 0 also loads group 32 (rows 256 .. 263, of which 256 and 257 are read)
  const char* pa[5];
  int ia[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = ((i < 4 ? wave * 4 + i : 32) * 8) + (lane >> 3);       // panel row; activation row m0 - 1 + r
    const int pos = pos0 - 1 + r;
    const bool ok = r < 258 && pos >= 0 && pos < L;
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    pa[i] = ok ? reinterpret_cast<const char*>(A + (long)(m0 - 1 + r) * g.lda + c * EPC) : zero;
    ia[i] = ok ? 128 : 0;
  }
  // B tiles: this lane's row of W for DMA instruction i, tap 0, K-step 0
  const char* pb[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    b_ok[i] = n0 + row < g.N;
    pb[i] = b_ok[i] ? reinterpret_cast<const char*>(W + (long)(n0 + row) * g.ldw + c * EPC) : zero;
  }
  const long tapb = g.tapstride * (long)sizeof(T);
  char* panels = smem;
  char* bring = smem + 2 * kPanelBytes;
  auto issue_a = [&](int panel, int i) {                      // group i of this wave into panel `panel`; advances to the next K-step
    const int gi = i < 4 ? wave * 4 + i : 32;
    __builtin_amdgcn_global_load_lds((gas_ptr)pa[i], (las_ptr)(panels + panel * kPanelBytes + gi * 1024), 16, 0, 0);
    pa[i] += ia[i];
  };
  auto issue_b = [&](int slot, int i, int t, int kb) {
    const char* q = b_ok[i] ? pb[i] + (long)t * tapb + (long)kb * 128 : zero;
    __builtin_amdgcn_global_load_lds((gas_ptr)q, (las_ptr)(bring + slot * 32768 + (wave * 4 + i) * 1024), 16, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_a(0, i);
  if (wave == 0) issue_a(0, 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue_b(0, i, 0, 0);
  __syncthreads();

  for (int kb = 0; kb < ksteps; ++kb) {
    const char* pan = panels + (kb & 1) * kPanelBytes;
    const bool more_k = kb + 1 < ksteps;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int slot = (kb + t) & 1;                          // (3 kb + t) & 1
      const char* sb = bring + slot * 32768;
      const bool more = t < 2 || more_k;
      const int tn = t < 2 ? t + 1 : 0, kn = t < 2 ? kb : kb + 1;
      if constexpr (X3) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (more) { issue_b(slot ^ 1, 2 * ks, tn, kn); issue_b(slot ^ 1, 2 * ks + 1, tn, kn); }
          if (more_k) {
            if (t == 0) issue_a((kb + 1) & 1, ks);
            if (t == 1 && ks == 0) issue_a((kb + 1) & 1, 2);
            if (t == 2 && ks == 0) issue_a((kb + 1) & 1, 3);
            if (t == 2 && ks == 1 && wave == 0) issue_a((kb + 1) & 1, 4);
          }
          bf16x8 ah[4], al[4], bh[2], bl[2];
          auto frag = [&](const char* base, int row, bf16x8& hi, bf16x8& lo) {      // k = 16 ks + 8 lh + 0..7 of this lane's row
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(base + swz_off(row, 4 * ks + 2 * lh));
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(base + swz_off(row, 4 * ks + 2 * lh + 1));
            const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            split_bf16x8(x, hi, lo);
          };
#pragma unroll
          for (int i = 0; i < 4; ++i) frag(pan, wr * 128 + i * 32 + lr + t, ah[i], al[i]);
#pragma unroll
          for (int j = 0; j < 2; ++j) frag(sb, wc * 64 + j * 32 + lr, bh[j], bl[j]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mfma_x3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (more) issue_b(slot ^ 1, ks, tn, kn);
        if (more_k) {                                        // next panel: two groups beside tap 0, one beside taps 1 and 2
          if (t == 0 && ks == 1) issue_a((kb + 1) & 1, 0);
          if (t == 0 && ks == 3) issue_a((kb + 1) & 1, 1);
          if (t == 1 && ks == 1) issue_a((kb + 1) & 1, 2);
          if (t == 2 && ks == 1) issue_a((kb + 1) & 1, 3);
          if (t == 2 && ks == 3 && wave == 0) issue_a((kb + 1) & 1, 4);
        }
        u32x4 fa[4], fb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u32x4*>(pan + swz_off(wr * 128 + i * 32 + lr + t, 2 * ks + lh));
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sb + swz_off(wc * 64 + j * 32 + lr, 2 * ks + lh));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
      }
      }
      __syncthreads();
    }
  }
  gemm_big_epilogue<T>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
}

// ---------------------------------------------------------------------------------------------------------
// The shared-panel k = 3 kernel on the 8-phase schedule of gemm_nt_big8_kernel (bf16).  A "K-tile" is one (K-step, tap): four phases over
// the K-step's activation panel (rows shifted by the tap) and the tap's B tile.  B half-tiles are staged two K-tiles ahead as there (P2
// stages B0(t+2), P4 B1(t+2)); the NEXT K-step's panel (33 pieces of 8 rows) rides the free staging slots of the three taps (P1 / P3 of taps
// 0 and 1, P1 of tap 2) into the other panel buffer, whose last reader was the previous K-step's tap 2 / P3.  One counted wait per K-tile
// (P4): everything up to B1(t+1) -- and, at tap 2, the whole next panel -- has landed, the 4 (6 with panel pieces behind it) youngest
// LDS-DMA stay in flight.  Accumulation order = gemm_nt_big_halo3_kernel's (K-steps outer, taps inner): bit-identical.
// LDS: panel 0 | panel 1 (264 rows x 128 B each) | B buf 0 | B buf 1 (32 KiB each) | GroupNorm slots.
// ---------------------------------------------------------------------------------------------------------
static constexpr int kH8B = 2 * kPanelBytes;                 // 67,584: B ring behind the two panels

template <int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_big8_halo3_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.N + kBig - 1) / kBig;
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;       // tile order as in gemm_nt_big_kernel
  const int mt = (qid / tiles_n) * 8 + xcd;
  const int m0 = mt * kBig, n0 = (qid % tiles_n) * kBig;
  if (m0 >= g.M) return;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_row);
  const int L = g.rm.Lout;
  const int pos0 = m0 % L;                                   // position of the tile's first row inside its sample (a tile never straddles samples)
  const int ksteps = g.K / 64;
  const int n = 3 * ksteps;                                  // K-tiles
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;

  // A panel: group gi (8 rows) of wave w is 4 w + i, i < 4; wave 0 also loads group 32 (rows 256 .. 263, of which 256 and 257 are read).
  // Rows outside the sample walk the zero row.
  const char* pa[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = ((i < 4 ? wave * 4 + i : 32) * 8) + (lane >> 3);       // panel row; activation row m0 - 1 + r
    const int pos = pos0 - 1 + r;
    const bool ok = r < 258 && pos >= 0 && pos < L;
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    pa[i] = (ok ? reinterpret_cast<const char*>(A + (long)(m0 - 1 + r) * g.lda) : zero) + c * 16;
  }
  // B: half-tile s = tile rows wc' * 64 + s * 32 + 0..31; this wave's two pieces j: rows (wave >> 1) * 64 + s * 32 + (wave & 1) * 16 + 8 j + lane / 8.
  // Running pointers through (K-step, tap) order; W rows beyond N are clamped (their columns are never stored).
  const char* pb[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rb = (wave >> 1) * 64 + s * 32 + (wave & 1) * 16 + j * 8 + (lane >> 3);
      pb[s][j] = reinterpret_cast<const char*>(W + (long)min(n0 + rb, g.N - 1) * g.ldw + ((lane & 7) ^ ((rb >> 1) & 7)) * 8);
    }
  const long tapb = g.tapstride * (long)sizeof(T);
  const long next_k = 128 - 2 * tapb;                        // from tap 2 of a K-step to tap 0 of the next
  const int bw = kH8B + ((wave >> 1) * 64 + (wave & 1) * 16) * 128;
  int st_tap = 0;                                            // tap of the B tile being staged
  auto stage_b = [&](auto sc, auto bufc) {                    // half-tile S of the B tile being staged, into buffer BUF; behind B1: on to the next B tile
    constexpr int S = decltype(sc)::value, BUF = decltype(bufc)::value;
    if (DBG != 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gas_ptr)pb[S][j], (las_ptr)(smem + bw + BUF * 32768 + S * 4096 + j * 1024), 16, 0, 0);
    }
    if (S == 1) {
      const long d = st_tap < 2 ? tapb : next_k;
      st_tap = st_tap < 2 ? st_tap + 1 : 0;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) pb[s][j] += d;
    }
  };
  auto stage_a = [&](auto ic, auto panelc) {                  // piece I of this wave into panel PANEL; advances to the next K-step
    constexpr int I = decltype(ic)::value, PANEL = decltype(panelc)::value;
    const int gi = I < 4 ? wave * 4 + I : 32;
    if (DBG != 3) __builtin_amdgcn_global_load_lds((gas_ptr)pa[I], (las_ptr)(smem + PANEL * kPanelBytes + gi * 1024), 16, 0, 0);
    pa[I] += 128;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  uint32_t kb_[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kb_[ks] = lds0 + (uint32_t)kH8B + (uint32_t)((wc * 64 + lr) * 128) + (uint32_t)(((2 * ks + lh) ^ ((lr >> 1) & 7)) << 4);
  u32x4 fa[2][4], fb0[4], fb1[4];
#define H8_WAIT_A(cnt)                                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), \
               "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]));
#define H8_WAIT_B(cnt, fb) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
#define H8_MFMA(i0, j, fbv)                                                                                                   \
  if (DBG != 1) {                                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                        \
      acc[i0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[0][ks]), __builtin_bit_cast(bf16x8, fbv[ks]), acc[i0][j], 0, 0, 0); \
      acc[i0 + 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[1][ks]), __builtin_bit_cast(bf16x8, fbv[ks]), acc[i0 + 1][j], 0, 0, 0); \
    }                                                                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                                                            \
  } else {                                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) { asm volatile("" ::"v"(fa[0][ks]), "v"(fa[1][ks]), "v"(fbv[ks])); }         \
  }
  // one K-tile = (K-step in panel PANEL, tap TAP), B tile in buffer BUF; t = its index, more_k: another K-step follows this one
  auto tile = [&](auto panelc, auto bufc, auto tapc, int t, bool more_k) {
    constexpr int PANEL = decltype(panelc)::value, BUF = decltype(bufc)::value, TAP = decltype(tapc)::value;
    constexpr int OA = PANEL * kPanelBytes, OB = BUF * 32768;
    // fragment rows of this tap: panel row wr * 128 + i * 32 + lr + TAP (the swizzle follows the shifted row)
    uint32_t ka[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      ka[ks] = lds0 + (uint32_t)((wr * 128 + lr + TAP) * 128) + (uint32_t)(((2 * ks + lh) ^ (((lr + TAP) >> 1) & 7)) << 4);
    // ---- P1: B0, A rows 0..63 -> acc[0..1][0]; a piece of the next panel
    lds_read_b128<OB>(fb0[0], kb_[0]); lds_read_b128<OB>(fb0[1], kb_[1]); lds_read_b128<OB>(fb0[2], kb_[2]); lds_read_b128<OB>(fb0[3], kb_[3]);
    __builtin_amdgcn_sched_barrier(0);
    lds_read_b128<OA>(fa[0][0], ka[0]); lds_read_b128<OA>(fa[0][1], ka[1]); lds_read_b128<OA>(fa[0][2], ka[2]); lds_read_b128<OA>(fa[0][3], ka[3]);
    lds_read_b128<OA + 4096>(fa[1][0], ka[0]); lds_read_b128<OA + 4096>(fa[1][1], ka[1]); lds_read_b128<OA + 4096>(fa[1][2], ka[2]);
    lds_read_b128<OA + 4096>(fa[1][3], ka[3]);
    if (more_k) {
      if (TAP == 0) stage_a(IntC<0>{}, IntC<PANEL ^ 1>{});
      if (TAP == 1) stage_a(IntC<2>{}, IntC<PANEL ^ 1>{});
      if (TAP == 2 && wave == 0) stage_a(IntC<4>{}, IntC<PANEL ^ 1>{});
    }
    H8_WAIT_B(8, fb0)                                           // B0's reads are done before anyone passes the barrier: P2 re-stages B0's slot
    __builtin_amdgcn_s_barrier();
    H8_WAIT_A(0)
    __builtin_amdgcn_sched_barrier(0);
    H8_MFMA(0, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P2: B1 -> acc[0..1][1]; stage B0(t + 2)
    lds_read_b128<OB + 4096>(fb1[0], kb_[0]); lds_read_b128<OB + 4096>(fb1[1], kb_[1]); lds_read_b128<OB + 4096>(fb1[2], kb_[2]); lds_read_b128<OB + 4096>(fb1[3], kb_[3]);
    if (t + 2 < n) stage_b(IntC<0>{}, IntC<BUF>{});
    __builtin_amdgcn_s_barrier();
    H8_WAIT_B(0, fb1)
    __builtin_amdgcn_sched_barrier(0);
    H8_MFMA(0, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P3: A rows 64..127 -> acc[2..3][1]; a piece of the next panel
    lds_read_b128<OA + 8192>(fa[0][0], ka[0]); lds_read_b128<OA + 8192>(fa[0][1], ka[1]); lds_read_b128<OA + 8192>(fa[0][2], ka[2]);
    lds_read_b128<OA + 8192>(fa[0][3], ka[3]);
    lds_read_b128<OA + 12288>(fa[1][0], ka[0]); lds_read_b128<OA + 12288>(fa[1][1], ka[1]); lds_read_b128<OA + 12288>(fa[1][2], ka[2]);
    lds_read_b128<OA + 12288>(fa[1][3], ka[3]);
    if (more_k) {
      if (TAP == 0) stage_a(IntC<1>{}, IntC<PANEL ^ 1>{});
      if (TAP == 1) stage_a(IntC<3>{}, IntC<PANEL ^ 1>{});
    }
    __builtin_amdgcn_s_barrier();
    H8_WAIT_A(0)
    __builtin_amdgcn_sched_barrier(0);
    H8_MFMA(2, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- P4: (B0 still in registers) -> acc[2..3][0]; stage B1(t + 2); B(t + 1) -- at tap 2 also the next panel -- has landed behind the wait
    if (t + 2 < n) {
      stage_b(IntC<1>{}, IntC<BUF>{});
      // younger than B1(t + 1): this tile's panel pieces (taps 0, 1: two) and B0 / B1(t + 2) (two each); tap 2's piece must land too
      if (TAP < 2 && more_k) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    H8_MFMA(2, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // prologue: panel 0, B tiles 0 and 1
  stage_a(IntC<0>{}, IntC<0>{}); stage_a(IntC<1>{}, IntC<0>{}); stage_a(IntC<2>{}, IntC<0>{}); stage_a(IntC<3>{}, IntC<0>{});
  if (wave == 0) stage_a(IntC<4>{}, IntC<0>{});
  stage_b(IntC<0>{}, IntC<0>{}); stage_b(IntC<1>{}, IntC<0>{});
  stage_b(IntC<0>{}, IntC<1>{}); stage_b(IntC<1>{}, IntC<1>{});
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // panel 0 and B tile 0; B tile 1 stays in flight
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                   // waves 4-7 run one barrier behind
  int kb = 0;
  for (; kb + 1 < ksteps; kb += 2) {
    tile(IntC<0>{}, IntC<0>{}, IntC<0>{}, 3 * kb, true);
    tile(IntC<0>{}, IntC<1>{}, IntC<1>{}, 3 * kb + 1, true);
    tile(IntC<0>{}, IntC<0>{}, IntC<2>{}, 3 * kb + 2, true);
    const bool more = kb + 2 < ksteps;
    tile(IntC<1>{}, IntC<1>{}, IntC<0>{}, 3 * kb + 3, more);
    tile(IntC<1>{}, IntC<0>{}, IntC<1>{}, 3 * kb + 4, more);
    tile(IntC<1>{}, IntC<1>{}, IntC<2>{}, 3 * kb + 5, more);
  }
  if (kb < ksteps) {
    tile(IntC<0>{}, IntC<0>{}, IntC<0>{}, 3 * kb, false);
    tile(IntC<0>{}, IntC<1>{}, IntC<1>{}, 3 * kb + 1, false);
    tile(IntC<0>{}, IntC<0>{}, IntC<2>{}, 3 * kb + 2, false);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef H8_WAIT_A
#undef H8_WAIT_B
#undef H8_MFMA
  gemm_big_epilogue<T>(g, acc, smem, m0, n0, tid, lane, wave, wr, wc);
}

// ---------------------------------------------------------------------------------------------------------
// Skinny-N variant (N <= 32: the rank-r LoRA products u = A(x) and du = dy (s g B), functional.adapter_grads): a 256 x 32 tile per
// workgroup -- 8 waves x one 32x32 MFMA tile -- so the DMA traffic is the A panel only (the 128 / 256-wide tiles spend 4-8x the
// MFMAs and B-side DMA slots on zero columns).  Same loader (per-tap row maps, incremental pointers), 2 x 36 KiB ring -> two
// workgroups per CU.  No epilogue options: C = A W^T in the storage type.
// ---------------------------------------------------------------------------------------------------------
static constexpr int kSkStage = kBig * 128 + 32 * 128;    // A 256 rows + B 32 rows, 128 B of K each

__global__ __launch_bounds__(512, 2) void gemm_nt_skinny_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef bf16_t T;
  constexpr int BK = 64, EPC = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * kBig;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  int a_base[4], a_pos[4], koff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    koff[i] = ((lane & 7) ^ ((row >> 1) & 7)) * EPC;
    const int m = m0 + row;
    if (m < g.M) { int b = m / g.rm.Lout; a_base[i] = b * g.rm.Lin; a_pos[i] = m - b * g.rm.Lout; }
    else { a_base[i] = 0; a_pos[i] = -1; }
  }
  // B loader: waves 0-3, one instruction each (8 rows); its swizzle follows the B tile's own row index
  const int brow = wave * 8 + (lane >> 3);
  const int bkoff = ((lane & 7) ^ ((brow >> 1) & 7)) * EPC;
  const bool b_ok = wave < 4 && brow < g.N;
  const int ksteps = (g.K + BK - 1) / BK;
  const int nsteps = g.taps * ksteps;
  const bool ktail = (g.K % BK) != 0;

  const char* pa[4];
  const char* pb = zero;
  int ia[4], ib = 0;
  auto set_tap = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pa[i] = zero; ia[i] = 0;
      if (a_pos[i] >= 0) {
        int s2 = map_row(g.rm, a_pos[i], t);
        if (s2 >= 0) { pa[i] = reinterpret_cast<const char*>(A + (long)(a_base[i] + s2) * g.lda + koff[i]); ia[i] = BK * (int)sizeof(T); }
      }
    }
    pb = zero; ib = 0;
    if (b_ok) { pb = reinterpret_cast<const char*>(W + (long)t * g.tapstride + (long)brow * g.ldw + bkoff); ib = BK * (int)sizeof(T); }
  };
  int itap = 0, ikb = 0;
  set_tap(0);
  auto issue = [&](int buf) {
    char* sa = smem + buf * kSkStage + wave * 4096;
    char* sb = smem + buf * kSkStage + kBig * 128 + wave * 1024;
    const bool last = ktail && ikb == ksteps - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* qa = pa[i];
      if (last && ikb * BK + koff[i] >= g.K) qa = zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qa, (las_ptr)(sa + i * 1024), 16, 0, 0);
      pa[i] += ia[i];
    }
    if (wave < 4) {
      const char* qb = pb;
      if (last && ikb * BK + bkoff >= g.K) qb = zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qb, (las_ptr)sb, 16, 0, 0);
      pb += ib;
    }
    if (++ikb == ksteps) {
      ikb = 0;
      if (++itap < g.taps) set_tap(itap);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  issue(0);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
    if (step + 1 < nsteps) issue(buf ^ 1);
    const char* sa = smem + buf * kSkStage;
    const char* sb = sa + kBig * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 fa = *reinterpret_cast<const u32x4*>(sa + swz_off(wave * 32 + lr, 2 * ks + lh));
      const u32x4 fb = *reinterpret_cast<const u32x4*>(sb + swz_off(lr, 2 * ks + lh));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: 32x32 fp32 through a 4 KiB per-wave LDS slice -> 8 columns per lane (4 lanes per row, 16 rows per pass)
  float* cs = reinterpret_cast<float*>(smem + wave * 4096);
#pragma unroll
  for (int r = 0; r < 16; ++r) cs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[r];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  T* C = reinterpret_cast<T*>(g.C);
  const int col8 = (lane & 3) * 8;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = it * 16 + (lane >> 2);
    const int m = m0 + wave * 32 + row;
    if (m < g.M && col8 < g.N) {
      float v[8];
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(cs + row * 32 + col8);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(cs + row * 32 + col8 + 4);
      v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3]; v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
      store8(C + (long)m * g.ldc + col8, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad: dW[t][n1][n2] += sum_m dY[m][n1] * X[rowmap(m,t)][n2]
// ---------------------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* dY; const void* X; float* dW;
  float* ws;                       // optional [splits][taps][N1][N2] partial tiles (plain stores) instead of atomics
  int ws_pk;                       // the partial tiles are bf16 pairs: dword [split][tap][n1 / 2][n2] = (row n1 even | row n1 + 1 << 16)  (round 4)
  long es;                         // element stride of n2 in dW (1, or `taps` for torch's (Cout, Cin, k) weight layout)
  long ldy, ldx, ldw, tapstride;
  int M, N1, N2, taps;
  RowMap rm;
  int rows_per_split;
  float* dbias;                    // optional: column sums of dY (the bias gradient) += , from the dY fragments the kernel already holds
  int xcd_chunks;                  // 256x256 wgrad kernels: unit list dealt to the XCDs in eighths (1) or round-robin (0: A/B switch OSUF_TN_RR)
};

// bf16 tile: [64 rows][128 cols] (256 B/row), byte-in-row ^= (row&3)<<6 -> ds_read_b64_tr_b16 conflict-free
// f32  tile: [32 rows][128 cols] (512 B/row), plain ds_read_b32 (lanes = consecutive columns)
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool kBF = sizeof(T) == 2;
  constexpr int BKM = kBF ? 64 : 32;                // rows of m per step
  constexpr int EPC = ElemTraits<T>::kPer16B;
  constexpr int CPR = kTile / EPC;                  // 16-B chunks per tile row: 16 (bf16) / 32 (f32)
  constexpr int ROWB = kTile * (int)sizeof(T);      // bytes per tile row
  constexpr int TILEB = BKM * ROWB;                 // 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n2 = (g.N2 + kTile - 1) / kTile;
  const int n1_0 = (blockIdx.x / tiles_n2) * kTile, n2_0 = (blockIdx.x % tiles_n2) * kTile;
  const int t = blockIdx.y;
  const int m_begin = blockIdx.z * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  if (m_begin >= m_end) return;
  const T* dY = reinterpret_cast<const T*>(g.dY);
  const T* X = reinterpret_cast<const T*>(g.X);

  const int c = tid % CPR, r0 = tid / CPR;          // rows r0 + (256/CPR)*i
  constexpr int RSTEP = 256 / CPR;                  // 16 (bf16) / 8 (f32)
  const bool y_ok = (n1_0 + c * EPC) < g.N1, x_ok = (n2_0 + c * EPC) < g.N2;

  u32x4 ry0[4], rx0[4], ry1[4], rx1[4];
  auto load_regs = [&](int mb, u32x4 (&ry)[4], u32x4 (&rx)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 z = {0u, 0u, 0u, 0u};
      ry[i] = z; rx[i] = z;
      int m = mb + r0 + RSTEP * i;
      if (m < m_end) {
        if (y_ok) ry[i] = *reinterpret_cast<const u32x4*>(dY + (long)m * g.ldy + n1_0 + c * EPC);
        if (x_ok) {
          int b = m / g.rm.Lout;
          int s = map_row(g.rm, m - b * g.rm.Lout, t);
          if (s >= 0) rx[i] = *reinterpret_cast<const u32x4*>(X + (long)(b * g.rm.Lin + s) * g.ldx + n2_0 + c * EPC);
        }
      }
    }
  };
  auto store_lds = [&](int buf, const u32x4 (&ry)[4], const u32x4 (&rx)[4]) {
    char* sy = smem + buf * 2 * TILEB;
    char* sx = sy + TILEB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = r0 + RSTEP * i;
      int off = row * ROWB + (kBF ? ((c * 16) ^ ((row & 3) << 6)) : c * 16);
      *reinterpret_cast<u32x4*>(sy + off) = ry[i];
      *reinterpret_cast<u32x4*>(sx + off) = rx[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  // transposed-read lane roles (ds_read_b64_tr_b16 works per 16-lane group; see cdna_hip_programming.md T10)
  const int ip = lane & 15, cb = ((lane >> 4) & 1) * 16, tq = ip >> 2, tp = ip & 3;

  auto compute = [&](int buf) {
    const char* sy = smem + buf * 2 * TILEB;
    const char* sx = sy + TILEB;
    if constexpr (kBF) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          s16x4 lo[2], hi[2];
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            int row = 16 * ks + 8 * lh + 4 * q4 + tq;
            int sw = (row & 3) << 6;
            int cola = (wr * 64 + i * 32 + cb + 4 * tp) * 2;
            int colb = (wc * 64 + i * 32 + cb + 4 * tp) * 2;
            lo[q4] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(sy + row * ROWB + (cola ^ sw)));
            hi[q4] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(sx + row * ROWB + (colb ^ sw)));
          }
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          s16x8 va = {lo[0][0], lo[0][1], lo[0][2], lo[0][3], lo[1][0], lo[1][1], lo[1][2], lo[1][3]};
          s16x8 vb = {hi[0][0], hi[0][1], hi[0][2], hi[0][3], hi[1][0], hi[1][1], hi[1][2], hi[1][3]};
          fa[i] = __builtin_bit_cast(bf16x8, va);
          fb[i] = __builtin_bit_cast(bf16x8, vb);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (SPLIT) {
      // two 16-deep bf16 k-steps per 32-row stage: lane (r, h) takes rows 16 kk + 8 h .. +7 of its column (one ds_read_b32 each)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float xa[8], xb[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int row = 16 * kk + 8 * lh + e;
            xa[e] = *reinterpret_cast<const float*>(sy + row * ROWB + (wr * 64 + i * 32 + lr) * 4);
            xb[e] = *reinterpret_cast<const float*>(sx + row * ROWB + (wc * 64 + i * 32 + lr) * 4);
          }
          split_bf16x8(xa, ah[i], al[i]);
          split_bf16x8(xb, bh[i], bl[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mfma_x3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int row = 2 * ks + lh;
        float fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const float*>(sy + row * ROWB + (wr * 64 + i * 32 + lr) * 4);
          fb[i] = *reinterpret_cast<const float*>(sx + row * ROWB + (wc * 64 + i * 32 + lr) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  load_regs(m_begin, ry0, rx0);
  if (m_begin + BKM < m_end) load_regs(m_begin + BKM, ry1, rx1);
  store_lds(0, ry0, rx0);
  __syncthreads();
  for (int mb = m_begin; mb < m_end; mb += 2 * BKM) {
    if (mb + 2 * BKM < m_end) load_regs(mb + 2 * BKM, ry0, rx0);
    compute(0);
    if (mb + BKM < m_end) store_lds(1, ry1, rx1);
    __syncthreads();
    if (mb + BKM >= m_end) break;
    if (mb + 3 * BKM < m_end) load_regs(mb + 3 * BKM, ry1, rx1);
    compute(1);
    if (mb + 2 * BKM < m_end) store_lds(0, ry0, rx0);
    __syncthreads();
  }

  float* dW = g.dW + (long)t * g.tapstride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n2 = n2_0 + wc * 64 + j * 32 + lr;
      if (n2 < g.N2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int n1 = n1_0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n1 < g.N1) atomic_add_f32(dW + (long)n1 * g.ldw + (long)n2 * g.es, acc[i][j][r]);
        }
      }
    }
}

// the 256 x 256 wgrad tile leaves either as a partial tile of its m-split (plain 128-B-segment stores, 5x the fp32-atomic rate; summed
// by wgrad_reduce_kernel) or by fp32 atomics into dW
// (IL: the x3 kernel's n1 mapping -- MFMA row rho of tile i is column 4 rho + i of the wave's 128, see its fragment reads)
template <bool IL = false>
__device__ __forceinline__ void tn_big_store(const WgradArgs& g, const f32x16 (&acc)[4][2], int split, int t, int n1_0, int n2_0, int wr, int wc,
                                             int lr, int lh) {
  auto n1_of = [&](int i, int r) {
    const int rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
    return n1_0 + wr * 128 + (IL ? 4 * rho + i : i * 32 + rho);
  };
  if (g.ws && g.ws_pk && !IL) {
    // bf16 partial tiles: registers r, r + 1 (r even) are rows n1, n1 + 1 of this lane's column -> one dword; half the bytes and half the
    // store instructions of the fp32 tiles (19 GB of partials per step were written here and read back by the reduce kernels)
    uint32_t* out = reinterpret_cast<uint32_t*>(g.ws) + ((long)split * g.taps + t) * (g.N1 >> 1) * g.N2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n2 = n2_0 + wc * 64 + j * 32 + lr;
        if (n2 < g.N2) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int n1 = n1_of(i, r);
            if (n1 < g.N1) out[(long)(n1 >> 1) * g.N2 + n2] = pack_bf16x2(acc[i][j][r], acc[i][j][r + 1]);
          }
        }
      }
    return;
  }
  if (g.ws) {
    float* out = g.ws + ((long)split * g.taps + t) * g.N1 * g.N2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n2 = n2_0 + wc * 64 + j * 32 + lr;
        if (n2 < g.N2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n1 = n1_of(i, r);
            if (n1 < g.N1) out[(long)n1 * g.N2 + n2] = acc[i][j][r];
          }
        }
      }
    return;
  }
  float* dW = g.dW + (long)t * g.tapstride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n2 = n2_0 + wc * 64 + j * 32 + lr;
      if (n2 < g.N2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n1 = n1_of(i, r);
          if (n1 < g.N1) atomic_add_f32(dW + (long)n1 * g.ldw + (long)n2 * g.es, acc[i][j][r]);
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 256x256 output tile wgrad (bf16): 8 waves (2x4), wave tile 128 (n1) x 64 (n2), 64 rows of m per step, LDS-DMA staging into a
// 2 x 64 KiB ring.  Tiles are row-major as stored ([64 rows][256 cols] = 512 B per row); chunk ^= (row&3)<<2 keeps the
// ds_read_b64_tr_b16 fragment reads conflict-free.  The transposed reads are issued through inline asm (hipcc guards the
// ds_read_tr16 intrinsic with vmcnt(0) while an LDS-DMA is in flight) and software-pipelined one k-step ahead of the MFMAs.
// ---------------------------------------------------------------------------------------------------------
// sum of the 8 bf16 of an A-operand fragment (two 8-byte halves), added to acc: v_dot2c_f32_bf16 against (1, 1).
// (The words are taken out of the vectors as scalars first: `__builtin_bit_cast(bf2, lo[1])` on a `const u32x2&` made hipcc use word 0
//  twice -- rows 0, 1 counted double, rows 2, 3 never; found by the row-indicator case of the parity test.)
__device__ __forceinline__ float bf16x2_sum(unsigned w, float acc) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, w), __builtin_bit_cast(bf2, 0x3F803F80u), acc, false);
}
__device__ __forceinline__ float bf16x8_sum(u32x2 lo, u32x2 hi, float acc) {
  const unsigned w0 = lo.x, w1 = lo.y, w2 = hi.x, w3 = hi.y;
  return bf16x2_sum(w3, bf16x2_sum(w2, bf16x2_sum(w1, bf16x2_sum(w0, acc))));
}

#define OSUF_TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

__global__ __launch_bounds__(512, 2) void gemm_tn_big_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKM = 64, ROWB = 512, TILEB = BKM * ROWB;      // 32 KiB per operand tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n2 = (g.N2 + kBig - 1) / kBig;
  const int tiles_n1 = (g.N1 + kBig - 1) / kBig;
  // XCD-aware order (speed only): the taps of one (split, tile) unit re-read identical dY / X bytes; give them ids 8 apart
  // so they share one XCD's L2 and run back to back, and deal the units themselves round-robin over the 8 XCDs.
  const int ntile = tiles_n1 * tiles_n2;
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  // (units are (split, tile) pairs, split-major: XCD x takes the x-th eighth of the list, so the tiles of one m-split -- which re-read the
  //  same dY / X rows -- and their taps run back to back on one XCD.  The first form, unit = (qid / taps) * 8 + xcd, dealt a split's tiles
  //  to eight different L2s: PMC FETCH_SIZE of these launches summed to 4.5 TB/s HBM-side, twice the algorithmic bytes.)
  const int unit = g.xcd_chunks ? xcd * (((int)gridDim.x >> 3) / g.taps) + qid / g.taps : (qid / g.taps) * 8 + xcd;
  const int t = qid % g.taps;
  const int split = unit / ntile, tile = unit % ntile;
  const int n1_0 = (tile / tiles_n2) * kBig, n2_0 = (tile % tiles_n2) * kBig;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  if (m_begin >= m_end) return;
  const bf16_t* dY = reinterpret_cast<const bf16_t*>(g.dY);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // DMA role: instruction i of this wave fills tile rows (wave*4+i)*2 + (lane>>5), LDS chunk position lane&31
  int srow[4], scol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    srow[i] = (wave * 4 + i) * 2 + (lane >> 5);
    scol[i] = ((lane & 31) ^ ((srow[i] & 3) << 2)) * 8;        // logical column (elements) stored at this LDS position
  }
  // Per-lane DMA sources, advanced incrementally: 64 rows of m per step.  (sample, position) of each lane's row are carried
  // instead of re-divided every step, and the dY pointer is a plain `+= 64 rows` (same triage finding as gemm_nt_big_kernel).
  int sb_[4], sp_[4];
  const char* py[4];
  bool y_ok[4], x_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_begin + srow[i];
    sb_[i] = m / g.rm.Lout;
    sp_[i] = m - sb_[i] * g.rm.Lout;
    y_ok[i] = n1_0 + scol[i] < g.N1;
    x_ok[i] = n2_0 + scol[i] < g.N2;
    py[i] = reinterpret_cast<const char*>(dY + (long)m * g.ldy + n1_0 + scol[i]);
  }
  const long ystep = (long)BKM * g.ldy * (long)sizeof(bf16_t);
  auto issue = [&](int mb, int buf) {
    char* sy = smem + buf * 2 * TILEB + wave * 4096;
    char* sx = sy + TILEB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = mb + srow[i] < m_end;
      const char* qy = (in && y_ok[i]) ? py[i] : zero;
      const char* qx = zero;
      if (in && x_ok[i]) {
        const int s2 = map_row(g.rm, sp_[i], t);
        if (s2 >= 0) qx = reinterpret_cast<const char*>(X + (long)(sb_[i] * g.rm.Lin + s2) * g.ldx + n2_0 + scol[i]);
      }
      __builtin_amdgcn_global_load_lds((gas_ptr)qy, (las_ptr)(sy + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gas_ptr)qx, (las_ptr)(sx + i * 1024), 16, 0, 0);
      py[i] += ystep;
      sp_[i] += BKM;
      while (sp_[i] >= g.rm.Lout) { sp_[i] -= g.rm.Lout; ++sb_[i]; }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const int ip = lane & 15, cb = ((lane >> 4) & 1) * 16, tq = ip >> 2, tp = ip & 3;
  // per-lane byte offsets of the transposed reads (k-step / q4 row bases are immediates)
  uint32_t offA[4], offB[2];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
#pragma unroll
  for (int i = 0; i < 4; ++i) offA[i] = (8 * lh + tq) * ROWB + ((((wr * 128 + i * 32 + cb + 4 * tp) * 2)) ^ (tq << 6));
#pragma unroll
  for (int j = 0; j < 2; ++j) offB[j] = TILEB + (8 * lh + tq) * ROWB + ((((wc * 64 + j * 32 + cb + 4 * tp) * 2)) ^ (tq << 6));

  typedef u32x2 frag_half;
  frag_half fa[2][4][2], fb[2][2][2];                           // [pipeline slot][tile][lo/hi]
#define TN_READS(slot, ks, base)                                                                                   \
  OSUF_TR_READ(fa[slot][0][0], base + offA[0], (ks) * 16 * 512); OSUF_TR_READ(fa[slot][0][1], base + offA[0], ((ks) * 16 + 4) * 512); \
  OSUF_TR_READ(fa[slot][1][0], base + offA[1], (ks) * 16 * 512); OSUF_TR_READ(fa[slot][1][1], base + offA[1], ((ks) * 16 + 4) * 512); \
  OSUF_TR_READ(fa[slot][2][0], base + offA[2], (ks) * 16 * 512); OSUF_TR_READ(fa[slot][2][1], base + offA[2], ((ks) * 16 + 4) * 512); \
  OSUF_TR_READ(fa[slot][3][0], base + offA[3], (ks) * 16 * 512); OSUF_TR_READ(fa[slot][3][1], base + offA[3], ((ks) * 16 + 4) * 512); \
  OSUF_TR_READ(fb[slot][0][0], base + offB[0], (ks) * 16 * 512); OSUF_TR_READ(fb[slot][0][1], base + offB[0], ((ks) * 16 + 4) * 512); \
  OSUF_TR_READ(fb[slot][1][0], base + offB[1], (ks) * 16 * 512); OSUF_TR_READ(fb[slot][1][1], base + offB[1], ((ks) * 16 + 4) * 512);
#define TN_WAIT(slot, n)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                        \
               : "+v"(fa[slot][0][0]), "+v"(fa[slot][0][1]), "+v"(fa[slot][1][0]), "+v"(fa[slot][1][1]), "+v"(fa[slot][2][0]),  \
                 "+v"(fa[slot][2][1]), "+v"(fa[slot][3][0]), "+v"(fa[slot][3][1]), "+v"(fb[slot][0][0]), "+v"(fb[slot][0][1]),  \
                 "+v"(fb[slot][1][0]), "+v"(fb[slot][1][1]));                                                     \
  __builtin_amdgcn_sched_barrier(0);
#define TN_MFMA(slot)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                   \
    u32x4 va = {fa[slot][i][0][0], fa[slot][i][0][1], fa[slot][i][1][0], fa[slot][i][1][1]};                       \
    u32x4 vb = {fb[slot][j][0][0], fb[slot][j][0][1], fb[slot][j][1][0], fb[slot][j][1][1]};                       \
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc[i][j], 0, 0, 0); \
  }

  // bias gradient (g.dbias): the column sums of dY are the row sums of the dY^T fragments this wave holds anyway -- one wave column of
  // the workgroups of the first n2 tile and tap adds them up (4 v_dot2c per fragment), replacing a separate pass over dY (osuf_colsum)
  const bool do_bias = g.dbias != nullptr && n2_0 == 0 && t == 0 && wc == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#define TN_BIAS(slot)                                                                                             \
  if (do_bias) { _Pragma("unroll") for (int i = 0; i < 4; ++i) bsum[i] = bf16x8_sum(fa[slot][i][0], fa[slot][i][1], bsum[i]); }
  issue(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int mb = m_begin; mb < m_end; mb += BKM, buf ^= 1) {
    const bool more = mb + BKM < m_end;
    if (more) issue(mb + BKM, buf ^ 1);
    const uint32_t base = lds0 + buf * 2 * TILEB;
    TN_READS(0, 0, base)
    TN_READS(1, 1, base)
    TN_WAIT(0, 12)
    TN_MFMA(0)
    TN_BIAS(0)
    TN_READS(0, 2, base)
    TN_WAIT(1, 12)
    TN_MFMA(1)
    TN_BIAS(1)
    TN_READS(1, 3, base)
    TN_WAIT(0, 12)
    TN_MFMA(0)
    TN_BIAS(0)
    TN_WAIT(1, 0)
    TN_MFMA(1)
    TN_BIAS(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef TN_READS
#undef TN_WAIT
#undef TN_MFMA
#undef TN_BIAS
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);        // the two k halves of the fragment rows
      const int n1 = n1_0 + wr * 128 + i * 32 + lr;
      if (lh == 0 && n1 < g.N1) atomic_add_f32(g.dbias + n1, v);
    }
  }

  tn_big_store(g, acc, split, t, n1_0, n2_0, wr, wc, lr, lh);
}

// ---------------------------------------------------------------------------------------------------------
// The same 256 x 256 wgrad tile for OSUF_DT_F32X3: fp32 stages of 32 rows of m ([32][256] floats = 1 KiB per row and operand, one
// LDS-DMA instruction per row), fragments gathered down the columns with ds_read_b32 (lanes = consecutive columns: conflict-free
// without a swizzle), split into bf16 hi + lo in registers, three bf16 MFMAs per product (mfma_x3).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn_big_x3_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKM = 32, ROWB = 1024, TILEB = BKM * ROWB;      // 32 KiB per operand tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n2 = (g.N2 + kBig - 1) / kBig;
  const int tiles_n1 = (g.N1 + kBig - 1) / kBig;
  const int ntile = tiles_n1 * tiles_n2;                        // block -> (split, tile, tap) as in gemm_tn_big_kernel
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int unit = g.xcd_chunks ? xcd * (((int)gridDim.x >> 3) / g.taps) + qid / g.taps : (qid / g.taps) * 8 + xcd;
  const int t = qid % g.taps;
  const int split = unit / ntile, tile = unit % ntile;
  const int n1_0 = (tile / tiles_n2) * kBig, n2_0 = (tile % tiles_n2) * kBig;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  if (m_begin >= m_end) return;
  const float* dY = reinterpret_cast<const float*>(g.dY);
  const float* X = reinterpret_cast<const float*>(g.X);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // DMA role: instruction i of this wave fills tile row wave*4 + i, this lane its 16-B chunk `lane` (4 columns)
  const int scol = lane * 4;
  const bool y_ok = n1_0 + scol < g.N1, x_ok = n2_0 + scol < g.N2;
  int sb_[4], sp_[4];
  const char* py[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_begin + wave * 4 + i;
    sb_[i] = m / g.rm.Lout;
    sp_[i] = m - sb_[i] * g.rm.Lout;
    py[i] = reinterpret_cast<const char*>(dY + (long)m * g.ldy + n1_0 + scol);
  }
  const long ystep = (long)BKM * g.ldy * (long)sizeof(float);
  auto issue = [&](int mb, int buf) {
    char* sy = smem + buf * 2 * TILEB + wave * 4096;
    char* sx = sy + TILEB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = mb + wave * 4 + i < m_end;
      const char* qy = (in && y_ok) ? py[i] : zero;
      const char* qx = zero;
      if (in && x_ok) {
        const int s2 = map_row(g.rm, sp_[i], t);
        if (s2 >= 0) qx = reinterpret_cast<const char*>(X + (long)(sb_[i] * g.rm.Lin + s2) * g.ldx + n2_0 + scol);
      }
      __builtin_amdgcn_global_load_lds((gas_ptr)qy, (las_ptr)(sy + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gas_ptr)qx, (las_ptr)(sx + i * 1024), 16, 0, 0);
      py[i] += ystep;
      sp_[i] += BKM;
      while (sp_[i] >= g.rm.Lout) { sp_[i] -= g.rm.Lout; ++sb_[i]; }
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  issue(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int mb = m_begin; mb < m_end; mb += BKM, buf ^= 1) {
    if (mb + BKM < m_end) issue(mb + BKM, buf ^ 1);
    const char* sy = smem + buf * 2 * TILEB;
    const char* sx = sy + TILEB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 ah[4], al[4], bh[2], bl[2];
      // dY side: one ds_read_b128 per row of m hands this lane columns 4 lr .. 4 lr + 3 of the wave's 128 = its k-element for all
      // four M-tiles (tile i owns the columns congruent i mod 4; tn_big_store<true> undoes the interleave) -- 8 wide reads instead of
      // 32 ds_read_b32, which had the LDS array as busy as the matrix cores.  X side: lanes = consecutive columns (coalesced stores).
      {
        f32x4 y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = *reinterpret_cast<const f32x4*>(sy + (16 * kk + 8 * lh + e) * ROWB + (wr * 128 + 4 * lr) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x[8] = {y[0][i], y[1][i], y[2][i], y[3][i], y[4][i], y[5][i], y[6][i], y[7][i]};
          split_bf16x8(x, ah[i], al[i]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<const float*>(sx + (16 * kk + 8 * lh + e) * ROWB + (wc * 64 + j * 32 + lr) * 4);
        split_bf16x8(x, bh[j], bl[j]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mfma_x3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  tn_big_store<true>(g, acc, split, t, n1_0, n2_0, wr, wc, lr, lh);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the k = 3 'same' convolutions with the three taps in ONE workgroup (mode 0, stride 1, pad 1, L % 128 == 0): a
// 128 (n1) x 128 (n2) tile per tap, 8 waves (2 x 4), wave tile 64 x 32 x 3 taps = 96 accumulator registers.  A stage is 128 rows of m:
// the dY tile (32 KiB) and ONE X panel of 130 rows (rows -1 .. 128; the two halo rows come from the zero page at a sample edge, a stage
// never straddles samples) that serves all three taps -- 65 KiB through the L2 -> LDS feed per 6.3 M MACs, two thirds of the bytes per MAC
// of gemm_tn_big_kernel's one-workgroup-per-tap tiles (the loop is bound by that feed: DESIGN.md section 4).  Tiles are row-major as
// stored (256 B per row), byte-in-row ^= (row & 3) << 6 keeps the ds_read_b64_tr_b16 fragment reads conflict-free; transposed reads by
// inline asm, software-pipelined one k-step ahead, as in gemm_tn_big_kernel.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn_taps3_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKM = 128, ROWB = 256, YB = BKM * ROWB, XROWS = 132, STAGE = YB + XROWS * ROWB;      // 32 KiB + 33 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n2 = (g.N2 + 127) / 128, tiles_n1 = (g.N1 + 127) / 128;
  const int ntile = tiles_n1 * tiles_n2;
  // XCD-aware order (speed only): the tiles of one m-split re-read the same dY / X rows.  Block ids are dealt round-robin to the 8 XCDs, so
  // XCD x takes the x-th eighth of the (split-major) unit list: a split's tiles run back to back on one XCD and meet in its L2
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int unit = xcd * ((int)gridDim.x >> 3) + local;
  const int split = unit / ntile, tile = unit % ntile;
  const int n1_0 = (tile / tiles_n2) * 128, n2_0 = (tile % tiles_n2) * 128;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  if (m_begin >= m_end) return;
  const bf16_t* dY = reinterpret_cast<const bf16_t*>(g.dY);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int L = g.rm.Lout;

  // DMA roles: an instruction moves 4 rows x 256 B; instruction i < 4 of wave w is tile / panel rows (4 w + i) * 4 + (lane >> 4);
  // wave 0's fifth X instruction is panel rows 128 .. 131.  LDS position lane & 15 of a row holds logical chunk (lane & 15) ^ ((row & 3) << 2).
  int yrow[4], prow[5], ycol[4], pcol[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = ((i < 4 ? wave * 4 + i : 32) * 4) + (lane >> 4);
    const int col = ((lane & 15) ^ ((r & 3) << 2)) * 8;
    prow[i] = r; pcol[i] = col;
    if (i < 4) { yrow[i] = r; ycol[i] = col; }
  }
  int pos = m_begin % L;                                      // position of the stage's first row inside its sample
  auto issue = [&](int mb, int buf) {
    char* sy = smem + buf * STAGE;
    char* sx = sy + YB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + yrow[i];
      const char* qy = (m < m_end && n1_0 + ycol[i] < g.N1) ? reinterpret_cast<const char*>(dY + (long)m * g.ldy + n1_0 + ycol[i]) : zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qy, (las_ptr)(sy + (wave * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i == 4 && wave != 0) break;
      const int pr = prow[i];                                 // panel row pr = X row mb - 1 + pr
      const int q = pos - 1 + pr;
      const bool ok = pr < 130 && q >= 0 && q < L && mb - 1 + pr < g.M && n2_0 + pcol[i] < g.N2;
      const char* qx = ok ? reinterpret_cast<const char*>(X + (long)(mb - 1 + pr) * g.ldx + n2_0 + pcol[i]) : zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qx, (las_ptr)(sx + (i < 4 ? wave * 4 + i : 32) * 1024), 16, 0, 0);
    }
    pos += BKM;
    if (pos >= L) pos -= L;
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const int ip = lane & 15, cb = ((lane >> 4) & 1) * 16, tq = ip >> 2, tp = ip & 3;
  uint32_t offA[2], offB[3];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (8 * lh + tq) * ROWB + (((wr * 64 + i * 32 + cb + 4 * tp) * 2) ^ (tq << 6));
#pragma unroll
  for (int t = 0; t < 3; ++t) offB[t] = YB + (8 * lh + tq + t) * ROWB + (((wc * 32 + cb + 4 * tp) * 2) ^ (((tq + t) & 3) << 6));

  typedef u32x2 frag_half;
  frag_half fa[2][2][2], fb[2][3][2];                           // [pipeline slot][tile / tap][lo/hi]
#define T3_READS(slot, ks, base)                                                                                   \
  OSUF_TR_READ(fa[slot][0][0], base + offA[0], (ks) * 16 * 256); OSUF_TR_READ(fa[slot][0][1], base + offA[0], ((ks) * 16 + 4) * 256); \
  OSUF_TR_READ(fa[slot][1][0], base + offA[1], (ks) * 16 * 256); OSUF_TR_READ(fa[slot][1][1], base + offA[1], ((ks) * 16 + 4) * 256); \
  OSUF_TR_READ(fb[slot][0][0], base + offB[0], (ks) * 16 * 256); OSUF_TR_READ(fb[slot][0][1], base + offB[0], ((ks) * 16 + 4) * 256); \
  OSUF_TR_READ(fb[slot][1][0], base + offB[1], (ks) * 16 * 256); OSUF_TR_READ(fb[slot][1][1], base + offB[1], ((ks) * 16 + 4) * 256); \
  OSUF_TR_READ(fb[slot][2][0], base + offB[2], (ks) * 16 * 256); OSUF_TR_READ(fb[slot][2][1], base + offB[2], ((ks) * 16 + 4) * 256);
#define T3_WAIT(slot, n)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                        \
               : "+v"(fa[slot][0][0]), "+v"(fa[slot][0][1]), "+v"(fa[slot][1][0]), "+v"(fa[slot][1][1]), "+v"(fb[slot][0][0]),  \
                 "+v"(fb[slot][0][1]), "+v"(fb[slot][1][0]), "+v"(fb[slot][1][1]), "+v"(fb[slot][2][0]), "+v"(fb[slot][2][1]));  \
  __builtin_amdgcn_sched_barrier(0);
#define T3_MFMA(slot)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int t = 0; t < 3; ++t) {                   \
    u32x4 va = {fa[slot][i][0][0], fa[slot][i][0][1], fa[slot][i][1][0], fa[slot][i][1][1]};                       \
    u32x4 vb = {fb[slot][t][0][0], fb[slot][t][0][1], fb[slot][t][1][0], fb[slot][t][1][1]};                       \
    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc[i][t], 0, 0, 0); \
  }

  const bool do_bias = g.dbias != nullptr && n2_0 == 0 && wc == 0;      // (see gemm_tn_big_kernel)
  float bsum[2] = {0.f, 0.f};
#define T3_BIAS(slot)                                                                                             \
  if (do_bias) { _Pragma("unroll") for (int i = 0; i < 2; ++i) bsum[i] = bf16x8_sum(fa[slot][i][0], fa[slot][i][1], bsum[i]); }
  issue(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int mb = m_begin; mb < m_end; mb += BKM, buf ^= 1) {
    if (mb + BKM < m_end) issue(mb + BKM, buf ^ 1);
    const uint32_t base = lds0 + buf * STAGE;
    T3_READS(0, 0, base)
    T3_READS(1, 1, base)
    T3_WAIT(0, 10)  T3_MFMA(0)  T3_BIAS(0)  T3_READS(0, 2, base)
    T3_WAIT(1, 10)  T3_MFMA(1)  T3_BIAS(1)  T3_READS(1, 3, base)
    T3_WAIT(0, 10)  T3_MFMA(0)  T3_BIAS(0)  T3_READS(0, 4, base)
    T3_WAIT(1, 10)  T3_MFMA(1)  T3_BIAS(1)  T3_READS(1, 5, base)
    T3_WAIT(0, 10)  T3_MFMA(0)  T3_BIAS(0)  T3_READS(0, 6, base)
    T3_WAIT(1, 10)  T3_MFMA(1)  T3_BIAS(1)  T3_READS(1, 7, base)
    T3_WAIT(0, 10)  T3_MFMA(0)  T3_BIAS(0)
    T3_WAIT(1, 0)   T3_MFMA(1)  T3_BIAS(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef T3_READS
#undef T3_WAIT
#undef T3_MFMA
#undef T3_BIAS
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      const int n1 = n1_0 + wr * 64 + i * 32 + lr;
      if (lh == 0 && n1 < g.N1) atomic_add_f32(g.dbias + n1, v);
    }
  }

  // partial tiles of this m-split ([split][tap][N1][N2], plain stores; summed by the wgrad_reduce kernels) or fp32 atomics into dW
  if (g.ws && g.ws_pk) {                                             // bf16 pairs of rows (see tn_big_store)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      uint32_t* out = reinterpret_cast<uint32_t*>(g.ws) + ((long)split * 3 + t) * (g.N1 >> 1) * g.N2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n2 = n2_0 + wc * 32 + lr;
        if (n2 < g.N2) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int n1 = n1_0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (n1 < g.N1) out[(long)(n1 >> 1) * g.N2 + n2] = pack_bf16x2(acc[i][t][r], acc[i][t][r + 1]);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* out = g.ws ? g.ws + ((long)split * 3 + t) * g.N1 * g.N2 : g.dW + (long)t * g.tapstride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n2 = n2_0 + wc * 32 + lr;
      if (n2 < g.N2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n1 = n1_0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n1 < g.N1) {
            if (g.ws) out[(long)n1 * g.N2 + n2] = acc[i][t][r];
            else atomic_add_f32(out + (long)n1 * g.ldw + (long)n2 * g.es, acc[i][t][r]);
          }
        }
      }
    }
  }
}

// The same three-taps-per-workgroup weight gradient for OSUF_DT_F32X3: fp32 stages of 64 rows of m (dY tile [64][128] + X panel of 66
// rows, 512 B per row, one LDS-DMA instruction per two rows), fragments gathered down the columns -- dY with one ds_read_b64 per row (the
// lane's k-element for both M-tiles: tile i owns the columns congruent i mod 2), X with ten ds_read_b32 per k-step that serve all three
// taps' 8-row windows -- split into bf16 hi + lo in registers, three bf16 MFMAs per product.
__global__ __launch_bounds__(512, 2) void gemm_tn_taps3_x3_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKM = 64, ROWB = 512, YB = BKM * ROWB, XROWS = 66, STAGE = YB + XROWS * ROWB;      // 32 KiB + 33 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n2 = (g.N2 + 127) / 128, tiles_n1 = (g.N1 + 127) / 128;
  const int ntile = tiles_n1 * tiles_n2;
  // XCD-aware order (speed only): the tiles of one m-split re-read the same dY / X rows.  Block ids are dealt round-robin to the 8 XCDs, so
  // XCD x takes the x-th eighth of the (split-major) unit list: a split's tiles run back to back on one XCD and meet in its L2
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int unit = xcd * ((int)gridDim.x >> 3) + local;
  const int split = unit / ntile, tile = unit % ntile;
  const int n1_0 = (tile / tiles_n2) * 128, n2_0 = (tile % tiles_n2) * 128;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  if (m_begin >= m_end) return;
  const float* dY = reinterpret_cast<const float*>(g.dY);
  const float* X = reinterpret_cast<const float*>(g.X);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int L = g.rm.Lout;
  // DMA: an instruction moves 2 rows x 512 B; instruction i < 4 of wave w is rows (4 w + i) * 2 + (lane >> 5), this lane its 16-B chunk
  // lane & 31 (4 columns); wave 0's fifth X instruction is panel rows 64, 65
  const int col = (lane & 31) * 4;
  const bool y_ok = n1_0 + col < g.N1, x_ok = n2_0 + col < g.N2;
  int pos = m_begin % L;
  auto issue = [&](int mb, int buf) {
    char* sy = smem + buf * STAGE;
    char* sx = sy + YB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + (wave * 4 + i) * 2 + (lane >> 5);
      const char* qy = (m < m_end && y_ok) ? reinterpret_cast<const char*>(dY + (long)m * g.ldy + n1_0 + col) : zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qy, (las_ptr)(sy + (wave * 4 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i == 4 && wave != 0) break;
      const int gi = i < 4 ? wave * 4 + i : 32;
      const int pr = gi * 2 + (lane >> 5);                     // panel row pr = X row mb - 1 + pr
      const int q = pos - 1 + pr;
      const bool ok = q >= 0 && q < L && mb - 1 + pr < g.M && x_ok;
      const char* qx = ok ? reinterpret_cast<const char*>(X + (long)(mb - 1 + pr) * g.ldx + n2_0 + col) : zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qx, (las_ptr)(sx + gi * 1024), 16, 0, 0);
    }
    pos += BKM;
    if (pos >= L) pos -= L;
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  issue(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int mb = m_begin; mb < m_end; mb += BKM, buf ^= 1) {
    if (mb + BKM < m_end) issue(mb + BKM, buf ^ 1);
    const char* sy = smem + buf * STAGE;
    const char* sx = sy + YB;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 ah[2], al[2], bh[3], bl[3];
      {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = *reinterpret_cast<const f32x2*>(sy + (16 * kk + 8 * lh + e) * ROWB + (wr * 64 + 2 * lr) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float x[8] = {y[0][i], y[1][i], y[2][i], y[3][i], y[4][i], y[5][i], y[6][i], y[7][i]};
          split_bf16x8(x, ah[i], al[i]);
        }
      }
      {
        float xw[10];                                          // panel rows 16 kk + 8 lh + 0 .. 9 of this lane's column: tap t's window starts at t
#pragma unroll
        for (int e = 0; e < 10; ++e) xw[e] = *reinterpret_cast<const float*>(sx + (16 * kk + 8 * lh + e) * ROWB + (wc * 32 + lr) * 4);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float x[8] = {xw[t], xw[t + 1], xw[t + 2], xw[t + 3], xw[t + 4], xw[t + 5], xw[t + 6], xw[t + 7]};
          split_bf16x8(x, bh[t], bl[t]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) mfma_x3(acc[i][t], ah[i], al[i], bh[t], bl[t]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* out = g.ws ? g.ws + ((long)split * 3 + t) * g.N1 * g.N2 : g.dW + (long)t * g.tapstride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n2 = n2_0 + wc * 32 + lr;
      if (n2 < g.N2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n1 = n1_0 + wr * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
          if (n1 < g.N1) {
            if (g.ws) out[(long)n1 * g.N2 + n2] = acc[i][t][r];
            else atomic_add_f32(out + (long)n1 * g.ldw + (long)n2 * g.es, acc[i][t][r]);
          }
        }
      }
    }
  }
}

// deterministic second stage of the split wgrad: dW (+)= sum_s ws[s][t][i], i = n1*N2 + n2.
// LAYOUT 0: dW[t][i] (the kernel's own order)   LAYOUT 1: dW[i][t] = torch's (Cout, Cin, k) conv weight layout -- the permute
// is free here: a thread owns one i and writes its `taps` values contiguously.
// ---------------------------------------------------------------------------------------------------------
// Skinny wgrad (N2 <= 32: the rank-r LoRA gradients dB = dy^T u and, with the operands' roles swapped, dA^T = x^T du): a 256 (n1) x
// 32 (n2) output tile per workgroup, 8 waves x one 32x32 MFMA tile, split over m with fp32 atomics into the small result
// (32 consecutive floats per wave-instruction: the full-rate atomic shape).  Same dY loader and transposed fragment reads as
// gemm_tn_big_kernel; the X tile is compact ([64 rows][64 B], no swizzle).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn_skinny_kernel(WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BKM = 64, ROWB = 512, TILEB = BKM * ROWB, XTILEB = BKM * 64, STAGE = TILEB + XTILEB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n1 = (g.N1 + kBig - 1) / kBig;
  const int tile = blockIdx.x % tiles_n1;
  const int t = (blockIdx.x / tiles_n1) % g.taps;
  const int split = blockIdx.x / (tiles_n1 * g.taps);
  const int n1_0 = tile * kBig;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);
  if (m_begin >= m_end) return;
  const bf16_t* dY = reinterpret_cast<const bf16_t*>(g.dY);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);

  // dY loader (as gemm_tn_big_kernel): instruction i of this wave fills tile rows (wave*4+i)*2 + (lane>>5), LDS chunk lane&31
  int srow[4], scol[4];
  const char* py[4];
  bool y_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    srow[i] = (wave * 4 + i) * 2 + (lane >> 5);
    scol[i] = ((lane & 31) ^ ((srow[i] & 3) << 2)) * 8;
    y_ok[i] = n1_0 + scol[i] < g.N1;
    py[i] = reinterpret_cast<const char*>(dY + (long)(m_begin + srow[i]) * g.ldy + n1_0 + scol[i]);
  }
  const long ystep = (long)BKM * g.ldy * (long)sizeof(bf16_t);
  // X loader: waves 0-3, one instruction each: rows wave*16 + (lane>>2), 16-B chunk lane&3 (8 columns of n2)
  const int xrow = wave * 16 + (lane >> 2), xcol = (lane & 3) * 8;
  const bool x_ok = wave < 4 && xcol < g.N2;
  int xb = 0, xp = 0;
  if (wave < 4) { const int m = m_begin + xrow; xb = m / g.rm.Lout; xp = m - xb * g.rm.Lout; }
  auto issue = [&](int mb, int buf) {
    char* sy = smem + buf * STAGE + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* qy = (mb + srow[i] < m_end && y_ok[i]) ? py[i] : zero;
      __builtin_amdgcn_global_load_lds((gas_ptr)qy, (las_ptr)(sy + i * 1024), 16, 0, 0);
      py[i] += ystep;
    }
    if (wave < 4) {
      const char* qx = zero;
      if (x_ok && mb + xrow < m_end) {
        const int s2 = map_row(g.rm, xp, t);
        if (s2 >= 0) qx = reinterpret_cast<const char*>(X + (long)(xb * g.rm.Lin + s2) * g.ldx + xcol);
      }
      __builtin_amdgcn_global_load_lds((gas_ptr)qx, (las_ptr)(smem + buf * STAGE + TILEB + wave * 1024), 16, 0, 0);
      xp += BKM;
      while (xp >= g.rm.Lout) { xp -= g.rm.Lout; ++xb; }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;
  const int ip = lane & 15, cb = ((lane >> 4) & 1) * 16, tq = ip >> 2, tp = ip & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
  const uint32_t offA = (8 * lh + tq) * ROWB + ((((wave * 32 + cb + 4 * tp) * 2)) ^ (tq << 6));
  const uint32_t offB = TILEB + (8 * lh + tq) * 64 + (cb + 4 * tp) * 2;

  issue(m_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int mb = m_begin; mb < m_end; mb += BKM, buf ^= 1) {
    if (mb + BKM < m_end) issue(mb + BKM, buf ^ 1);
    const uint32_t base = lds0 + buf * STAGE;
    u32x2 a0[4], a1[4], b0[4], b1[4];
    OSUF_TR_READ(a0[0], base + offA, 0 * 16 * 512);  OSUF_TR_READ(a1[0], base + offA, (0 * 16 + 4) * 512);
    OSUF_TR_READ(b0[0], base + offB, 0 * 16 * 64);   OSUF_TR_READ(b1[0], base + offB, (0 * 16 + 4) * 64);
    OSUF_TR_READ(a0[1], base + offA, 1 * 16 * 512);  OSUF_TR_READ(a1[1], base + offA, (1 * 16 + 4) * 512);
    OSUF_TR_READ(b0[1], base + offB, 1 * 16 * 64);   OSUF_TR_READ(b1[1], base + offB, (1 * 16 + 4) * 64);
    OSUF_TR_READ(a0[2], base + offA, 2 * 16 * 512);  OSUF_TR_READ(a1[2], base + offA, (2 * 16 + 4) * 512);
    OSUF_TR_READ(b0[2], base + offB, 2 * 16 * 64);   OSUF_TR_READ(b1[2], base + offB, (2 * 16 + 4) * 64);
    OSUF_TR_READ(a0[3], base + offA, 3 * 16 * 512);  OSUF_TR_READ(a1[3], base + offA, (3 * 16 + 4) * 512);
    OSUF_TR_READ(b0[3], base + offB, 3 * 16 * 64);   OSUF_TR_READ(b1[3], base + offB, (3 * 16 + 4) * 64);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a0[0]), "+v"(a1[0]), "+v"(b0[0]), "+v"(b1[0]), "+v"(a0[1]), "+v"(a1[1]), "+v"(b0[1]), "+v"(b1[1]),
                   "+v"(a0[2]), "+v"(a1[2]), "+v"(b0[2]), "+v"(b1[2]), "+v"(a0[3]), "+v"(a1[3]), "+v"(b0[3]), "+v"(b1[3]));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 va = {a0[ks][0], a0[ks][1], a1[ks][0], a1[ks][1]};
      const u32x4 vb = {b0[ks][0], b0[ks][1], b1[ks][0], b1[ks][1]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, vb), acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float* dW = g.dW + (long)t * g.tapstride;
  if (lr < g.N2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n1 = n1_0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n1 < g.N1) atomic_add_f32(dW + (long)n1 * g.ldw + (long)lr * g.es, acc[r]);
    }
  }
}

// sum of `splits` partial float4s that lie `stride` floats apart; four independent chains so that four loads are in flight per
// lane (a single dependent chain waits out one L2 / HBM round trip per split)
__device__ __forceinline__ f32x4 sum_partials(const float* p, long stride, int splits) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int sidx = 0;
  for (; sidx + 4 <= splits; sidx += 4) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p + (long)sidx * stride);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p + (long)(sidx + 1) * stride);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(p + (long)(sidx + 2) * stride);
    const f32x4 b3 = *reinterpret_cast<const f32x4*>(p + (long)(sidx + 3) * stride);
    a0 += b0; a1 += b1; a2 += b2; a3 += b3;
  }
  for (; sidx < splits; ++sidx) a0 += *reinterpret_cast<const f32x4*>(p + (long)sidx * stride);
  return (a0 + a1) + (a2 + a3);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, float* dW, long n12, int taps, int splits, int accumulate) {
  if constexpr (LAYOUT == 0) {
    const long n = n12 * taps, n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      f32x4 a = sum_partials(ws + 4 * i, n, splits);
      if (accumulate) a += reinterpret_cast<f32x4*>(dW)[i];
      reinterpret_cast<f32x4*>(dW)[i] = a;
    }
  } else {
    // a thread owns 4 consecutive i: float4 loads of the partials (16 B/lane), then 4*taps contiguous outputs
    const long n = n12 * taps, q12 = n12 >> 2;            // n12 % 4 == 0 is guaranteed by the launcher
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < q12; q += (long)gridDim.x * blockDim.x) {
      if (taps == 3) {                                     // k3 convs (all but the stems): 12 outputs = three 16-B stores
        f32x4 a[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) a[t] = sum_partials(ws + (long)t * n12 + 4 * q, n, splits);
        f32x4* dst = reinterpret_cast<f32x4*>(dW + 12 * q);
        f32x4 o0 = {a[0][0], a[1][0], a[2][0], a[0][1]}, o1 = {a[1][1], a[2][1], a[0][2], a[1][2]}, o2 = {a[2][2], a[0][3], a[1][3], a[2][3]};
        if (accumulate) { o0 += dst[0]; o1 += dst[1]; o2 += dst[2]; }
        dst[0] = o0; dst[1] = o1; dst[2] = o2;
      } else {
        for (int t = 0; t < taps; ++t) {
          const f32x4 a = sum_partials(ws + (long)t * n12 + 4 * q, n, splits);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float* dst = dW + (4 * q + e) * taps + t;
            *dst = accumulate ? *dst + a[e] : a[e];
          }
        }
      }
    }
  }
}

// The same reduce for many splits and few outputs (the 256-channel k3 convs of the top level: 85 partial tiles of 768 KB each, but
// only 16 K float4 groups -> 64 workgroups in the kernel above, each lane walking 3 x 85 slabs in 64 dependent round trips:
// 25 us).  Here a workgroup is 64 output groups x 4 split lanes: wave s sums splits s, s+4, ... (a quarter of the walk, four times
// the workgroups), the four partial sums meet in LDS in a fixed order (s = 0..3: still deterministic).
template <int LAYOUT>
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* ws, float* dW, long n12, int taps, int splits, int accumulate) {
  __shared__ f32x4 red[3][3][64];                          // [split lane 1..3][tap][output group]
  const int tq = threadIdx.x & 63, ts = threadIdx.x >> 6;
  const long n = n12 * taps;
  const int mine = (splits - ts + 3) >> 2;                  // splits ts, ts + 4, ...
  if constexpr (LAYOUT == 0) {
    const long n4 = n >> 2;
    const long i = (long)blockIdx.x * 64 + tq;
    const long ic = min(i, n4 - 1);
    f32x4 a = sum_partials(ws + (long)ts * n + 4 * ic, 4 * n, mine);
    if (ts) red[ts - 1][0][tq] = a;
    __syncthreads();
    if (ts == 0 && i < n4) {
      a = ((a + red[0][0][tq]) + red[1][0][tq]) + red[2][0][tq];
      if (accumulate) a += reinterpret_cast<f32x4*>(dW)[i];
      reinterpret_cast<f32x4*>(dW)[i] = a;
    }
  } else {                                                  // taps == 3 only (the launcher checks)
    const long q12 = n12 >> 2;
    const long q = (long)blockIdx.x * 64 + tq;
    const long qc = min(q, q12 - 1);
    f32x4 a[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) a[t] = sum_partials(ws + (long)ts * n + (long)t * n12 + 4 * qc, 4 * n, mine);
    if (ts) {
#pragma unroll
      for (int t = 0; t < 3; ++t) red[ts - 1][t][tq] = a[t];
    }
    __syncthreads();
    if (ts == 0 && q < q12) {
#pragma unroll
      for (int t = 0; t < 3; ++t) a[t] = ((a[t] + red[0][t][tq]) + red[1][t][tq]) + red[2][t][tq];
      f32x4* dst = reinterpret_cast<f32x4*>(dW + 12 * q);
      f32x4 o0 = {a[0][0], a[1][0], a[2][0], a[0][1]}, o1 = {a[1][1], a[2][1], a[0][2], a[1][2]}, o2 = {a[2][2], a[0][3], a[1][3], a[2][3]};
      if (accumulate) { o0 += dst[0]; o1 += dst[1]; o2 += dst[2]; }
      dst[0] = o0; dst[1] = o1; dst[2] = o2;
    }
  }
}

// The reduce for bf16-pair partial tiles: a workgroup is 64 output groups x 4 split lanes (wave s sums splits s, s + 4, ... in two chains,
// the four sums meet in LDS in the fixed order s = 0..3: deterministic); a group is 4 columns of a row PAIR: one 16-byte load per split gives
// rows n1, n1 + 1 of columns 4 q .. 4 q + 3.  NT = taps a thread carries: 1 (group = (tap, pair, quad); LAYOUT 0, or any tap count in LAYOUT 1
// with scattered stores) or 3 (LAYOUT 1, k3 convs: the twelve outputs of a row's quad are three 16-byte stores, as wgrad_reduce_kernel<1>).
template <int LAYOUT, int NT>
__global__ __launch_bounds__(256) void wgrad_reduce_pk_kernel(const uint32_t* ws, float* dW, int N1, int N2, int taps, int splits, int accumulate) {
  __shared__ f32x4 red[3][2 * NT][64];
  const int tq = threadIdx.x & 63, ts = threadIdx.x >> 6;
  const long quads = N2 >> 2, per_tap = (long)(N1 >> 1) * quads, groups = NT == 3 ? per_tap : per_tap * taps;
  const long gi = (long)blockIdx.x * 64 + tq, gc = min(gi, groups - 1);
  const int t0 = NT == 3 ? 0 : (int)(gc / per_tap);
  const long rem = gc - (long)t0 * per_tap;
  const long p = rem / quads, q = rem - p * quads;
  const long tile = (long)(N1 >> 1) * N2, stride = tile * taps;                              // dwords per (split, tap) / per split
  const uint32_t* src = ws + (long)t0 * tile + p * N2 + 4 * q;
  f32x4 ev[NT], od[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { ev[t] = f32x4{0.f, 0.f, 0.f, 0.f}; od[t] = ev[t]; }
  for (int sidx = ts; sidx < splits; sidx += 4) {
    u32x4 a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const u32x4*>(src + (long)sidx * stride + (long)t * tile);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ev[t][e] += __builtin_bit_cast(float, a[t][e] << 16);
        od[t][e] += __builtin_bit_cast(float, a[t][e] & 0xffff0000u);
      }
  }
  if (ts) {
#pragma unroll
    for (int t = 0; t < NT; ++t) { red[ts - 1][2 * t][tq] = ev[t]; red[ts - 1][2 * t + 1][tq] = od[t]; }
  }
  __syncthreads();
  if (ts == 0 && gi < groups) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ev[t] = ((ev[t] + red[0][2 * t][tq]) + red[1][2 * t][tq]) + red[2][2 * t][tq];
      od[t] = ((od[t] + red[0][2 * t + 1][tq]) + red[1][2 * t + 1][tq]) + red[2][2 * t + 1][tq];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long n1 = 2 * p + h;
      if constexpr (NT == 3) {
        const f32x4 a0 = h ? od[0] : ev[0], a1 = h ? od[1] : ev[1], a2 = h ? od[2] : ev[2];
        f32x4* dst = reinterpret_cast<f32x4*>(dW + (n1 * N2 + 4 * q) * 3);
        f32x4 o0 = {a0[0], a1[0], a2[0], a0[1]}, o1 = {a1[1], a2[1], a0[2], a1[2]}, o2 = {a2[2], a0[3], a1[3], a2[3]};
        if (accumulate) { o0 += dst[0]; o1 += dst[1]; o2 += dst[2]; }
        dst[0] = o0; dst[1] = o1; dst[2] = o2;
      } else if constexpr (LAYOUT == 0) {
        const f32x4 v = h ? od[0] : ev[0];
        f32x4* dst = reinterpret_cast<f32x4*>(dW + ((long)t0 * N1 + n1) * N2 + 4 * q);
        *dst = accumulate ? *dst + v : v;
      } else {
        const f32x4 v = h ? od[0] : ev[0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* dst = dW + (n1 * N2 + 4 * q + e) * taps + t0;
          *dst = accumulate ? *dst + v[e] : v[e];
        }
      }
    }
  }
}

// launches the reduce of the partial tiles of a split weight gradient: bf16 pairs (pk) or the fp32 tiles' kernels
static void launch_wgrad_reduce(const WgradArgs& gb, float* dW, int N1, int N2, int taps, int sp, int out_layout, int accumulate, bool many_splits_ok,
                                hipStream_t stream) {
  const long n12 = (long)N1 * N2, n = n12 * taps;
  if (gb.ws_pk) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(gb.ws);
    const long per_tap = (long)(N1 >> 1) * (N2 >> 2);
    if (out_layout == 1 && taps == 3) hipLaunchKernelGGL((wgrad_reduce_pk_kernel<1, 3>), dim3((int)((per_tap + 63) / 64)), dim3(256), 0, stream, w, dW, N1, N2, taps, sp, accumulate);
    else if (out_layout == 1) hipLaunchKernelGGL((wgrad_reduce_pk_kernel<1, 1>), dim3((int)((per_tap * taps + 63) / 64)), dim3(256), 0, stream, w, dW, N1, N2, taps, sp, accumulate);
    else hipLaunchKernelGGL((wgrad_reduce_pk_kernel<0, 1>), dim3((int)((per_tap * taps + 63) / 64)), dim3(256), 0, stream, w, dW, N1, N2, taps, sp, accumulate);
    return;
  }
  const long groups = out_layout == 1 ? n12 / 4 : n / 4;            // float4 output groups (x taps partial reads each in layout 1)
  long blocks = (groups + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (sp >= 8 && blocks < 512 && many_splits_ok) {                  // many splits, few outputs: four split lanes per output group
    const long b4 = (groups + 63) / 64;
    if (out_layout == 1) hipLaunchKernelGGL(wgrad_reduce4_kernel<1>, dim3((int)b4), dim3(256), 0, stream, gb.ws, dW, n12, taps, sp, accumulate);
    else hipLaunchKernelGGL(wgrad_reduce4_kernel<0>, dim3((int)b4), dim3(256), 0, stream, gb.ws, dW, n12, taps, sp, accumulate);
  } else if (out_layout == 1) hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((int)blocks), dim3(256), 0, stream, gb.ws, dW, n12, taps, sp, accumulate);
  else hipLaunchKernelGGL(wgrad_reduce_kernel<0>, dim3((int)blocks), dim3(256), 0, stream, gb.ws, dW, n12, taps, sp, accumulate);
}

// column sums: out[n] += sum_m Y[m][n]   (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* Y, long ldy, int M, int N, float* out, int rows_per_block) {
  // block: 32 column-chunks (8 elems) x 8 row lanes
  const int cchunk = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  const int n = cchunk * 8;
  const int m_begin = blockIdx.y * rows_per_block;
  const int m_end = min(M, m_begin + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    int m = m_begin + rl;
    for (; m + 24 < m_end; m += 32) {                  // four independent 16-B loads in flight per lane
      float v0[8], v1[8], v2[8], v3[8];
      load8(Y + (long)m * ldy + n, v0);
      load8(Y + (long)(m + 8) * ldy + n, v1);
      load8(Y + (long)(m + 16) * ldy + n, v2);
      load8(Y + (long)(m + 24) * ldy + n, v3);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
    }
    for (; m < m_end; m += 8) {
      float v[8];
      load8(Y + (long)m * ldy + n, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
  __shared__ float red[8][32 * 8 + 1];
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][(threadIdx.x & 31) * 8 + e] = acc[e];
  __syncthreads();
  const int col = threadIdx.x;                       // 256 columns of this block
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) s += red[r][col];
  const int nn = blockIdx.x * 256 + col;
  if (nn < N) atomic_add_f32(out + nn, s);
}

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int gemm_nt_launch(int dtype, const void* A, long lda, const void* W, long ldw, long tapstride,
                          void* C, long ldc, void* C2, long ldc2, const void* R, long ldr, const void* U, long ldu,
                          const float* bias, const float* rscale, double* stats,
                          int M, int N, int K, int taps, int Lin, int Lout, int stride, int pad, int mode, int act,
                          float* delta, int heads, hipStream_t stream);

extern "C" int osuf_gemm_nt(int dtype, const void* A, long lda, const void* W, long ldw, long tapstride,
                            void* C, long ldc, void* C2, long ldc2, const void* R, long ldr, const void* U, long ldu,
                            const float* bias, const float* rscale, double* stats,
                            int M, int N, int K, int taps, int Lin, int Lout, int stride, int pad, int mode, int act,
                            hipStream_t stream) {
  return gemm_nt_launch(dtype, A, lda, W, ldw, tapstride, C, ldc, C2, ldc2, R, ldr, U, ldu, bias, rscale, stats, M, N, K, taps, Lin, Lout,
                        stride, pad, mode, act, nullptr, 0, stream);
}

// C = A W^T (one tap, no bias) and, from the same epilogue, delta[b][h][l] = sum_{d < 64} bf16(C[b*L + l][h*64 + d]) * O[b*L + l][h*64 + d]:
// the to_out input-gradient GEMM of the attention block hands the flash backward its row constants (sum_d dO * O) without a
// second pass over dO and O.  N = heads * 64; M % L == 0.
extern "C" int osuf_gemm_nt_rowdot(int dtype, const void* A, long lda, const void* W, long ldw, void* C, long ldc, const void* O, long ldo,
                                   float* delta, int M, int N, int K, int L, int heads, hipStream_t stream) {
  if (!O || !delta || heads <= 0 || N != heads * 64 || L <= 0 || !aligned16(O) || ldo % 8) return OSUF_EINVAL;
  return gemm_nt_launch(dtype, A, lda, W, ldw, 0, C, ldc, nullptr, 0, O, ldo, nullptr, 0, nullptr, nullptr, nullptr, M, N, K, 1, L, L, 1, 0, 0, 0,
                        delta, heads, stream);
}

static int gemm_nt_launch(int dtype, const void* A, long lda, const void* W, long ldw, long tapstride,
                          void* C, long ldc, void* C2, long ldc2, const void* R, long ldr, const void* U, long ldu,
                          const float* bias, const float* rscale, double* stats,
                          int M, int N, int K, int taps, int Lin, int Lout, int stride, int pad, int mode, int act,
                          float* delta, int heads, hipStream_t stream) {
  const int epc = dtype == OSUF_DT_BF16 ? 8 : 4;
  if (dtype != OSUF_DT_BF16 && dtype != OSUF_DT_F32 && dtype != OSUF_DT_F32X3) return OSUF_EUNSUPPORTED;
  if (M <= 0 || N <= 0 || K <= 0 || taps <= 0 || Lout <= 0 || Lin <= 0) return OSUF_EINVAL;
  if (M % Lout != 0) return OSUF_EINVAL;
  if (K % epc || lda % epc || ldw % epc || tapstride % epc || N % 4 || ldc % 4) return OSUF_EINVAL;
  if ((C2 && ldc2 % 4) || (R && ldr % 4) || (U && ldu % 4)) return OSUF_EINVAL;
  if (!aligned16(A) || !aligned16(W) || !aligned16(C) || (bias && !aligned16(bias)) || (rscale && !aligned16(rscale))) return OSUF_EINVAL;
  if ((long)(M / Lout) * Lin >= (1L << 31)) return OSUF_EINVAL;
  GemmArgs g;
  g.A = A; g.W = W; g.C = C; g.C2 = C2; g.R = R; g.U = U; g.bias = bias; g.rscale = rscale; g.stats = stats;
  g.delta = delta; g.heads = heads;
  g.lda = lda; g.ldw = ldw; g.tapstride = tapstride; g.ldc = ldc; g.ldc2 = ldc2; g.ldr = ldr; g.ldu = ldu;
  g.M = M; g.N = N; g.K = K; g.taps = taps;
  g.rm = RowMap{Lin, Lout, stride, pad, mode};
  g.act = act;
  const int grid = ((M + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
  const int lds = 2 * kStageBytes + 512;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_nt_glds_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_nt_glds_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)(gemm_nt_glds_kernel<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const bool noskinny = getenv("OSUF_GEMM_NOSKINNY") != nullptr;
  if (!noskinny && dtype == OSUF_DT_BF16 && N <= 32 && N % 8 == 0 && ldc % 8 == 0 && M >= 4096 && !C2 && !R && !U && !bias && !rscale && !stats &&
      act == 0) {
    const int lds_sk = 2 * kSkStage;
    static bool sk_attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_skinny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_sk), true);
    (void)sk_attr;
    hipLaunchKernelGGL(gemm_nt_skinny_kernel, dim3((M + kBig - 1) / kBig), dim3(512), lds_sk, stream, g);
    return osuf_launch_status();
  }
  const bool regstage = getenv("OSUF_GEMM_REGSTAGE") != nullptr;      // A/B switches for profiling only
  // 256^2 tiles once they fill most of the 256 CUs; OSUF_GEMM_BIG_MIN_TILES overrides the threshold (tests force 1, "off" = never)
  const char* bigenv = getenv("OSUF_GEMM_BIG_MIN_TILES");
  const long min_tiles = bigenv ? atol(bigenv) : 192;
  const long big_tiles = (long)((M + kBig - 1) / kBig) * ((N + kBig - 1) / kBig);
  const bool use_big = !regstage && (dtype == OSUF_DT_BF16 || dtype == OSUF_DT_F32X3) && min_tiles > 0 && big_tiles >= min_tiles && (N >= 192 || bigenv) && N % 8 == 0 &&
                       ldc % 8 == 0 && (!C2 || ldc2 % 8 == 0) && (!R || ldr % 8 == 0) && (!U || ldu % 8 == 0) &&
                       (!bias || (reinterpret_cast<uintptr_t>(bias) & 31) == 0) && (!rscale || (reinterpret_cast<uintptr_t>(rscale) & 31) == 0);
  if (use_big) {
    const int lds_big = 2 * kBigStage + 2112;          // ring + per-tile GroupNorm stat slots (2 x 258 floats)
    const int lds_p8 = kP8Tbl + taps * kBig * 4;
    static bool big_attr = false;
    if (!big_attr) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      (void)hipFuncSetAttribute((const void*)(gemm_nt_big_kernel<float, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<bf16_t, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<bf16_t, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<bf16_t, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
      big_attr = true;
    }
    const long tm = (M + kBig - 1) / kBig, tn = (N + kBig - 1) / kBig;
    const int dbg = getenv("OSUF_GEMM_DBG") ? atoi(getenv("OSUF_GEMM_DBG")) : 0;
    const dim3 grid_big((int)(((tm + 7) / 8) * 8 * tn));
    const bool halo3 = taps == 3 && mode == 0 && stride == 1 && pad == 1 && Lin == Lout && Lout % kBig == 0 && K % (dtype == OSUF_DT_BF16 ? 64 : 32) == 0 &&
                       dbg == 0 && getenv("OSUF_GEMM_NOHALO") == nullptr;
    if (halo3) {
      static bool halo_attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_big_halo3_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big),
                               (void)hipFuncSetAttribute((const void*)gemm_nt_big_halo3_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big), true);
      (void)halo_attr;
      if (dtype == OSUF_DT_BF16 && getenv("OSUF_GEMM_NO8P") == nullptr && K <= 8192) {
        const int lds_h8 = kH8B + 65536 + 2112;
        static bool h8_attr = ((void)hipFuncSetAttribute((const void*)gemm_nt_big8_halo3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_h8), true);
        (void)h8_attr;
        hipLaunchKernelGGL(gemm_nt_big8_halo3_kernel<0>, grid_big, dim3(512), lds_h8, stream, g);
      } else if (dtype == OSUF_DT_BF16) hipLaunchKernelGGL(gemm_nt_big_halo3_kernel<bf16_t>, grid_big, dim3(512), lds_big, stream, g);
      else hipLaunchKernelGGL(gemm_nt_big_halo3_kernel<float>, grid_big, dim3(512), lds_big, stream, g);
    } else if (dtype == OSUF_DT_BF16 && getenv("OSUF_GEMM_NO8P") == nullptr && taps <= kP8MaxTaps && K <= 8192 && K % 64 == 0) {
      // (the 8-phase loop; OSUF_GEMM_NO8P=1 = the one-barrier-per-K-step loop above, for A/B runs and for K % 64 != 0 / more than 16 taps)
      const int lds_p8_max = kP8Tbl + kP8MaxTaps * kBig * 4;
#define P8_CASE(D) case D: { static bool a_ = ((void)hipFuncSetAttribute((const void*)gemm_nt_big8_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_p8_max), true); (void)a_; \
                             hipLaunchKernelGGL(gemm_nt_big8_kernel<D>, grid_big, dim3(512), lds_p8, stream, g); } break;
      switch (dbg) { P8_CASE(1) P8_CASE(2) P8_CASE(3) P8_CASE(4) P8_CASE(5) P8_CASE(6) default: P8_CASE(0) }
#undef P8_CASE
    } else if (dtype == OSUF_DT_F32X3) hipLaunchKernelGGL((gemm_nt_big_kernel<float, 0, true>), grid_big, dim3(512), lds_big, stream, g);
    else if (dbg == 1) hipLaunchKernelGGL((gemm_nt_big_kernel<bf16_t, 1>), grid_big, dim3(512), lds_big, stream, g);
    else if (dbg == 2) hipLaunchKernelGGL((gemm_nt_big_kernel<bf16_t, 2>), grid_big, dim3(512), lds_big, stream, g);
    else if (dbg == 3) hipLaunchKernelGGL((gemm_nt_big_kernel<bf16_t, 3>), grid_big, dim3(512), lds_big, stream, g);
    else if (dbg == 4) hipLaunchKernelGGL((gemm_nt_big_kernel<bf16_t, 4>), grid_big, dim3(512), lds_big, stream, g);
    else if (dbg == 5) hipLaunchKernelGGL((gemm_nt_big_kernel<bf16_t, 5>), grid_big, dim3(512), lds_big, stream, g);
    else hipLaunchKernelGGL(gemm_nt_big_kernel<bf16_t>, grid_big, dim3(512), lds_big, stream, g);
  } else if (regstage) {
    if (dtype == OSUF_DT_BF16) hipLaunchKernelGGL(gemm_nt_kernel<bf16_t>, dim3(grid), dim3(256), lds, stream, g);
    else hipLaunchKernelGGL(gemm_nt_kernel<float>, dim3(grid), dim3(256), lds, stream, g);
  } else {
    if (dtype == OSUF_DT_BF16) hipLaunchKernelGGL(gemm_nt_glds_kernel<bf16_t>, dim3(grid), dim3(256), lds, stream, g);
    else if (dtype == OSUF_DT_F32X3) hipLaunchKernelGGL((gemm_nt_glds_kernel<float, true>), dim3(grid), dim3(256), lds, stream, g);
    else hipLaunchKernelGGL(gemm_nt_glds_kernel<float>, dim3(grid), dim3(256), lds, stream, g);
  }
  return osuf_launch_status();
}

// split plan of the 256x256 wgrad kernel: rows of m per split and number of splits (one workgroup per CU, ~1.25 rounds)
static bool tn_big_plan(int dtype, int M, int N1, int N2, int taps, int* rows_out, int* splits_out) {
  const char* bigenv = getenv("OSUF_GEMM_BIG_MIN_TILES");
  const bool forced = bigenv && atol(bigenv) == 1, off = bigenv && atol(bigenv) <= 0;
  const int lo = getenv("OSUF_TN_BIG_MIN_N") ? atoi(getenv("OSUF_TN_BIG_MIN_N")) : 64;
  if ((dtype != OSUF_DT_BF16 && dtype != OSUF_DT_F32X3) || off || !(forced || (N1 >= lo && N2 >= lo && (N1 >= 192 || N2 >= 192)))) return false;
  const int btiles = ((N1 + kBig - 1) / kBig) * ((N2 + kBig - 1) / kBig);
  int sp = (256 + btiles * taps / 2) / (btiles * taps);        // one workgroup per CU: about one round of the 256 CUs
  if (sp < 1) sp = 1;
  // ... and never a few workgroups MORE than one round: the launch pads sp * btiles to a multiple of 8 per tap, which turned
  // 85 x 3 = 255 into 264 workgroups -- the last 8 ran alone in a second round and the launch took 130 us instead of 59
  while (sp > 1 && ((sp * btiles + 7) / 8) * 8 * taps > 256) --sp;
  int rows = (M + sp - 1) / sp;
  rows = ((rows + 63) / 64) * 64;
  if (rows_out) *rows_out = rows;
  if (splits_out) *splits_out = (M + rows - 1) / rows;
  return true;
}

// bytes of fp32 workspace that let osuf_gemm_tn avoid atomics for this shape (0: the atomic path is used anyway)
extern "C" long osuf_gemm_tn_workspace_bytes(int dtype, int M, int N1, int N2, int taps) {
  int rows, sp;
  if (M <= 0 || N1 <= 0 || N2 <= 0 || taps <= 0 || !tn_big_plan(dtype, M, N1, N2, taps, &rows, &sp)) return 0;
  if ((N1 * (long)N2 * taps) % 4) return 0;
  return (long)sp * taps * N1 * N2 * (long)sizeof(float);            // (bf16-pair partial tiles need half of it; one size keeps callers simple)
}
// bf16 kernels write their partial tiles as bf16 pairs of rows (half the bytes of the write-out and of the reduce's reads); the fp32
// modes keep fp32 tiles.  OSUF_WGRAD_F32_PARTIALS=1 restores fp32 tiles for A/B.
static bool wgrad_pk(int dtype, int N1, int N2) {
  return dtype == OSUF_DT_BF16 && (N1 % 2) == 0 && (N2 % 4) == 0 && getenv("OSUF_WGRAD_F32_PARTIALS") == nullptr;
}

extern "C" int osuf_colsum(int dtype, const void* Y, long ldy, int M, int N, float* out, hipStream_t stream);

// dbias (optional): += the column sums of dY (the bias gradient of the layer).  The bf16 256x256 and merged-taps kernels take them from the
// dY fragments they hold anyway; every other path runs osuf_colsum on dY.
static int gemm_tn_launch(int dtype, const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, long tapstride,
                          int M, int N1, int N2, int taps, int Lin, int Lout, int stride, int pad, int mode,
                          int splits, int out_layout, int accumulate, float* workspace, long workspace_bytes, float* dbias, hipStream_t stream) {
  // out_layout 0: dW[t][n1][n2] with (ldw, tapstride) as given; 1: dense torch conv layout dW[n1][n2][t] (ldw/tapstride ignored)
  long es = 1;
  if (out_layout == 1) { ldw = (long)N2 * taps; tapstride = 1; es = taps; }
  else if (out_layout != 0) return OSUF_EINVAL;
  const int epc = dtype == OSUF_DT_BF16 ? 8 : 4;
  if (dtype != OSUF_DT_BF16 && dtype != OSUF_DT_F32 && dtype != OSUF_DT_F32X3) return OSUF_EUNSUPPORTED;
  if (M <= 0 || N1 <= 0 || N2 <= 0 || taps <= 0 || Lout <= 0 || Lin <= 0 || M % Lout) return OSUF_EINVAL;
  if (N1 % epc || N2 % epc || ldy % epc || ldx % epc) return OSUF_EINVAL;
  if (!aligned16(dY) || !aligned16(X)) return OSUF_EINVAL;
  auto bias_by_colsum = [&]() -> int {                       // paths whose kernel does not produce the sums
    return dbias ? osuf_colsum(dtype == OSUF_DT_BF16 ? OSUF_DT_BF16 : OSUF_DT_F32, dY, ldy, M, N1, dbias, stream) : OSUF_OK;
  };
  const int bkm = dtype == OSUF_DT_BF16 ? 64 : 32;
  const bool noskinny = getenv("OSUF_GEMM_NOSKINNY") != nullptr;
  if (!noskinny && splits <= 0 && dtype == OSUF_DT_BF16 && N2 <= 32 && M >= 4096) {
    WgradArgs gs{};
    gs.dY = dY; gs.X = X; gs.dW = dW; gs.ws = nullptr; gs.es = es; gs.ldy = ldy; gs.ldx = ldx; gs.ldw = ldw; gs.tapstride = tapstride;
    gs.M = M; gs.N1 = N1; gs.N2 = N2; gs.taps = taps; gs.rm = RowMap{Lin, Lout, stride, pad, mode};
    const int tiles_n1 = (N1 + kBig - 1) / kBig;
    int sp = 512 / (tiles_n1 * taps);                                  // 2 workgroups per CU, and not one workgroup more than that
    if (sp < 1) sp = 1;                                                // (rounding up gave 171 x 3 = 513: a second round for one block)
    int rows = (M + sp - 1) / sp;
    rows = ((rows + 63) / 64) * 64;
    if (rows < 256) rows = 256;
    sp = (M + rows - 1) / rows;
    gs.rows_per_split = rows;
    if (!accumulate) {
      if (out_layout == 0 && !(ldw == N2 && (taps == 1 || tapstride == (long)N1 * N2))) return OSUF_EINVAL;
      (void)hipMemsetAsync(dW, 0, (size_t)taps * N1 * N2 * sizeof(float), stream);
    }
    const int lds_sk = 2 * (64 * 512 + 64 * 64);
    static bool sk_attr = ((void)hipFuncSetAttribute((const void*)gemm_tn_skinny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_sk), true);
    (void)sk_attr;
    hipLaunchKernelGGL(gemm_tn_skinny_kernel, dim3(tiles_n1 * taps * sp), dim3(512), lds_sk, stream, gs);
    if (int rc = bias_by_colsum()) return rc;
    return osuf_launch_status();
  }
  if (splits <= 0 && (dtype == OSUF_DT_BF16 || dtype == OSUF_DT_F32X3) && taps == 3 && mode == 0 && stride == 1 && pad == 1 && Lin == Lout && Lout % 128 == 0 && N1 >= 64 &&
      N2 >= 64 && tn_big_plan(dtype, M, N1, N2, taps, nullptr, nullptr) && getenv("OSUF_GEMM_NOHALO") == nullptr) {
    // the three taps in one workgroup (gemm_tn_taps3_kernel): 128 x 128 tiles, about one round of the 256 CUs
    const int btiles = ((N1 + 127) / 128) * ((N2 + 127) / 128);
    int sp = 256 / btiles;
    if (sp < 1) sp = 1;
    int rows = (M + sp - 1) / sp;
    rows = ((rows + 127) / 128) * 128;
    sp = (M + rows - 1) / rows;
    WgradArgs gb{};
    gb.dY = dY; gb.X = X; gb.dW = dW; gb.ldy = ldy; gb.ldx = ldx; gb.ldw = ldw; gb.tapstride = tapstride;
    gb.M = M; gb.N1 = N1; gb.N2 = N2; gb.taps = taps; gb.rm = RowMap{Lin, Lout, stride, pad, mode};
    gb.rows_per_split = rows;
    const long n = (long)taps * N1 * N2;
    const bool dense = (out_layout == 1 && ((long)N1 * N2) % 4 == 0) || (out_layout == 0 && ldw == N2 && tapstride == (long)N1 * N2 && n % 4 == 0 && aligned16(dW));
    gb.ws = (workspace && dense && aligned16(workspace) && workspace_bytes >= (long)sp * n * (long)sizeof(float)) ? workspace : nullptr;
    gb.es = es;
    gb.ws_pk = gb.ws && wgrad_pk(dtype, N1, N2);
    gb.dbias = dtype == OSUF_DT_BF16 ? dbias : nullptr;
    if (dtype != OSUF_DT_BF16) { if (int rc = bias_by_colsum()) return rc; }
    if (!gb.ws && !accumulate) (void)hipMemsetAsync(dW, 0, (size_t)n * sizeof(float), stream);    // atomic path needs zeros
    const int lds_t3 = 2 * (128 * 256 + 132 * 256);
    static bool t3_attr = ((void)hipFuncSetAttribute((const void*)gemm_tn_taps3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_t3),
                           (void)hipFuncSetAttribute((const void*)gemm_tn_taps3_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_t3), true);
    (void)t3_attr;
    const dim3 grid_t3(((sp * btiles + 7) / 8) * 8);
    if (dtype == OSUF_DT_F32X3) hipLaunchKernelGGL(gemm_tn_taps3_x3_kernel, grid_t3, dim3(512), lds_t3, stream, gb);
    else hipLaunchKernelGGL(gemm_tn_taps3_kernel, grid_t3, dim3(512), lds_t3, stream, gb);
    if (gb.ws) launch_wgrad_reduce(gb, dW, N1, N2, taps, sp, out_layout, accumulate, true, stream);
    return osuf_launch_status();
  }
  {
    int rows, sp;
    if (splits <= 0 && tn_big_plan(dtype, M, N1, N2, taps, &rows, &sp)) {
      WgradArgs gb{};
      gb.dY = dY; gb.X = X; gb.dW = dW; gb.ldy = ldy; gb.ldx = ldx; gb.ldw = ldw; gb.tapstride = tapstride;
      gb.M = M; gb.N1 = N1; gb.N2 = N2; gb.taps = taps; gb.rm = RowMap{Lin, Lout, stride, pad, mode};
      gb.rows_per_split = rows;
      gb.xcd_chunks = getenv("OSUF_TN_RR") == nullptr;
      const long n = (long)taps * N1 * N2;
      const bool dense = (out_layout == 1 && ((long)N1 * N2) % 4 == 0) || (out_layout == 0 && ldw == N2 && (taps == 1 || tapstride == (long)N1 * N2) && n % 4 == 0 && aligned16(dW));
      gb.ws = (workspace && dense && aligned16(workspace) && workspace_bytes >= (long)sp * n * (long)sizeof(float)) ? workspace : nullptr;
      gb.es = es;
      gb.ws_pk = gb.ws && wgrad_pk(dtype, N1, N2);
      gb.dbias = dtype == OSUF_DT_BF16 ? dbias : nullptr;
      if (dtype != OSUF_DT_BF16) { if (int rc = bias_by_colsum()) return rc; }
      if (!gb.ws && !accumulate) (void)hipMemsetAsync(dW, 0, (size_t)n * sizeof(float), stream);    // atomic path needs zeros
      const int btiles = ((N1 + kBig - 1) / kBig) * ((N2 + kBig - 1) / kBig);
      const int lds_big = 2 * 65536;
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
        (void)hipFuncSetAttribute((const void*)gemm_tn_big_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big);
        attr = true;
      }
      if (dtype == OSUF_DT_F32X3) hipLaunchKernelGGL(gemm_tn_big_x3_kernel, dim3(((sp * btiles + 7) / 8) * 8 * taps), dim3(512), lds_big, stream, gb);
      else hipLaunchKernelGGL(gemm_tn_big_kernel, dim3(((sp * btiles + 7) / 8) * 8 * taps), dim3(512), lds_big, stream, gb);
      if (gb.ws) launch_wgrad_reduce(gb, dW, N1, N2, taps, sp, out_layout, accumulate, out_layout == 0 || taps == 3, stream);
      return osuf_launch_status();
    }
  }
  const int tiles = ((N1 + kTile - 1) / kTile) * ((N2 + kTile - 1) / kTile);
  if (splits <= 0) {                                  // aim at ~3 workgroups per CU
    splits = (768 + tiles * taps - 1) / (tiles * taps);
  }
  int rows = (M + splits - 1) / splits;
  rows = ((rows + bkm - 1) / bkm) * bkm;
  splits = (M + rows - 1) / rows;
  WgradArgs g{};
  g.dY = dY; g.X = X; g.dW = dW; g.ldy = ldy; g.ldx = ldx; g.ldw = ldw; g.tapstride = tapstride;
  g.M = M; g.N1 = N1; g.N2 = N2; g.taps = taps; g.rm = RowMap{Lin, Lout, stride, pad, mode};
  g.rows_per_split = rows;
  g.ws = nullptr;
  g.es = es;
  if (!accumulate) {
    if (out_layout == 0 && !(ldw == N2 && (taps == 1 || tapstride == (long)N1 * N2))) return OSUF_EINVAL;   // needs a dense dW to clear
    (void)hipMemsetAsync(dW, 0, (size_t)taps * N1 * N2 * sizeof(float), stream);
  }
  if (int rc = bias_by_colsum()) return rc;
  const int lds = 4 * 16384;
  if (dtype == OSUF_DT_BF16) {
    hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(tiles, taps, splits), dim3(256), lds, stream, g);
  } else if (dtype == OSUF_DT_F32X3) {
    hipLaunchKernelGGL((gemm_tn_kernel<float, true>), dim3(tiles, taps, splits), dim3(256), lds, stream, g);
  } else {
    hipLaunchKernelGGL(gemm_tn_kernel<float>, dim3(tiles, taps, splits), dim3(256), lds, stream, g);
  }
  return osuf_launch_status();
}

extern "C" int osuf_gemm_tn(int dtype, const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, long tapstride,
                            int M, int N1, int N2, int taps, int Lin, int Lout, int stride, int pad, int mode,
                            int splits, int out_layout, int accumulate, float* workspace, long workspace_bytes, hipStream_t stream) {
  return gemm_tn_launch(dtype, dY, ldy, X, ldx, dW, ldw, tapstride, M, N1, N2, taps, Lin, Lout, stride, pad, mode, splits, out_layout, accumulate,
                        workspace, workspace_bytes, nullptr, stream);
}

// osuf_gemm_tn + the bias gradient of the same layer: dbias[n1] += sum_m dY[m][n1] (fp32, N1 entries; needs N1 % 8 == 0 on the paths that
// fall back to osuf_colsum)
extern "C" int osuf_gemm_tn_bias(int dtype, const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, long tapstride,
                                 int M, int N1, int N2, int taps, int Lin, int Lout, int stride, int pad, int mode,
                                 int splits, int out_layout, int accumulate, float* workspace, long workspace_bytes, float* dbias,
                                 hipStream_t stream) {
  if (!dbias) return OSUF_EINVAL;
  return gemm_tn_launch(dtype, dY, ldy, X, ldx, dW, ldw, tapstride, M, N1, N2, taps, Lin, Lout, stride, pad, mode, splits, out_layout, accumulate,
                        workspace, workspace_bytes, dbias, stream);
}

extern "C" int osuf_colsum(int dtype, const void* Y, long ldy, int M, int N, float* out, hipStream_t stream) {
  if (M <= 0 || N <= 0 || N % 8 || ldy % 8 || !aligned16(Y)) return OSUF_EINVAL;
  const int rows_per_block = 256;
  dim3 grid((N + 255) / 256, (M + rows_per_block - 1) / rows_per_block);
  if (dtype == OSUF_DT_BF16) {
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)Y, ldy, M, N, out, rows_per_block);
  } else if (dtype == OSUF_DT_F32) {
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)Y, ldy, M, N, out, rows_per_block);
  } else {
    return OSUF_EUNSUPPORTED;
  }
  return osuf_launch_status();
}
