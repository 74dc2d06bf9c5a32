// MQA flash attention for the denoiser's self-attention (unet.py:125-141, attention.py:84-101), gfx950.
//   q: [B*N][ldq] bf16, head h at columns h*64..h*64+63 ; k, v: [B*N][ldk/ldv] bf16, ONE kv head (64 columns)
//   softmax(q k^T / 8) v, non-causal, no mask.  bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 softmax/accumulate.
// The 16 query heads share one K/V head, so a workgroup stages each 64-key K/V tile in LDS once and all of
// its 8 waves (= 8 (head, 32-query block) pairs) consume it: 1/16 of the K/V bytes of the reference's
// repeat()-ed tensors.  Everything is computed "transposed" (S^T = K Q^T, O^T = V^T P^T) so that a query row
// lives on one lane: row max / row sum / rescale are lane-local (one xor-32 shuffle), and the S^T accumulator
// is directly the B operand of the PV product (k order permuted consistently on the V^T side, read with
// ds_read_b64_tr_b16 from the row-major V tile).
// Backward = two kernels without atomics: dQ (query-stationary, same loop as forward) and dK/dV
// (key-stationary: a wave owns 32 keys, dK^T/dV^T stay in registers while it sweeps heads x query blocks).
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

static constexpr int D = 64;                      // head dim (bytes per tile row = 128)
static constexpr float kLog2e = 1.4426950408889634f;

// tile image [rows][64 bf16]: 16-B chunk index xor-swizzled so that BOTH ds_read_b128 row reads and
// ds_read_b64_tr_b16 column reads are bank-conflict free (f differs in bit 2 between rows r and r+2).
__device__ __forceinline__ int tile_off(int row, int colbyte) {
  const int x = (row >> 1) & 7;
  const int f = ((x & 1) << 2) | (x >> 1);
  return row * 128 + ((((colbyte >> 4) ^ f) << 4) | (colbyte & 15));
}

// Loop-invariant per-lane LDS byte offsets (the XOR swizzle makes them non-affine in the k-step, so they are precomputed
// once; everything that varies inside the tile loop is a compile-time row base folded into the ds instruction's offset).
struct LaneOffs {
  int row[4];      // row read of tile row (lane&31) (+32*n rows = +4096*n B), K chunk 2*ks + (lane>>5)
  int tr[2][2];    // transposed read, [dt][variant]: variant = bit 3 of the row base (it flips one swizzle bit)
  __device__ __forceinline__ explicit LaneOffs(int lane) {
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) row[ks] = tile_off(lr, (2 * ks + lh) * 16);
    const int cb = ((lane >> 4) & 1) * 16, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int var = 0; var < 2; ++var) tr[dt][var] = tile_off(8 * var + 4 * lh + tq, (dt * 32 + cb + 4 * tp) * 2) - 8 * var * 128;
  }
};

__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, const LaneOffs& lo, int ks, int rowblock32) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + lo.row[ks] + rowblock32 * 4096));
}

// A-operand fragment of a TRANSPOSED tile read: element e of lane (r = lane&31, h = lane>>5) =
//   tile[rowbase + 8*(e>>2) + 4*h + (e&3)][dt*32 + r]      (the k order of an accumulator used as B operand)
// rowbase must be a multiple of 8 (compile-time in the unrolled loops).
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, const LaneOffs& lo, int rowbase, int dt) {
  const int v0 = (rowbase >> 3) & 1;
  s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + lo.tr[dt][v0] + rowbase * 128));
  s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(tile + lo.tr[dt][v0 ^ 1] + (rowbase + 8) * 128));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// registers 8*sub .. 8*sub+7 of a 32x32 accumulator -> bf16x8 B-operand fragment (k order permuted, see above);
// __builtin_convertvector lowers to four two-operand v_cvt_pk_bf16_f32
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int sub) {
  typedef __attribute__((ext_vector_type(8))) float f32x8;
  f32x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = a[8 * sub + i];
  return __builtin_convertvector(v, bf16x8);
}

// Epilogue of the backward kernels: one row's 64 gradient values (acc[0]: d < 32, acc[1]: d >= 32, this lane holding
// d = 8g + 4*(lane>>5) + e of each half) times `mul`, optionally mapped through the transpose of the RoPE rotation
// (apply_rotary_pos_emb, attention.py:52-58: y1 = x1 c - x2 s, y2 = x2 c + x1 s  =>  dx1 = dy1 c + dy2 s, dx2 = dy2 c - dy1 s),
// stored as fp32 or bf16.  Fusing this here removes the fp32 round trip of the whole (B, N, (H+2)*64) gradient through HBM.
template <typename TO>
__device__ __forceinline__ void store_grad_row(TO* row, const f32x16 (&acc)[2], float mul, const float* cs_row, const float* sn_row, int lh) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d0 = 8 * g + 4 * lh;
    float y1[4] = {acc[0][4 * g] * mul, acc[0][4 * g + 1] * mul, acc[0][4 * g + 2] * mul, acc[0][4 * g + 3] * mul};
    float y2[4] = {acc[1][4 * g] * mul, acc[1][4 * g + 1] * mul, acc[1][4 * g + 2] * mul, acc[1][4 * g + 3] * mul};
    if (cs_row != nullptr) {
      float cs[4], sn[4];
      load4(cs_row + d0, cs);
      load4(sn_row + d0, sn);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = y1[e] * cs[e] + y2[e] * sn[e];
        const float a2 = y2[e] * cs[e] - y1[e] * sn[e];
        y1[e] = a1; y2[e] = a2;
      }
    }
    store4(row + d0, y1);
    store4(row + 32 + d0, y2);
  }
}
__device__ __forceinline__ void store_grad(void* base, long ld, long m, int col0, int is_bf16, const f32x16 (&acc)[2], float mul,
                                           const float* rcos, const float* rsin, int n, int lh) {
  const float* cs = rcos ? rcos + (long)n * 32 : nullptr;
  const float* sn = rcos ? rsin + (long)n * 32 : nullptr;
  if (is_bf16) store_grad_row(reinterpret_cast<bf16_t*>(base) + m * ld + col0, acc, mul, cs, sn, lh);
  else store_grad_row(reinterpret_cast<float*>(base) + m * ld + col0, acc, mul, cs, sn, lh);
}

#ifdef OSUF_FWD_TRIAGE_NOEXP
__device__ __forceinline__ float fast_exp2(float x) { return x * 0.001f; }     // timing-only triage builds (tools/build_fwd_triage.sh): never in the product
#else
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  long ldq, ldk, ldv;
  void* o; long ldo; int o_is_f32;              // forward output (bf16-rounded values; stored as bf16 or f32)
  float* lse2;                                  // [B][H][N] log2-domain logsumexp of (s * scale * log2e)
  const bf16_t* dout; long lddo;                // backward
  const float* delta;                           // [B][H][N]
  void* dq; long lddq;                          // [B*N][lddq], head h at h*64; fp32, or bf16 when g_bf16
  void* dk; void* dv; long lddk;                // [B*N][lddk]
  int g_bf16;                                   // gradient output element type
  const float* rcos; const float* rsin;         // [N][32] RoPE tables: when set, dQ / dK are stored as gradients of the UN-rotated q / k
  int B, H, N;
  float scale;
  bf16_t* qout; long ldqo; float qmul;          // osuf_mqa_fwd_rope: q arrives un-rotated; rotated * qmul it is used here and (qout != null) stored for the backward
  float* zdq;                                   // osuf_mqa_fwd_zdq: the backward's fp32 dQ accumulator [B*N][H*64], zero-filled by the forward kernel (null: not)
  float cexp, kmul;                             // exponent multiplier of S (scale * log2 e; 1 when q arrives pre-scaled by it: the *_qs entry points)
                                                // and the multiplier of the dK sums (scale; scale / cexp = 1 / log2 e for pre-scaled q)
  int qsplit;                                   // dK/dV: > 1 = the query range is cut into qsplit parts (short sequences: more workgroups),
  float* wsk; float* wsv;                       //        per-part partial sums in [qsplit][B*N][64] workspaces, summed by dkv_finish_kernel
  const bf16_t* mask; long mask_b, mask_h, mask_q, mask_k;   // osuf_mqa_fwd_masked: additive bf16 score bias, element strides (0 = broadcast)
};

// cooperative K/V tile stage: NT threads move one 64-key tile (64 x 128 B of K and of V = 512 + 512 16-B chunks)
template <int NT>
struct KVStage {
  static constexpr int PER = 512 / NT;
  u32x4 rk[PER], rv[PER];
  __device__ __forceinline__ void load(const AttnArgs& a, int b, int key0, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cid = tid + i * NT, row = cid >> 3, chunk = cid & 7;
      const int key = key0 + row;
      u32x4 z = {0u, 0u, 0u, 0u};
      rk[i] = z; rv[i] = z;
      if (key < a.N) {
        const long m = (long)b * a.N + key;
        rk[i] = *reinterpret_cast<const u32x4*>(a.k + m * a.ldk + chunk * 8);
        rv[i] = *reinterpret_cast<const u32x4*>(a.v + m * a.ldv + chunk * 8);
      }
    }
  }
  __device__ __forceinline__ void store(char* ks, char* vs, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cid = tid + i * NT, row = cid >> 3, chunk = cid & 7;
      *reinterpret_cast<u32x4*>(ks + tile_off(row, chunk * 16)) = rk[i];
      *reinterpret_cast<u32x4*>(vs + tile_off(row, chunk * 16)) = rv[i];
    }
  }
};

// the same stage with RUNNING source pointers (forward kernel, round 5): the tile loop's loads are `pointer += 64 rows` instead of two 64-bit
// multiply-adds per tile (the compiled loop spent ~21 vector instructions per tile, four of them quarter-rate v_mul_lo_u32, on these addresses)
template <int NT>
struct KVStageInc {
  static constexpr int PER = 512 / NT;
  u32x4 rk[PER], rv[PER];
  const bf16_t* pk[PER];
  const bf16_t* pv[PER];
  int row[PER];
  long stepk, stepv;
  __device__ __forceinline__ void init(const AttnArgs& a, int b, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cid = tid + i * NT, chunk = cid & 7;
      row[i] = cid >> 3;
      const long m = (long)b * a.N + row[i];
      pk[i] = a.k + m * a.ldk + chunk * 8;
      pv[i] = a.v + m * a.ldv + chunk * 8;
    }
    stepk = 64 * a.ldk; stepv = 64 * a.ldv;
  }
  template <bool WHOLE = false>                                            // WHOLE: N % 64 == 0, no row of any tile lies beyond N (no zero fill, no exec mask)
  __device__ __forceinline__ void load(const AttnArgs& a, int key0) {      // tiles in order: key0 = 0, 64, 128, ...
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if constexpr (WHOLE) {
        rk[i] = *reinterpret_cast<const u32x4*>(pk[i]);
        rv[i] = *reinterpret_cast<const u32x4*>(pv[i]);
      } else {
        u32x4 z = {0u, 0u, 0u, 0u};
        rk[i] = z; rv[i] = z;
        if (key0 + row[i] < a.N) {
          rk[i] = *reinterpret_cast<const u32x4*>(pk[i]);
          rv[i] = *reinterpret_cast<const u32x4*>(pv[i]);
        }
      }
      pk[i] += stepk; pv[i] += stepv;
    }
  }
  __device__ __forceinline__ void store(char* ks, char* vs, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cid = tid + i * NT, r = cid >> 3, chunk = cid & 7;
      *reinterpret_cast<u32x4*>(ks + tile_off(r, chunk * 16)) = rk[i];
      *reinterpret_cast<u32x4*>(vs + tile_off(r, chunk * 16)) = rv[i];
    }
  }
};

// ------------------------------------------------------------------------------------------------------
// forward
// (A two-wave-group variant skewed by half a tile with LDS-DMA staging was measured at 590-605 TFLOP/s against 803 for
//  this single-phase loop -- two barriers per tile and the serial MFMA -> max -> exp chain cost more than the overlap won.)
// ------------------------------------------------------------------------------------------------------
// QS (queries pre-scaled by scale * log2 e, osuf_mqa_fwd_qs): the scores ARE the exponents, so the running maximum rides the S chain as its C
// operand (S' = K Qs^T - m_run: 16 registers that change only when a row is rescaled) and p = exp2(S') needs no per-element fma
// WHOLE (N % 64 == 0): the K / V loads carry no bounds check and there is no masked copy of the tile body
// ROPE (osuf_mqa_fwd_rope): the query tile is rotated (rcos / rsin) and multiplied by qmul in the prologue, rounded to bf16 once -- the arithmetic of
// rope_cast_kernel, which then only has the K | V columns left to do -- and written to qout for the backward when that is given
template <int NW, bool QS = false, bool WHOLE = false, bool ROPE = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void mqa_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2][K 8K | V 8K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int nqb = (a.N + 31) >> 5;
  const int vb = blockIdx.x * NW + wave;
  const bool active = vb < nqb * a.H;
  const int h = active ? vb % a.H : 0, pb = active ? vb / a.H : 0;
  const int qrow = pb * 32 + lr;
  const bool qok = active && qrow < a.N;
  const float c = a.cexp;

  // Q tile of this wave: 32 rows x 128 B in a wave-private LDS slice behind the K/V ring (re-read once per k-step).  Keeping
  // the 16 fragment registers live across the loop pushed the kernel over 128 VGPRs; at <= 128 four waves per SIMD fit
  // (two workgroups per CU): measured 812 -> ~890 TFLOP/s.
  char* qtile = smem + 32768 + wave * 4096;
  // the first K / V tile is requested BEFORE the query tile is built: behind the ROPE form's rotation (loads -> wait -> arithmetic) its latency
  // came on top of the queries' (+1.75 us per workgroup, measured)
  KVStageInc<NW * 64> st;
  st.init(a, b, tid);
  st.template load<WHOLE>(a, 0);
  {
    const bf16_t* qp = a.q + ((long)b * a.N + qrow) * a.ldq + h * D;
    if constexpr (ROPE) {
      // the lane's chunks ks and ks + 2 hold columns c and c + 32 of the head: the rotation's partners are lane-local
      u32x4 zq[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
      if (qok) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float x1[8], x2[8], cs[8], sn[8];
          load8(qp + 16 * pr + 8 * lh, x1);
          load8(qp + 32 + 16 * pr + 8 * lh, x2);
          load8(a.rcos + (long)qrow * 32 + 16 * pr + 8 * lh, cs);
          load8(a.rsin + (long)qrow * 32 + 16 * pr + 8 * lh, sn);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float a1 = x1[e] * cs[e] - x2[e] * sn[e];
            const float a2 = x2[e] * cs[e] + x1[e] * sn[e];
            x1[e] = a1 * a.qmul; x2[e] = a2 * a.qmul;
          }
          store8(reinterpret_cast<bf16_t*>(&zq[pr]), x1);
          store8(reinterpret_cast<bf16_t*>(&zq[pr + 2]), x2);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<u32x4*>(qtile + tile_off(lr, (2 * ks + lh) * 16)) = zq[ks];
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u};
      if (qok) z = *reinterpret_cast<const u32x4*>(qp + 16 * ks + 8 * lh);
      *reinterpret_cast<u32x4*>(qtile + tile_off(lr, (2 * ks + lh) * 16)) = z;
    }
    }
  }
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = QS ? 0.f : -INFINITY, l_run = 0.f;

  const LaneOffs lo(lane);
  const int ntiles = (a.N + 63) >> 6;
  st.store(smem, smem + 8192, tid);
  __syncthreads();
  // one 64-key tile; MASK only for a ragged last tile (a branch-free mask on every tile costs 64 VALU ops per tile)
  auto tile = [&](int j, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const char* ks_ = smem + (j & 1) * 16384;
    const char* vs_ = ks_ + 8192;
#ifndef OSUF_FWD_TRIAGE_NOLOAD
    if (j + 1 < ntiles) st.template load<WHOLE>(a, (j + 1) * 64);
#endif
    // S^T = K Q^T  (two 32-key tiles)
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
    f32x16 negm;                                            // QS: -m_run in every register, the C operand of the S chain's first MFMA
    if constexpr (QS) {
      float nm = -m_run;
      asm volatile("" : "+v"(nm));                          // rebuilt per tile -- 16 moves -- instead of 16 registers live across the loop: kept
#pragma unroll                                              // live the kernel needs 128 VGPRs + 40 B of scratch (971 vs 993 TFLOP/s at N = 4096)
      for (int r = 0; r < 16; ++r) negm[r] = nm;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 qf = lds_row_frag(qtile, lo, ks, 0);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(ks_, lo, ks, kt), qf, (QS && ks == 0) ? negm : s[kt], 0, 0, 0);
    }
    if constexpr (MASK) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = j * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= a.N) s[kt][r] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    float ps0 = 0.f, ps1 = 0.f;
    if constexpr (QS) {
      // s holds S - m_run already.  Rescale (wave-uniform branch) when a row's maximum grew by more than 2^6 -- and on the first tile, which
      // sets the running maximum exactly (its accumulators are still zero: no alpha there)
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (j == 0 || __any(mx > 6.0f)) {
        const float d = j == 0 ? mx : fmaxf(mx, 0.f);
        const float alpha = j == 0 ? 1.f : fast_exp2(-d);
        m_run += d;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] -= d;
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float p0 = fast_exp2(s[kt][r]);
          float p1 = fast_exp2(s[kt][r + 1]);
          s[kt][r] = p0; s[kt][r + 1] = p1;
          ps0 += p0; ps1 += p1;
        }
    } else {
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
    // lazy rescale: keep the running max while no row's max grew by more than 2^6 (P <= 64: bf16 rounding is scale-free);
    // the branch is wave-uniform
    if (__any(mx > m_run + 6.0f)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    // (the packed forms v_pk_fma_f32 / v_pk_add_f32 of these two elementwise streams were measured 4 % SLOWER: N = 8192 9.36 -> 9.75 ms)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float p0 = fast_exp2(fmaf(s[kt][r], c, -m_run));
        float p1 = fast_exp2(fmaf(s[kt][r + 1], c, -m_run));
        s[kt][r] = p0; s[kt][r + 1] = p1;
        ps0 += p0; ps1 += p1;
      }
    }
    l_run += ps0 + ps1;
    // O^T += V^T P^T
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const bf16x8 pf = acc_to_frag(s[s4 >> 1], s4 & 1);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(vs_, lo, (s4 >> 1) * 32 + (s4 & 1) * 16, dt), pf, o[dt], 0, 0, 0);
    }
#ifndef OSUF_FWD_TRIAGE_NOLOAD
    if (j + 1 < ntiles) st.store(smem + ((j + 1) & 1) * 16384, smem + ((j + 1) & 1) * 16384 + 8192, tid);
#endif
#ifndef OSUF_FWD_TRIAGE_NOBARRIER
    __syncthreads();
#endif
  };
  const bool ragged = !WHOLE && (a.N & 63) != 0;
  const int nfull = ragged ? ntiles - 1 : ntiles;                   // one unmasked copy of the tile body: with a loop AND a peeled unmasked twin
  for (int j = 0; j < nfull; ++j) tile(j, std::false_type{});      // hipcc spilled 200 B per lane around the twins (round 4: 128 -> 122 VGPRs, no scratch)
  if constexpr (!WHOLE) { if (ragged) tile(ntiles - 1, std::true_type{}); }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qok) {
    const float inv = 1.f / l_tot;
    if (lh == 0) a.lse2[((long)b * a.H + h) * a.N + qrow] = m_run + __builtin_amdgcn_logf(l_tot);   // v_log_f32 = log2
    const long m = (long)b * a.N + qrow;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = round_bf16(o[dt][4 * g + e] * inv);
        const int d0 = dt * 32 + 8 * g + 4 * lh;
        if (a.o_is_f32) store4(reinterpret_cast<float*>(a.o) + m * a.ldo + h * D + d0, v4);
        else store4(reinterpret_cast<bf16_t*>(a.o) + m * a.ldo + h * D + d0, v4);
      }
  }
  // osuf_mqa_fwd_rope: the rotated query tile goes to memory for the backward from its LDS slice HERE, behind the loop -- stored in the prologue, the
  // stores' completion sat in front of the first barrier (vmcnt counts stores): +1.75 us per workgroup, the whole gain of the fusion
  if constexpr (ROPE) {
    if (a.qout != nullptr && qok) {
      bf16_t* qo = a.qout + ((long)b * a.N + qrow) * a.ldqo + h * D;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        *reinterpret_cast<u32x4*>(qo + 16 * ks + 8 * lh) = *reinterpret_cast<const u32x4*>(qtile + tile_off(lr, (2 * ks + lh) * 16));
    }
  }
  // osuf_mqa_fwd_zdq: this wave's 32 rows x 256 B of the backward's dQ accumulator are cleared here -- the loop above is bound by the vector pipe with
  // HBM idle (0.16 TB/s), so the 8 stores per wave are free where the hipMemsetAsync in front of the backward sweep cost 66 us per N = 4096 layer
  if (a.zdq != nullptr && active) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = pb * 32 + i * 4 + (lane >> 4);
      if (r < a.N) *reinterpret_cast<u32x4*>(a.zdq + ((long)b * a.N + r) * ((long)a.H * D) + h * D + (lane & 15) * 4) = z;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// backward, dQ: query-stationary.  dQ^T[d][q] = sum_key K^T[d][key] * dS^T[key][q]
// ------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void mqa_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int nqb = (a.N + 31) >> 5;
  const int vb = blockIdx.x * NW + wave;
  const bool active = vb < nqb * a.H;
  const int h = active ? vb % a.H : 0, pb = active ? vb / a.H : 0;
  const int qrow = pb * 32 + lr;
  const bool qok = active && qrow < a.N;
  const float c = a.cexp;

  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qp = a.q + ((long)b * a.N + qrow) * a.ldq + h * D;
    const bf16_t* dp = a.dout + ((long)b * a.N + qrow) * a.lddo + h * D;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u}, z2 = {0u, 0u, 0u, 0u};
      if (qok) { z = *reinterpret_cast<const u32x4*>(qp + 16 * ks + 8 * lh); z2 = *reinterpret_cast<const u32x4*>(dp + 16 * ks + 8 * lh); }
      qf[ks] = __builtin_bit_cast(bf16x8, z);
      dof[ks] = __builtin_bit_cast(bf16x8, z2);
    }
  }
  const long sidx = ((long)b * a.H + h) * a.N + qrow;
  const float L2 = qok ? a.lse2[sidx] : INFINITY;
  const float dl = qok ? a.delta[sidx] : 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const LaneOffs lo(lane);
  const int ntiles = (a.N + 63) >> 6;
  KVStage<NW * 64> st;
  st.load(a, b, 0, tid);
  st.store(smem, smem + 8192, tid);
  __syncthreads();
  for (int j = 0; j < ntiles; ++j) {
    const char* ks_ = smem + (j & 1) * 16384;
    const char* vs_ = ks_ + 8192;
    if (j + 1 < ntiles) st.load(a, b, (j + 1) * 64, tid);
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(ks_, lo, ks, kt), qf[ks], s[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(vs_, lo, ks, kt), dof[ks], dp[kt], 0, 0, 0);
      }
    }
    // dS^T = P^T * (dP^T - delta) * scale   (zero K rows make masked keys contribute nothing)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = fast_exp2(fmaf(s[kt][r], c, -L2));
        s[kt][r] = p * (dp[kt][r] - dl);          // * scale folded into the final store
      }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const bf16x8 df = acc_to_frag(s[s4 >> 1], s4 & 1);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(ks_, lo, (s4 >> 1) * 32 + (s4 & 1) * 16, dt), df, acc[dt], 0, 0, 0);
    }
    if (j + 1 < ntiles) st.store(smem + ((j + 1) & 1) * 16384, smem + ((j + 1) & 1) * 16384 + 8192, tid);
    __syncthreads();
  }
  if (qok) store_grad(a.dq, a.lddq, (long)b * a.N + qrow, h * D, a.g_bf16, acc, a.scale, a.rcos, a.rsin, qrow, lh);
}

// ------------------------------------------------------------------------------------------------------
// dQ, software-pipelined inside the wave (8 waves; same finding as for dK/dV below: with one workgroup per CU the plain loop's
// MFMA time and exp/VALU time add up instead of overlapping).  Iteration j issues, interleaved by hand and pinned with
// sched_barrier fences: the S^T / dP^T products of key tile j+1 (16 MFMAs), the exp / dS arithmetic of tile j (32 elements per
// lane, in 8 slices), and the dQ products of tile j-1 (8 MFMAs, from the bf16 dS fragments carried over).  Three key tiles are
// live in LDS and a fourth is being filled: 4-slot ring of (K 8 KiB | V 8 KiB).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mqa_bwd_dq_pipe_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [4][K 8K | V 8K]
  constexpr int NW = 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int nqb = (a.N + 31) >> 5;
  const int vb = blockIdx.x * NW + wave;
  const bool active = vb < nqb * a.H;
  const int h = active ? vb % a.H : 0, pb = active ? vb / a.H : 0;
  const int qrow = pb * 32 + lr;
  const bool qok = active && qrow < a.N;
  const float c = a.cexp;

  bf16x8 qf[4], dof[4];
  {
    const bf16_t* qp = a.q + ((long)b * a.N + qrow) * a.ldq + h * D;
    const bf16_t* dp = a.dout + ((long)b * a.N + qrow) * a.lddo + h * D;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u}, z2 = {0u, 0u, 0u, 0u};
      if (qok) { z = *reinterpret_cast<const u32x4*>(qp + 16 * ks + 8 * lh); z2 = *reinterpret_cast<const u32x4*>(dp + 16 * ks + 8 * lh); }
      qf[ks] = __builtin_bit_cast(bf16x8, z);
      dof[ks] = __builtin_bit_cast(bf16x8, z2);
    }
  }
  const long sidx = ((long)b * a.H + h) * a.N + qrow;
  const float L2 = qok ? a.lse2[sidx] : INFINITY;
  const float dl = qok ? a.delta[sidx] : 0.f;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const LaneOffs lo(lane);
  const int ntiles = (a.N + 63) >> 6;
  KVStage<NW * 64> st;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto tile_k = [&](int slot) { return smem + slot * 16384; };

  // S^T / dP^T of one tile, unpipelined (prologue only)
  auto qk_plain = [&](int slot, f32x16 (&s)[2], f32x16 (&dp)[2]) {
    const char* ks_ = tile_k(slot);
    const char* vs_ = ks_ + 8192;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      s[kt] = zero16; dp[kt] = zero16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(ks_, lo, ks, kt), qf[ks], s[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(vs_, lo, ks, kt), dof[ks], dp[kt], 0, 0, 0);
      }
    }
  };
  // dQ products of one tile from its bf16 dS fragments, unpipelined (epilogue only)
  auto dq_plain = [&](int slot, const bf16x8 (&df)[4]) {
    const char* ks_ = tile_k(slot);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(ks_, lo, (s4 >> 1) * 32 + (s4 & 1) * 16, dt), df[s4], acc[dt], 0, 0, 0);
  };

#define OSUF_FENCE __builtin_amdgcn_sched_barrier(0)
  // iteration j: (sc, dpc) = S^T / dP^T of tile j (slot_c irrelevant: registers only); tile j+1 in slot_n; tile j-1 in slot_p with
  // its dS fragments in dfp; the tile loaded two ahead goes to slot_ld.  On return dfp holds tile j's fragments.
  auto iter = [&](int j, int slot_p, int slot_n, int slot_ld, f32x16 (&sc)[2], f32x16 (&dpc)[2], f32x16 (&sn)[2], f32x16 (&dpn)[2],
                  bf16x8 (&dfp)[4]) {
    const char* kn = tile_k(slot_n);
    const char* vn = kn + 8192;
    const char* kp = tile_k(slot_p);
    const bool more2 = j + 2 < ntiles;
    if (more2) st.load(a, b, (j + 2) * 64, tid);
    OSUF_FENCE;
    // 8 steps (kt = i>>2, ks = i&3): 3 MFMAs on the fragments fetched during the previous step, then the fetch for the next
    // step, then a 4-element slice of the exp / dS arithmetic (which hides that fetch's LDS latency)
    bf16x8 fk[2], fv[2], tk[2];
    auto fetch = [&](int i, int set) {
      const int kt = i >> 2, ks = i & 3, s4 = i >> 1;
      fk[set] = lds_row_frag(kn, lo, ks, kt);
      fv[set] = lds_row_frag(vn, lo, ks, kt);
      tk[set] = lds_tr_frag(kp, lo, (s4 >> 1) * 32 + (s4 & 1) * 16, i & 1);
    };
    fetch(0, 0);
    OSUF_FENCE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kt = i >> 2, ks = i & 3, set = i & 1;
      sn[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[set], qf[ks], ks == 0 ? zero16 : sn[kt], 0, 0, 0);
      dpn[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[set], dof[ks], ks == 0 ? zero16 : dpn[kt], 0, 0, 0);
      acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk[set], dfp[i >> 1], acc[i & 1], 0, 0, 0);
      if (i < 7) fetch(i + 1, set ^ 1);
      // dS^T = P^T * (dP^T - delta) * scale for 4 of this lane's 32 (key, query) elements of tile j
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * ks + e;
        float p = fast_exp2(fmaf(sc[kt][r], c, -L2));
        sc[kt][r] = p * (dpc[kt][r] - dl);       // * scale folded into the final store
        asm volatile("" : "+v"(sc[kt][r]));            // keep the slice HERE (its only consumer is the conversion at the end of the
      }                                                  // iteration, and code sinking would otherwise move all 32 exps behind the MFMAs)
      OSUF_FENCE;
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) dfp[s4] = acc_to_frag(sc[s4 >> 1], s4 & 1);
    OSUF_FENCE;
    if (more2) st.store(tile_k(slot_ld), tile_k(slot_ld) + 8192, tid);
    __syncthreads();
  };
#undef OSUF_FENCE

  st.load(a, b, 0, tid);
  st.store(tile_k(0), tile_k(0) + 8192, tid);
  if (ntiles > 1) { st.load(a, b, 64, tid); st.store(tile_k(1), tile_k(1) + 8192, tid); }
  __syncthreads();
  f32x16 s0[2], dp0[2], s1[2], dp1[2];
  bf16x8 dfp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dfp[i] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});   // "tile -1": adds zero
  qk_plain(0, s0, dp0);
  // slots: tile t lives in slot t & 3; on the first iteration the "previous" tile is a zero fragment times whatever slot 3 holds
  // -- uninitialised LDS could be NaN/Inf bits, so clear slot 3's K half once
  for (int i = tid; i < 8192 / 16; i += NW * 64) reinterpret_cast<u32x4*>(tile_k(3))[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  for (int j = 0; j < ntiles; j += 2) {
    iter(j, (j + 3) & 3, (j + 1) & 3, (j + 2) & 3, s0, dp0, s1, dp1, dfp);
    if (j + 1 < ntiles) iter(j + 1, j & 3, (j + 2) & 3, (j + 3) & 3, s1, dp1, s0, dp0, dfp);
  }
  dq_plain((ntiles - 1) & 3, dfp);
  if (qok) store_grad(a.dq, a.lddq, (long)b * a.N + qrow, h * D, a.g_bf16, acc, a.scale, a.rcos, a.rsin, qrow, lh);
}

// ------------------------------------------------------------------------------------------------------
// backward, dK/dV: key-stationary.  A wave owns 32 keys; the workgroup (8 waves = 256 keys) sweeps every
// (head, 32-query block) pair, staging Q / dO / lse / delta of the pair in LDS for all waves.
// ------------------------------------------------------------------------------------------------------
// DBG (triage builds via OSUF_ATTN_DBG, results garbage unless 0): 1 = no stage loads/stores (stage 0 reused), 2 = 1 + no barrier,
// 3 = no softmax VALU, 4 = no LDS fragment reads, 5 = no MFMA.
template <int NW, int DBG = 0>
__global__ __launch_bounds__(NW * 64) void mqa_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2][Q 4K | dO 4K | lse 128 | delta 128]
  constexpr int kStage = 4096 + 4096 + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware order (speed only): every key block of a sample sweeps the same Q / dO / lse / delta; block ids are dealt
  // round-robin over the 8 XCDs, so ids 8 apart share an L2.  PMC before this remap: 4.46 GB fetched per launch at
  // N=4096 against 0.54 GB of distinct Q + dO bytes.
  const int nkb = (a.N + 32 * NW - 1) / (32 * NW);
  const int xcd = blockIdx.x & 7, qid = blockIdx.x >> 3;
  const int b = (qid / nkb) * 8 + xcd;
  const int kb = qid % nkb;
  if (b >= a.B) return;
  const int key = kb * (32 * NW) + wave * 32 + lr;
  const bool kok = key < a.N;
  const float c = a.cexp;
  const int nqb = (a.N + 31) >> 5;
  const int niter = nqb * a.H;

  bf16x8 kf[4], vf[4];
  {
    const long m = (long)b * a.N + key;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u}, z2 = {0u, 0u, 0u, 0u};
      if (kok) { z = *reinterpret_cast<const u32x4*>(a.k + m * a.ldk + 16 * ks + 8 * lh); z2 = *reinterpret_cast<const u32x4*>(a.v + m * a.ldv + 16 * ks + 8 * lh); }
      kf[ks] = __builtin_bit_cast(bf16x8, z);
      vf[ks] = __builtin_bit_cast(bf16x8, z2);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  // stage loader: 512 16-B chunks per (head, query block): chunks 0..255 = Q tile, 256..511 = dO tile; threads 0..31 lse,
  // 32..63 delta
  constexpr int NT = NW * 64, PER = 512 / NT;
  u32x4 rt[PER]; float rs = 0.f;
  auto load_stage = [&](int it) {
    const int h = it % a.H, pb = it / a.H;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cid = tid + i * NT, t = cid & 255, row = t >> 3, chunk = t & 7;
      const int qrow = pb * 32 + row;
      u32x4 z = {0u, 0u, 0u, 0u};
      rt[i] = z;
      if (qrow < a.N) {
        const long m = (long)b * a.N + qrow;
        rt[i] = (cid < 256) ? *reinterpret_cast<const u32x4*>(a.q + m * a.ldq + h * D + chunk * 8)
                            : *reinterpret_cast<const u32x4*>(a.dout + m * a.lddo + h * D + chunk * 8);
      }
    }
    if (tid < 64) {
      const int qr = pb * 32 + (tid & 31);
      const long sidx = ((long)b * a.H + h) * a.N + qr;
      if (tid < 32) rs = qr < a.N ? a.lse2[sidx] : INFINITY;
      else rs = qr < a.N ? a.delta[sidx] : 0.f;
    }
  };
  auto store_stage = [&](int buf) {
    char* base = smem + buf * kStage;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int cid = tid + i * NT, t = cid & 255, row = t >> 3, chunk = t & 7;
      *reinterpret_cast<u32x4*>(base + (cid < 256 ? 0 : 4096) + tile_off(row, chunk * 16)) = rt[i];
    }
    if (tid < 64) reinterpret_cast<float*>(base + 8192)[tid] = rs;
  };

  const LaneOffs lo(lane);
  load_stage(0);
  store_stage(0);
  __syncthreads();
  bf16x8 cfrag = __builtin_bit_cast(bf16x8, u32x4{(uint32_t)lane, 1u, 2u, 3u});
  for (int it = 0; it < niter; ++it) {
    const char* qs = smem + ((DBG == 1 || DBG == 2) ? 0 : (it & 1)) * kStage;
    const char* dos = qs + 4096;
    const float* ls = reinterpret_cast<const float*>(qs + 8192);
    if (DBG != 1 && DBG != 2 && it + 1 < niter) load_stage(it + 1);
    // S[q][key] = Q K^T ; dP[q][key] = dO V^T    (q rows in registers, key on the lane)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    if (DBG == 4) asm volatile("" : "+v"(cfrag));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 fq = DBG == 4 ? cfrag : lds_row_frag(qs, lo, ks, 0);
      const bf16x8 fo = DBG == 4 ? cfrag : lds_row_frag(dos, lo, ks, 0);
      if (DBG == 5) { asm volatile("" ::"v"(fq), "v"(fo)); continue; }
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq, kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, vf[ks], dp, 0, 0, 0);
    }
    if (DBG == 5) asm volatile("" : "+v"(s), "+v"(dp));
    f32x16 ds;
    if (DBG == 3) {
      ds = dp;
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(ls + 8 * g + 4 * lh);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float p = fast_exp2(fmaf(s[r], c, -l4[e]));
          s[r] = p;
          ds[r] = p * (dp[r] - d4[e]);              // * scale folded into the final store
        }
      }
    }
    // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = acc_to_frag(s, s2);
      const bf16x8 df = acc_to_frag(ds, s2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16x8 fo = DBG == 4 ? cfrag : lds_tr_frag(dos, lo, s2 * 16, dt);
        const bf16x8 fq = DBG == 4 ? cfrag : lds_tr_frag(qs, lo, s2 * 16, dt);
        if (DBG == 5) { asm volatile("" ::"v"(fq), "v"(fo), "v"(pf), "v"(df)); continue; }
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq, df, dk[dt], 0, 0, 0);
      }
    }
    if (DBG != 1 && DBG != 2 && it + 1 < niter) store_stage((it + 1) & 1);
    if (DBG != 2) __syncthreads();
  }
  if (kok) {
    store_grad(a.dk, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dk, a.kmul, a.rcos, a.rsin, key, lh);
    store_grad(a.dv, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dv, 1.f, nullptr, nullptr, key, lh);
  }
}

// ------------------------------------------------------------------------------------------------------
// dK/dV, software-pipelined inside the wave (8 waves).  The triage builds of the plain kernel (OSUF_ATTN_DBG) showed its time
// is the SUM of the MFMA time and of everything else (5.16 ms = 1.8 + 3.05 at N=4096): with 211 VGPRs there are two waves per
// SIMD, both of one workgroup and in barrier lock-step, and inside a wave the stage is a strict chain
// MFMA(S, dP) -> exp/VALU -> MFMA(dV, dK), so the matrix pipe idles during the VALU phase and vice versa.  Here iteration `it`
// issues the S / dP products of stage it+1 (independent of everything else in the iteration) next to the exp/VALU work of
// stage it, then the dV / dK products of stage it.  Two stages are live in LDS, so the ring has three slots and the global
// prefetch runs two stages ahead.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mqa_bwd_dkv_pipe_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [3][Q 4K | dO 4K | lse 128 | delta 128]
  constexpr int NW = 8, kStage = 4096 + 4096 + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int nkb = (a.N + 32 * NW - 1) / (32 * NW);
  const int per_split = (int)gridDim.x / a.qsplit;                 // workgroups of one query part (a multiple of 8)
  const int part = (int)blockIdx.x / per_split, bid = (int)blockIdx.x - part * per_split;
  const int xcd = bid & 7, qid = bid >> 3;
  const int b = (qid / nkb) * 8 + xcd;
  const int kb = qid % nkb;
  if (b >= a.B) return;
  const int key = kb * (32 * NW) + wave * 32 + lr;
  const bool kok = key < a.N;
  const float c = a.cexp;
  const int nqb = (a.N + 31) >> 5;
  const int qb_per = (nqb + a.qsplit - 1) / a.qsplit;
  const int qb_begin = part * qb_per, qb_end = min(nqb, qb_begin + qb_per);
  if (qb_begin >= qb_end) return;                                  // uniform per workgroup, before any barrier
  const int niter = (qb_end - qb_begin) * a.H;

  bf16x8 kf[4], vf[4];
  {
    const long m = (long)b * a.N + key;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u}, z2 = {0u, 0u, 0u, 0u};
      if (kok) { z = *reinterpret_cast<const u32x4*>(a.k + m * a.ldk + 16 * ks + 8 * lh); z2 = *reinterpret_cast<const u32x4*>(a.v + m * a.ldv + 16 * ks + 8 * lh); }
      kf[ks] = __builtin_bit_cast(bf16x8, z);
      vf[ks] = __builtin_bit_cast(bf16x8, z2);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  // stage loader (512 threads, one 16-B chunk each): chunks 0..255 = Q tile, 256..511 = dO tile; threads 0..31 lse, 32..63 delta
  const int lt = tid & 255, lrow = lt >> 3, lchunk = lt & 7;
  const bf16_t* lsrc = tid < 256 ? a.q : a.dout;
  const long lld = tid < 256 ? a.ldq : a.lddo;
  const int lds_dst = (tid < 256 ? 0 : 4096) + tile_off(lrow, lchunk * 16);
  u32x4 rt; float rs = 0.f;
  int ih = 0, ipb = qb_begin;                                     // (head, query block) of the next stage to load
  auto load_stage = [&]() {
    const int qrow = ipb * 32 + lrow;
    u32x4 z = {0u, 0u, 0u, 0u};
    rt = z;
    if (qrow < a.N) rt = *reinterpret_cast<const u32x4*>(lsrc + ((long)b * a.N + qrow) * lld + ih * D + lchunk * 8);
    if (tid < 64) {
      const int qr = ipb * 32 + (tid & 31);
      const long sidx = ((long)b * a.H + ih) * a.N + qr;
      if (tid < 32) rs = qr < a.N ? a.lse2[sidx] : INFINITY;
      else rs = qr < a.N ? a.delta[sidx] : 0.f;
    }
    if (++ih == a.H) { ih = 0; ++ipb; }
  };
  auto store_stage = [&](int slot) {
    char* base = smem + slot * kStage;
    *reinterpret_cast<u32x4*>(base + lds_dst) = rt;
    if (tid < 64) reinterpret_cast<float*>(base + 8192)[tid] = rs;
  };

  const LaneOffs lo(lane);
  auto qk = [&](int slot, f32x16& s, f32x16& dp) {                 // S[q][key] = Q K^T ; dP[q][key] = dO V^T
    const char* qs = smem + slot * kStage;
    const char* dos = qs + 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(qs, lo, ks, 0), kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(dos, lo, ks, 0), vf[ks], dp, 0, 0, 0);
    }
  };
  // one pipelined iteration: (sc, dpc) hold S / dP of stage `it` (slot), (sn, dpn) receive those of stage it+1 (slot_n).
  // The issue order is written out by hand and pinned with sched_barrier fences (left alone, hipcc clusters the 16 exps first
  // and the 16 MFMAs last): an MFMA pair of the next stage, then a quarter of this stage's exp / VALU work, ...; the dV / dK
  // products of query rows 0-15 start as soon as their P / dS rows exist.
#define OSUF_FENCE __builtin_amdgcn_sched_barrier(0)
  auto iter = [&](int it, int slot, int slot_n, int slot_ld, f32x16& sc, f32x16& dpc, f32x16& sn, f32x16& dpn) {
    const char* qs = smem + slot * kStage;
    const char* dos = qs + 4096;
    const float* ls = reinterpret_cast<const float*>(qs + 8192);
    const char* qn = smem + slot_n * kStage;
    const char* don = qn + 4096;
    const bool more2 = it + 2 < niter;
    if (more2) load_stage();
    OSUF_FENCE;
    bf16x8 fq[4], fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { fq[ks] = lds_row_frag(qn, lo, ks, 0); fo[ks] = lds_row_frag(don, lo, ks, 0); }
    f32x4 l4[4], d4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      l4[g] = *reinterpret_cast<const f32x4*>(ls + 8 * g + 4 * lh);
      d4[g] = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * g + 4 * lh);
    }
    f32x16 ds;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto valu_quarter = [&](int g) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float p = fast_exp2(fmaf(sc[r], c, -l4[g][e]));
        sc[r] = p;
        ds[r] = p * (dpc[r] - d4[g][e]);         // * scale folded into the final store
      }
    };
    OSUF_FENCE;
    sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[0], kf[0], zero16, 0, 0, 0);
    dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[0], vf[0], zero16, 0, 0, 0);
    valu_quarter(0);
    OSUF_FENCE;
    sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[1], kf[1], sn, 0, 0, 0);
    dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[1], vf[1], dpn, 0, 0, 0);
    valu_quarter(1);
    OSUF_FENCE;
    // transposed fragments of rows 0-15 + their P / dS operands while the next MFMA pair runs
    sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[2], kf[2], sn, 0, 0, 0);
    dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[2], vf[2], dpn, 0, 0, 0);
    bf16x8 to0[2], tq0[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { to0[dt] = lds_tr_frag(dos, lo, 0, dt); tq0[dt] = lds_tr_frag(qs, lo, 0, dt); }
    const bf16x8 pf0 = acc_to_frag(sc, 0), df0 = acc_to_frag(ds, 0);
    OSUF_FENCE;
    sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[3], kf[3], sn, 0, 0, 0);
    dpn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo[3], vf[3], dpn, 0, 0, 0);
    valu_quarter(2);
    OSUF_FENCE;
    // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]   (query rows 0-15)
    dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to0[0], pf0, dv[0], 0, 0, 0);
    dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq0[0], df0, dk[0], 0, 0, 0);
    valu_quarter(3);
    OSUF_FENCE;
    dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to0[1], pf0, dv[1], 0, 0, 0);
    dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq0[1], df0, dk[1], 0, 0, 0);
    bf16x8 to1[2], tq1[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { to1[dt] = lds_tr_frag(dos, lo, 16, dt); tq1[dt] = lds_tr_frag(qs, lo, 16, dt); }
    const bf16x8 pf1 = acc_to_frag(sc, 1), df1 = acc_to_frag(ds, 1);
    OSUF_FENCE;
    dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to1[0], pf1, dv[0], 0, 0, 0);
    dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq1[0], df1, dk[0], 0, 0, 0);
    dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to1[1], pf1, dv[1], 0, 0, 0);
    dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq1[1], df1, dk[1], 0, 0, 0);
    OSUF_FENCE;
    if (more2) store_stage(slot_ld);
    __syncthreads();
  };
#undef OSUF_FENCE

  load_stage(); store_stage(0);
  if (niter > 1) { load_stage(); store_stage(1); }
  __syncthreads();
  f32x16 s0, dp0, s1, dp1;
  qk(0, s0, dp0);
  int slot = 0;                                                   // slot of stage `it`; the ring advances by one per iteration
  for (int it = 0; it < niter; it += 2) {
    const int sA = slot, sB = slot == 2 ? 0 : slot + 1, sC = sB == 2 ? 0 : sB + 1;
    iter(it, sA, sB, sC, s0, dp0, s1, dp1);
    if (it + 1 < niter) iter(it + 1, sB, sC, sA, s1, dp1, s0, dp0);
    slot = sC;
  }
  if (kok && a.qsplit > 1) {                                       // this part's partial sums, plain 16-byte stores (atomics measured
    const long prow = ((long)part * a.B + b) * a.N + key;          // as slow as the iterations the split saves: 38 G atomics/s)
    store_grad_row(a.wsk + prow * D, dk, 1.f, nullptr, nullptr, lh);
    store_grad_row(a.wsv + prow * D, dv, 1.f, nullptr, nullptr, lh);
  } else if (kok) {
    store_grad(a.dk, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dk, a.kmul, a.rcos, a.rsin, key, lh);
    store_grad(a.dv, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dv, 1.f, nullptr, nullptr, key, lh);
  }
}

// ------------------------------------------------------------------------------------------------------
// Fused backward: the key-stationary dK/dV sweep ALSO produces dQ, so S / dP / exp are computed once per (query, key) pair
// (5 matrix products per tile instead of the 7 of the dQ + dK/dV kernel pair).  A wave owns 32 keys as above (S, dP with the key on
// the lane: the P / dS accumulators are directly the B operands of the dV^T / dK^T products).  dQ sums over the KEY (lane) index,
// which needs one transpose: every wave stores its dS rows as bf16 into a [256 keys][32 queries] LDS image (double-buffered: the
// image written in iteration `it` is consumed in iteration it+1, behind the loop's one barrier), and in the next iteration the 8
// waves each take one 16 (query) x 16 (d) tile of dQ[32][64] = dS[32][256] K[256][64] -- v_mfma_f32_16x16x32_bf16 over the 256 keys,
// both operands read column-wise (ds_read_b64_tr_b16) from the dS image and from a resident K image of the workgroup's keys.
// The 256-key partial dQ is added to an fp32 dQ buffer with global float atomics (4 x 64-byte row segments per wave-instruction):
// 8 KiB per iteration per workgroup, i.e. B*H*N*64*4 bytes x N/256 per launch -- at the chip's ~1.3 TB/s atomic rate that is the
// floor of this kernel (6.6 ms at B=32, N=4096), still below the 9.0 ms of the two-kernel path.  A finishing pass scales, applies
// the RoPE transpose and casts (dq_finish_kernel).  Ragged shapes: padded query rows have Q = dO = 0 and lse = +inf (P = dS = 0),
// padded keys have zero K rows in the image (they add nothing to dQ) and are not stored.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ds_img_off(int key, int qchunk) {     // 8-byte chunk (4 queries) of a 64-byte row, xor-swizzled
  return key * 64 + ((qchunk ^ ((key ^ (key >> 3)) & 7)) << 3);
}

// DQ_MODE 0: dQ by fp32 atomics into dq32 [B*N][H*64] (measured 7.4 ms at B=32, N=4096: the atomic floor above); 1 / 2: every
// workgroup stores the dQ^T tiles of its 256 keys as bf16 / fp32 rows of its own slab ws[key block][b][n (padded to 32)][H*64]
// with plain 8 / 16-byte stores (5x the atomic byte rate, MI355X_MICROARCH "Global float atomics") and dq_reduce_kernel adds the
// slabs in key-block order -- deterministic, and the sweep is compute-bound again.  bf16 slabs round each 256-key partial once
// (relative 2^-9, the rounding the bf16 dq output gets anyway); fp32 slabs serve the fp32 compute mode.
// RAGGED = N is not a multiple of 32.  Whole-block shapes (every UNet level) take the lean form: no row clamps / selects, and
// every global address of the loop is a wave-uniform 64-bit base (SGPRs, advanced by scalar adds) plus a loop-invariant 32-bit
// lane offset -- the clamped per-lane 64-bit form cost ~50 of the loop's 134 VALU instructions.
template <int DQ_MODE, bool RAGGED>
__global__ __launch_bounds__(512) void mqa_bwd_fused_kernel(AttnArgs a, float* dq32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2][Q 4K | dO 4K | lse 128 | delta 128] | K image 32K | [2] dS image 16K
  constexpr int NW = 8, kStage = 4096 + 4096 + 256;
  char* kimg = smem + 2 * kStage;
  char* eimg = kimg + 32768;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int nkb = (a.N + 32 * NW - 1) / (32 * NW);
  const int per_split = (int)gridDim.x / a.qsplit;
  const int part = (int)blockIdx.x / per_split, bid = (int)blockIdx.x - part * per_split;
  const int xcd = bid & 7, qid = bid >> 3;
  const int b = (qid / nkb) * 8 + xcd;
  const int kb = qid % nkb;
  if (b >= a.B) return;
  const int key = kb * (32 * NW) + wave * 32 + lr;
  const bool kok = key < a.N;
  const float c = a.cexp;
  const int nqb = (a.N + 31) >> 5;
  const int qb_per = (nqb + a.qsplit - 1) / a.qsplit;
  const int qb_begin = part * qb_per, qb_end = min(nqb, qb_begin + qb_per);
  if (qb_begin >= qb_end) return;                                  // uniform per workgroup, before any barrier
  const int niter = (qb_end - qb_begin) * a.H;

  bf16x8 kf[4], vf[4];
  {
    const long m = (long)b * a.N + key;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u}, z2 = {0u, 0u, 0u, 0u};
      if (kok) { z = *reinterpret_cast<const u32x4*>(a.k + m * a.ldk + 16 * ks + 8 * lh); z2 = *reinterpret_cast<const u32x4*>(a.v + m * a.ldv + 16 * ks + 8 * lh); }
      kf[ks] = __builtin_bit_cast(bf16x8, z);
      vf[ks] = __builtin_bit_cast(bf16x8, z2);
      *reinterpret_cast<u32x4*>(kimg + tile_off(wave * 32 + lr, (2 * ks + lh) * 16)) = z;     // resident K image (zero rows beyond N)
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  // stage loader (512 threads, one 16-B chunk each): chunks 0..255 = Q tile, 256..511 = dO tile; threads 0..31 lse, 32..63 delta.
  // BRANCH-FREE on purpose: the kernel keeps float atomics in flight (they stay in the in-order vmcnt queue for ~3,000 cycles when
  // every CU issues them), and hipcc's wait-count pass falls back to vmcnt(0) -- i.e. a full atomic round trip per iteration --
  // as soon as a load or an atomic sits under an exec-masked or scalar branch.  So every lane always loads (row indices clamped,
  // the value zeroed by a select afterwards) and loads are issued BEFORE the atomics of the same iteration: waiting for them is
  // then a counted vmcnt(4) that leaves the four younger atomics in flight.
  const int lt = tid & 255, lrow = lt >> 3, lchunk = lt & 7;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // scalar copy: selects below stay in SGPRs
  const bf16_t* lsrc = wave_u < 4 ? a.q : a.dout;                 // waves 0-3 stage the Q tile, 4-7 the dO tile
  const long lld = wave_u < 4 ? a.ldq : a.lddo;
  const int lds_dst = (tid < 256 ? 0 : 4096) + tile_off(lrow, lchunk * 16);
  const float* ssrc = (RAGGED ? (tid & 32) != 0 : (wave_u & 1) != 0) ? a.delta : a.lse2;   // ragged: lanes 0-31 lse, 32-63 delta of wave 0;
  const float sfill = (tid & 32) ? 0.f : INFINITY;                                         // lean: wave 0 lse, wave 1 delta
  const unsigned loff = (unsigned)(lrow * (int)lld + lchunk * 8);   // (whole-block form) element offset of this lane's chunk in the tile
  const int last_pb = qb_end - 1;
  u32x4 rt; float rs = 0.f;
  int ih = 0, ipb = qb_begin;                                     // (head, query block) of the next stage to load
  auto load_stage = [&]() {
    const int pbc = min(ipb, last_pb);                             // past the end: reload the last block (never consumed)
    if constexpr (RAGGED) {
      const int qrow = pbc * 32 + lrow, qr = pbc * 32 + (tid & 31);
      const u32x4 v = *reinterpret_cast<const u32x4*>(lsrc + ((long)b * a.N + min(qrow, a.N - 1)) * lld + ih * D + lchunk * 8);
      const float sv = ssrc[((long)b * a.H + ih) * a.N + min(qr, a.N - 1)];
      const u32x4 z = {0u, 0u, 0u, 0u};
      rt = qrow < a.N ? v : z;
      rs = qr < a.N ? sv : sfill;
    } else {
      const bf16_t* tb = lsrc + ((long)b * a.N + pbc * 32) * lld + ih * D;           // scalar
      const float* sb = ssrc + ((long)b * a.H + ih) * a.N + pbc * 32;               // scalar
      rt = *reinterpret_cast<const u32x4*>(tb + loff);
      rs = sb[(unsigned)(tid & 31)];
    }
    if (++ih == a.H) { ih = 0; ++ipb; }
  };
  auto store_stage = [&](int slot) {
    char* base = smem + slot * kStage;
    *reinterpret_cast<u32x4*>(base + lds_dst) = rt;
    if constexpr (RAGGED) { if (tid < 64) reinterpret_cast<float*>(base + 8192)[tid] = rs; }
    else { if (tid < 128 && !(tid & 32)) reinterpret_cast<float*>(base + 8192)[(tid >> 6) * 32 + (tid & 31)] = rs; }
  };

  // dQ tile of this wave: queries qh*16 .. +15, head-dim columns dq4*16 .. +15 of the pair handled one iteration earlier
  const int qh = wave & 1, dq4 = wave >> 1;
  const int g4 = lane >> 4, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
  // The K-side operand of this wave's dQ tile (K[256 keys][16 d-columns], read column-wise) never changes during the sweep: its
  // eight k-step fragments are read from the K image once, after the prologue barrier, and stay in registers (32 VGPRs; re-reading
  // them every iteration was half of the dQ phase's LDS traffic, 8 KiB per wave and iteration).
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 kcol[8];
  // (whole-block form) element offset of this lane's first dQ value from the (query block, head) base; rows r = 1..3 follow H*64 apart
  const unsigned aoff = (unsigned)((qh * 16 + 4 * g4) * (a.H * D) + dq4 * 16 + ip);
  auto dq_tile = [&](const char* eb, int ph, int ppb) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two chains: a single one waits out each MFMA's latency 8 times
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      // k-step ks covers keys 32*ks .. +31: rows 32*ks + 8*g4 + {0..7} of the dS image, whose swizzle (row ^ row >> 3) & 7 changes with
      // ks (+32 rows flips bit 2 of row >> 3), so the offsets are formed per k-step
      const int r0 = 32 * ks + 8 * g4 + tq;
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(eb + ds_img_off(r0, qh * 4 + tp)));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(eb + ds_img_off(r0 + 4, qh * 4 + tp)));
      const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const s16x8 bv = kcol[ks];
      f32x4& c4 = (ks & 1) ? acc1 : acc;
      // atomics: dQ tile (row = query); slabs: dQ^T tile (row = d), so that a lane ends up with 4 consecutive d of one query row
      if constexpr (DQ_MODE == 0) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c4, 0, 0, 0);
      else c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv), __builtin_bit_cast(bf16x8, av), c4, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
    if constexpr (DQ_MODE != 0) {
      // accumulator: column n = lane & 15 -> query, row m = 4 * (lane >> 4) + r -> d.  Slab rows are padded to whole 32-query
      // blocks, so no store is masked (a branch around it would cost the counted waits, see the stage loader)
      const long npad = (long)nqb * 32;
      const long row = ((long)kb * a.B + b) * npad + ppb * 32 + qh * 16 + ip;
      const float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
      if constexpr (DQ_MODE == 1) store4(reinterpret_cast<bf16_t*>(dq32) + row * (a.H * D) + ph * D + dq4 * 16 + 4 * g4, v4);
      else store4(dq32 + row * (a.H * D) + ph * D + dq4 * 16 + 4 * g4, v4);
      return;
    }
    // accumulator: column n = lane & 15 -> d, row m = 4 * (lane >> 4) + r -> query
    if constexpr (RAGGED) {
      // (no branch around the atomics, see the stage loader: a padded query row adds 0.0 to the sample's last row instead)
      const int qrow0 = ppb * 32 + qh * 16 + 4 * g4;
      float* dst = dq32 + (long)b * a.N * (a.H * D) + ph * D + dq4 * 16 + ip;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = qrow0 + r < a.N;
        atomic_add_f32(dst + (long)min(qrow0 + r, a.N - 1) * (a.H * D), ok ? acc[r] : 0.f);
      }
    } else {
      float* sb = dq32 + ((long)b * a.N + ppb * 32) * (a.H * D) + ph * D;             // scalar
#pragma unroll
      for (int r = 0; r < 4; ++r) atomic_add_f32(sb + (aoff + (unsigned)(r * (a.H * D))), acc[r]);
    }
  };

  const LaneOffs lo(lane);
  for (int i = tid; i < 16384 / 16; i += NW * 64) reinterpret_cast<u32x4*>(eimg + 16384)[i] = u32x4{0u, 0u, 0u, 0u};
  load_stage(); store_stage(0);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {                                 // tile_off swizzles by (row >> 1) & 7: a +32-row step leaves it alone
    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(kimg + tile_off(8 * g4 + tq, (dq4 * 16 + 4 * tp) * 2) + ks * 4096));
    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(kimg + tile_off(8 * g4 + 4 + tq, (dq4 * 16 + 4 * tp) * 2) + ks * 4096));
    kcol[ks] = s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  }
  int ph = 0, ppb = qb_begin;                                      // (head, query block) of the previous iteration's pair
  int ch = 0, cpb = qb_begin;                                      // ... of the current one
  for (int it = 0; it < niter; ++it) {
    const char* qs = smem + (it & 1) * kStage;
    const char* dos = qs + 4096;
    const float* ls = reinterpret_cast<const float*>(qs + 8192);
    // global prefetch of the next pair FIRST, then the previous pair's dQ (its four atomics are younger than the loads)
    load_stage();
    dq_tile(eimg + ((it + 1) & 1) * 16384, ph, ppb);               // it = 0: the zeroed image -> adds 0.0
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(qs, lo, ks, 0), kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(dos, lo, ks, 0), vf[ks], dp, 0, 0, 0);
    }
    f32x16 ds;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(ls + 8 * g + 4 * lh);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * g + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float p = fast_exp2(fmaf(s[r], c, -l4[e]));
        s[r] = p;
        ds[r] = p * (dp[r] - d4[e]);                // * scale folded into the finishing passes
      }
    }
    // dS rows of this wave's keys -> the transposing image (queries 8g + 4lh .. +3 = one 8-byte chunk)
    {
      char* eb = eimg + (it & 1) * 16384;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w2;
        w2[0] = pack_bf16x2(ds[4 * g], ds[4 * g + 1]);
        w2[1] = pack_bf16x2(ds[4 * g + 2], ds[4 * g + 3]);
        *reinterpret_cast<u32x2*>(eb + ds_img_off(wave * 32 + lr, 2 * g + lh)) = w2;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 pf = acc_to_frag(s, s2);
      const bf16x8 df = acc_to_frag(ds, s2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(dos, lo, s2 * 16, dt), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(qs, lo, s2 * 16, dt), df, dk[dt], 0, 0, 0);
      }
    }
    store_stage((it + 1) & 1);                                     // (after the last pair: a stage nobody reads)
    __syncthreads();
    ph = ch; ppb = cpb;
    if (++ch == a.H) { ch = 0; ++cpb; }
  }
  dq_tile(eimg + ((niter - 1) & 1) * 16384, ph, ppb);
  if (kok && a.qsplit > 1) {
    const long prow = ((long)part * a.B + b) * a.N + key;
    store_grad_row(a.wsk + prow * D, dk, 1.f, nullptr, nullptr, lh);
    store_grad_row(a.wsv + prow * D, dv, 1.f, nullptr, nullptr, lh);
  } else if (kok) {
    store_grad(a.dk, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dk, a.kmul, a.rcos, a.rsin, key, lh);
    store_grad(a.dv, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dv, 1.f, nullptr, nullptr, key, lh);
  }
}

// (Tried and removed, round 2: the same sweep as 4 waves x 64 keys, one wave per SIMD with the whole 512-register file -- two 32-key
//  halves per wave so that S / dP of one half overlap the exp / dS arithmetic of the other, Q / dO fragments read once per
//  iteration for both halves.  Correct, but 9.95 ms per backward at B=32, N=4096 against 7.34 ms for the kernel above: past 256
//  registers hipcc parks accumulators in AGPRs, and the loop spent 200 of its ~700 instructions on v_accvgpr_read / _mov / _write
//  moving S, dP between the two halves of the file for the VALU.)
// ------------------------------------------------------------------------------------------------------
// Fused backward, 512 keys per workgroup (round 3): the same key-stationary sweep with HALF the dQ atomic bytes -- the float-atomic
// floor of the 256-key kernel above (B*H*N*64*4 bytes x N/256 at the chip's ~1.3 TB/s: 6.6 ms at B=32, N=4096) is what bounds it.
// Four waves, ONE per SIMD, each with the whole 512-register file: a wave owns 128 keys = four 32-key tiles, and its dK^T / dV^T
// accumulators (2 x [64 d][128 keys] fp32 = 256 registers) live in the AGPR half of the file for the whole sweep.  hipcc cannot be
// left to place them: as soon as a kernel may need AGPRs it selects the AGPR form for EVERY MFMA builtin and then shuttles S / dP
// through v_accvgpr_* for the VALU (round 2's 4-wave try: 200 of 700 loop instructions; reproduced with a 40-line probe this round).
// So every MFMA of this kernel is an inline-asm statement that names its register class: "+a" accumulators for dK^T / dV^T (they
// are touched by nothing else until the epilogue), "+v" for S, dP and the dQ tiles, which the VALU consumes.  What hipcc does not
// do for an asm MFMA is done by hand (cdna_hip_programming.md 5.7): `s_nop 1` in front of each one (an operand may have been
// written by the VALU instruction just before it), and a fence statement -- s_nop N with the accumulator as a "+v" operand -- before
// the first non-MFMA reader of a result (8-pass 32x32x16: 12 wait states; 4-pass 16x16x32: 8).
// Per (head, 32-query block) pair: S^T / dP^T of the wave's four key tiles (their accumulators start from -lse/c and -delta, staged
// with the Q / dO tile, so p = exp2(c S') and dS = p dP' need no subtraction), dS rows -> the shared [512 keys][32 queries] image,
// dV^T / dK^T MFMAs; one barrier; one pair later wave w takes the two 16x16 tiles (both query halves, head-dim columns 16w..16w+15)
// of dQ[32][64] = dS[32][512] K[512][64] and adds them with float atomics: 8 KiB per pair and workgroup, as above, but for twice the
// keys.  K fragments are read from the resident K image (S needs its rows, dQ its columns), V fragments stay in 64 VGPRs.
// Whole-block shapes only (N % 32 == 0: every UNet level); keys past N have zero K / V rows (their dS rows meet zero K rows).
// ------------------------------------------------------------------------------------------------------
// dS image of the 512-key sweep: [key][32 queries] bf16, 8-byte chunks xor-swizzled by the key's position INSIDE its 32-key tile only, so
// that a step of 32 keys is a plain +2048 bytes (an instruction offset): with the 256-key image's swizzle (key ^ key >> 3) every one of the
// 16 k-steps x 4 reads of the dQ phase needed its own per-lane offset register -- 64 of them, which hipcc spilled to scratch
__device__ __forceinline__ int ds_img_off512(int key, int qchunk) {
  return key * 64 + ((qchunk ^ ((key ^ ((key >> 3) & 3)) & 7)) << 3);
}
__device__ __forceinline__ void mfma32_agpr(f32x16& acc, const bf16x8 a, const bf16x8 b) {      // (P / dS fragments: converted >= 2 instructions earlier, see the slots)
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// (no leading s_nop on these two: their A / B operands come out of LDS reads or loop-invariant registers and their accumulators out of LDS
//  reads or the previous MFMA of the chain -- never out of a VALU instruction issued just before; tools/check_mfma_hazards.py checks the
//  emitted code for a VALU write of an MFMA source within the two preceding instructions)
__device__ __forceinline__ void mfma32_vgpr(f32x16& acc, const bf16x8 a, const bf16x8 b) {
  asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_vgpr(f32x4& acc, const bf16x8 a, const bf16x8 b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// an MFMA right behind the v_cvt_pk that produced its B operand (hipcc is free to sink the fillers written between them)
__device__ __forceinline__ void mfma32_agpr_nop(f32x16& acc, const bf16x8 a, const bf16x8 b) {
  asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// first MFMA of a dQ chain: its accumulator was zeroed by VALU moves that hipcc may place right in front of it
__device__ __forceinline__ void mfma16_vgpr_first(f32x4& acc, const bf16x8 a, const bf16x8 b) {
  asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// results of asm MFMAs become readable by the VALU / stores only after the matrix pipe has written them back
// (none of these statements is `volatile`: a volatile asm is a barrier for every memory operation in hipcc's scheduling graph, which pinned
//  each LDS fragment read between two MFMAs -- read, s_waitcnt lgkmcnt(0), MFMA, 32 times over -- and cost 2 ms of 8.8; as pure functions of
//  their operands they are ordered by data dependences alone: chain -> fence -> reader)
__device__ __forceinline__ void mfma32_fence(f32x16& acc) { asm("s_nop 11" : "+v"(acc)); }
__device__ __forceinline__ void mfma16_fence(f32x4& a0, f32x4& a1) { asm("s_nop 7" : "+v"(a0), "+v"(a1)); }
// the same ordering point where the program order already puts >= 2 MFMAs (64 cycles) between the chain's last MFMA and its reader
__device__ __forceinline__ void mfma32_fence_short(f32x16& acc) { asm("s_nop 3" : "+v"(acc)); }
__device__ __forceinline__ void mfma32_fence_agpr(f32x16& acc) { asm("s_nop 15\n\ts_nop 3" : "+a"(acc)); }

// ATOMICS = false: timing-only build of the sweep (dQ tiles are computed and dropped) -- prices the loop without its atomics
template <bool ATOMICS>
__global__ __launch_bounds__(256, 1) void mqa_bwd_fused512_kernel(AttnArgs a, float* dq32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2][Q 4K | dO 4K | -lse/c 128 | -delta 128] | K image 64K | [2] dS image 32K
  constexpr int kStage = 4096 + 4096 + 256;
  char* kimg = smem + 2 * kStage;
  char* eimg = kimg + 65536;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int nkb = (a.N + 511) / 512;
  const int per_split = (int)gridDim.x / a.qsplit;
  const int part = (int)blockIdx.x / per_split, bid = (int)blockIdx.x - part * per_split;
  const int xcd = bid & 7, qid = bid >> 3;
  const int b = (qid / nkb) * 8 + xcd;
  const int kb = qid % nkb;
  if (b >= a.B) return;
  const float c = a.cexp;
  const int nqb = a.N >> 5;
  const int qb_per = (nqb + a.qsplit - 1) / a.qsplit;
  const int qb_begin = part * qb_per, qb_end = min(nqb, qb_begin + qb_per);
  if (qb_begin >= qb_end) return;                                  // uniform per workgroup, before any barrier
  const int niter = (qb_end - qb_begin) * a.H;
  const int key0 = kb * 512 + wave * 128;                          // this wave's keys: key0 + 32 t + lr, t = 0..3

  bf16x8 vf[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int key = key0 + t * 32 + lr;
    const long m = (long)b * a.N + key;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 z = {0u, 0u, 0u, 0u}, z2 = {0u, 0u, 0u, 0u};
      if (key < a.N) { z = *reinterpret_cast<const u32x4*>(a.k + m * a.ldk + 16 * ks + 8 * lh); z2 = *reinterpret_cast<const u32x4*>(a.v + m * a.ldv + 16 * ks + 8 * lh); }
      vf[t][ks] = __builtin_bit_cast(bf16x8, z2);
      *reinterpret_cast<u32x4*>(kimg + tile_off(wave * 128 + t * 32 + lr, (2 * ks + lh) * 16)) = z;   // resident K image (zero rows beyond N)
    }
  }
  f32x16 dk[4][2], dv[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[t][i][r] = 0.f; dv[t][i][r] = 0.f; }

  // stage loader (256 threads): thread -> one 16-B chunk of the Q tile and the same chunk of the dO tile; threads 0..31 / 32..63 the
  // pair's lse / delta rows, stored as the INITIAL ACCUMULATORS -lse2 / c and -delta.  Branch-free, loads before the iteration's atomics
  // (see mqa_bwd_fused_kernel).
  const int lrow = tid >> 3, lchunk = tid & 7;
  const unsigned qoff = (unsigned)(lrow * (int)a.ldq + lchunk * 8), dooff = (unsigned)(lrow * (int)a.lddo + lchunk * 8);
  const int lds_dst = tile_off(lrow, lchunk * 16);
  const float smul = (tid & 32) ? -1.f : -1.f / c;
  const int last_pb = qb_end - 1;
  u32x4 rq, rd; float rs = 0.f;
  // (head, query block) of the next stage to load, as running wave-uniform pointers: +64 columns per head, and at the last head on to the
  // next block's rows (past the end: the last pair again -- loaded, never consumed)
  int ih = 0, ipb = qb_begin;
  const bf16_t* qp_ = a.q + ((long)b * a.N + qb_begin * 32) * a.ldq;
  const bf16_t* dp_ = a.dout + ((long)b * a.N + qb_begin * 32) * a.lddo;
  const float* lp_ = a.lse2 + (long)b * a.H * a.N + qb_begin * 32;
  const float* tp_ = a.delta + (long)b * a.H * a.N + qb_begin * 32;
  const long q_wrap = 32 * a.ldq - (long)(a.H - 1) * D, d_wrap = 32 * a.lddo - (long)(a.H - 1) * D, s_wrap = 32 - (long)(a.H - 1) * a.N;
  auto load_stage = [&]() {
    rq = *reinterpret_cast<const u32x4*>(qp_ + qoff);
    rd = *reinterpret_cast<const u32x4*>(dp_ + dooff);
    rs = ((tid & 32) ? tp_ : lp_)[(unsigned)(tid & 31)] * smul;
    const bool wrap = ih + 1 == a.H;
    const bool stay = wrap && ipb == last_pb;                      // scalar selects
    ih = wrap ? 0 : ih + 1;
    ipb += (wrap && !stay) ? 1 : 0;
    const long dq_ = stay ? -(long)(a.H - 1) * D : (wrap ? q_wrap : (long)D);
    const long dd_ = stay ? -(long)(a.H - 1) * D : (wrap ? d_wrap : (long)D);
    const long ds_ = stay ? -(long)(a.H - 1) * a.N : (wrap ? s_wrap : (long)a.N);
    qp_ += dq_; dp_ += dd_; lp_ += ds_; tp_ += ds_;
  };
  auto store_stage = [&](int slot) {
    char* base = smem + slot * kStage;
    *reinterpret_cast<u32x4*>(base + lds_dst) = rq;
    *reinterpret_cast<u32x4*>(base + 4096 + lds_dst) = rd;
    if (tid < 64) reinterpret_cast<float*>(base + 8192)[tid] = rs;
  };

  // dQ of the pair handled one iteration earlier: this wave's tiles = queries 16 qh .. +15 (qh = 0, 1) x head-dim columns 16 wave .. +15.
  // Its 16 k-steps (32 keys each: 6 transposed reads, 2 MFMAs) are spread over the four key tiles of the iteration, one k-step per
  // stage, operands read one stage ahead.
  const int g4 = lane >> 4, ip = lane & 15, tq = ip >> 2, tp = ip & 3;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int koff0 = tile_off(8 * g4 + tq, (wave * 16 + 4 * tp) * 2), koff1 = tile_off(8 * g4 + 4 + tq, (wave * 16 + 4 * tp) * 2);
  // float atomics of a dQ tile pair: accumulator column n = lane & 15 -> d, row m = 4 * (lane >> 4) + r -> query (of its half).  Addresses are
  // a wave-uniform 64-bit base + a 32-bit per-lane BYTE offset (the saddr form): eight zero-extended 64-bit lane offsets would not fit the file
  const unsigned aoffb = (unsigned)((4 * g4) * (a.H * D) + wave * 16 + ip) * 4u;
  const unsigned rowb = (unsigned)(a.H * D) * 4u;
  // wave-uniform base of the PREVIOUS pair's dQ rows, carried from iteration to iteration (re-deriving it from (b, block, head) for each of
  // the four atomic pairs was ~40 scalar instructions per iteration in a loop that is bound by its instruction count)
  char* dq_prev = reinterpret_cast<char*>(dq32 + ((long)b * a.N + qb_begin * 32) * (a.H * D));
  char* dq_cur = dq_prev;
  auto dq_add2 = [&](const f32x4& q0, const f32x4& q1, int r, int, int) {      // row r of both query halves
    if constexpr (ATOMICS) {
      char* sb = dq_prev;
      unsigned ob = aoffb;
      asm volatile("" : "+v"(ob));
      atomic_add_f32(reinterpret_cast<float*>(sb + (ob + (unsigned)r * rowb)), q0[r]);
      atomic_add_f32(reinterpret_cast<float*>(sb + (ob + (unsigned)(16 + r) * rowb)), q1[r]);
    } else {
      asm volatile("" :: "v"(q0[r]), "v"(q1[r]));
    }
  };
  auto dq_add = [&](const f32x4& q0, const f32x4& q1, int, int) {
    if constexpr (ATOMICS) {
      char* sb = dq_prev;
      unsigned ob = aoffb;
      asm volatile("" : "+v"(ob));                                  // opaque per call: keeps the eight offsets out of loop-invariant registers
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomic_add_f32(reinterpret_cast<float*>(sb + (ob + (unsigned)r * rowb)), q0[r]);
        atomic_add_f32(reinterpret_cast<float*>(sb + (ob + (unsigned)(16 + r) * rowb)), q1[r]);
      }
    } else {
      asm volatile("" :: "v"(q0), "v"(q1));
    }
  };
  const int eoff[2][2] = {{ds_img_off512(8 * g4 + tq, tp), ds_img_off512(8 * g4 + tq + 4, tp)},
                          {ds_img_off512(8 * g4 + tq, 4 + tp), ds_img_off512(8 * g4 + tq + 4, 4 + tp)}};
  struct DqOps { bf16x8 bv, av0, av1; };
  auto dq_read = [&](const char* ep, int ks) {
    DqOps o;
    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(kimg + koff0 + ks * 4096));
    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(kimg + koff1 + ks * 4096));
    o.bv = __builtin_bit_cast(bf16x8, (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(ep + eoff[0][0] + ks * 2048));
    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(ep + eoff[0][1] + ks * 2048));
    o.av0 = __builtin_bit_cast(bf16x8, (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
    const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(ep + eoff[1][0] + ks * 2048));
    const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(ep + eoff[1][1] + ks * 2048));
    o.av1 = __builtin_bit_cast(bf16x8, (s16x8){c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]});
    return o;
  };
#define OSUF_FENCE __builtin_amdgcn_sched_barrier(0)

  const LaneOffs lo(lane);
  for (int i = tid; i < 32768 / 16; i += 256) reinterpret_cast<u32x4*>(eimg + 32768)[i] = u32x4{0u, 0u, 0u, 0u};
  load_stage(); store_stage(0);
  __syncthreads();
  int ph = 0, ppb = qb_begin;                                      // (head, query block) of the previous iteration's pair
  int ch = 0, cpb = qb_begin;                                      // ... of the current one
  for (int it = 0; it < niter; ++it) {
    const char* qs = smem + (it & 1) * kStage;
    const char* dos = qs + 4096;
    const float* ls = reinterpret_cast<const float*>(qs + 8192);
    char* eb = eimg + (it & 1) * 32768;
    const char* ep = eimg + ((it + 1) & 1) * 32768;                // the previous pair's dS image (it = 0: zeros -> adds 0.0)
    load_stage();                                                  // global prefetch of the next pair FIRST (older than this iteration's atomics)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;                // dQ tiles of the two query halves (one chain each: a dependent
    auto dq_mma = [&](const DqOps& o) {                            // 16x16x32 issues at the rate of an independent one, and successive
      mfma16_vgpr(acc0, o.av0, o.bv);                              // k-steps sit a whole stage apart)
      mfma16_vgpr(acc1, o.av1, o.bv);
    };
    bf16x8 qa[4], da[4], kf[4];
    f32x16 s, dp;
    auto read_consts_s = [&]() {                                   // the initial accumulators of a key tile's S / dP chains: -lse/c, -delta
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(ls + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[4 * g + e] = l4[e];
      }
    };
    auto read_consts_dp = [&]() {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[4 * g + e] = d4[e];
      }
    };
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { qa[ks] = lds_row_frag(qs, lo, ks, 0); da[ks] = lds_row_frag(dos, lo, ks, 0); }
    read_consts_s(); read_consts_dp();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = lds_row_frag(kimg + wave * 4 * 4096, lo, ks, 0);
    DqOps o0 = dq_read(ep, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // One wave per SIMD: nothing overlaps unless it is interleaved in program order, so every group below pairs MFMAs with the
      // VALU / LDS work that is independent of them (the first all-MFMAs-then-all-VALU order ran 8.6 ms at N = 4096: the sum of the
      // matrix, vector and LDS times).  Within a tile: S chain | dP chain beside exp2(c S') | dV beside dS = p dP' | dK beside the next
      // tile's operand reads; the 8 small dQ MFMAs and their 24 transposed reads are spread over the four stages.
      bf16x8 trd[2][2], trq[2][2];                                 // transposed dO / Q fragments [s2][dt]
      bf16x8 pf[2], df[2];
      DqOps o1;
      typedef __attribute__((ext_vector_type(4))) float f4;
      auto exp4 = [&](int r0) {
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) s[r] = fast_exp2(s[r] * c);
      };
      auto ds4 = [&](int r0) {
#pragma unroll
        for (int r = r0; r < r0 + 4; ++r) dp[r] = s[r] * dp[r];    // (x scale folded into the finishing passes)
      };
      auto ds_write2 = [&](int g0) {                               // the dS fragment's words ARE the packed row pieces of the image
#pragma unroll
        for (int g = g0; g < g0 + 2; ++g) {
          u32x2 w2;
          w2[0] = __builtin_bit_cast(u32x4, df[g >> 1])[2 * (g & 1)];
          w2[1] = __builtin_bit_cast(u32x4, df[g >> 1])[2 * (g & 1) + 1];
          *reinterpret_cast<u32x2*>(eb + ds_img_off512(wave * 128 + t * 32 + lr, 2 * g + lh)) = w2;
        }
      };
      // slot = one matrix instruction (or a dQ pair), then the fillers that issue while it executes; OSUF_FENCE pins the order.
      // Fragment reads sit as late as their consumers allow (two slots ahead or more): the register file is full
      const char* krow = kimg + (wave * 4 + t) * 4096;
      // dQ k-steps: tiles 0-2 run theirs in slots 2, 9, 12, 16; the last tile runs them early (slots 2, 6, 9, 10) so that the eight float
      // atomics of the finished tiles can leave two per slot behind its dK MFMAs instead of as one burst before the barrier.  Every
      // operand read sits at least three slots ahead of its consumer.
      const bool last = t == 3;
      OSUF_FENCE;
      mfma32_vgpr(s, qa[0], kf[0]);                 OSUF_FENCE;  o1 = dq_read(ep, 4 * t + 1);                                   OSUF_FENCE;
      mfma32_vgpr(s, qa[1], kf[1]);                 OSUF_FENCE;
      if (t == 0) { mfma16_vgpr_first(acc0, o0.av0, o0.bv); mfma16_vgpr_first(acc1, o0.av1, o0.bv); } else dq_mma(o0);          OSUF_FENCE;   // k-step 4t
      mfma32_vgpr(s, qa[2], kf[2]);                 OSUF_FENCE;  trd[0][0] = lds_tr_frag(dos, lo, 0, 0); trd[0][1] = lds_tr_frag(dos, lo, 0, 1);   OSUF_FENCE;
      mfma32_vgpr(s, qa[3], kf[3]);                 OSUF_FENCE;  o0 = dq_read(ep, 4 * t + 2);                                   OSUF_FENCE;
      mfma32_vgpr(dp, da[0], vf[t][0]);             OSUF_FENCE;  trd[1][0] = lds_tr_frag(dos, lo, 16, 0); trd[1][1] = lds_tr_frag(dos, lo, 16, 1); OSUF_FENCE;
      mfma32_vgpr(dp, da[1], vf[t][1]);             OSUF_FENCE;  if (last) dq_mma(o1);                                          OSUF_FENCE;   // (last tile: k-step 13)
      mfma32_vgpr(dp, da[2], vf[t][2]);             OSUF_FENCE;  mfma32_fence_short(s); exp4(0); if (last) o1 = dq_read(ep, 15); OSUF_FENCE;
      mfma32_vgpr(dp, da[3], vf[t][3]);             OSUF_FENCE;  exp4(4);                                                       OSUF_FENCE;
      if (last) dq_mma(o0); else dq_mma(o1);        OSUF_FENCE;  pf[0] = acc_to_frag(s, 0); OSUF_FENCE; exp4(8);                OSUF_FENCE;   // k-step 4t+1 (last tile: 14)
      mfma32_agpr_nop(dv[t][0], trd[0][0], pf[0]);  OSUF_FENCE;  exp4(12); if (last) dq_mma(o1); else o1 = dq_read(ep, 4 * t + 3);   OSUF_FENCE;   // (last tile: k-step 15)
      mfma32_agpr(dv[t][1], trd[0][1], pf[0]);      OSUF_FENCE;  pf[1] = acc_to_frag(s, 1); OSUF_FENCE; mfma32_fence_short(dp); ds4(0);
                                                                 trq[0][0] = lds_tr_frag(qs, lo, 0, 0); trq[0][1] = lds_tr_frag(qs, lo, 0, 1);     OSUF_FENCE;
      mfma32_agpr(dv[t][0], trd[1][0], pf[1]);      OSUF_FENCE;  ds4(4); ds4(8); if (last) mfma16_fence(acc0, acc1); else dq_mma(o0);   OSUF_FENCE;   // k-step 4t+2
      mfma32_agpr(dv[t][1], trd[1][1], pf[1]);      OSUF_FENCE;  ds4(12); df[0] = acc_to_frag(dp, 0); OSUF_FENCE;
                                                                 trq[1][0] = lds_tr_frag(qs, lo, 16, 0); trq[1][1] = lds_tr_frag(qs, lo, 16, 1);
                                                                 if (!last) o0 = dq_read(ep, 4 * t + 4);                        OSUF_FENCE;
      mfma32_agpr(dk[t][0], trq[0][0], df[0]);      OSUF_FENCE;  df[1] = acc_to_frag(dp, 1); OSUF_FENCE; ds_write2(0);
                                                                 if (last) dq_add2(acc0, acc1, 0, ph, ppb); else read_consts_s();                   OSUF_FENCE;
      mfma32_agpr(dk[t][1], trq[0][1], df[0]);      OSUF_FENCE;  ds_write2(2);
                                                                 if (last) dq_add2(acc0, acc1, 1, ph, ppb); else read_consts_dp();                  OSUF_FENCE;
      mfma32_agpr(dk[t][0], trq[1][0], df[1]);      OSUF_FENCE;
      if (last) dq_add2(acc0, acc1, 2, ph, ppb);
      else {
        dq_mma(o1);                                                                                                                           // k-step 4t+3
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = lds_row_frag(krow + 4096, lo, ks, 0);
      }
      OSUF_FENCE;
      mfma32_agpr(dk[t][1], trq[1][1], df[1]);      OSUF_FENCE;
      if (last) dq_add2(acc0, acc1, 3, ph, ppb);
    }
    OSUF_FENCE;
    store_stage((it + 1) & 1);                                     // (after the last pair: a stage nobody reads)
    __syncthreads();
    ph = ch; ppb = cpb;
    dq_prev = dq_cur;
    if (++ch == a.H) { ch = 0; ++cpb; dq_cur += (32L * a.H - (a.H - 1)) * (D * 4); }
    else dq_cur += D * 4;
  }
#undef OSUF_FENCE
  {                                                                // dQ of the last pair
    const char* ep = eimg + ((niter - 1) & 1) * 32768;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const DqOps o = dq_read(ep, ks);
      if (ks == 0) { mfma16_vgpr_first(acc0, o.av0, o.bv); mfma16_vgpr_first(acc1, o.av1, o.bv); }
      else { mfma16_vgpr(acc0, o.av0, o.bv); mfma16_vgpr(acc1, o.av1, o.bv); }
    }
    mfma16_fence(acc0, acc1);
    dq_add(acc0, acc1, ph, ppb);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int key = key0 + t * 32 + lr;
#pragma unroll
    for (int i = 0; i < 2; ++i) { mfma32_fence_agpr(dk[t][i]); mfma32_fence_agpr(dv[t][i]); }     // written back before the accumulators are read
    if (key < a.N && a.qsplit > 1) {
      const long prow = ((long)part * a.B + b) * a.N + key;
      store_grad_row(a.wsk + prow * D, dk[t], 1.f, nullptr, nullptr, lh);
      store_grad_row(a.wsv + prow * D, dv[t], 1.f, nullptr, nullptr, lh);
    } else if (key < a.N) {
      store_grad(a.dk, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dk[t], a.kmul, a.rcos, a.rsin, key, lh);
      store_grad(a.dv, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dv[t], 1.f, nullptr, nullptr, key, lh);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Fused backward, 512 keys per workgroup, HAND-PLACED LOOP (round 4): the algorithm, LDS images, fragment maps and accumulation orders of
// mqa_bwd_fused512_kernel above (dK / dV are bit-identical to it), with the loop between the prologue and the epilogue emitted by
// tools/gen_attn_bwd512.py as one asm statement on fixed physical registers (attn_bwd512_asm.inc; the register map is at the top of the
// generator): 538 instead of 764 non-MFMA instructions per (head, 32-query block) pair, LDS-DMA stages, tile-pipelined matrix order.
// Here: everything that is per-lane arithmetic once per workgroup (the loop's lane offsets, the resident K image, the V fragments, stage 0).
// Whole 512-key blocks only (N % 512 == 0), hence niter = (N / 32 / qsplit) * H is even: the loop is unrolled by two.
// ------------------------------------------------------------------------------------------------------
#include "attn_bwd512_asm.inc"
#include "attn_bwd512qs_asm.inc"      // the same loop for queries pre-scaled by c (tools/gen_attn_bwd512.py --qs): no v_mul of the scores
typedef __attribute__((ext_vector_type(32))) uint32_t u32x32;
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;
typedef __attribute__((ext_vector_type(32))) float f32x32;
template <bool QS>
__global__ __launch_bounds__(256, 1) void mqa_bwd_fused512a_kernel(AttnArgs a, float* dq32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2][Q 4K | dO 4K | -lse/c 128 | -delta 128] | K image 64K | [2] dS image 32K
  constexpr int kStage = 4096 + 4096 + 256;
  char* kimg = smem + 2 * kStage;
  char* eimg = kimg + 65536;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int nkb = a.N / 512;
  const int per_split = (int)gridDim.x / a.qsplit;
  const int part = (int)blockIdx.x / per_split, bid = (int)blockIdx.x - part * per_split;
  const int xcd = bid & 7, qid = bid >> 3;
  const int b = (qid / nkb) * 8 + xcd;
  const int kb = qid % nkb;
  if (b >= a.B) return;
  float c = a.cexp;
  // c is pinned into a VECTOR register, and the v_readfirstlane statements below carry their own s_nop pads: a vector write of the source
  // register DIRECTLY in front of an inline-asm v_readfirstlane is a hazard hipcc pads for its own instructions but not around an asm statement
  // (cdna_hip_programming.md 5.7).  With the kernel argument left in its SGPR hipcc emitted `v_mov_b32 v, s` right before the statement and
  // the non-QS loop ran with a wrong c (every gradient ~0.6 N times too large); bisected on the GPU: unpinned fails, unpinned + s_nop pads
  // passes, pinning only the two values handed to the statements passes (round 5; tests/test_asm_hazards.py refuses the failing shape).
  asm volatile("" : "+v"(c));
  const int nqb = a.N >> 5;
  const int qb_per = nqb / a.qsplit;
  const int qb_begin = part * qb_per;
  const int niter = qb_per * a.H;                                  // even (host-checked)
  const int key0 = kb * 512 + wave * 128;

  // V fragments -> registers (v0..v63 of the loop), K rows -> the resident image
  u32x32 vf0, vf1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const long m = (long)b * a.N + key0 + t * 32 + lr;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 z = *reinterpret_cast<const u32x4*>(a.k + m * a.ldk + 16 * ks + 8 * lh);
      const u32x4 z2 = *reinterpret_cast<const u32x4*>(a.v + m * a.ldv + 16 * ks + 8 * lh);
#pragma unroll
      for (int j = 0; j < 4; ++j) { if (t < 2) vf0[16 * t + 4 * ks + j] = z2[j]; else vf1[16 * (t - 2) + 4 * ks + j] = z2[j]; }
      *reinterpret_cast<u32x4*>(kimg + tile_off(wave * 128 + t * 32 + lr, (2 * ks + lh) * 16)) = z;
    }
  }
  // stage 0 = the first pair, as mqa_bwd_fused512_kernel stages it; the previous pair's dS image (slot 1) = zeros: the first iteration adds 0.0
  {
    const int lrow = tid >> 3, lchunk = tid & 7;
    const long row0 = (long)b * a.N + qb_begin * 32;
    *reinterpret_cast<u32x4*>(smem + tile_off(lrow, lchunk * 16)) = *reinterpret_cast<const u32x4*>(a.q + (row0 + lrow) * a.ldq + lchunk * 8);
    *reinterpret_cast<u32x4*>(smem + 4096 + tile_off(lrow, lchunk * 16)) = *reinterpret_cast<const u32x4*>(a.dout + (row0 + lrow) * a.lddo + lchunk * 8);
    if (tid < 64) {
      const float* src = (tid & 32) ? a.delta : a.lse2;
      reinterpret_cast<float*>(smem + 8192)[tid] = src[(long)b * a.H * a.N + qb_begin * 32 + (tid & 31)] * ((tid & 32) ? -1.f : -1.f / c);
    }
    for (int i = tid; i < 32768 / 16; i += 256) reinterpret_cast<u32x4*>(eimg + 32768)[i] = u32x4{0u, 0u, 0u, 0u};
  }
  // the loop's lane offsets (LDS byte addresses; immediates carry slot, tile and k-step): v224..v255 of the loop, see tools/gen_attn_bwd512.py
  const uint32_t sb = (uint32_t)(uintptr_t)(LDS_PTR(char))smem;
  const int g4 = lane >> 4, ip = lane & 15, tq = ip >> 2, tp = ip & 3, cb = ((lane >> 4) & 1) * 16;
  u32x32 adr;
#pragma unroll
  for (int i = 0; i < 32; ++i) adr[i] = 0u;
  {
    const uint32_t rowb_ = (uint32_t)(a.H * D) * 4u;
    const uint32_t aoffb = (uint32_t)((4 * g4) * (a.H * D) + wave * 16 + ip) * 4u;
    adr[4] = aoffb;                                                                         // AO: byte offset of this lane's dQ element, query half 0 / 1 (rows: scalar bases)
    adr[5] = aoffb + 16u * rowb_;
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) adr[8 + ks] = sb + 2 * kStage + wave * 16384 + tile_off(lr, (2 * ks + lh) * 16);   // RK: row fragments of this wave's K rows (stage tiles: less SKOF)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int var = 0; var < 2; ++var) adr[12 + 2 * dt + var] = sb + tile_off(8 * var + 4 * lh + tq, (dt * 32 + cb + 4 * tp) * 2) - 8 * var * 128;   // T (LaneOffs::tr)
  adr[16] = sb + 2 * kStage + tile_off(8 * g4 + tq, (wave * 16 + 4 * tp) * 2);            // KO: K columns of the dQ tiles
  adr[17] = sb + 2 * kStage + tile_off(8 * g4 + 4 + tq, (wave * 16 + 4 * tp) * 2);
#pragma unroll
  for (int qh = 0; qh < 2; ++qh)
#pragma unroll
    for (int j = 0; j < 2; ++j) adr[18 + 2 * qh + j] = sb + 2 * kStage + 65536 + ds_img_off512(8 * g4 + tq + 4 * j, 4 * qh + tp);                    // EO: dS image, transposed
#pragma unroll
  for (int g = 0; g < 4; ++g) adr[22 + g] = sb + 2 * kStage + 65536 + ds_img_off512(wave * 128 + lr, 2 * g + lh);                                    // EW: dS rows of this lane's key
  adr[26] = sb + 8192 + 16 * lh;                                                            // CR: row constants (b128 reads)
  adr[27] = sb + 8192 + lane * 4;                                                           // CW: ... written (every wave the same 256 bytes)
  {
    const int row = 8 * wave + (lane >> 3), x = (row >> 1) & 7, f = ((x & 1) << 2) | (x >> 1), chunk = (lane & 7) ^ f;   // source-side swizzle of tile_off
    adr[28] = (uint32_t)(row * (int)a.ldq + chunk * 8) * 2u;                                // QOFF / DOOFF: byte offsets of this lane's 16 B of the DMA piece
    adr[29] = (uint32_t)(row * (int)a.lddo + chunk * 8) * 2u;
  }
  adr[30] = (uint32_t)(lane & 31) * 4u;                                                     // COFF
  // loop state: bases = the part's FIRST pair (the lowest address of each tensor's part: the running 32-bit offsets stay non-negative); the
  // request offsets start at the second pair (the first is staged above), the dQ offsets at the first
  const long hn = (long)b * a.H * a.N;
  auto uni = [](uint64_t x) {                                                               // block-uniform by construction; tell hipcc (physical SGPR operands)
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) |      // (the builtin returns int: widen it unsigned)
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
  };
  const uint64_t pq = uni((uint64_t)(a.q + ((long)b * a.N + qb_begin * 32) * a.ldq));
  const uint64_t pdo = uni((uint64_t)(a.dout + ((long)b * a.N + qb_begin * 32) * a.lddo));
  const uint64_t pls = uni((uint64_t)(a.lse2 + hn + qb_begin * 32));
  const uint64_t pdl = uni((uint64_t)(a.delta + hn + qb_begin * 32));
  const uint64_t pdq = uni((uint64_t)(dq32 + ((long)b * a.N + qb_begin * 32) * (a.H * D)));
  const uint64_t wq = uni((uint64_t)((32 * a.ldq - (long)(a.H - 1) * D) * 2)), wdo = uni((uint64_t)((32 * a.lddo - (long)(a.H - 1) * D) * 2));
  const uint64_t wls = uni((uint64_t)((32 - (long)(a.H - 1) * a.N) * 4)), wdq = uni((uint64_t)((32L * a.H * D - (long)(a.H - 1) * D) * 4));
  adr[28] += (uint32_t)(a.H > 1 ? (uint64_t)(D * 2) : wq);                                 // second pair: the next head, or (one head) the next block
  adr[29] += (uint32_t)(a.H > 1 ? (uint64_t)(D * 2) : wdo);
  adr[30] += (uint32_t)(a.H > 1 ? (uint64_t)a.N * 4u : wls);
  uint32_t cnt = __builtin_amdgcn_readfirstlane((uint32_t)(niter / 2));
  uint32_t srem = __builtin_amdgcn_readfirstlane((uint32_t)(qb_per - 1 - (a.H > 1 ? 0 : 1)));    // block steps left for the request offsets (they sit at the second pair)
  const uint32_t hh = __builtin_amdgcn_readfirstlane((uint32_t)a.H), n4 = __builtin_amdgcn_readfirstlane((uint32_t)a.N * 4u);
  const uint32_t rowb = __builtin_amdgcn_readfirstlane((uint32_t)(a.H * D) * 4u);
  const uint32_t skof = __builtin_amdgcn_readfirstlane((uint32_t)(2 * kStage + wave * 16384));   // RK - (row-fragment offsets of a stage tile)
  // (floats come out of VALU instructions, and hipcc folds __builtin_amdgcn_readfirstlane of a value it knows to be uniform: an opaque one)
  auto rfl = [](float x) { uint32_t r; asm volatile("s_nop 4\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(r) : "v"(x)); return r; };
  const uint32_t cbits = rfl(c), nrc = rfl(-1.f / c);
  const uint32_t m0base = __builtin_amdgcn_readfirstlane(sb + (uint32_t)wave * 1024u);
  __syncthreads();
  f32x32 acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7;                                    // dK^T tiles 0..3, dV^T tiles 0..3 (the loop's a[0:255], copied to v[0:255] at its end)
  if constexpr (QS) {
  asm volatile(OSUF_BWD512AQS_ASM
                 : "={v[0:31]}"(acc0), "={v[32:63]}"(acc1), "={v[64:95]}"(acc2), "={v[96:127]}"(acc3), "={v[128:159]}"(acc4), "={v[160:191]}"(acc5),
                   "={v[192:223]}"(acc6), "={v[224:255]}"(acc7), "+{s60}"(cnt), "+{s63}"(srem)
                 : "{v[0:31]}"(vf0), "{v[32:63]}"(vf1), "{v[224:255]}"(adr), "{s[48:49]}"(pq), "{s[50:51]}"(pdo), "{s[52:53]}"(pls), "{s[54:55]}"(pdl),
                   "{s[56:57]}"(pdq), "{s61}"(hh), "{s65}"(n4), "{s66}"((uint32_t)wq), "{s68}"((uint32_t)wdo), "{s70}"((uint32_t)wls), "{s72}"((uint32_t)wdq),
                   "{s74}"(cbits), "{s75}"(nrc), "{s76}"(m0base), "{s77}"(skof), "{s80}"(rowb)
                 : OSUF_BWD512AQS_CLOBBERS);
  } else {
  asm volatile(OSUF_BWD512A_ASM
                 : "={v[0:31]}"(acc0), "={v[32:63]}"(acc1), "={v[64:95]}"(acc2), "={v[96:127]}"(acc3), "={v[128:159]}"(acc4), "={v[160:191]}"(acc5),
                   "={v[192:223]}"(acc6), "={v[224:255]}"(acc7), "+{s60}"(cnt), "+{s63}"(srem)
                 : "{v[0:31]}"(vf0), "{v[32:63]}"(vf1), "{v[224:255]}"(adr), "{s[48:49]}"(pq), "{s[50:51]}"(pdo), "{s[52:53]}"(pls), "{s[54:55]}"(pdl),
                   "{s[56:57]}"(pdq), "{s61}"(hh), "{s65}"(n4), "{s66}"((uint32_t)wq), "{s68}"((uint32_t)wdo), "{s70}"((uint32_t)wls), "{s72}"((uint32_t)wdq),
                   "{s74}"(cbits), "{s75}"(nrc), "{s76}"(m0base), "{s77}"(skof), "{s80}"(rowb)
                 : OSUF_BWD512A_CLOBBERS);
  }
  const f32x32 dkv[8] = {acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int key = key0 + t * 32 + lr;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = dkv[t][r]; dk[1][r] = dkv[t][16 + r]; dv[0][r] = dkv[4 + t][r]; dv[1][r] = dkv[4 + t][16 + r]; }
    if (a.qsplit > 1) {
      const long prow = ((long)part * a.B + b) * a.N + key;
      store_grad_row(a.wsk + prow * D, dk, 1.f, nullptr, nullptr, lh);
      store_grad_row(a.wsv + prow * D, dv, 1.f, nullptr, nullptr, lh);
    } else {
      store_grad(a.dk, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dk, a.kmul, a.rcos, a.rsin, key, lh);
      store_grad(a.dv, a.lddk, (long)b * a.N + key, 0, a.g_bf16, dv, 1.f, nullptr, nullptr, key, lh);
    }
  }
}

// finishing pass of the fused backward's dQ: fp32 sums [M][H*64] -> scale, RoPE transpose (as store_grad_row), cast, into dq [M][lddq]
template <typename TO>
__global__ __launch_bounds__(256) void dq_finish_kernel(const float* __restrict__ dq32, TO* dq, long lddq, long M, int N, int H, float scale,
                                                        const float* __restrict__ rcos, const float* __restrict__ rsin) {
  // one thread: 4 columns d0..d0+3 (d0 < 32) of one head and their partners d0+32..; 8 threads per (row, head)
  const long total = M * H * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int sub = (int)(idx & 7);
    const long rh = idx >> 3;
    const long m = rh / H;
    const int h = (int)(rh - m * H), n = (int)(m % N), d0 = sub * 4;
    float y1[4], y2[4];
    load4(dq32 + m * ((long)H * D) + h * D + d0, y1);
    load4(dq32 + m * ((long)H * D) + h * D + 32 + d0, y2);
#pragma unroll
    for (int e = 0; e < 4; ++e) { y1[e] *= scale; y2[e] *= scale; }
    if (rcos) {
      float cs[4], sn[4];
      load4(rcos + (long)n * 32 + d0, cs);
      load4(rsin + (long)n * 32 + d0, sn);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = y1[e] * cs[e] + y2[e] * sn[e];
        const float a2 = y2[e] * cs[e] - y1[e] * sn[e];
        y1[e] = a1; y2[e] = a2;
      }
    }
    store4(dq + m * lddq + h * D + d0, y1);
    store4(dq + m * lddq + h * D + 32 + d0, y2);
  }
}

// dQ of the fused backward from the key-block slabs: sum the nkb slabs in key-block order (fixed: bit-reproducible), scale, RoPE
// transpose (as store_grad_row), cast.  ws: [nkb][B][npad][H*64] TP; dq: [B*N][lddq] TO.
template <typename TP, typename TO>
__global__ __launch_bounds__(256) void dq_reduce_kernel(const TP* __restrict__ ws, int nkb, TO* dq, long lddq, int B, int N, int npad, int H,
                                                        float scale, const float* __restrict__ rcos, const float* __restrict__ rsin) {
  const long total = (long)B * N * H * 8;                          // one thread: d0..d0+3 (d0 < 32) of one head and the partners d0+32..
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int sub = (int)(idx & 7);
    const long rh = idx >> 3;
    const long m = rh / H;
    const int h = (int)(rh - m * H), d0 = sub * 4;
    const int bb = (int)(m / N), n = (int)(m - (long)bb * N);
    const long slab = (long)B * npad * (H * D);
    const TP* src = ws + ((long)bb * npad + n) * (H * D) + h * D + d0;
    float y1[4] = {0.f, 0.f, 0.f, 0.f}, y2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < nkb; ++p) {
      float u1[4], u2[4];
      load4(src + p * slab, u1);
      load4(src + p * slab + 32, u2);
#pragma unroll
      for (int e = 0; e < 4; ++e) { y1[e] += u1[e]; y2[e] += u2[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { y1[e] *= scale; y2[e] *= scale; }
    if (rcos) {
      float cs[4], sn[4];
      load4(rcos + (long)n * 32 + d0, cs);
      load4(rsin + (long)n * 32 + d0, sn);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a1 = y1[e] * cs[e] + y2[e] * sn[e];
        const float a2 = y2[e] * cs[e] - y1[e] * sn[e];
        y1[e] = a1; y2[e] = a2;
      }
    }
    store4(dq + m * lddq + h * D + d0, y1);
    store4(dq + m * lddq + h * D + 32 + d0, y2);
  }
}

// second half of the query-split dK/dV path: sum the parts (fixed order), scale, un-rotate (RoPE backward, as store_grad_row), cast
template <typename TO>
__global__ __launch_bounds__(256) void dkv_finish_kernel(const float* __restrict__ wsk, const float* __restrict__ wsv, int parts, TO* dk, TO* dv,
                                                         long ld, long M, int N, float scale, const float* __restrict__ rcos,
                                                         const float* __restrict__ rsin) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;           // one thread per (row, d < 32)
  if (idx >= M * 32) return;
  const long m = idx >> 5;
  const int d = (int)(idx & 31), n = (int)(m % N);
  float k1 = 0.f, k2 = 0.f, v1 = 0.f, v2 = 0.f;
  for (int p = 0; p < parts; ++p) {
    const long o = ((long)p * M + m) * D + d;
    k1 += wsk[o]; k2 += wsk[o + 32]; v1 += wsv[o]; v2 += wsv[o + 32];
  }
  k1 *= scale; k2 *= scale;
  if (rcos) {
    const float cs = rcos[(long)n * 32 + d], sn = rsin[(long)n * 32 + d];
    const float a1 = k1 * cs + k2 * sn, a2 = k2 * cs - k1 * sn;
    k1 = a1; k2 = a2;
  }
  ElemTraits<TO>::store(dk + m * ld + d, k1);
  ElemTraits<TO>::store(dk + m * ld + 32 + d, k2);
  ElemTraits<TO>::store(dv + m * ld + d, v1);
  ElemTraits<TO>::store(dv + m * ld + 32 + d, v2);
}

// delta[b][h][n] = sum_d dO[b,n,h,d] * O[b,n,h,d]   (8 lanes per (row, head), 8 elements each)
template <typename TO>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* dout, long lddo, const TO* o, long ldo, float* delta,
                                                         int B, int H, int N) {
  const long total = (long)B * N * H * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total + 7; idx += (long)gridDim.x * blockDim.x) {
    const bool ok = idx < total;
    const long rh = ok ? idx >> 3 : 0;
    const int sub = (int)(idx & 7);
    const long m = rh / H;
    const int h = (int)(rh - m * H);
    float s = 0.f;
    if (ok) {
      float x[8], y[8];
      load8(dout + m * lddo + h * D + sub * 8, x);
      load8(o + m * ldo + h * D + sub * 8, y);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[e] * y[e];
    }
    s = group_sum<8>(s);
    if (ok && sub == 0) {
      const long bb = m / N, n = m - bb * N;
      delta[(bb * H + h) * N + n] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// RoPE (attention.py:33-58, utils.py:25-32; half-split layout) fused with the bf16 cast of q, k, v
//   in : [M][ld_in]  T (q heads | kv heads k | kv heads v), tab: [N][32] cos, [N][32] sin (fp32)
//   out: [M][ld_out] bf16, same column layout
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_cast_kernel(const T* in, long ld_in, bf16_t* out, long ld_out, const float* cosb,
                                                        const float* sinb, int M, int N, int n_rot_heads, int n_heads_total,
                                                        float qmul, int n_q_heads) {
  // one thread: 8 columns d0..d0+7 (d0 < 32) of one head and their partners d0+32..; 4 threads per (row, head)
  const long total = (long)M * n_heads_total * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int sub = (int)(idx & 3);
    const long rh = idx >> 2;
    const long m = rh / n_heads_total;
    const int hd = (int)(rh - m * n_heads_total);
    const int n = (int)(m % N);
    float x1[8], x2[8];
    load8(in + m * ld_in + hd * D + sub * 8, x1);
    load8(in + m * ld_in + hd * D + 32 + sub * 8, x2);
    if (hd < n_rot_heads) {
      float cs[8], sn[8];
      load8(cosb + (long)n * 32 + sub * 8, cs);
      load8(sinb + (long)n * 32 + sub * 8, sn);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a1 = x1[e] * cs[e] - x2[e] * sn[e];
        const float a2 = x2[e] * cs[e] + x1[e] * sn[e];
        x1[e] = a1; x2[e] = a2;
      }
    }
    if (hd < n_q_heads) {                                       // osuf_rope_cast_qs: the softmax scale rides the queries' one bf16 rounding
#pragma unroll
      for (int e = 0; e < 8; ++e) { x1[e] *= qmul; x2[e] *= qmul; }
    }
    store8(out + m * ld_out + hd * D + sub * 8, x1);
    store8(out + m * ld_out + hd * D + 32 + sub * 8, x2);
  }
}

// inverse rotation of fp32 grads -> T:  dx1 = dy1*cos + dy2*sin ; dx2 = dy2*cos - dy1*sin
template <typename T>
__global__ __launch_bounds__(256) void rope_bwd_kernel(const float* in, long ld_in, T* out, long ld_out, const float* cosb,
                                                       const float* sinb, int M, int N, int n_rot_heads, int n_heads_total) {
  const long total = (long)M * n_heads_total * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int sub = (int)(idx & 3);
    const long rh = idx >> 2;
    const long m = rh / n_heads_total;
    const int hd = (int)(rh - m * n_heads_total);
    const int n = (int)(m % N);
    float y1[8], y2[8];
    load8(in + m * ld_in + hd * D + sub * 8, y1);
    load8(in + m * ld_in + hd * D + 32 + sub * 8, y2);
    if (hd < n_rot_heads) {
      float cs[8], sn[8];
      load8(cosb + (long)n * 32 + sub * 8, cs);
      load8(sinb + (long)n * 32 + sub * 8, sn);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a1 = y1[e] * cs[e] + y2[e] * sn[e];
        const float a2 = y2[e] * cs[e] - y1[e] * sn[e];
        y1[e] = a1; y2[e] = a2;
      }
    }
    store8(out + m * ld_out + hd * D + sub * 8, y1);
    store8(out + m * ld_out + hd * D + 32 + sub * 8, y2);
  }
}

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
#include "attn_generic.hpp"

static inline int ew_grid(long total_threads) {
  long blocks = (total_threads + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// head dims served by the generic kernels of attn_generic.hpp (64 has the tuned kernels of this file): padded tile width, 0 = unsupported
static int gen_dp(int head_dim) { return head_dim == 16 || head_dim == 32 ? 32 : head_dim == 128 ? 128 : 0; }
struct FwdRope { const float* cos; const float* sin; float qmul; void* qout; long ldqo; };   // osuf_mqa_fwd_rope's extra operands
static int mqa_fwd_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                        float* lse2, int B, int H, int N, int head_dim, float scale, bool qs, hipStream_t stream, float* zdq, const FwdRope* rope = nullptr);
extern "C" int osuf_mqa_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                            float* lse2, int B, int H, int N, int head_dim, float scale, hipStream_t stream) {
  return mqa_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, o_dtype, lse2, B, H, N, head_dim, scale, false, stream, nullptr);
}
// q holds the rotated queries ALREADY multiplied by scale * log2 e (osuf_rope_cast_qs): the scores leave the MFMA chain in the log2 domain
extern "C" int osuf_mqa_fwd_qs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                               float* lse2, int B, int H, int N, int head_dim, float scale, hipStream_t stream) {
  if (head_dim != D) return OSUF_EUNSUPPORTED;
  return mqa_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, o_dtype, lse2, B, H, N, head_dim, scale, true, stream, nullptr);
}
// osuf_mqa_fwd (qs = 0) / osuf_mqa_fwd_qs (qs = 1) that ALSO zero-fills zero_dq[B*N][H*64] fp32 -- the first osuf_mqa_bwd_fused_workspace_bytes'
// dQ accumulator of the layer's backward, which is then called with dq_mode | OSUF_DQ_PREZEROED and skips its memset.  head_dim 64 only.
extern "C" int osuf_mqa_fwd_zdq(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                                float* lse2, int B, int H, int N, int head_dim, float scale, int qs, float* zero_dq, hipStream_t stream) {
  if (head_dim != D || !zero_dq) return OSUF_EUNSUPPORTED;
  return mqa_fwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, o_dtype, lse2, B, H, N, head_dim, scale, qs != 0, stream, zero_dq);
}
// The forward on UN-rotated queries: q_raw is the q block of the q|kv projection as the GEMM left it (bf16); the kernel rotates each wave's 32 x 64 query
// tile (rope_cos / rope_sin, [N][32] fp32), multiplies it by q_mul = scale * log2 e and rounds it to bf16 once -- exactly osuf_rope_cast_qs' arithmetic
// -- before the tile loop, and stores it to q_out (may be NULL: inference) for the backward.  k / v: rotated / cast by osuf_rope_cast on their two
// head blocks alone.  zero_dq as osuf_mqa_fwd_zdq (may be NULL).  head_dim 64.
extern "C" int osuf_mqa_fwd_rope(const void* q_raw, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                                 float* lse2, int B, int H, int N, int head_dim, float scale, const float* rope_cos, const float* rope_sin,
                                 float q_mul, void* q_out, long ldqo, float* zero_dq, hipStream_t stream) {
  if (head_dim != D) return OSUF_EUNSUPPORTED;
  const FwdRope r = {rope_cos, rope_sin, q_mul, q_out, ldqo};
  return mqa_fwd_impl(q_raw, ldq, k, ldk, v, ldv, o, ldo, o_dtype, lse2, B, H, N, head_dim, scale, true, stream, zero_dq, &r);
}
static int mqa_fwd_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                        float* lse2, int B, int H, int N, int head_dim, float scale, bool qs, hipStream_t stream, float* zdq, const FwdRope* rope) {
  if (head_dim != D && !gen_dp(head_dim)) return OSUF_EUNSUPPORTED;
  if (zdq && (head_dim != D || !al16(zdq))) return OSUF_EINVAL;
  if (rope && (head_dim != D || !qs || !rope->cos || !rope->sin || !(rope->qmul > 0.f) || (rope->qout && (rope->ldqo % 8 || !al16(rope->qout))))) return OSUF_EINVAL;
  if (B <= 0 || H <= 0 || N <= 0 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || !al16(q) || !al16(k) || !al16(v) || !al16(o)) return OSUF_EINVAL;
  AttnArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.o = o; a.ldo = ldo; a.o_is_f32 = o_dtype == OSUF_DT_F32; a.lse2 = lse2; a.B = B; a.H = H; a.N = N; a.scale = scale; a.cexp = scale * kLog2e; a.kmul = scale;
  if (qs) a.cexp = 1.f;
  a.zdq = zdq;
  if (rope) { a.rcos = rope->cos; a.rsin = rope->sin; a.qmul = rope->qmul; a.qout = (bf16_t*)rope->qout; a.ldqo = rope->ldqo; }
  const int nvb = ((N + 31) / 32) * H;
  if (head_dim != D) {
    const dim3 grid((nvb + 3) / 4, B);
    if (gen_dp(head_dim) == 32) hipLaunchKernelGGL(mqa_gen_fwd_kernel<32>, grid, dim3(256), 2 * 64 * 64, stream, a, head_dim);
    else hipLaunchKernelGGL(mqa_gen_fwd_kernel<128>, grid, dim3(256), 2 * 64 * 256, stream, a, head_dim);
    return osuf_launch_status();
  }
  const bool whole = (N & 63) == 0 && getenv("OSUF_ATTN_FWD_NOWHOLE") == nullptr;
  const dim3 grid((nvb + 7) / 8, B);
  const int lds = 32768 + 8 * 4096;
  if (rope) {
    if (whole) hipLaunchKernelGGL((mqa_fwd_kernel<8, true, true, true>), grid, dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((mqa_fwd_kernel<8, true, false, true>), grid, dim3(512), lds, stream, a);
  } else if (qs && getenv("OSUF_ATTN_FWD_NOQSK") == nullptr) {
    if (whole) hipLaunchKernelGGL((mqa_fwd_kernel<8, true, true>), grid, dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((mqa_fwd_kernel<8, true>), grid, dim3(512), lds, stream, a);
  } else if (whole) hipLaunchKernelGGL((mqa_fwd_kernel<8, false, true>), grid, dim3(512), lds, stream, a);
  else hipLaunchKernelGGL(mqa_fwd_kernel<8>, grid, dim3(512), lds, stream, a);
  return osuf_launch_status();
}

// Attend(q, k, v, attn_mask) (attention.py:77-99): the forward with an additive bf16 score bias, element strides over (b, h, q, key) with 0
// for broadcast dimensions.  Inference path of the stand-alone Attend module; every head dim (64 included) runs the generic kernel.
extern "C" int osuf_mqa_fwd_masked(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                                   float* lse2, const void* mask, long mask_b, long mask_h, long mask_q, long mask_k, int B, int H, int N,
                                   int head_dim, float scale, hipStream_t stream) {
  if (head_dim != D && !gen_dp(head_dim)) return OSUF_EUNSUPPORTED;
  if (B <= 0 || H <= 0 || N <= 0 || ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4 || !al16(q) || !al16(k) || !al16(v) || !al16(o) || !mask || scale == 0.f)
    return OSUF_EINVAL;
  AttnArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.o = o; a.ldo = ldo; a.o_is_f32 = o_dtype == OSUF_DT_F32; a.lse2 = lse2; a.B = B; a.H = H; a.N = N; a.scale = scale; a.cexp = scale * kLog2e; a.kmul = scale;
  a.mask = (const bf16_t*)mask; a.mask_b = mask_b; a.mask_h = mask_h; a.mask_q = mask_q; a.mask_k = mask_k;
  const int nvb = ((N + 31) / 32) * H;
  const dim3 grid((nvb + 3) / 4, B);
  const int dp = head_dim == D ? 64 : gen_dp(head_dim);
  if (dp == 32) hipLaunchKernelGGL((mqa_gen_fwd_kernel<32, true>), grid, dim3(256), 2 * 64 * 64, stream, a, head_dim);
  else if (dp == 64) hipLaunchKernelGGL((mqa_gen_fwd_kernel<64, true>), grid, dim3(256), 2 * 64 * 128, stream, a, head_dim);
  else hipLaunchKernelGGL((mqa_gen_fwd_kernel<128, true>), grid, dim3(256), 2 * 64 * 256, stream, a, head_dim);
  return osuf_launch_status();
}

static int fill_bwd_args(AttnArgs& a, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                         const float* lse2, const float* delta, int B, int H, int N, int head_dim, float scale) {
  if (head_dim != D && !gen_dp(head_dim)) return OSUF_EUNSUPPORTED;
  if (B <= 0 || H <= 0 || N <= 0 || ldq % 8 || ldk % 8 || ldv % 8 || lddo % 8) return OSUF_EINVAL;
  if (!al16(q) || !al16(k) || !al16(v) || !al16(dout)) return OSUF_EINVAL;
  a = AttnArgs{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.lse2 = const_cast<float*>(lse2); a.dout = (const bf16_t*)dout; a.lddo = lddo; a.delta = delta;
  a.B = B; a.H = H; a.N = N; a.scale = scale; a.cexp = scale * kLog2e; a.kmul = scale; a.qsplit = 1;
  return OSUF_OK;
}

// delta[b][h][n] = sum_d dO * O   (o: bf16 or f32 storage of the forward output)
extern "C" int osuf_attn_delta(const void* dout, long lddo, const void* o, long ldo, int o_dtype, float* delta, int B, int H, int N,
                               int head_dim, hipStream_t stream) {
  if (head_dim != D && !gen_dp(head_dim)) return OSUF_EUNSUPPORTED;
  if (B <= 0 || H <= 0 || N <= 0 || ldo % 8 || lddo % 8 || !al16(o) || !al16(dout)) return OSUF_EINVAL;
  if (head_dim != D) {
    const int gb = ew_grid((long)B * N * H);
    if (o_dtype == OSUF_DT_F32) hipLaunchKernelGGL(attn_delta_gen_kernel<float>, dim3(gb), dim3(256), 0, stream, (const bf16_t*)dout, lddo, (const float*)o, ldo, delta, B, H, N, head_dim);
    else hipLaunchKernelGGL(attn_delta_gen_kernel<bf16_t>, dim3(gb), dim3(256), 0, stream, (const bf16_t*)dout, lddo, (const bf16_t*)o, ldo, delta, B, H, N, head_dim);
    return osuf_launch_status();
  }
  const long tot = (long)B * N * H * 8;
  if (o_dtype == OSUF_DT_F32) {
    hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(ew_grid(tot)), dim3(256), 0, stream, (const bf16_t*)dout, lddo, (const float*)o, ldo, delta, B, H, N);
  } else {
    hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3(ew_grid(tot)), dim3(256), 0, stream, (const bf16_t*)dout, lddo, (const bf16_t*)o, ldo, delta, B, H, N);
  }
  return osuf_launch_status();
}

// dq: [B*N][lddq] in out_dtype, head h at columns h*64; rope_cos / rope_sin ([N][32], or both NULL): store the gradient of the un-rotated q
// variant: OSUF_ATTN_AUTO picks by shape; _PLAIN / _PIPE force one kernel (parity tests force each on every shape)
extern "C" int osuf_mqa_bwd_dq(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                               const float* lse2, const float* delta, void* dq, long lddq, int B, int H, int N, int head_dim, float scale,
                               int out_dtype, const float* rope_cos, const float* rope_sin, int variant, hipStream_t stream) {
  AttnArgs a;
  int rc = fill_bwd_args(a, q, ldq, k, ldk, v, ldv, dout, lddo, lse2, delta, B, H, N, head_dim, scale);
  if (rc) return rc;
  if (lddq % 8 || !al16(dq) || (out_dtype != OSUF_DT_F32 && out_dtype != OSUF_DT_BF16) || ((rope_cos == nullptr) != (rope_sin == nullptr)) ||
      variant < OSUF_ATTN_AUTO || variant > OSUF_ATTN_PIPE)
    return OSUF_EINVAL;
  a.dq = dq; a.lddq = lddq; a.g_bf16 = out_dtype == OSUF_DT_BF16; a.rcos = rope_cos; a.rsin = rope_sin;
  const int nvb = ((N + 31) / 32) * H;
  if (head_dim != D) {                                            // generic head dims: gradient of the rotated q; the caller un-rotates (osuf_rope_bwd)
    if (rope_cos) return OSUF_EUNSUPPORTED;
    const dim3 grid((nvb + 3) / 4, B);
    if (gen_dp(head_dim) == 32) hipLaunchKernelGGL(mqa_gen_bwd_dq_kernel<32>, grid, dim3(256), 2 * 64 * 64, stream, a, head_dim);
    else hipLaunchKernelGGL(mqa_gen_bwd_dq_kernel<128>, grid, dim3(256), 2 * 64 * 256, stream, a, head_dim);
    return osuf_launch_status();
  }
  // the pipelined kernel (2 waves/SIMD, 256 VGPRs) wins once the key loop is long: +4.5 % at N=4096, +3 % at 2048, -2 % at <= 1024
  if (variant == OSUF_ATTN_AUTO) variant = N >= 2048 ? OSUF_ATTN_PIPE : OSUF_ATTN_PLAIN;
  if (variant == OSUF_ATTN_PLAIN) hipLaunchKernelGGL(mqa_bwd_dq_kernel<8>, dim3((nvb + 7) / 8, B), dim3(512), 32768, stream, a);
  else {
    static bool once = ((void)hipFuncSetAttribute((const void*)mqa_bwd_dq_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), true);
    (void)once;
    hipLaunchKernelGGL(mqa_bwd_dq_pipe_kernel, dim3((nvb + 7) / 8, B), dim3(512), 65536, stream, a);
  }
  return osuf_launch_status();
}

// dk, dv: [B*N][lddk] in out_dtype; rope tables as for dq (applied to dk only)
// query parts of the dK/dV kernel for this shape: short sequences give few 256-key workgroups (B=32, N=512: 64 on 256 CUs)
static int dkv_qsplit(int B, int N, int forced) {
  if (forced > 0) return forced;
  const int blocks = ((N + 255) / 256) * ((B + 7) / 8 * 8);
  int sp = 1;
  while (sp < 4 && blocks * sp < 256 && ((N + 31) / 32) / (sp * 2) >= 4) sp *= 2;
  return sp;
}

// fp32 workspace that lets osuf_mqa_bwd_dkv split the query range for this shape (0: no split; the call works without it, unsplit)
// qsplit: 0 = the split this shape gets by default, > 0 = force that many parts (tests)
extern "C" long osuf_mqa_bwd_dkv_workspace_bytes(int B, int N, int qsplit) {
  if (B <= 0 || N <= 0 || qsplit < 0 || qsplit > 16) return 0;
  const int sp = dkv_qsplit(B, N, qsplit);
  return sp > 1 ? 2L * sp * B * N * D * (long)sizeof(float) : 0;
}

extern "C" int osuf_mqa_bwd_dkv(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                                const float* lse2, const float* delta, void* dk, void* dv, long lddk, int B, int H, int N, int head_dim,
                                float scale, int out_dtype, const float* rope_cos, const float* rope_sin, float* workspace,
                                long workspace_bytes, int qsplit, int variant, hipStream_t stream) {
  AttnArgs a;
  int rc = fill_bwd_args(a, q, ldq, k, ldk, v, ldv, dout, lddo, lse2, delta, B, H, N, head_dim, scale);
  if (rc) return rc;
  if (lddk % 8 || !al16(dk) || !al16(dv) || (out_dtype != OSUF_DT_F32 && out_dtype != OSUF_DT_BF16) ||
      ((rope_cos == nullptr) != (rope_sin == nullptr)) || variant < OSUF_ATTN_AUTO || variant > OSUF_ATTN_PIPE || qsplit < 0 || qsplit > 16)
    return OSUF_EINVAL;
  a.dk = dk; a.dv = dv; a.lddk = lddk; a.g_bf16 = out_dtype == OSUF_DT_BF16; a.rcos = rope_cos; a.rsin = rope_sin;
  if (head_dim != D) {                                            // generic head dims: gradients of the rotated k and of v, unsplit
    if (rope_cos) return OSUF_EUNSUPPORTED;
    const dim3 ggrid(((N + 127) / 128) * B);
    if (gen_dp(head_dim) == 32) hipLaunchKernelGGL(mqa_gen_bwd_dkv_kernel<32>, ggrid, dim3(256), 2 * 32 * 64 + 256, stream, a, head_dim);
    else hipLaunchKernelGGL(mqa_gen_bwd_dkv_kernel<128>, ggrid, dim3(256), 2 * 32 * 256 + 256, stream, a, head_dim);
    return osuf_launch_status();
  }
  const int b8 = (B + 7) / 8 * 8;
  const dim3 grid(((N + 255) / 256) * b8);
  if (variant == OSUF_ATTN_PLAIN) {
    hipLaunchKernelGGL(mqa_bwd_dkv_kernel<8>, grid, dim3(512), 2 * (4096 + 4096 + 256), stream, a);
    return osuf_launch_status();
  }
  const long need = osuf_mqa_bwd_dkv_workspace_bytes(B, N, qsplit);
  if (need > 0 && workspace && workspace_bytes >= need && al16(workspace)) {
    a.qsplit = dkv_qsplit(B, N, qsplit);
    a.wsk = workspace; a.wsv = workspace + (long)a.qsplit * B * N * D;
    hipLaunchKernelGGL(mqa_bwd_dkv_pipe_kernel, dim3(grid.x * a.qsplit), dim3(512), 3 * (4096 + 4096 + 256), stream, a);
    const long M = (long)B * N;
    const unsigned fb = (unsigned)((M * 32 + 255) / 256);
    if (a.g_bf16) hipLaunchKernelGGL(dkv_finish_kernel<bf16_t>, dim3(fb), dim3(256), 0, stream, a.wsk, a.wsv, a.qsplit, (bf16_t*)dk, (bf16_t*)dv, lddk, M, N, a.kmul, rope_cos, rope_sin);
    else hipLaunchKernelGGL(dkv_finish_kernel<float>, dim3(fb), dim3(256), 0, stream, a.wsk, a.wsv, a.qsplit, (float*)dk, (float*)dv, lddk, M, N, a.kmul, rope_cos, rope_sin);
  } else {
    hipLaunchKernelGGL(mqa_bwd_dkv_pipe_kernel, grid, dim3(512), 3 * (4096 + 4096 + 256), stream, a);
  }
  return osuf_launch_status();
}

// ---- fused backward (one sweep; dQ through key-block slabs or fp32 atomics, dK / dV in registers) ------------------------
// dq_mode: OSUF_DQ_ATOMIC (default: fp32 atomics -- the faster one as measured, 7.8 vs 10.1 ms per backward at B=32, N=4096: the slab
// stores are 32-byte pieces of 128-byte lines) or OSUF_DQ_SLABS (slabs in the output's element type, summed in a fixed order)
// workspace = the dQ slabs / sums, followed, for query-split shapes, by the dK / dV partial sums of osuf_mqa_bwd_dkv
static long fused_dq_bytes(int B, int H, int N, int out_dtype, int dq_mode) {
  if (dq_mode != OSUF_DQ_SLABS) return (long)B * N * H * D * (long)sizeof(float);
  const long npad = ((long)N + 31) / 32 * 32, nkb = ((long)N + 255) / 256;
  return ((nkb * B * npad * H * D * (out_dtype == OSUF_DT_BF16 ? 2 : 4)) + 15) / 16 * 16;
}
// Which sweep an atomic-dQ call runs, and in how many query parts.  512 keys per workgroup (mqa_bwd_fused512_kernel) halves the dQ
// atomic bytes but gives half as many workgroups: it is taken where whole 32-query blocks and enough work per part remain.
static bool fused_use512(int B, int N, int dq_mode) {
  if (dq_mode == OSUF_DQ_ATOMIC_512 || dq_mode == OSUF_DQ_TIMING_512 || dq_mode == OSUF_DQ_ATOMIC_512A) return true;
  if (dq_mode != OSUF_DQ_ATOMIC) return false;
  // whole 512-key blocks only: a padded key's K / V rows are zero, but its S accumulator starts at -lse2 / c, so p = exp2(-lse2) and
  // dS = p * (-delta) are not -- for a row with lse2 < -128 p overflows and inf x 0 lands in dQ as NaN
  return (N % 512) == 0 && N >= 1024;
}
// the timing-only build returns OSUF_OK with dq all zeros: refused unless the caller's environment says it is pricing a loop
static bool timing_builds_allowed() { const char* e = getenv("OSUF_ALLOW_TIMING_BUILDS"); return e && e[0] == '1'; }
static int fused512_qsplit(int B, int N, int forced) {
  if (forced > 0) return forced;
  const int blocks = ((N + 511) / 512) * ((B + 7) / 8 * 8);
  int sp = 1;
  while (sp < 8 && blocks * sp < 256 && (N / 32) / (sp * 2) >= 8) sp *= 2;
  return sp;
}
static bool fused_mode_ok(int dq_mode) { return dq_mode >= OSUF_DQ_ATOMIC && dq_mode <= OSUF_DQ_ATOMIC_512A; }
// dq_mode | OSUF_DQ_PREZEROED: the dQ accumulator at the head of the workspace was zero-filled by osuf_mqa_fwd_zdq (no memset here)
static int strip_prezeroed(int dq_mode, bool* prezeroed) {
  if (prezeroed) *prezeroed = (dq_mode & OSUF_DQ_PREZEROED) != 0;
  return dq_mode & ~OSUF_DQ_PREZEROED;
}
// the hand-placed loop (mqa_bwd_fused512a_kernel) walks the pairs two at a time and carries 32-bit byte offsets from the first row of a
// sample's query part: whole query parts of an even number of pairs, and every running offset -- Q / dO requests (up to N rows of ldq / lddo
// elements), dQ atomics (up to N + 32 rows of H * 64 floats: row bases + 16 rows of the second query half + the lane's 4 g4 rows), row
// constants (H * N floats) -- below 2^31 (sign-safe).  tools/check_bwd512a_addresses.py replays the emitted text against exactly this guard.
static bool fused512a_ok(const AttnArgs& a, int N) {
  const int nqb = N / 32;
  return (N % 512) == 0 && a.qsplit >= 1 && nqb % a.qsplit == 0 && (((nqb / a.qsplit) * a.H) % 2) == 0 &&
         (long)N * a.ldq * 2 < (1L << 31) && (long)N * a.lddo * 2 < (1L << 31) && ((long)N + 32) * a.H * D * 4 < (1L << 31) &&
         (long)a.H * N * 4 < (1L << 31);
}
static long fused_dkv_ws_bytes(int B, int N, int qsplit, int dq_mode) {
  if (!fused_use512(B, N, dq_mode)) return osuf_mqa_bwd_dkv_workspace_bytes(B, N, qsplit);
  const int sp = fused512_qsplit(B, N, qsplit);
  return sp > 1 ? 2L * sp * B * N * D * (long)sizeof(float) : 0;
}
extern "C" long osuf_mqa_bwd_fused_workspace_bytes(int B, int H, int N, int out_dtype, int qsplit, int dq_mode) {
  dq_mode = strip_prezeroed(dq_mode, nullptr);
  if (B <= 0 || H <= 0 || N <= 0 || qsplit < 0 || qsplit > 16 || !fused_mode_ok(dq_mode)) return 0;
  return fused_dq_bytes(B, H, N, out_dtype, dq_mode) + fused_dkv_ws_bytes(B, N, qsplit, dq_mode);
}

static int mqa_bwd_fused_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                              const float* lse2, const float* delta, void* dq, long lddq, void* dk, void* dv, long lddk, int B, int H,
                              int N, int head_dim, float scale, int out_dtype, const float* rope_cos, const float* rope_sin,
                              float* workspace, long workspace_bytes, int qsplit, int dq_mode, bool qs, hipStream_t stream);
extern "C" int osuf_mqa_bwd_fused(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                                  const float* lse2, const float* delta, void* dq, long lddq, void* dk, void* dv, long lddk, int B, int H,
                                  int N, int head_dim, float scale, int out_dtype, const float* rope_cos, const float* rope_sin,
                                  float* workspace, long workspace_bytes, int qsplit, int dq_mode, hipStream_t stream) {
  return mqa_bwd_fused_impl(q, ldq, k, ldk, v, ldv, dout, lddo, lse2, delta, dq, lddq, dk, dv, lddk, B, H, N, head_dim, scale, out_dtype, rope_cos,
                            rope_sin, workspace, workspace_bytes, qsplit, dq_mode, false, stream);
}
// q pre-scaled by c = scale * log2 e (osuf_rope_cast_qs; lse2 from osuf_mqa_fwd_qs): p = exp2(Qs K^T - lse2) with no multiply; dq = scale dS K
// as before (the gradient of the UN-scaled rotated q), dk = (scale / c) dS^T Qs.  The generated 512-key loop drops its 64 v_mul per pair.
extern "C" int osuf_mqa_bwd_fused_qs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                                     const float* lse2, const float* delta, void* dq, long lddq, void* dk, void* dv, long lddk, int B, int H,
                                     int N, int head_dim, float scale, int out_dtype, const float* rope_cos, const float* rope_sin,
                                     float* workspace, long workspace_bytes, int qsplit, int dq_mode, hipStream_t stream) {
  return mqa_bwd_fused_impl(q, ldq, k, ldk, v, ldv, dout, lddo, lse2, delta, dq, lddq, dk, dv, lddk, B, H, N, head_dim, scale, out_dtype, rope_cos,
                            rope_sin, workspace, workspace_bytes, qsplit, dq_mode, true, stream);
}
static int mqa_bwd_fused_impl(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                              const float* lse2, const float* delta, void* dq, long lddq, void* dk, void* dv, long lddk, int B, int H,
                              int N, int head_dim, float scale, int out_dtype, const float* rope_cos, const float* rope_sin,
                              float* workspace, long workspace_bytes, int qsplit, int dq_mode, bool qs, hipStream_t stream) {
  bool prezeroed = false;
  dq_mode = strip_prezeroed(dq_mode, &prezeroed);
  if (prezeroed && dq_mode == OSUF_DQ_SLABS) return OSUF_EINVAL;     // the slabs are written whole: nothing to pre-zero
  AttnArgs a;
  int rc = fill_bwd_args(a, q, ldq, k, ldk, v, ldv, dout, lddo, lse2, delta, B, H, N, head_dim, scale);
  if (rc) return rc;
  if (qs) { a.cexp = 1.f; a.kmul = 1.f / kLog2e; }
  if (lddq % 8 || lddk % 8 || !al16(dq) || !al16(dk) || !al16(dv) || (out_dtype != OSUF_DT_F32 && out_dtype != OSUF_DT_BF16) ||
      ((rope_cos == nullptr) != (rope_sin == nullptr)) || qsplit < 0 || qsplit > 16 || !workspace || !al16(workspace) ||
      !fused_mode_ok(dq_mode) || workspace_bytes < osuf_mqa_bwd_fused_workspace_bytes(B, H, N, out_dtype, qsplit, dq_mode))
    return OSUF_EINVAL;
  if (head_dim != D) return OSUF_EUNSUPPORTED;                     // the fused sweeps are written for 64-wide heads (others: dq + dkv kernels)
  const bool use512 = fused_use512(B, N, dq_mode);
  if (use512 && (N % 512) != 0) return OSUF_EUNSUPPORTED;          // the 512-key sweep is written for whole 512-key blocks (see fused_use512)
  if (dq_mode == OSUF_DQ_TIMING_512 && !timing_builds_allowed()) return OSUF_EUNSUPPORTED;
  a.dk = dk; a.dv = dv; a.lddk = lddk; a.g_bf16 = out_dtype == OSUF_DT_BF16; a.rcos = rope_cos; a.rsin = rope_sin;
  const long M = (long)B * N;
  const long dq_bytes = fused_dq_bytes(B, H, N, out_dtype, dq_mode);
  float* dq32 = workspace;
  float* wsp = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + dq_bytes);
  a.qsplit = use512 ? fused512_qsplit(B, N, qsplit) : dkv_qsplit(B, N, qsplit);
  if (a.qsplit > 1) { a.wsk = wsp; a.wsv = wsp + (long)a.qsplit * B * N * D; }
  const int b8 = (B + 7) / 8 * 8;
  if (dq_mode != OSUF_DQ_SLABS && !prezeroed) {
    hipError_t e = hipMemsetAsync(dq32, 0, (size_t)dq_bytes, stream);
    if (e != hipSuccess) return (int)e;
  }
  // the 512-key sweep runs its hand-placed loop wherever that loop's shape rules hold (every UNet level does), the compiled loop elsewhere
  const bool use512a = use512 && (dq_mode == OSUF_DQ_ATOMIC_512A || (dq_mode == OSUF_DQ_ATOMIC && fused512a_ok(a, N)));
  if (use512a) {
    if (!fused512a_ok(a, N)) return OSUF_EUNSUPPORTED;
    const int lds = 2 * (4096 + 4096 + 256) + 65536 + 2 * 32768;
    static bool attr512a = false;
    if (!attr512a) {
      (void)hipFuncSetAttribute((const void*)mqa_bwd_fused512a_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute((const void*)mqa_bwd_fused512a_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr512a = true;
    }
    if (qs) hipLaunchKernelGGL(mqa_bwd_fused512a_kernel<true>, dim3((N / 512) * b8 * a.qsplit), dim3(256), lds, stream, a, dq32);
    else hipLaunchKernelGGL(mqa_bwd_fused512a_kernel<false>, dim3((N / 512) * b8 * a.qsplit), dim3(256), lds, stream, a, dq32);
  } else if (use512) {
    const int lds = 2 * (4096 + 4096 + 256) + 65536 + 2 * 32768;
    void (*kern)(AttnArgs, float*) = dq_mode == OSUF_DQ_TIMING_512 ? mqa_bwd_fused512_kernel<false> : mqa_bwd_fused512_kernel<true>;
    static bool attr512[2] = {false, false};
    if (!attr512[dq_mode == OSUF_DQ_TIMING_512]) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr512[dq_mode == OSUF_DQ_TIMING_512] = true;
    }
    hipLaunchKernelGGL(kern, dim3(((N + 511) / 512) * b8 * a.qsplit), dim3(256), lds, stream, a, dq32);
  } else {
    const int lds = 2 * (4096 + 4096 + 256) + 32768 + 2 * 16384;
    const bool ragged = (N % 32) != 0;
    const int mode = dq_mode != OSUF_DQ_SLABS ? 0 : (a.g_bf16 ? 1 : 2);
    void (*kern)(AttnArgs, float*) =
        mode == 0 ? (ragged ? mqa_bwd_fused_kernel<0, true> : mqa_bwd_fused_kernel<0, false>)
      : mode == 1 ? (ragged ? mqa_bwd_fused_kernel<1, true> : mqa_bwd_fused_kernel<1, false>)
                  : (ragged ? mqa_bwd_fused_kernel<2, true> : mqa_bwd_fused_kernel<2, false>);
    static bool attr_set[6] = {false, false, false, false, false, false};
    if (!attr_set[2 * mode + ragged]) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set[2 * mode + ragged] = true;
    }
    hipLaunchKernelGGL(kern, dim3(((N + 255) / 256) * b8 * a.qsplit), dim3(512), lds, stream, a, dq32);
  }
  if (a.qsplit > 1) {
    const unsigned fb = (unsigned)((M * 32 + 255) / 256);
    if (a.g_bf16) hipLaunchKernelGGL(dkv_finish_kernel<bf16_t>, dim3(fb), dim3(256), 0, stream, a.wsk, a.wsv, a.qsplit, (bf16_t*)dk, (bf16_t*)dv, lddk, M, N, a.kmul, rope_cos, rope_sin);
    else hipLaunchKernelGGL(dkv_finish_kernel<float>, dim3(fb), dim3(256), 0, stream, a.wsk, a.wsv, a.qsplit, (float*)dk, (float*)dv, lddk, M, N, a.kmul, rope_cos, rope_sin);
  }
  const int qb = ew_grid(M * H * 8);
  const int nkb = (N + 255) / 256, npad = (N + 31) / 32 * 32;
  if (dq_mode != OSUF_DQ_SLABS) {
    if (a.g_bf16) hipLaunchKernelGGL(dq_finish_kernel<bf16_t>, dim3(qb), dim3(256), 0, stream, dq32, (bf16_t*)dq, lddq, M, N, H, scale, rope_cos, rope_sin);
    else hipLaunchKernelGGL(dq_finish_kernel<float>, dim3(qb), dim3(256), 0, stream, dq32, (float*)dq, lddq, M, N, H, scale, rope_cos, rope_sin);
  } else if (a.g_bf16) {
    hipLaunchKernelGGL((dq_reduce_kernel<bf16_t, bf16_t>), dim3(qb), dim3(256), 0, stream, (const bf16_t*)dq32, nkb, (bf16_t*)dq, lddq, B, N, npad, H, scale, rope_cos, rope_sin);
  } else {
    hipLaunchKernelGGL((dq_reduce_kernel<float, float>), dim3(qb), dim3(256), 0, stream, (const float*)dq32, nkb, (float*)dq, lddq, B, N, npad, H, scale, rope_cos, rope_sin);
  }
  return osuf_launch_status();
}

static int rope_cast_impl(int dtype, const void* in, long ld_in, void* out, long ld_out, const float* cosb, const float* sinb,
                          int M, int N, int n_rot_heads, int n_heads_total, int head_dim, float qmul, int n_q_heads, hipStream_t stream);
extern "C" int osuf_rope_cast(int dtype, const void* in, long ld_in, void* out, long ld_out, const float* cosb, const float* sinb,
                              int M, int N, int n_rot_heads, int n_heads_total, int head_dim, hipStream_t stream) {
  return rope_cast_impl(dtype, in, ld_in, out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, head_dim, 1.f, 0, stream);
}
// as osuf_rope_cast, and the first n_q_heads heads (the queries) are multiplied by q_mul before their bf16 rounding (head_dim 64 only)
extern "C" int osuf_rope_cast_qs(int dtype, const void* in, long ld_in, void* out, long ld_out, const float* cosb, const float* sinb,
                                 int M, int N, int n_rot_heads, int n_heads_total, int head_dim, float q_mul, int n_q_heads, hipStream_t stream) {
  if (head_dim != D || n_q_heads < 0 || n_q_heads > n_heads_total) return OSUF_EUNSUPPORTED;
  return rope_cast_impl(dtype, in, ld_in, out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, head_dim, q_mul, n_q_heads, stream);
}
static int rope_cast_impl(int dtype, const void* in, long ld_in, void* out, long ld_out, const float* cosb, const float* sinb,
                          int M, int N, int n_rot_heads, int n_heads_total, int head_dim, float qmul, int n_q_heads, hipStream_t stream) {
  if (head_dim != D && (head_dim <= 0 || head_dim % 16)) return OSUF_EUNSUPPORTED;
  if (M <= 0 || N <= 0 || M % N || ld_in % 8 || ld_out % 8 || !al16(in) || !al16(out)) return OSUF_EINVAL;
  if (head_dim != D) {                                            // any head dim that is a multiple of 16: tables [N][head_dim / 2]
    const int gb = ew_grid((long)M * n_heads_total * (head_dim / 16));
    if (dtype == OSUF_DT_BF16) hipLaunchKernelGGL((rope_gen_kernel<bf16_t, bf16_t, 1>), dim3(gb), dim3(256), 0, stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, head_dim);
    else if (dtype == OSUF_DT_F32) hipLaunchKernelGGL((rope_gen_kernel<float, bf16_t, 1>), dim3(gb), dim3(256), 0, stream, (const float*)in, ld_in, (bf16_t*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, head_dim);
    else return OSUF_EUNSUPPORTED;
    return osuf_launch_status();
  }
  const long tot = (long)M * n_heads_total * 4;
  if (dtype == OSUF_DT_BF16) {
    hipLaunchKernelGGL(rope_cast_kernel<bf16_t>, dim3(ew_grid(tot)), dim3(256), 0, stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, qmul, n_q_heads);
  } else if (dtype == OSUF_DT_F32) {
    hipLaunchKernelGGL(rope_cast_kernel<float>, dim3(ew_grid(tot)), dim3(256), 0, stream, (const float*)in, ld_in, (bf16_t*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, qmul, n_q_heads);
  } else return OSUF_EUNSUPPORTED;
  return osuf_launch_status();
}

extern "C" int osuf_rope_bwd(int dtype, const float* in, long ld_in, void* out, long ld_out, const float* cosb, const float* sinb,
                             int M, int N, int n_rot_heads, int n_heads_total, int head_dim, hipStream_t stream) {
  if (head_dim != D && (head_dim <= 0 || head_dim % 16)) return OSUF_EUNSUPPORTED;
  if (M <= 0 || N <= 0 || M % N || ld_in % 8 || ld_out % 8 || !al16(in) || !al16(out)) return OSUF_EINVAL;
  if (head_dim != D) {
    const int gb = ew_grid((long)M * n_heads_total * (head_dim / 16));
    if (dtype == OSUF_DT_BF16) hipLaunchKernelGGL((rope_gen_kernel<float, bf16_t, -1>), dim3(gb), dim3(256), 0, stream, in, ld_in, (bf16_t*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, head_dim);
    else if (dtype == OSUF_DT_F32) hipLaunchKernelGGL((rope_gen_kernel<float, float, -1>), dim3(gb), dim3(256), 0, stream, in, ld_in, (float*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total, head_dim);
    else return OSUF_EUNSUPPORTED;
    return osuf_launch_status();
  }
  const long tot = (long)M * n_heads_total * 4;
  if (dtype == OSUF_DT_BF16) {
    hipLaunchKernelGGL(rope_bwd_kernel<bf16_t>, dim3(ew_grid(tot)), dim3(256), 0, stream, in, ld_in, (bf16_t*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total);
  } else if (dtype == OSUF_DT_F32) {
    hipLaunchKernelGGL(rope_bwd_kernel<float>, dim3(ew_grid(tot)), dim3(256), 0, stream, in, ld_in, (float*)out, ld_out, cosb, sinb, M, N, n_rot_heads, n_heads_total);
  } else return OSUF_EUNSUPPORTED;
  return osuf_launch_status();
}
