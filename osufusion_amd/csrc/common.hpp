// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// descriptor of osuf_pack_weight_group (include/osufusion_hip.h holds the same definition for callers)
#ifndef OSUF_PACK_DESC_DEFINED
#define OSUF_PACK_DESC_DEFINED
typedef struct osuf_pack_desc {
  const float* w;
  void* F;
  void* D;
  long f_ld, f_tapstride, d_ld, d_tapstride;
  int O, I, k, dkind;
  int block0, reserved;
} osuf_pack_desc;
#endif
static_assert(sizeof(osuf_pack_desc) == 80, "osuf_pack_desc is part of the C ABI");
// descriptor of the osuf_skinny_*_group entry points (same definition in include/osufusion_hip.h)
#ifndef OSUF_LINEAR_DESC_DEFINED
#define OSUF_LINEAR_DESC_DEFINED
typedef struct osuf_linear_desc {
  const float* W;          /* (N, K) fp32 master weight */
  const float* bias;       /* (N) or NULL */
  float* y;                /* forward output rows, row stride ldy */
  const float* dy;         /* backward: gradient of y, row stride lddy */
  long ldy, lddy;
  int N, block0;
} osuf_linear_desc;                                 /* 56 bytes */
#endif
static_assert(sizeof(osuf_linear_desc) == 56, "osuf_linear_desc is part of the C ABI");

#define OSUF_DT_F32 0
#define OSUF_DT_BF16 1
#define OSUF_DT_F32X3 2   /* GEMM entry points only: fp32 storage, products as three bf16 MFMAs on split (hi + lo) operands */
// kernel choice of the attention-backward entry points (per call; nothing is read from the environment)
#define OSUF_ATTN_AUTO 0
#define OSUF_ATTN_PLAIN 1
#define OSUF_ATTN_PIPE 2
// how the fused attention backward sums dQ over its 256-key workgroups
#define OSUF_DQ_ATOMIC 0          /* fp32 atomics; the sweep (256 or 512 keys per workgroup) is picked by shape */
#define OSUF_DQ_SLABS 1
#define OSUF_DQ_ATOMIC_256 2      /* force the 8-wave, 256-key sweep */
#define OSUF_DQ_ATOMIC_512 3      /* force the 4-wave, 512-key sweep (N % 32 == 0) */
#define OSUF_DQ_TIMING_512 4      /* the 512-key sweep WITHOUT its atomics: timing only, dq is zero */
#define OSUF_DQ_ATOMIC_512A 5     /* the 512-key sweep with the generated, hand-placed loop (attn_bwd512_asm.inc) */
#define OSUF_DQ_PREZEROED 0x100   /* flag: the dQ accumulator was zero-filled by osuf_mqa_fwd_zdq (no memset in the backward entry point) */

typedef uint16_t bf16_t;                                   // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T*

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even via the hardware cast (keeps NaN a NaN; see MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));     // one v_cvt_pk_bf16_f32
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kPer16B = 4;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float rnd(float v) { return v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ float rnd(float v) { return round_bf16(v); }
};

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32); pointers must be 16-B aligned
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r[i] << 16);
    v[2 * i + 1] = __uint_as_float(r[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
// 4 consecutive elements
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  u32x2 r = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xFFFF0000u);
  v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xFFFF0000u);
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<u32x2*>(p) = r;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  f32x4 a; a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
  *reinterpret_cast<f32x4*>(p) = a;
}

// v_rcp_f32 (<= 1 ulp) instead of an IEEE division: the division expands to ~10 VALU ops per element (v_div_scale / v_div_fmas /
// v_div_fixup + Newton steps), and the GroupNorm / SiLU kernels evaluate it for every element of every activation
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx silu(x) = s * (1 + x * (1 - s)), s = sigmoid(x)
__device__ __forceinline__ float silu_grad_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// wave64 butterfly reductions over the low `width` lanes of each aligned group (width: power of 2 <= 64)
template <int WIDTH = 64> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int WIDTH = 64> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = WIDTH >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group_sum_dyn(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group_max_dyn(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// C-ABI status: 0 ok, negative = argument error, positive = hipError_t
#define OSUF_OK 0
#define OSUF_EINVAL (-1)
#define OSUF_EUNSUPPORTED (-2)

static inline int osuf_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? OSUF_OK : (int)e;
}
