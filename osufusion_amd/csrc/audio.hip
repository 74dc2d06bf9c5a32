// Audio front end: log-magnitude variable-Q transform of a mono waveform, the feature the denoiser is conditioned on
// (reference: osu_fusion/scripts/dataset_creator.py:36-55 -- np.log(np.abs(librosa.vqt(...)) + 1e-10)).
//
// Direct form.  With the complex wavelet bank laid out as rows of a [2*bins][K] fp32 matrix (real parts, then imaginary
// parts, already time-reversed and centred on the host -- osufusion_amd/audio.py), frame t of the transform is
//     spec[t][j] = sum_n wave_pad[t*hop + n] * bank[j][n]
// i.e. ONE tap-GEMM whose A operand is the padded waveform read with a row stride of `hop` samples (rows overlap; nothing
// is materialised), run on the fp32 MFMA path of gemm.hip.  The kernel below then folds |.|, the per-bin sqrt(length)
// scale and the log into the (frames, 2*bins) -> (bins, frames) transpose.
#include "common.hpp"

extern "C" int osuf_gemm_nt(int dtype, const void* A, long lda, const void* W, long ldw, long tapstride,
                            void* C, long ldc, void* C2, long ldc2, const void* R, long ldr, const void* U, long ldu,
                            const float* bias, const float* rscale, double* stats,
                            int M, int N, int K, int taps, int Lin, int Lout, int stride, int pad, int mode, int act,
                            hipStream_t stream);

namespace {

constexpr int kFrames = 64;       // frames per block

// out[k][t] = log(scale[k] * |spec[t][k] + i spec[t][bins + k]| + eps)
// Reads are row-contiguous (2*bins floats per frame), writes are frame-contiguous (64 floats = 256 B per bin row);
// the transpose goes through LDS with an odd row pitch.
__global__ __launch_bounds__(256) void vqt_logmag_kernel(const float* __restrict__ spec, long ld, float* __restrict__ out,
                                                         long ldo, const float* __restrict__ scale, int bins, long frames,
                                                         float eps) {
  extern __shared__ float tile[];                          // [kFrames][2*bins + 1]
  const int cols = 2 * bins, pitch = cols + 1;
  const long t0 = (long)blockIdx.x * kFrames;
  const int nt = (int)((frames - t0) < kFrames ? (frames - t0) : kFrames);
  for (int e = threadIdx.x; e < nt * cols; e += blockDim.x) {
    const int r = e / cols, c = e - r * cols;
    tile[r * pitch + c] = spec[(t0 + r) * ld + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < bins * kFrames; e += blockDim.x) {
    const int k = e / kFrames, r = e - k * kFrames;
    if (r < nt) {
      const float re = tile[r * pitch + k], im = tile[r * pitch + bins + k];
      out[(long)k * ldo + t0 + r] = logf(scale[k] * sqrtf(re * re + im * im) + eps);
    }
  }
}

// Octave recursion of librosa.vqt (core/constantq.py): between octaves the signal is low-passed and decimated by 2
//   out[m] = sum_j taps[j] * in[2m + j - (ntaps-1)/2]     (zeros outside [0, n_in); taps hold the sqrt(2) of resample(scale=True))
// One output per thread, taps staged in LDS; a 4-minute song is 5.3 M samples x ~375 taps at the first stage: ~2 GFLOP, HBM-trivial.
__global__ __launch_bounds__(256) void fir_decimate2_kernel(const float* __restrict__ in, long n_in, const float* __restrict__ taps, int ntaps,
                                                            float* __restrict__ out, long n_out) {
  extern __shared__ float st[];
  for (int j = threadIdx.x; j < ntaps; j += blockDim.x) st[j] = taps[j];
  __syncthreads();
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_out) return;
  const long base = 2 * m - (ntaps - 1) / 2;
  float acc = 0.f;
  for (int j = 0; j < ntaps; ++j) {
    const long i = base + j;
    if (i >= 0 && i < n_in) acc = fmaf(st[j], in[i], acc);
  }
  out[m] = acc;
}

// rows[t][i] = in[t*hop + i] (0 beyond n_in): explicit framing for the low octaves, whose hops (22, 11 samples) are not 16-byte
// multiples and so cannot be read in place by the GEMM loader
__global__ __launch_bounds__(256) void frame_rows_kernel(const float* __restrict__ in, long n_in, int hop, int K, float* __restrict__ out,
                                                         long frames) {
  const long total = frames * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long t = e / K;
    const long i = t * hop + (e - t * K);
    out[e] = i < n_in ? in[i] : 0.f;
  }
}

}  // namespace

extern "C" int osuf_fir_decimate2(const float* in, long n_in, const float* taps, int ntaps, float* out, long n_out, hipStream_t stream) {
  if (!in || !taps || !out || n_in <= 0 || n_out <= 0 || ntaps <= 0 || ntaps > 8192 || (ntaps & 1) == 0) return OSUF_EINVAL;
  fir_decimate2_kernel<<<(unsigned)((n_out + 255) / 256), 256, ntaps * sizeof(float), stream>>>(in, n_in, taps, ntaps, out, n_out);
  return osuf_launch_status();
}

extern "C" int osuf_frame_rows(const float* in, long n_in, int hop, int K, float* out, long frames, hipStream_t stream) {
  if (!in || !out || n_in <= 0 || hop <= 0 || K <= 0 || frames <= 0) return OSUF_EINVAL;
  long blocks = (frames * K + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  frame_rows_kernel<<<(unsigned)blocks, 256, 0, stream>>>(in, n_in, hop, K, out, frames);
  return osuf_launch_status();
}

extern "C" int osuf_vqt_logmag(const float* spec, long ld, float* out, long ldo, const float* scale, int bins, long frames,
                               float eps, hipStream_t stream) {
  if (!spec || !out || !scale || bins <= 0 || frames <= 0 || ld < 2 * bins || ldo < frames) return OSUF_EINVAL;
  const size_t lds = (size_t)kFrames * (2 * bins + 1) * sizeof(float);
  if (lds > 160 * 1024) return OSUF_EUNSUPPORTED;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)vqt_logmag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = (unsigned)((frames + kFrames - 1) / kFrames);
  vqt_logmag_kernel<<<grid, 256, lds, stream>>>(spec, ld, out, ldo, scale, bins, frames, eps);
  return osuf_launch_status();
}

// wave_pad: the zero-padded waveform, at least (frames - 1) * hop + K floats; bank: [2*bins][K]; spec_ws: workspace of
// frames * 2*bins floats; out: [bins][ldo].  hop and K must be multiples of 4 (16-byte rows for the fp32 GEMM loader).
extern "C" int osuf_log_vqt(const float* wave_pad, long n_pad, const float* bank, int K, int bins, int hop, const float* scale,
                            float eps, float* spec_ws, float* out, long ldo, long frames, hipStream_t stream) {
  if (!wave_pad || !bank || !spec_ws || frames <= 0 || frames >= (1L << 31) || hop <= 0 || K <= 0) return OSUF_EINVAL;
  if ((frames - 1) * (long)hop + K > n_pad) return OSUF_EINVAL;       // the last frame would read past the buffer
  const int M = (int)frames, N = 2 * bins;
  int rc = osuf_gemm_nt(OSUF_DT_F32, wave_pad, hop, bank, K, 0, spec_ws, N, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr,
                        nullptr, M, N, K, 1, M, M, 1, 0, 0, 0, stream);
  if (rc != OSUF_OK) return rc;
  return osuf_vqt_logmag(spec_ws, N, out, ldo, scale, bins, frames, eps, stream);
}
