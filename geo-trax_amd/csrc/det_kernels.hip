// Non-GEMM detector kernels for gfx950. Behaviour specifications are restated in
// oracle/yolov8_ref.py (the checker); the third-party calls they stand in for are listed in
// SURVEY.md §2b (K2 preprocess, K3 stem/SPPF/upsample, K4 decode + NMS).
#include <hip/hip_runtime.h>

#include <mutex>
#include <hip/hip_fp16.h>

#include <cmath>

#include "det_kernels.hpp"

namespace gtx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// ============================================================================ letterbox
namespace {
// Python's round(): ties to even.
inline double py_round(double v) { return std::nearbyint(v); }
}  // namespace

Letterbox letterbox_geometry(int src_h, int src_w, int imgsz, bool rect, int stride) {
  Letterbox lb{};
  lb.src_h = src_h;
  lb.src_w = src_w;
  const double r = std::min((double)imgsz / src_h, (double)imgsz / src_w);
  lb.new_w = (int)py_round(src_w * r);
  lb.new_h = (int)py_round(src_h * r);
  double dw = imgsz - lb.new_w, dh = imgsz - lb.new_h;
  if (rect) {
    dw = std::fmod(dw, (double)stride);
    dh = std::fmod(dh, (double)stride);
  }
  dw /= 2;
  dh /= 2;
  const int top = (int)py_round(dh - 0.1), bottom = (int)py_round(dh + 0.1);
  const int left = (int)py_round(dw - 0.1), right = (int)py_round(dw + 0.1);
  lb.top = top;
  lb.left = left;
  lb.net_h = lb.new_h + top + bottom;
  lb.net_w = lb.new_w + left + right;
  lb.gain = std::min((double)lb.net_h / src_h, (double)lb.net_w / src_w);
  return lb;
}

// ============================================================================ preprocess
// cv2.cvtColor(BGR2GRAY) 8-bit fixed point: (B*1868 + G*9617 + R*4899 + 8192) >> 14.
__device__ __forceinline__ int bgr2gray(int b, int g, int r) {
  return (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14;
}

struct PreParams {
  const uint8_t* frames;
  void* img;
  uint8_t* gray;
  int src_h, src_w, net_h, net_w, new_h, new_w, top, left, gh, gw;
  int exact2x;       // new == src/2: cv2.resize INTER_LINEAR degenerates to the rounded 2x2 mean
  float scale_x, scale_y;  // src/new for the general bilinear path
};

// The network input is kept as RGB0 bytes: the /255 of ultralytics' preprocess maps 256 possible values, so the stem
// kernels apply it through a 256-entry table as they read (the same correctly rounded fp32 division, per table entry
// instead of per pixel), and this pass writes 4 bytes per pixel instead of 16 (fp32) or 8 (fp16): 29.5 MB instead of
// 118 MB per two 1920x1920 frames, and as much less for the stem to read back.
// One thread per network-input pixel. Exact-2x path: out = (a+b+c+d+2)>>2 per channel, which is
// what cv2.resize(INTER_LINEAR) yields for an exact 0.5 scale (OpenCV switches to its 2x2 area
// kernel; the fixed-point bilinear gives the same integers). General path: OpenCV's 11-bit
// fixed-point bilinear, restated from the published resize.cpp arithmetic (cv2 itself is not in this image); held against
// an independent implementation of the same half-pixel-centre mapping, skimage.transform.resize(order=1), to <= 1 grey
// level on a colour crop at the 0.711 ratio of a 2.7K source (tests/test_independent.py, GPU and oracle).
__global__ __launch_bounds__(256) void preprocess_kernel(const PreParams p) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  const int n = blockIdx.z;
  if (ox >= p.net_w) return;
  const uint8_t* __restrict__ src = p.frames + (size_t)n * p.src_h * p.src_w * 3;
  uchar4* __restrict__ dst = static_cast<uchar4*>(p.img) + ((size_t)n * p.net_h + oy) * p.net_w + ox;
  const int ry = oy - p.top, rx = ox - p.left;
  if (ry < 0 || ry >= p.new_h || rx < 0 || rx >= p.new_w) {
    *dst = make_uchar4(114, 114, 114, 0);
    return;
  }
  int B, G, R;
  if (p.exact2x) {
    const uint8_t* r0 = src + ((size_t)(2 * ry) * p.src_w + 2 * rx) * 3;
    const uint8_t* r1 = r0 + (size_t)p.src_w * 3;
    B = (r0[0] + r0[3] + r1[0] + r1[3] + 2) >> 2;
    G = (r0[1] + r0[4] + r1[1] + r1[4] + 2) >> 2;
    R = (r0[2] + r0[5] + r1[2] + r1[5] + 2) >> 2;
    if (p.gray) {
      const int g00 = bgr2gray(r0[0], r0[1], r0[2]), g01 = bgr2gray(r0[3], r0[4], r0[5]);
      const int g10 = bgr2gray(r1[0], r1[1], r1[2]), g11 = bgr2gray(r1[3], r1[4], r1[5]);
      p.gray[((size_t)n * p.gh + ry) * p.gw + rx] = (uint8_t)((g00 + g01 + g10 + g11 + 2) >> 2);
    }
  } else {
    float fx = (rx + 0.5f) * p.scale_x - 0.5f;
    float fy = (ry + 0.5f) * p.scale_y - 0.5f;
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= sx;
    fy -= sy;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= p.src_w - 1) { sx = p.src_w - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= p.src_h - 1) { sy = p.src_h - 1; fy = 0.f; }
    const int sx1 = min(sx + 1, p.src_w - 1), sy1 = min(sy + 1, p.src_h - 1);
    const int a1 = __float2int_rn(fx * 2048.f), a0 = __float2int_rn((1.f - fx) * 2048.f);
    const int b1 = __float2int_rn(fy * 2048.f), b0 = __float2int_rn((1.f - fy) * 2048.f);
    const uint8_t* r0 = src + (size_t)sy * p.src_w * 3;
    const uint8_t* r1 = src + (size_t)sy1 * p.src_w * 3;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
      const int h1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
      v[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    }
    B = v[0]; G = v[1]; R = v[2];
  }
  *dst = make_uchar4((unsigned char)R, (unsigned char)G, (unsigned char)B, 0);
}

// Gray-only pass for frames whose letterbox is not the exact-2x case but the stabilizer still
// wants full-res gray -> 2x2 mean.
__global__ __launch_bounds__(256) void gray_half_kernel(const uint8_t* __restrict__ frames, int n_img,
                                                        int h, int w, uint8_t* __restrict__ gray,
                                                        int gh, int gw) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
  if (x >= gw) return;
  const uint8_t* r0 = frames + ((size_t)n * h * w + (size_t)(2 * y) * w + 2 * x) * 3;
  const uint8_t* r1 = r0 + (size_t)w * 3;
  const int g00 = bgr2gray(r0[0], r0[1], r0[2]), g01 = bgr2gray(r0[3], r0[4], r0[5]);
  const int g10 = bgr2gray(r1[0], r1[1], r1[2]), g11 = bgr2gray(r1[3], r1[4], r1[5]);
  gray[((size_t)n * gh + y) * gw + x] = (uint8_t)((g00 + g01 + g10 + g11 + 2) >> 2);
}

// The exact-2x case (a 3840 x 2160 frame into a 1920-wide letterbox: every frame of the headline workload) with 4 output pixels
// per thread: the 2 x 24 source bytes of a thread arrive as six 8-byte loads instead of 48 byte loads, the four RGB0 pixels
// leave as one 16-byte store and the four gray pixels as one word. Same integers as preprocess_kernel.
__global__ __launch_bounds__(256) void preprocess2x_kernel(const PreParams p) {
  const int groups = p.net_w >> 2;                                   // 4-pixel groups per row (launch_preprocess checks divisibility)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (gid >= groups * p.net_h) return;
  const int oy = gid / groups, ox = (gid - oy * groups) << 2;
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(static_cast<uchar4*>(p.img) + ((size_t)n * p.net_h + oy) * p.net_w + ox);
  const int ry = oy - p.top, rx = ox - p.left;
  if (ry < 0 || ry >= p.new_h || rx < 0 || rx >= p.new_w) {
    const unsigned pad = 114u | (114u << 8) | (114u << 16);
    *dst = make_uint4(pad, pad, pad, pad);
    return;
  }
  const uint8_t* r0p = p.frames + (size_t)n * p.src_h * p.src_w * 3 + ((size_t)(2 * ry) * p.src_w + 2 * rx) * 3;
  const uint2* r0 = reinterpret_cast<const uint2*>(r0p);
  const uint2* r1 = reinterpret_cast<const uint2*>(r0p + (size_t)p.src_w * 3);
  const uint2 a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
  const unsigned wa[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y}, wb[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
  auto A = [&](int i) { return (int)((wa[i >> 2] >> (8 * (i & 3))) & 255u); };     // byte i of the 24 of the upper row
  auto Bq = [&](int i) { return (int)((wb[i >> 2] >> (8 * (i & 3))) & 255u); };    // ... of the lower row
  unsigned out[4], gray4 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = 6 * k;                                              // source pixels 2k and 2k + 1: bytes o .. o + 5 (B G R B G R)
    const int B = (A(o) + A(o + 3) + Bq(o) + Bq(o + 3) + 2) >> 2;
    const int G = (A(o + 1) + A(o + 4) + Bq(o + 1) + Bq(o + 4) + 2) >> 2;
    const int R = (A(o + 2) + A(o + 5) + Bq(o + 2) + Bq(o + 5) + 2) >> 2;
    out[k] = (unsigned)R | ((unsigned)G << 8) | ((unsigned)B << 16);
    if (p.gray) {
      const int g = (bgr2gray(A(o), A(o + 1), A(o + 2)) + bgr2gray(A(o + 3), A(o + 4), A(o + 5)) + bgr2gray(Bq(o), Bq(o + 1), Bq(o + 2)) +
                     bgr2gray(Bq(o + 3), Bq(o + 4), Bq(o + 5)) + 2) >> 2;
      gray4 |= (unsigned)g << (8 * k);
    }
  }
  *dst = make_uint4(out[0], out[1], out[2], out[3]);
  if (p.gray) *reinterpret_cast<unsigned*>(p.gray + ((size_t)n * p.gh + ry) * p.gw + rx) = gray4;
}

void launch_preprocess(int dtype, const uint8_t* frames, int n, const Letterbox& lb, void* img,
                       uint8_t* gray, int gh, int gw, hipStream_t s) {
  PreParams p{};
  p.frames = frames;
  p.img = img;
  p.src_h = lb.src_h; p.src_w = lb.src_w;
  p.net_h = lb.net_h; p.net_w = lb.net_w;
  p.new_h = lb.new_h; p.new_w = lb.new_w;
  p.top = lb.top; p.left = lb.left;
  p.gh = gh; p.gw = gw;
  p.exact2x = (lb.new_h * 2 == lb.src_h && lb.new_w * 2 == lb.src_w) ? 1 : 0;
  p.scale_x = (float)((double)lb.src_w / lb.new_w);
  p.scale_y = (float)((double)lb.src_h / lb.new_h);
  const bool fuse_gray = gray && p.exact2x && gh == lb.new_h && gw == lb.new_w;
  p.gray = fuse_gray ? gray : nullptr;
  dim3 grid(cdiv(lb.net_w, 256), lb.net_h, n);
  (void)dtype;                 // the image is RGB0 bytes for every arithmetic
  const bool wide = p.exact2x && lb.net_w % 4 == 0 && lb.left % 4 == 0 && lb.new_w % 4 == 0 && lb.src_w % 8 == 0 &&
                    reinterpret_cast<uintptr_t>(frames) % 8 == 0 && (!p.gray || (gw % 4 == 0 && reinterpret_cast<uintptr_t>(gray) % 4 == 0));
  if (wide)
    hipLaunchKernelGGL(preprocess2x_kernel, dim3(cdiv((lb.net_w / 4) * lb.net_h, 256), n), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, s, p);
  GTX_HIP(hipGetLastError());
  if (gray && !fuse_gray) {
    GTX_CHECK(gh * 2 == lb.src_h && gw * 2 == lb.src_w, "gray output must be half the frame size");
    hipLaunchKernelGGL(gray_half_kernel, dim3(cdiv(gw, 256), gh, n), dim3(256), 0, s, frames, n,
                       lb.src_h, lb.src_w, gray, gh, gw);
    GTX_HIP(hipGetLastError());
  }
}

// ============================================================================ stem conv
// x * sigmoid(x) with v_exp_f32 and v_rcp_f32 (1 ulp each); hipcc expands __fdividef to a full IEEE division (10 instructions)
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// value of an image byte as the network sees it: ultralytics' `im / 255` in fp32, rounded to fp16 for half=True
template <typename T> __device__ __forceinline__ float px_value(int v);
template <> __device__ __forceinline__ float px_value<float>(int v) { return (float)v / 255.f; }
template <> __device__ __forceinline__ float px_value<_Float16>(int v) { return (float)(_Float16)((float)v / 255.f); }
template <typename T> __device__ __forceinline__ void store8(T* dst, const float* v);
template <> __device__ __forceinline__ void store8<_Float16>(_Float16* dst, const float* v) {
  half8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
  *reinterpret_cast<half8*>(dst) = h;
}
template <> __device__ __forceinline__ void store8<float>(float* dst, const float* v) {
  reinterpret_cast<float4*>(dst)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(dst)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// One thread = one output pixel, all C0 output channels (direct 27-tap conv on the VALU; the
// layer is bound by its 2*C0-byte-per-pixel store, not by arithmetic). Weights are read through
// wave-uniform addresses, i.e. the scalar cache.
template <typename T, int C0>
__global__ __launch_bounds__(256) void stem_kernel(const uchar4* __restrict__ img, int h, int w,
                                                   const float* __restrict__ w27,
                                                   const float* __restrict__ bias, T* __restrict__ out,
                                                   int ho, int wo) {
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = px_value<T>(threadIdx.x);
  __syncthreads();
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y, n = blockIdx.z;
  if (ox >= wo) return;
  float acc[C0];
#pragma unroll
  for (int c = 0; c < C0; ++c) acc[c] = bias[c];
  const uchar4* base = img + (size_t)n * h * w;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      float px[3] = {0.f, 0.f, 0.f};
      if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
        const uchar4 u = base[(size_t)iy * w + ix];
        px[0] = s_lut[u.x]; px[1] = s_lut[u.y]; px[2] = s_lut[u.z];
      }
      const float* wt = w27 + (ky * 3 + kx) * 3 * C0;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int c = 0; c < C0; ++c) acc[c] = fmaf(px[ci], wt[ci * C0 + c], acc[c]);
    }
  }
  T* o = out + (((size_t)n * ho + oy) * wo + ox) * C0;
#pragma unroll
  for (int c = 0; c < C0; c += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = silu_f(acc[c + i]);
    store8<T>(o + c, v);
  }
}

typedef float floatx16_t __attribute__((ext_vector_type(16)));

// fp16 stem on the matrix cores: implicit GEMM with K = 12 taps x 4 channels (RGB0; taps 9..11 and
// channel 3 are zero weights), D[cout 32][pixel 32] += W[cout][k] X[k][pixel]. A wave takes 32
// consecutive output pixels of a row; each lane gathers its B fragments straight from the NHWC4
// image (two 8-byte pixels = 16 bytes = one k-step half), the weights stay in registers for all the
// tiles a wave walks, the epilogue is bias + SiLU and 8-byte NHWC stores. No LDS.
__global__ __launch_bounds__(256) void stem_mfma_kernel(const uchar4* __restrict__ img, int h, int w,
                                                        const _Float16* __restrict__ wpk /*[groups][3][64][8]*/,
                                                        const float* __restrict__ bias, _Float16* __restrict__ out,
                                                        int ho, int wo, int c0, int groups, int tiles_per_row, int n_tiles) {
  constexpr int kStemPitch = 64 + 16;                   // bytes per staged pixel (+16: conflict-free 8-B writes)
  __shared__ __attribute__((aligned(16))) char s_stage[4 * 32 * kStemPitch];
  __shared__ _Float16 s_lut[256];                       // byte -> fp16(byte / 255)
  s_lut[threadIdx.x] = (_Float16)((float)threadIdx.x / 255.f);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const int n = blockIdx.y;
  const uchar4* base = img + (size_t)n * h * w;
  for (int g = 0; g < groups; ++g) {
    half8 wf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) wf[s] = *reinterpret_cast<const half8*>(wpk + (((size_t)g * 3 + s) * 64 + lane) * 8);
    // as in stem_split_kernel: the next tile's pixels are gathered before the current tile is worked on
    auto gather = [&](int t, unsigned (&raw)[6]) {
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) raw[kk] = 0u;
      if (t >= n_tiles) return;
      const int oy = t / tiles_per_row, tx = t - oy * tiles_per_row, ox = tx * 32 + r;
      const bool interior = oy > 0 && tx > 0 && oy * 2 + 1 < h && tx * 64 + 64 < w && tx * 32 + 32 <= wo;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int tap = 2 * (2 * s + hh) + q;
          if (tap < 9) {
            const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
            if (interior || (iy >= 0 && iy < h && ix >= 0 && ix < w && ox < wo))
              raw[2 * s + q] = reinterpret_cast<const unsigned*>(base)[(unsigned)(iy * w + ix)];
          }
        }
    };
    unsigned raw[6], nxt[6];
    gather(wave, raw);
    for (int t = wave; t < n_tiles; t += nwaves) {
      const int oy = t / tiles_per_row, ox = (t - oy * tiles_per_row) * 32 + r;
      gather(t + nwaves, nxt);
      floatx16_t acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        half8 xf;
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[e] = (_Float16)0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned px = raw[2 * s + q];
          xf[4 * q] = s_lut[px & 255u]; xf[4 * q + 1] = s_lut[(px >> 8) & 255u]; xf[4 * q + 2] = s_lut[(px >> 16) & 255u];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], xf, acc, 0, 0, 0);
      }
      if (c0 == 32) {
        // 32 channels = one 64-B line per pixel: the wave transposes its 32 pixels through LDS and stores 16 B per
        // lane, 1 KB contiguous per instruction (the direct form below writes each line as eight 8-B pieces)
        char* stg = s_stage + (threadIdx.x >> 6) * (32 * kStemPitch);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int cl = 8 * g4 + 4 * hh;
          half4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (_Float16)silu_f(acc[4 * g4 + i] + bias[cl + i]);
          *reinterpret_cast<half4*>(stg + r * kStemPitch + cl * 2) = v;
        }
        const int ox0 = (t - oy * tiles_per_row) * 32;
        _Float16* orow = out + (((size_t)n * ho + oy) * wo + ox0) * 32;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int p = it * 16 + (lane >> 2), q = lane & 3;
          const uint4 val = *reinterpret_cast<const uint4*>(stg + p * kStemPitch + q * 16);
          if (ox0 + p < wo) *reinterpret_cast<uint4*>(orow + p * 32 + q * 8) = val;
        }
      } else if (ox < wo) {
        _Float16* o = out + (((size_t)n * ho + oy) * wo + ox) * c0 + g * 32;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int cl = 8 * g4 + 4 * hh;
          if (g * 32 + cl < c0) {
            half4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (_Float16)silu_f(acc[4 * g4 + i] + bias[g * 32 + cl + i]);
            *reinterpret_cast<half4*>(o + cl) = v;
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) raw[kk] = nxt[kk];
    }
  }
}

// fp32 stem on the fp16 matrix pipe ("split-f16x3", see conv_igemm_split.hip): the same implicit GEMM as
// stem_mfma_kernel on RGB0 bytes, output in the pair format of split_format.hpp. Every lane splits the pixels it gathers into hi + lo fp16
// parts, the weights arrive pre-split (scaled by an exact power of two, undone by `acc_scale`), a product costs three
// MFMAs (w_lo x_hi + w_hi x_lo + w_hi x_hi). The 32-channel fp32 output row of a pixel is 128 B: the wave transposes
// its 32 pixels through LDS and stores whole lines.
__global__ __launch_bounds__(256) void stem_split_kernel(const uchar4* __restrict__ img, int h, int w,
                                                         const _Float16* __restrict__ wpk /*[groups][3][hi|lo][64][8]*/,
                                                         const float* __restrict__ bias, float acc_scale, float* __restrict__ out,
                                                         int ho, int wo, int c0, int groups, int tiles_per_row, int n_tiles) {
  constexpr int kPitch = 128 + 16;                      // bytes per staged pixel (32 fp32 channels + pad)
  __shared__ __attribute__((aligned(16))) char s_stage[4 * 32 * kPitch];
  __shared__ unsigned s_lut[256];                       // byte -> hi | lo << 16 of byte / 255 (already split)
  {
    const float f = (float)threadIdx.x / 255.f;
    const _Float16 hi = (_Float16)f, lo = (_Float16)(f - (float)hi);
    s_lut[threadIdx.x] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const int n = blockIdx.y;
  const uchar4* base = img + (size_t)n * h * w;     // offsets below fit 32 bits (one image)
  for (int g = 0; g < groups; ++g) {
    half8 wh[3], wl[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      wh[s] = *reinterpret_cast<const half8*>(wpk + ((((size_t)g * 3 + s) * 2 + 0) * 64 + lane) * 8);
      wl[s] = *reinterpret_cast<const half8*>(wpk + ((((size_t)g * 3 + s) * 2 + 1) * 64 + lane) * 8);
    }
    // A wave walks several tiles and gathers the next tile's pixels (6 RGB0 words per lane) before it works on the current
    // one: alone on the GPU the kernel is bound by the latency of that gather, and next to the other stream's convolutions
    // it was three times slower than alone (320 vs 99 us) with nothing in flight to cover it.
    auto gather = [&](int t, unsigned (&raw)[6]) {
#pragma unroll
      for (int k = 0; k < 6; ++k) raw[k] = 0u;             // byte 0 = value 0.0 = the convolution's zero padding
      if (t >= n_tiles) return;
      const int oy = t / tiles_per_row, tx = t - oy * tiles_per_row, ox = tx * 32 + r;
      // tiles whose 3x3 stride-2 footprint lies inside the image skip the per-tap bounds tests (all but the first row / column)
      const bool interior = oy > 0 && tx > 0 && oy * 2 + 1 < h && tx * 64 + 64 < w && tx * 32 + 32 <= wo;
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int tap = 2 * (2 * s + hh) + q;
          if (tap < 9) {
            const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
            if (interior || (iy >= 0 && iy < h && ix >= 0 && ix < w && ox < wo))
              raw[2 * s + q] = reinterpret_cast<const unsigned*>(base)[(unsigned)(iy * w + ix)];
          }
        }
    };
    unsigned raw[6], nxt[6];
    gather(wave, raw);
    for (int t = wave; t < n_tiles; t += nwaves) {
      const int oy = t / tiles_per_row, tx = t - oy * tiles_per_row, ox = tx * 32 + r;
      gather(t + nwaves, nxt);
      floatx16_t acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        half8 xh, xl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[e] = (_Float16)0.f; xl[e] = (_Float16)0.f; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const unsigned px = raw[2 * s + q];
          const unsigned e3[3] = {s_lut[px & 255u], s_lut[(px >> 8) & 255u], s_lut[(px >> 16) & 255u]};
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            xh[4 * q + e] = __builtin_bit_cast(_Float16, (unsigned short)(e3[e] & 0xffffu));
            xl[4 * q + e] = __builtin_bit_cast(_Float16, (unsigned short)(e3[e] >> 16));
          }
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], xh, acc, 0, 0, 0);
      }
      // Output in pair format (split_format.hpp): lanes l and l + 32 hold the two halves of an 8-channel group; two
      // v_permlane32_swap make the group's 16-byte hi chunk (lane l) and lo chunk (lane l + 32), as in the conv epilogue.
      uint4 chunk[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = g * 32 + 8 * g4 + 4 * hh;
        half4 hv, lv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // SiLU of a sum of 27 products with inputs in [0, 1]: inside fp16's range for any sane weights; the clamp only keeps
          // an absurd checkpoint from producing inf - inf (bias is padded to whole 32-channel groups)
          const float v = __builtin_amdgcn_fmed3f(silu_f(fmaf(acc[4 * g4 + i], acc_scale, bias[cl + i])), -65504.f, 65504.f);
          const _Float16 hi = (_Float16)v;
          hv[i] = hi;
          lv[i] = (_Float16)(v - (float)hi);
        }
        const uint2 hu = *reinterpret_cast<const uint2*>(&hv), lu = *reinterpret_cast<const uint2*>(&lv);
        const auto sx = __builtin_amdgcn_permlane32_swap(hu.x, lu.x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(hu.y, lu.y, false, false);
        chunk[g4] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
      if (c0 == 32) {
        char* stg = s_stage + (threadIdx.x >> 6) * (32 * kPitch);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<uint4*>(stg + r * kPitch + (8 * g4 + 4 * hh) * 4) = chunk[g4];
        const int ox0 = tx * 32;
        float* orow = out + (((size_t)n * ho + oy) * wo + ox0) * 32;
#pragma unroll
        for (int it = 0; it < 4; ++it) {                  // 8 pixels x 128 B per store instruction
          const int p = it * 8 + (lane >> 3), q = lane & 7;
          const uint4 val = *reinterpret_cast<const uint4*>(stg + p * kPitch + q * 16);
          if (ox0 + p < wo) *reinterpret_cast<uint4*>(orow + p * 32 + q * 4) = val;
        }
      } else if (ox < wo) {
        float* o = out + (((size_t)n * ho + oy) * wo + ox) * c0 + g * 32;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          if (g * 32 + 8 * g4 < c0) *reinterpret_cast<uint4*>(o + 8 * g4 + 4 * hh) = chunk[g4];   // c0 is a multiple of 8: whole groups
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) raw[k] = nxt[k];
    }
  }
}

void launch_stem(int dtype, const void* img, int n, int h, int w, const float* w27, const float* bias,
                 const void* wpk_f16, int c0, void* out, int ho, int wo, hipStream_t s, float acc_scale) {
  if (dtype == DT_F32S) {
    GTX_CHECK(wpk_f16 != nullptr, "stem: split weights missing");
    const int tiles_per_row = cdiv(wo, 32);
    const int n_tiles = tiles_per_row * ho;
    const int blocks = std::min((n_tiles + 15) / 16, 4096);      // four tiles per wave: something to prefetch
    hipLaunchKernelGGL(stem_split_kernel, dim3(blocks, n), dim3(256), 0, s, (const uchar4*)img, h, w, (const _Float16*)wpk_f16,
                       bias, acc_scale, (float*)out, ho, wo, c0, cdiv(c0, 32), tiles_per_row, n_tiles);
    GTX_HIP(hipGetLastError());
    return;
  }
  if (dtype == DT_F16 && wpk_f16) {
    const int tiles_per_row = cdiv(wo, 32);
    const int n_tiles = tiles_per_row * ho;
    const int blocks = std::min((n_tiles + 15) / 16, 4096);
    hipLaunchKernelGGL(stem_mfma_kernel, dim3(blocks, n), dim3(256), 0, s, (const uchar4*)img, h, w, (const _Float16*)wpk_f16,
                       bias, (_Float16*)out, ho, wo, c0, cdiv(c0, 32), tiles_per_row, n_tiles);
    GTX_HIP(hipGetLastError());
    return;
  }
  dim3 grid(cdiv(wo, 256), ho, n), block(256);
#define GTX_STEM(C)                                                                              \
  if (c0 == C) {                                                                                 \
    if (dtype == DT_F16)                                                                         \
      hipLaunchKernelGGL((stem_kernel<_Float16, C>), grid, block, 0, s, (const uchar4*)img, h,   \
                         w, w27, bias, (_Float16*)out, ho, wo);                                  \
    else                                                                                         \
      hipLaunchKernelGGL((stem_kernel<float, C>), grid, block, 0, s, (const uchar4*)img, h, w,   \
                         w27, bias, (float*)out, ho, wo);                                        \
    GTX_HIP(hipGetLastError());                                                                  \
    return;                                                                                      \
  }
  GTX_STEM(16) GTX_STEM(32) GTX_STEM(48) GTX_STEM(64) GTX_STEM(80)
#undef GTX_STEM
  fail(-3, "stem: unsupported channel count %d", c0);
}

// Packs the stem weights for stem_mfma_kernel: [group of 32 couts][k-step 3][lane 64][8 halves],
// lane (r = cout in group, hh) holds taps 2*(2s+hh) and 2*(2s+hh)+1, 4 channels each (RGB0).
std::vector<uint16_t> pack_stem_weights_f16(const float* w27 /*[27][c0], (tap*3+ch) major*/, int c0) {
  const int groups = cdiv(c0, 32);
  std::vector<uint16_t> out((size_t)groups * 3 * 64 * 8, 0);
  for (int g = 0; g < groups; ++g)
    for (int s = 0; s < 3; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, hh = lane >> 5, co = g * 32 + r;
        for (int q = 0; q < 2; ++q) {
          const int tap = 2 * (2 * s + hh) + q;
          for (int ch = 0; ch < 4; ++ch) {
            float v = 0.f;
            if (tap < 9 && ch < 3 && co < c0) v = w27[(size_t)(tap * 3 + ch) * c0 + co];
            const _Float16 hv = (_Float16)v;
            uint16_t bits;
            memcpy(&bits, &hv, 2);
            out[(((size_t)g * 3 + s) * 64 + lane) * 8 + 4 * q + ch] = bits;
          }
        }
      }
  return out;
}

// Split (hi + lo) packing for stem_split_kernel: [group][k-step 3][hi | lo][lane 64][8 halves], weights scaled by the
// power of two that puts max |w| in [2^13, 2^14); *acc_scale receives its inverse.
std::vector<uint16_t> pack_stem_weights_split(const float* w27, int c0, float* acc_scale) {
  const int groups = cdiv(c0, 32);
  float wmax = 0.f;
  for (int i = 0; i < 27 * c0; ++i) wmax = std::max(wmax, std::fabs(w27[i]));
  int shift = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    int e;
    std::frexp(wmax, &e);
    shift = std::max(-100, std::min(100, 14 - e));
  }
  const float up = std::ldexp(1.f, shift);
  *acc_scale = std::ldexp(1.f, -shift);
  std::vector<uint16_t> out((size_t)groups * 3 * 2 * 64 * 8, 0);
  for (int g = 0; g < groups; ++g)
    for (int s = 0; s < 3; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, hh = lane >> 5, co = g * 32 + r;
        for (int q = 0; q < 2; ++q) {
          const int tap = 2 * (2 * s + hh) + q;
          for (int ch = 0; ch < 4; ++ch) {
            float v = 0.f;
            if (tap < 9 && ch < 3 && co < c0) v = w27[(size_t)(tap * 3 + ch) * c0 + co] * up;
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            uint16_t bh, bl;
            memcpy(&bh, &hi, 2);
            memcpy(&bl, &lo, 2);
            out[((((size_t)g * 3 + s) * 2 + 0) * 64 + lane) * 8 + 4 * q + ch] = bh;
            out[((((size_t)g * 3 + s) * 2 + 1) * 64 + lane) * 8 + 4 * q + ch] = bl;
          }
        }
      }
  return out;
}

// ============================================================================ SPPF pools
template <typename T> struct Vec16;  // 16-byte vector of T
template <> struct Vec16<_Float16> { using type = half8; static constexpr int N = 8; };
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };

__device__ __forceinline__ half8 vmax(const half8& a, const half8& b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ float4 vmax(const float4& a, const float4& b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ half8 vneg_inf(half8) {
  half8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (_Float16)(-INFINITY);
  return v;
}
__device__ __forceinline__ float4 vneg_inf(float4) { return make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); }

// Pair format (split_format.hpp): a "vector" is 4 channels of one pixel = 8 bytes of hi halves and, 16 bytes further, 8 bytes
// of lo halves. The pools work on one 32-bit KEY per channel: (hi, lo) with both fp16 patterns mapped to unsigned codes that
// order like the numbers (sign bit flipped for positive, all bits for negative values), hi in the upper half. Keys order
// like hi + lo: hi = fp16(x) and |lo| is at most half the spacing on its side of hi, so a larger hi is never the smaller
// value, and with equal hi the lo decides. A maximum is one v_max_u32 per channel (the arithmetic form -- two conversions
// and an add per operand, a compare and two selects -- made the three cascaded pools the most expensive 30 MB of the pass),
// the winner's two halves travel inside the key unchanged, and the keys are made / unmade once per element.
struct PairTag {};
template <> struct Vec16<PairTag> { using type = uint4; static constexpr int N = 4; };
__device__ __forceinline__ uint4 vmax(const uint4& a, const uint4& b) { return make_uint4(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z), max(a.w, b.w)); }
__device__ __forceinline__ uint4 vneg_inf(uint4) { return make_uint4(0x03FF8000u, 0x03FF8000u, 0x03FF8000u, 0x03FF8000u); }   // (-inf, +0)
// two fp16 bit patterns -> two 16-bit codes that compare like the numbers as unsigned integers, and back
__device__ __forceinline__ unsigned pair_code2(unsigned x) { return x ^ ((((x >> 15) & 0x00010001u) * 0xFFFFu) | 0x80008000u); }
__device__ __forceinline__ unsigned pair_uncode2(unsigned s) { return s ^ ((((~s >> 15) & 0x00010001u) * 0xFFFFu) | 0x80008000u); }
// 8 bytes of hi halves + 8 bytes of lo halves (4 channels) <-> 4 keys
__device__ __forceinline__ uint4 pair_keys(uint2 hi, uint2 lo) {
  const unsigned h0 = pair_code2(hi.x), h1 = pair_code2(hi.y), l0 = pair_code2(lo.x), l1 = pair_code2(lo.y);
  return make_uint4((h0 << 16) | (l0 & 0xFFFFu), (h0 & 0xFFFF0000u) | (l0 >> 16), (h1 << 16) | (l1 & 0xFFFFu), (h1 & 0xFFFF0000u) | (l1 >> 16));
}
__device__ __forceinline__ void pair_unkeys(const uint4& k, uint2& hi, uint2& lo) {
  hi = make_uint2(pair_uncode2((k.x >> 16) | (k.y & 0xFFFF0000u)), pair_uncode2((k.z >> 16) | (k.w & 0xFFFF0000u)));
  lo = make_uint2(pair_uncode2((k.x & 0xFFFFu) | (k.y << 16)), pair_uncode2((k.z & 0xFFFFu) | (k.w << 16)));
}
// element type in memory, and the load / store of vector `v` (in units of N channels) of the pixel record at `px`
template <typename T> struct PoolMem { using elem = T; };
template <> struct PoolMem<PairTag> { using elem = float; };
template <typename T> __device__ __forceinline__ typename Vec16<T>::type pool_load(const typename PoolMem<T>::elem* px, int v) {
  return *reinterpret_cast<const typename Vec16<T>::type*>(px + v * Vec16<T>::N);
}
template <> __device__ __forceinline__ uint4 pool_load<PairTag>(const float* px, int v) {
  const char* g = reinterpret_cast<const char*>(px) + (v >> 1) * 32 + (v & 1) * 8;
  return pair_keys(*reinterpret_cast<const uint2*>(g), *reinterpret_cast<const uint2*>(g + 16));
}
template <typename T> __device__ __forceinline__ void pool_store(typename PoolMem<T>::elem* px, int v, const typename Vec16<T>::type& val) {
  *reinterpret_cast<typename Vec16<T>::type*>(px + v * Vec16<T>::N) = val;
}
template <> __device__ __forceinline__ void pool_store<PairTag>(float* px, int v, const uint4& val) {
  char* g = reinterpret_cast<char*>(px) + (v >> 1) * 32 + (v & 1) * 8;
  uint2 hi, lo;
  pair_unkeys(val, hi, lo);
  *reinterpret_cast<uint2*>(g) = hi;
  *reinterpret_cast<uint2*>(g + 16) = lo;
}

// The three cascaded 5x5/s1/p2 max-pools of SPPF in one pass: a block takes a 16x16 spatial tile of
// one 16-byte channel vector, stages the tile plus a 6-pixel halo in LDS (out-of-image = -inf,
// which is what the framework's padding does at every stage) and runs the three pools as separable
// row/column passes on shrinking regions (28 -> 24 -> 20 -> 16), storing the 5x5, 9x9 and 13x13
// results of the centre.
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_kernel(typename PoolMem<T>::elem* __restrict__ x, int h, int w, int c) {
  using V = typename Vec16<T>::type;
  using E = typename PoolMem<T>::elem;
  constexpr int VN = Vec16<T>::N, TS = 16, R = 6, P = TS + 2 * R;   // 28
  __shared__ V s_a[P * P], s_b[P * P];
  const int vecs = c / VN;
  const int tiles_x = (w + TS - 1) / TS;
  const int v = blockIdx.x % vecs, tile = blockIdx.x / vecs;
  const int tx0 = (tile % tiles_x) * TS, ty0 = (tile / tiles_x) * TS;
  const int n = blockIdx.y, cs = 4 * c;
  E* img = x + (size_t)n * h * w * cs;
  const V ninf = vneg_inf(V());
  for (int i = threadIdx.x; i < P * P; i += 256) {
    const int yy = ty0 - R + i / P, xx = tx0 - R + i % P;
    s_a[i] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? pool_load<T>(img + ((size_t)yy * w + xx) * cs, v) : ninf;
  }
  __syncthreads();
  int lo = 0;   // valid region of s_a is [lo, P-lo) in both axes
#pragma unroll 1
  for (int stage = 0; stage < 3; ++stage) {
    const int in_n = P - 2 * lo, out_n = in_n - 4;
    // rows: s_b[y][x'] = max over 5 horizontal neighbours, for all in_n rows and out_n columns
    for (int i = threadIdx.x; i < in_n * out_n; i += 256) {
      const int y = lo + i / out_n, xo = lo + 2 + i % out_n;
      const V* r = s_a + y * P + xo;
      s_b[y * P + xo] = vmax(vmax(vmax(r[-2], r[-1]), vmax(r[0], r[1])), r[2]);
    }
    __syncthreads();
    // columns, written back to s_a; positions outside the image become -inf again (padding of the next stage)
    for (int i = threadIdx.x; i < out_n * out_n; i += 256) {
      const int yo = lo + 2 + i / out_n, xo = lo + 2 + i % out_n;
      const V* q = s_b + yo * P + xo;
      V m = vmax(vmax(vmax(q[-2 * P], q[-P]), vmax(q[0], q[P])), q[2 * P]);
      const int yy = ty0 - R + yo, xx = tx0 - R + xo;
      const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
      s_a[yo * P + xo] = inside ? m : ninf;
      if (inside && yo >= R && yo < R + TS && xo >= R && xo < R + TS)
        pool_store<T>(img + ((size_t)yy * w + xx) * cs + (stage + 1) * c, v, m);
    }
    __syncthreads();
    lo += 2;
  }
}

// Pair format, maps that fit LDS whole (60 x 60 at 1920 input: 115 KB per 8-channel unit): one 512-thread workgroup per
// (image, 32-byte unit) holds the whole map as keys and runs the six separable passes in place. A thread owns a run of 9
// consecutive positions along the pass direction (of one row or column and one half of the unit): 13 LDS reads, nine
// 5-window maxima as v_max3_u32 pairs, a barrier, 9 writes -- 1.4 reads per result instead of 5, and LDS bandwidth is what
// bounds this kernel once the maxima are integer. Against the tiled kernel above: no halo (a 16 x 16 tile with its 6-pixel
// ring reads 3.06 x its own pixels), 32 contiguous bytes per pixel instead of two 8-byte pieces 16 bytes apart, 128
// workgroups instead of 2 048. Positions outside the map read the edge cell again, which is what -inf padding amounts to.
// (Runs of 9 pixels are 288 bytes apart: the 16 lanes of an LDS access cycle fall on 16 different bank groups.)
constexpr int kPoolRun = 9, kPoolMaxRuns = 2, kPoolThreads = 512;   // two runs per thread and direction: maps up to 64 x 64 (at 1024 threads the runs spill: 128 registers)
__global__ __launch_bounds__(kPoolThreads) void sppf_pool_image_kernel(float* __restrict__ x, int h, int w, int c) {
  extern __shared__ __attribute__((aligned(16))) char s_raw[];
  uint4* s_m = reinterpret_cast<uint4*>(s_raw);            // [pixel][2]: keys of channels 0-3 and 4-7 of the unit
  constexpr int NT = kPoolThreads, RUN = kPoolRun, LD_IT = 8;      // h * w <= NT * LD_IT (launch_sppf_pool checks)
  const int u = blockIdx.x, n = blockIdx.y, cs = 4 * c, npx = h * w;
  char* img = reinterpret_cast<char*>(x + (size_t)n * npx * cs) + (size_t)u * 32;
  const size_t pstride = (size_t)cs * 4;                   // bytes per pixel record [x | y1 | y2 | y3]
  {
    uint4 hi[LD_IT], lo[LD_IT];                            // every load of the map is issued before the first key is made
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
      const int p = min((int)threadIdx.x + NT * it, npx - 1);
      hi[it] = *reinterpret_cast<const uint4*>(img + p * pstride);
      lo[it] = *reinterpret_cast<const uint4*>(img + p * pstride + 16);
    }
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
      const int p = threadIdx.x + NT * it;
      if (p < npx) {
        s_m[2 * p] = pair_keys(make_uint2(hi[it].x, hi[it].y), make_uint2(lo[it].x, lo[it].y));
        s_m[2 * p + 1] = pair_keys(make_uint2(hi[it].z, hi[it].w), make_uint2(lo[it].z, lo[it].w));
      }
    }
  }
  // the thread's runs: [pass direction][run] -> LDS index of the run's first cell and its first position (-1: none)
  int r_base[2][kPoolMaxRuns], r_pos[2][kPoolMaxRuns];
  {
    const int nsx = (w + RUN - 1) / RUN, nsy = (h + RUN - 1) / RUN;
#pragma unroll
    for (int r = 0; r < kPoolMaxRuns; ++r) {
      const int idx = threadIdx.x + NT * r, k = idx & 1, t = idx >> 1;
      const int y = t / nsx, sx = t - y * nsx;              // along rows: lanes walk (half, segment) of one row
      r_pos[0][r] = y < h ? sx * RUN : -1;
      r_base[0][r] = (y * w + sx * RUN) * 2 + k;
      const int sy = t / w, xx = t - sy * w;                // along columns: lanes walk (half, x) of one band of rows
      r_pos[1][r] = sy < nsy ? sy * RUN : -1;
      r_base[1][r] = (sy * RUN * w + xx) * 2 + k;
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int stage = 0; stage < 3; ++stage) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {                 // 0: along rows, 1: along columns
      const int lim = pass == 0 ? w : h, step = pass == 0 ? 2 : 2 * w;
      uint4 out[kPoolMaxRuns][RUN];
#pragma unroll
      for (int r = 0; r < kPoolMaxRuns; ++r) {
        const int pos0 = r_pos[pass][r], base = r_base[pass][r];
        if (pos0 >= 0) {
          uint4 in[RUN + 4];
#pragma unroll
          for (int j = 0; j < RUN + 4; ++j)               // positions outside the map: the edge cell again (it is in every window that reaches them)
            in[j] = s_m[base + (min(max(pos0 - 2 + j, 0), lim - 1) - pos0) * step];
#pragma unroll
          for (int j = 0; j < RUN; ++j) out[r][j] = vmax(vmax(vmax(in[j], in[j + 1]), vmax(in[j + 2], in[j + 3])), in[j + 4]);
        }
        __builtin_amdgcn_sched_barrier(0);                  // one run's 13 inputs at a time (both at once do not fit the registers)
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < kPoolMaxRuns; ++r) {
        const int pos0 = r_pos[pass][r], base = r_base[pass][r];
        if (pos0 >= 0) {
#pragma unroll
          for (int j = 0; j < RUN; ++j)
            if (pos0 + j < lim) s_m[base + j * step] = out[r][j];
        }
      }
      __syncthreads();
    }
    char* dst = img + (size_t)(stage + 1) * c * 4;          // slice y1 / y2 / y3 of the record
    for (int p = threadIdx.x; p < npx; p += NT) {
      uint2 ah, al, bh, bl;
      pair_unkeys(s_m[2 * p], ah, al);
      pair_unkeys(s_m[2 * p + 1], bh, bl);
      *reinterpret_cast<uint4*>(dst + p * pstride) = make_uint4(ah.x, ah.y, bh.x, bh.y);
      *reinterpret_cast<uint4*>(dst + p * pstride + 16) = make_uint4(al.x, al.y, bl.x, bl.y);
    }
  }
}

void launch_sppf_pool(int dtype, void* x, int n, int h, int w, int c, hipStream_t s) {
  const int vn = dtype == DT_F16 ? 8 : 4;
  GTX_CHECK(c % (dtype == DT_F32 ? 4 : 8) == 0, "sppf: channels %d not a multiple of %d", c, dtype == DT_F32 ? 4 : 8);
  const bool runs_fit = h * 2 * cdiv(w, kPoolRun) <= kPoolThreads * kPoolMaxRuns && w * 2 * cdiv(h, kPoolRun) <= kPoolThreads * kPoolMaxRuns;
  if (dtype == DT_F32S && runs_fit && h * w <= 8 * kPoolThreads && (size_t)h * w * 32 <= 144 * 1024 && c % 8 == 0) {      // the whole map in LDS (<= 64 x 64)
    static const bool tiled = [] { const char* e = getenv("GTX_POOL_TILED"); return e && e[0] == '1'; }();
    if (!tiled) {
      const int lds = h * w * 32;
      static std::once_flag once;
      std::call_once(once, [] {
        GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool_image_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      });
      hipLaunchKernelGGL(sppf_pool_image_kernel, dim3((unsigned)(c / 8), n), dim3(kPoolThreads), lds, s, (float*)x, h, w, c);
      GTX_HIP(hipGetLastError());
      return;
    }
  }
  const int tiles = cdiv(w, 16) * cdiv(h, 16);
  dim3 grid((unsigned)(tiles * (c / vn)), n), block(256);
  if (dtype == DT_F16) hipLaunchKernelGGL(sppf_pool_kernel<_Float16>, grid, block, 0, s, (_Float16*)x, h, w, c);
  else if (dtype == DT_F32S) hipLaunchKernelGGL(sppf_pool_kernel<PairTag>, grid, block, 0, s, (float*)x, h, w, c);   // pair format
  else hipLaunchKernelGGL(sppf_pool_kernel<float>, grid, block, 0, s, (float*)x, h, w, c);
  GTX_HIP(hipGetLastError());
}

// ============================================================================ upsample
__global__ __launch_bounds__(256) void upsample2x_kernel(const uint4* __restrict__ x, int h, int w,
                                                         int vecs, int in_vstride, int in_voff,
                                                         uint4* __restrict__ y, int out_vstride,
                                                         int out_voff) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  const int ho = 2 * h, wo = 2 * w;
  if (gid >= (long)ho * wo * vecs) return;
  const int v = (int)(gid % vecs);
  const int pix = (int)(gid / vecs);
  const int oy = pix / wo, ox = pix % wo;
  const uint4 val = x[((size_t)(n * h + (oy >> 1)) * w + (ox >> 1)) * in_vstride + in_voff + v];
  y[((size_t)(n * ho + oy) * wo + ox) * out_vstride + out_voff + v] = val;
}

void launch_upsample2x(int dtype, const void* x, int n, int h, int w, int c, int in_cstride,
                       int in_coff, void* y, int out_cstride, int out_coff, hipStream_t s) {
  const int vn = dtype == DT_F16 ? 8 : 4;
  GTX_CHECK(c % vn == 0 && in_cstride % vn == 0 && in_coff % vn == 0 && out_cstride % vn == 0 &&
                out_coff % vn == 0, "upsample: channel layout must be 16-byte aligned");
  const long work = (long)4 * h * w * (c / vn);
  dim3 grid((unsigned)((work + 255) / 256), n), block(256);
  hipLaunchKernelGGL(upsample2x_kernel, grid, block, 0, s, (const uint4*)x, h, w, c / vn,
                     in_cstride / vn, in_coff / vn, (uint4*)y, out_cstride / vn, out_coff / vn);
  GTX_HIP(hipGetLastError());
}

// ============================================================================ head decode
template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

__device__ __forceinline__ int find_level(const HeadParams& hp, int a) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxLevels; ++i)
    if (i < hp.n_levels && a >= hp.lv[i].anchor_begin) l = i;
  return l;
}

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<_Float16>(const _Float16* p, float (&v)[8]) {
  const half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Class branch of Detect folded into the score gate: final 1x1 conv (nc x cc) + sigmoid + max
// over classes + threshold + compaction. 16 lanes share an anchor, each owning 8-channel chunks of
// its cls features (one fully coalesced 16-B load per lane per 128 channels), partial dot products
// are combined with 4 xor-shuffles.
template <typename T>
__global__ __launch_bounds__(256) void head_candidates_kernel(const HeadParams hp, const NmsBuffers nb) {
  const int sub = threadIdx.x & 15;
  const int a = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int n = blockIdx.y;
  const bool valid = a < hp.n_anchors;
  const int l = find_level(hp, valid ? a : 0);
  const HeadLevel& L = hp.lv[l];
  const int la = (valid ? a : 0) - L.anchor_begin;
  const T* fc = static_cast<const T*>(L.feat) + ((size_t)n * L.h * L.w + la) * L.cstride + L.cb;
  const int chunks = L.cc >> 3;
  float best = -1.f;
  int best_c = 0;
  for (int c0 = 0; c0 < hp.nc; c0 += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = sub; ch < chunks; ch += 16) {
      float f[8];
      load8<T>(fc + ch * 8, f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c0 + j < hp.nc) {
          float w[8];
          load8<float>(L.wc + (size_t)(c0 + j) * L.cc + ch * 8, w);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j] = fmaf(f[e], w[e], acc[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
      if (c0 + j < hp.nc) {
        const float sc = 1.f / (1.f + expf(-(acc[j] + L.bc[c0 + j])));
        if (sc > best) { best = sc; best_c = c0 + j; }
      }
    }
  }
  if (valid && sub == 0 && best > hp.conf && ((hp.class_mask[best_c >> 6] >> (best_c & 63)) & 1ull)) {
    const int idx = atomicAdd(&nb.count[n], 1);
    if (idx < nb.cap) {
      const size_t o = (size_t)n * nb.cap + idx;
      nb.cand_score[o] = best;
      nb.cand_anchor[o] = a;
      nb.cand_cls[o] = best_c;
      if (nb.lvl_count && idx < nb.lvl_cap) {      // filed under its level for the sparse box branch (head_sparse.hip)
        const int k = atomicAdd(&nb.lvl_count[n * kMaxLevels + l], 1);
        nb.lvl_list[((size_t)n * kMaxLevels + l) * nb.lvl_cap + k] = idx;   // k <= idx < lvl_cap
      }
    }
  }
}

// DFL tail shared by the decode kernels: lane = side*16 + bin holds one box logit; softmax
// expectation per side, dist2bbox(xywh) * stride. Returns xywh in network pixels on every lane.
__device__ __forceinline__ float4 dfl_box(const HeadLevel& L, float acc, int la, int lane) {
  // softmax over the 16 lanes of a side
  float m = acc;
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const float e = expf(acc - m);
  float se = e, sw = e * (float)(lane & 15);
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) {
    se += __shfl_xor(se, o, 64);
    sw += __shfl_xor(sw, o, 64);
  }
  const float d = sw / se;
  const float d0 = __shfl(d, 0, 64), d1 = __shfl(d, 16, 64), d2 = __shfl(d, 32, 64), d3 = __shfl(d, 48, 64);
  const float ax = (float)(la % L.w) + 0.5f, ay = (float)(la / L.w) + 0.5f;
  const float x1 = ax - d0, y1 = ay - d1, x2 = ax + d2, y2 = ay + d3;
  return make_float4((x1 + x2) * 0.5f * L.stride, (y1 + y2) * 0.5f * L.stride, (x2 - x1) * L.stride,
                     (y2 - y1) * L.stride);
}

// Box branch for one anchor by one wave: 64 box logits (lane = side*16 + bin), DFL softmax
// expectation per side, dist2bbox(xywh) * stride. Returns xywh in network pixels on every lane.
template <typename T>
__device__ __forceinline__ float4 anchor_box(const HeadLevel& L, const T* f, int la, int lane) {
  // lane k holds box feature k (and k+64); the 64x cb mat-vec then takes each feature by shuffle and
  // one coalesced 256-B line of the transposed weights [cb][64] per feature.
  const float f0 = lane < L.cb ? ldf(f + lane) : 0.f;
  const float f1 = lane + 64 < L.cb ? ldf(f + lane + 64) : 0.f;
  float acc = L.bb[lane];
  const float* wr = L.wb + lane;
  const int k0 = min(L.cb, 64);
  for (int k = 0; k < k0; ++k) acc = fmaf(__shfl(f0, k, 64), wr[(size_t)k * 64], acc);
  for (int k = 64; k < L.cb; ++k) acc = fmaf(__shfl(f1, k - 64, 64), wr[(size_t)k * 64], acc);
  return dfl_box(L, acc, la, lane);
}

// One wave per candidate; the transposed box weights of every level ([cb][64] each) are staged in
// LDS once per block, so the 64 x cb mat-vec reads one conflict-free LDS row per feature.
template <typename T>
__global__ __launch_bounds__(256) void head_boxes_kernel(const HeadParams hp, const NmsBuffers nb) {
  extern __shared__ __attribute__((aligned(16))) float s_wb[];   // [levels][cb][64]
  const int n = blockIdx.y;
  const int cnt = min(nb.count[n], nb.cap);
  const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  if ((int)(blockIdx.x * (blockDim.x >> 6)) >= cnt) return;      // no candidate for this block
  const int cbmax = hp.lv[0].cb;
  for (int l = 0; l < hp.n_levels; ++l)
    for (int i = threadIdx.x; i < hp.lv[l].cb * 64; i += blockDim.x) s_wb[l * cbmax * 64 + i] = hp.lv[l].wb[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int i = wave_g; i < cnt; i += nwaves) {
    const size_t o = (size_t)n * nb.cap + i;
    const int a = nb.cand_anchor[o];
    const int l = find_level(hp, a);
    const HeadLevel& L = hp.lv[l];
    const int la = a - L.anchor_begin;
    const T* f = static_cast<const T*>(L.feat) + ((size_t)n * L.h * L.w + la) * L.cstride;
    const float f0 = lane < L.cb ? ldf(f + lane) : 0.f;
    const float f1 = lane + 64 < L.cb ? ldf(f + lane + 64) : 0.f;
    float acc = L.bb[lane];
    const float* wr = s_wb + l * cbmax * 64 + lane;
    const int k0 = min(L.cb, 64);
    for (int k = 0; k < k0; ++k) acc = fmaf(__shfl(f0, k, 64), wr[k * 64], acc);
    for (int k = 64; k < L.cb; ++k) acc = fmaf(__shfl(f1, k - 64, 64), wr[k * 64], acc);
    const float4 b = dfl_box(L, acc, la, lane);
    if (lane == 0) {
      const float hw = b.z / 2.f, hh = b.w / 2.f;   // xywh2xyxy
      reinterpret_cast<float4*>(nb.cand_box)[o] = make_float4(b.x - hw, b.y - hh, b.x + hw, b.y + hh);
    }
  }
}

// Debug / parity: full decode of every anchor -> [A][4+nc].
template <typename T>
__global__ __launch_bounds__(256) void head_raw_kernel(const HeadParams hp, float* __restrict__ out, int logits) {
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int a = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (a >= hp.n_anchors) return;
  const int l = find_level(hp, a);
  const HeadLevel& L = hp.lv[l];
  const int la = a - L.anchor_begin;
  const T* f = static_cast<const T*>(L.feat) + ((size_t)n * L.h * L.w + la) * L.cstride;
  const float4 b = anchor_box<T>(L, f, la, lane);
  float* o = out + ((size_t)n * hp.n_anchors + a) * (4 + hp.nc);
  if (lane == 0) { o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; }
  const T* fc = f + L.cb;
  for (int c = lane; c < hp.nc; c += 64) {
    float acc = L.bc[c];
    for (int k = 0; k < L.cc; ++k) acc = fmaf(ldf(fc + k), L.wc[c * L.cc + k], acc);
    o[4 + c] = logits ? acc : 1.f / (1.f + expf(-acc));
  }
}

void launch_head_candidates(int dtype, const HeadParams& hp, int n, const NmsBuffers& nb, hipStream_t s) {
  launch_head_gate(dtype, hp, n, nb, s);
  launch_head_boxes(dtype, hp, n, nb, s);
}

void launch_head_gate(int dtype, const HeadParams& hp, int n, const NmsBuffers& nb, hipStream_t s) {
  GTX_HIP(hipMemsetAsync(nb.count, 0, sizeof(int) * n, s));
  if (nb.lvl_count) GTX_HIP(hipMemsetAsync(nb.lvl_count, 0, sizeof(int) * n * kMaxLevels, s));
  dim3 grid(cdiv(hp.n_anchors, 16), n), block(256);
  if (dtype == DT_F16) hipLaunchKernelGGL(head_candidates_kernel<_Float16>, grid, block, 0, s, hp, nb);
  else hipLaunchKernelGGL(head_candidates_kernel<float>, grid, block, 0, s, hp, nb);
  GTX_HIP(hipGetLastError());
}

void launch_head_boxes(int dtype, const HeadParams& hp, int n, const NmsBuffers& nb, hipStream_t s) {
  dim3 grid2(64, n), block(256);
  const size_t lds = (size_t)hp.n_levels * hp.lv[0].cb * 64 * sizeof(float);
  static std::once_flag once;     // detectors run on several host threads (engine stage 1, set_reference)
  std::call_once(once, [] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_boxes_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * 64 * 4));
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_boxes_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * 64 * 4));
  });
  if (dtype == DT_F16) hipLaunchKernelGGL(head_boxes_kernel<_Float16>, grid2, block, lds, s, hp, nb);
  else hipLaunchKernelGGL(head_boxes_kernel<float>, grid2, block, lds, s, hp, nb);
  GTX_HIP(hipGetLastError());
}

void launch_head_raw(int dtype, const HeadParams& hp, int n, float* out, bool logits, hipStream_t s) {
  dim3 grid(cdiv(hp.n_anchors * 64, 256), n), block(256);
  if (dtype == DT_F16) hipLaunchKernelGGL(head_raw_kernel<_Float16>, grid, block, 0, s, hp, out, logits ? 1 : 0);
  else hipLaunchKernelGGL(head_raw_kernel<float>, grid, block, 0, s, hp, out, logits ? 1 : 0);
  GTX_HIP(hipGetLastError());
}

// ============================================================================ NMS
constexpr int kSmallNms = 4096;    // candidates the single-workgroup path handles
constexpr int kSmallKeep = 2048;   // max_det it handles

// (1) rank: position of each candidate in (score desc, anchor asc) order = stable descending
//     sort of the anchor-ordered candidate list, which is what torchvision.ops.nms applies to
//     the boolean-mask-filtered predictions. O(n^2) counting, keys tiled through LDS.
__global__ __launch_bounds__(256) void nms_rank_kernel(const NmsBuffers nb, int limit, float cls_offset) {
  __shared__ float s_sc[256];
  __shared__ int s_an[256];
  const int n = blockIdx.y;
  const int cnt = min(nb.count[n], nb.cap);
  if (cnt <= kSmallNms && nb.max_det <= kSmallKeep) return;   // nms_small_kernel did the image (sorted_n = 0)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) nb.sorted_n[n] = min(cnt, limit);
  if (blockIdx.x * blockDim.x >= cnt) return;  // whole block idle
  const size_t base = (size_t)n * nb.cap;
  const bool act = i < cnt;
  const float si = act ? nb.cand_score[base + i] : 0.f;
  const int ai = act ? nb.cand_anchor[base + i] : 0;
  int rank = 0;
  for (int t0 = 0; t0 < cnt; t0 += 256) {
    const int j = t0 + threadIdx.x;
    s_sc[threadIdx.x] = j < cnt ? nb.cand_score[base + j] : -1.f;
    s_an[threadIdx.x] = j < cnt ? nb.cand_anchor[base + j] : 0x7fffffff;
    __syncthreads();
    const int lim = min(256, cnt - t0);
    for (int k = 0; k < lim; ++k) {
      const float sj = s_sc[k];
      rank += (sj > si || (sj == si && s_an[k] < ai)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (act && rank < limit) {
    const size_t so = (size_t)n * nb.nms_cap + rank;
    const float4 b = reinterpret_cast<const float4*>(nb.cand_box)[base + i];
    const int c = nb.cand_cls[base + i];
    reinterpret_cast<float4*>(nb.s_box)[so] = b;
    nb.s_score[so] = si;
    nb.s_cls[so] = c;
    if (nb.s_anchor) nb.s_anchor[so] = ai;
    (void)cls_offset;
  }
}

// (2) suppression bit matrix, full and symmetric, stored word-major: bit (j&63) of
//     mask[j/64][i] = IoU(i,j) > thr, j != i.
//     One wave per 64x64 tile; the 64 column boxes sit in LDS, each lane owns one row.
//     IoU arithmetic is torchvision's fp32 sequence: inter / (area_i + area_j - inter).
__global__ __launch_bounds__(256) void nms_mask_kernel(const NmsBuffers nb, float thr, float cls_offset) {
  __shared__ float4 s_box[4][64];
  const int n = blockIdx.y;
  const int cnt = nb.sorted_n[n];
  const int nblk = (cnt + 63) >> 6;
  const int rs = nb.nms_cap >> 6;
  const float4* boxes = reinterpret_cast<const float4*>(nb.s_box) + (size_t)n * nb.nms_cap;
  const int* cls = nb.s_cls + (size_t)n * nb.nms_cap;
  unsigned long long* mask = nb.mask + (size_t)n * nb.nms_cap * rs;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long ntiles = (long)nblk * nblk;
  for (long t = (long)blockIdx.x * 4 + wv; t < ntiles; t += (long)gridDim.x * 4) {
    const int bi = (int)(t / nblk), bj = (int)(t % nblk);
    const int i = bi * 64 + lane, jc = bj * 64 + lane;
    float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (jc < cnt) {
      bc = boxes[jc];
      const float o = cls_offset * (float)cls[jc];
      bc.x += o; bc.y += o; bc.z += o; bc.w += o;
    }
    s_box[wv][lane] = bc;   // wave-private row of LDS: no workgroup barrier needed
    float4 br = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < cnt) {
      br = boxes[i];
      const float o = cls_offset * (float)cls[i];
      br.x += o; br.y += o; br.z += o; br.w += o;
    }
    const float ai = (br.z - br.x) * (br.w - br.y);
    unsigned long long bits = 0ull;
    const int jend = min(64, cnt - bj * 64);
    for (int k = 0; k < jend; ++k) {
      const float4 q = s_box[wv][k];
      const float aj = (q.z - q.x) * (q.w - q.y);
      const float iw = fmaxf(0.f, fminf(br.z, q.z) - fmaxf(br.x, q.x));
      const float ih = fmaxf(0.f, fminf(br.w, q.w) - fmaxf(br.y, q.y));
      const float inter = iw * ih;
      const float ovr = inter / (ai + aj - inter);
      if (ovr > thr && (bj * 64 + k) != i) bits |= 1ull << k;
    }
    if (i < cnt) mask[(size_t)bj * nb.nms_cap + i] = bits;   // [word][box]: coalesced here and in the resolve
  }
}

// (3) greedy NMS (torchvision.ops.nms order) as a wave pipeline, one workgroup per image.
//     Boxes are ranked; block b = ranks [64b, 64b+64). A box is removed iff a kept higher-ranked
//     box overlaps it. Wave k owns blocks k, k+16, ...: each lane streams its box's mask words
//     over the already decided blocks (AND with their keep words, waiting on an in-order
//     "blocks done" counter only when it catches up with the block in front), then the 64 boxes
//     of the block are settled among themselves by a monotone fixpoint on wave ballots (removed
//     as soon as a kept higher-ranked overlap exists, kept as soon as all higher-ranked overlaps
//     are removed; rounds = longest suppression chain inside the block). The serial chain per
//     block is a few hundred cycles, independent of how many candidates overlap.
// (4) kept boxes are compacted in rank order, capped at max_det, and mapped back to frame
//     pixels (ultralytics scale_boxes + clip_boxes).
constexpr int kNmsWords = 512;  // >= nms_cap / 64
__global__ __launch_bounds__(1024) void nms_resolve_kernel(const NmsBuffers nb, float gain, float padx,
                                                           float pady, float fw, float fh) {
  __shared__ unsigned long long s_keep[kNmsWords];
  __shared__ int s_prefix[kNmsWords];
  __shared__ int s_done;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  if (min(nb.count[n], nb.cap) <= kSmallNms && nb.max_det <= kSmallKeep) return;   // nms_small_kernel did the image
  const int cnt = nb.sorted_n[n];
  const int nblk = (cnt + 63) >> 6;
  const unsigned long long* mask = nb.mask + (size_t)n * nb.nms_cap * (nb.nms_cap >> 6);
  if (tid == 0) s_done = 0;
  __syncthreads();
  volatile int* done = &s_done;
  volatile unsigned long long* keep = s_keep;
  for (int b = wave; b < nblk; b += nwaves) {
    const int i = b * 64 + lane;
    const bool valid = i < cnt;
    bool rem = false;
    int ready = *done;
    for (int w0 = 0; w0 < b; w0 += 8) {
      unsigned long long m[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = (valid && w0 + k < b) ? mask[(size_t)(w0 + k) * nb.nms_cap + i] : 0ull;
      const int need = min(w0 + 8, b);
      while (ready < need) { __builtin_amdgcn_s_sleep(1); ready = *done; }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w0 + k < b && (m[k] & keep[w0 + k])) rem = true;
    }
    const unsigned long long low = (1ull << lane) - 1ull;
    const unsigned long long d = valid ? (mask[(size_t)b * nb.nms_cap + i] & low) : 0ull;
    unsigned long long removed = __ballot(rem || !valid), kept = 0ull;
    const unsigned long long me = 1ull << lane;
    for (int round = 0; round < 64; ++round) {
      const unsigned long long und = ~(kept | removed);
      if (und == 0ull) break;
      bool nk = false, nr = false;
      if (und & me) {
        if (d & kept) nr = true;
        else if ((d & ~removed) == 0ull) nk = true;
      }
      kept |= __ballot(nk);
      removed |= __ballot(nr);
    }
    if (lane == 0) {
      keep[b] = kept;
      __threadfence_block();
      *done = b + 1;
    }
  }
  __syncthreads();
  // compaction in rank order
  if (tid == 0) {
    int acc = 0;
    for (int w = 0; w < nblk; ++w) { s_prefix[w] = acc; acc += __popcll(s_keep[w]); }
    nb.out_n[n] = min(acc, nb.max_det);
  }
  __syncthreads();
  const float4* boxes = reinterpret_cast<const float4*>(nb.s_box) + (size_t)n * nb.nms_cap;
  float* rows = nb.out_rows + (size_t)n * nb.max_det * 6;
  for (int i = tid; i < cnt; i += blockDim.x) {
    const int wl = i >> 6;
    const unsigned long long bit = 1ull << (i & 63);
    if (!(s_keep[wl] & bit)) continue;
    const int slot = s_prefix[wl] + __popcll(s_keep[wl] & (bit - 1ull));
    if (slot >= nb.max_det) continue;
    float4 b = boxes[i];
    b.x = (b.x - padx) / gain; b.y = (b.y - pady) / gain;
    b.z = (b.z - padx) / gain; b.w = (b.w - pady) / gain;
    b.x = fminf(fmaxf(b.x, 0.f), fw); b.z = fminf(fmaxf(b.z, 0.f), fw);
    b.y = fminf(fmaxf(b.y, 0.f), fh); b.w = fminf(fmaxf(b.w, 0.f), fh);
    float* r = rows + (size_t)slot * 6;
    r[0] = b.x; r[1] = b.y; r[2] = b.z; r[3] = b.w;
    r[4] = nb.s_score[(size_t)n * nb.nms_cap + i];
    r[5] = (float)nb.s_cls[(size_t)n * nb.nms_cap + i];
    if (nb.out_anchor) nb.out_anchor[(size_t)n * nb.max_det + slot] = nb.s_anchor[(size_t)n * nb.nms_cap + i];
  }
}

// ---- small-n path: everything after the score gate in ONE workgroup per image -----------------
// For n <= 4096 candidates (traffic scenes: a few hundred to ~2000): (1) stable rank by a bitonic
// sort of 64-bit keys {score bits | inverted anchor | slot} in LDS, (2) boxes gathered into LDS in
// rank order, (3) greedy NMS as the same in-order wave pipeline as nms_resolve_kernel, but with IoUs
// computed on the fly against the list of kept boxes (also in LDS) instead of an n x n bit matrix,
// (4) kept boxes written straight to their output slot. Larger n falls through to the general
// rank / mask / resolve kernels (which exit immediately when this path applies).

__device__ __forceinline__ bool iou_gt(const float4& a, float area_a, const float4& b, float thr) {
  const float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x);
  const float ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
  if (!(iw > 0.f && ih > 0.f)) return false;        // disjoint: IoU is 0 (or NaN), never > thr
  const float area_b = (b.z - b.x) * (b.w - b.y);
  const float inter = iw * ih;
  return inter / (area_a + area_b - inter) > thr;   // torchvision's fp32 sequence
}

__global__ __launch_bounds__(1024) void nms_small_kernel(const NmsBuffers nb, float thr, float cls_offset, float gain, float padx,
                                                         float pady, float fw, float fh) {
  __shared__ unsigned long long s_key[kSmallNms];
  __shared__ float4 s_box[kSmallNms];
  __shared__ float4 s_kbox[kSmallKeep];
  __shared__ int s_nkeep, s_done;
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  const int cnt = min(nb.count[n], nb.cap);
  if (cnt > kSmallNms || nb.max_det > kSmallKeep) {         // the general kernels handle it (Detector::collect runs them for such a batch)
    if (tid == 0) nb.out_n[n] = 0;                          // until then the image has no rows: nothing of the previous pass is read as this one's
    return;
  }
  if (tid == 0) { nb.sorted_n[n] = 0; s_nkeep = 0; s_done = 0; }
  const size_t base = (size_t)n * nb.cap;
  int P = 64;
  while (P < cnt) P <<= 1;
  for (int i = tid; i < P; i += blockDim.x) {
    unsigned long long k = 0ull;
    if (i < cnt) {
      const unsigned sb = __float_as_uint(nb.cand_score[base + i]);            // scores are positive: bits are monotone
      k = ((unsigned long long)sb << 32) | ((unsigned long long)(0x7FFFFu - (unsigned)nb.cand_anchor[base + i]) << 12) | (unsigned)i;
    }
    s_key[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < P / 2; t += blockDim.x) {
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool desc = ((lo & k) == 0);
        const unsigned long long a = s_key[lo], b = s_key[hi];
        if ((a < b) == desc) { s_key[lo] = b; s_key[hi] = a; }
      }
      __syncthreads();
    }
  for (int i = tid; i < cnt; i += blockDim.x) {
    const int slot = (int)(s_key[i] & 0xFFFu);
    float4 b = reinterpret_cast<const float4*>(nb.cand_box)[base + slot];
    const float o = cls_offset * (float)nb.cand_cls[base + slot];
    b.x += o; b.y += o; b.z += o; b.w += o;
    s_box[i] = b;
  }
  __syncthreads();
  volatile int* done = &s_done;
  volatile int* nkeep = &s_nkeep;
  const int nblk = (cnt + 63) >> 6;
  float* rows = nb.out_rows + (size_t)n * nb.max_det * 6;
  for (int b = wave; b < nblk; b += nwaves) {
    const int i = b * 64 + lane;
    const bool valid = i < cnt;
    const float4 bi = valid ? s_box[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float ai = (bi.z - bi.x) * (bi.w - bi.y);
    // overlaps inside the block (does not depend on earlier blocks: overlaps with their resolution)
    unsigned long long d = 0ull;
    const int jn = min(64, cnt - b * 64);
    for (int j = 0; j < jn; ++j)
      if (j < lane && valid && iou_gt(bi, ai, s_box[b * 64 + j], thr)) d |= 1ull << j;
    // overlaps with boxes kept by earlier blocks, consumed as they are published
    bool rem = false;
    int checked = 0;
    for (;;) {
      const int dn = *done;
      const int nk = *nkeep;
      for (int j = checked; j < nk; ++j)
        if (!rem && valid && iou_gt(bi, ai, s_kbox[j], thr)) rem = true;
      checked = nk;
      if (dn >= b) break;
      __builtin_amdgcn_s_sleep(1);
    }
    unsigned long long removed = __ballot(rem || !valid), kept = 0ull;
    const unsigned long long me = 1ull << lane;
    for (int round = 0; round < 64; ++round) {
      const unsigned long long und = ~(kept | removed);
      if (und == 0ull) break;
      bool nk2 = false, nr = false;
      if (und & me) {
        if (d & kept) nr = true;
        else if ((d & ~removed) == 0ull) nk2 = true;
      }
      kept |= __ballot(nk2);
      removed |= __ballot(nr);
    }
    const int base_keep = checked;                 // == number kept by all earlier blocks
    const int my_slot = base_keep + __popcll(kept & (me - 1ull));
    if ((kept & me) && my_slot < nb.max_det) {
      s_kbox[my_slot] = bi;
      const int slot = (int)(s_key[i] & 0xFFFu);
      float4 o = reinterpret_cast<const float4*>(nb.cand_box)[base + slot];
      o.x = (o.x - padx) / gain; o.y = (o.y - pady) / gain;
      o.z = (o.z - padx) / gain; o.w = (o.w - pady) / gain;
      o.x = fminf(fmaxf(o.x, 0.f), fw); o.z = fminf(fmaxf(o.z, 0.f), fw);
      o.y = fminf(fmaxf(o.y, 0.f), fh); o.w = fminf(fmaxf(o.w, 0.f), fh);
      float* r = rows + (size_t)my_slot * 6;
      r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = o.w;
      r[4] = nb.cand_score[base + slot];
      r[5] = (float)nb.cand_cls[base + slot];
      if (nb.out_anchor) nb.out_anchor[(size_t)n * nb.max_det + my_slot] = nb.cand_anchor[base + slot];
    }
    __threadfence_block();
    if (lane == 0) {
      const int total = min(base_keep + __popcll(kept), nb.max_det);
      *nkeep = total;
      __threadfence_block();
      *done = (total >= nb.max_det) ? nblk : b + 1;   // nothing beyond max_det can be output
    }
    if (*nkeep >= nb.max_det) break;
  }
  __syncthreads();
  if (tid == 0) nb.out_n[n] = s_nkeep;
}

// One workgroup of `dim` threads per output row: thread k averages the c / dim consecutive channels of group k at the row's anchor
__global__ void obj_feats_kernel(const FeatLevels fl, int dtype, const NmsBuffers nb, float* __restrict__ out) {
  const int n = blockIdx.y, slot = blockIdx.x;
  if (slot >= min(nb.out_n[n], nb.max_det)) return;
  const int a = nb.out_anchor[(size_t)n * nb.max_det + slot];
  int l = 0;
#pragma unroll
  for (int i = 1; i < kMaxLevels; ++i)
    if (i < fl.n_levels && a >= fl.anchor_begin[i]) l = i;
  const int la = a - fl.anchor_begin[l];
  const int g = fl.c[l] / fl.dim;
  const size_t e0 = ((size_t)n * fl.h[l] * fl.w[l] + la) * fl.cstride[l] + fl.coff[l];
  for (int k = threadIdx.x; k < fl.dim; k += blockDim.x) {
    float sum = 0.f;
    for (int j = 0; j < g; ++j) {
      const size_t e = e0 + (size_t)k * g + j;
      float v;
      if (dtype == DT_F16) {
        v = (float)static_cast<const _Float16*>(fl.feat[l])[e];
      } else if (dtype == DT_F32S) {                     // pair format (split_format.hpp): hi + lo is exact in fp32
        const char* b = static_cast<const char*>(fl.feat[l]) + (e & ~(size_t)7) * 4;
        v = (float)*reinterpret_cast<const _Float16*>(b + 2 * (e & 7)) + (float)*reinterpret_cast<const _Float16*>(b + 16 + 2 * (e & 7));
      } else {
        v = static_cast<const float*>(fl.feat[l])[e];
      }
      sum += v;
    }
    out[((size_t)n * nb.max_det + slot) * fl.dim + k] = sum / (float)g;
  }
}

void launch_obj_feats(int dtype, const FeatLevels& fl, int n, const NmsBuffers& nb, float* out, hipStream_t s) {
  GTX_CHECK(nb.out_anchor != nullptr && fl.dim > 0, "obj_feats: the NMS buffers keep no anchors");
  for (int l = 0; l < fl.n_levels; ++l) GTX_CHECK(fl.c[l] % fl.dim == 0, "obj_feats: level %d has %d channels, not a multiple of %d", l, fl.c[l], fl.dim);
  hipLaunchKernelGGL(obj_feats_kernel, dim3(nb.max_det, n), dim3(128), 0, s, fl, dtype, nb, out);
  GTX_HIP(hipGetLastError());
}

// which == 0: everything (the single-workgroup kernel and the general rank / mask / resolve kernels, each of which leaves at
// once when the other path has the image); 1: the single-workgroup kernel only; 2: the general kernels only. A caller that
// knows the candidate counts (Detector: copied back with the results) launches 1 in the pass and 2 afterwards for a batch
// that needs it (nms_small_covers): three launches less in every pass's dependent tail.
void launch_nms(const NmsBuffers& nb, int n, float iou_thr, bool agnostic, int max_nms,
                const Letterbox& lb, hipStream_t s, int which) {
  GTX_CHECK(nb.nms_cap <= kNmsWords * 64, "nms: capacity %d too large", nb.nms_cap);
  const int limit = std::min(max_nms, nb.nms_cap);
  const float cls_offset = agnostic ? 0.f : 7680.f;  // ultralytics max_wh
  // ultralytics scale_boxes: gain = min ratio, pad = round((net - src*gain)/2 - 0.1)
  const double gain = lb.gain;
  const float padx = (float)std::nearbyint((lb.net_w - lb.src_w * gain) / 2 - 0.1);
  const float pady = (float)std::nearbyint((lb.net_h - lb.src_h * gain) / 2 - 0.1);
  if (which != 2) {
    hipLaunchKernelGGL(nms_small_kernel, dim3(n), dim3(1024), 0, s, nb, iou_thr, cls_offset, (float)gain, padx, pady,
                       (float)lb.src_w, (float)lb.src_h);
    GTX_HIP(hipGetLastError());
  }
  if (which == 1) return;
  hipLaunchKernelGGL(nms_rank_kernel, dim3(cdiv(nb.cap, 256), n), dim3(256), 0, s, nb, limit, cls_offset);
  GTX_HIP(hipGetLastError());
  hipLaunchKernelGGL(nms_mask_kernel, dim3(256, n), dim3(256), 0, s, nb, iou_thr, cls_offset);
  GTX_HIP(hipGetLastError());
  hipLaunchKernelGGL(nms_resolve_kernel, dim3(n), dim3(1024), 0, s, nb, (float)gain, padx, pady,
                     (float)lb.src_w, (float)lb.src_h);
  GTX_HIP(hipGetLastError());
}

bool nms_small_covers(int candidates, int max_det) { return candidates <= kSmallNms && max_det <= kSmallKeep; }

}  // namespace gtx
