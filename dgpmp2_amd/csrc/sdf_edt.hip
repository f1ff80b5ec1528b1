// sdf_edt.hip -- dgp_sdf_2d: signed Euclidean distance fields of a batch of occupancy images, on the GPU (gfx950 / CDNA4).
//
// What it replaces: diff_gpmp2/utils/sdf_utils.py:6-21 (sdf_2d), i.e. two calls of scipy.ndimage.distance_transform_edt per image
// on the host -- 8 ms for a 256 x 256 image, 34 ms for 512 x 512 -- which is what produces the grids the obstacle factor reads
// (examples/diff_gpmp2_2d_step_example.py:39, datasets/generate_2d_dataset.py:211).  Here a batch of images becomes a batch of
// (B, H + 2 pad, W + 2 pad) fields in two launches, bit-identical to scipy's result (the squared distances are integers; the
// square root of an integer in fp64 is correctly rounded on both sides, and the sign / resolution are applied in sdf_2d's order).
//
// Exact two-pass transform (separable in the squared distance):
//   1. edt_columns: one lane per image column walks its column down and up: for every pixel the vertical distance to the
//      nearest OBSTACLE pixel and to the nearest FREE pixel of that column (0xFFFF: none), packed into one 32-bit word per pixel.
//      Adjacent lanes own adjacent columns, so every row step is one coalesced line; the image is read once.
//   2. edt_rows: one workgroup per image row; the row's words are staged in LDS and every lane resolves its pixels by an outward
//      search  D^2 = min_x' (x - x')^2 + g(x', y)^2  that stops as soon as (x - x')^2 alone reaches the best candidate -- the search
//      radius is the answer itself, a handful of pixels next to an obstacle.  A free pixel looks for the nearest obstacle and gets
//      +sqrt(D^2) res, an obstacle pixel for the nearest free pixel and gets -sqrt(D^2) res: of im_dist - inv_im_dist one term is 0.
// An image without any obstacle (or, unpadded, without any free pixel) has no nearest pixel of the other kind; scipy then measures
// from the pixel at (row -1, column 0), and so does this kernel (tests/test_sdf_edt.py pins that).
//
// HBM traffic per padded pixel: image in (once) + 3 x 4 bytes of the packed words (written, updated in the upward walk, read by
// the row pass) + the field out; no arithmetic to speak of -- an HBM-bound byte kernel, not MFMA work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dgp_host.h"

namespace {

using dgp_host::fail;

constexpr uint32_t kNone = 0xFFFFu;            // "no such pixel in this column"
constexpr int kMaxDim = 8192;                  // padded rows / columns: squared distances stay below 2^27 + 2^27, and the row pass keeps two rows of them in 64 KB of LDS

struct EdtArgs {
  const void* image;      // (B, rows, cols), contiguous
  void* out;              // (B, rows + 2 pad, cols + 2 pad)
  uint32_t* words;        // workspace: (B, Hp, Wp) packed column distances
  uint32_t* flags;        // workspace: per image, bit 0: has an obstacle pixel, bit 1: has a free pixel
  int32_t B, rows, cols, pad, Hp, Wp;
  double res;
};

// free space? (sdf_utils.py:13: image > 0.75; the padding is free space, :15)
template <typename T>
__device__ __forceinline__ bool is_free(const EdtArgs& a, int b, int y, int x) {
  const int yi = y - a.pad, xi = x - a.pad;
  if (yi < 0 || yi >= a.rows || xi < 0 || xi >= a.cols) return true;
  const T v = ((const T*)a.image)[((int64_t)b * a.rows + yi) * a.cols + xi];
  return (double)v > 0.75;
}

// word: low half = vertical distance to the nearest obstacle pixel of the column, high half = to the nearest free pixel
template <typename T>
__global__ void __launch_bounds__(64) edt_columns(const EdtArgs a) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int b = blockIdx.y;
  if (x >= a.Wp) return;
  uint32_t* w = a.words + (int64_t)b * a.Hp * a.Wp + x;
  uint32_t d_obs = kNone, d_free = kNone, seen = 0;
  // downward walk: distance to the nearest pixel of each kind ABOVE (or at) the current one
  for (int y0 = 0; y0 < a.Hp; y0 += 8) {
    bool f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (y0 + k < a.Hp) ? is_free<T>(a, b, y0 + k, x) : true;      // (eight independent loads in flight)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (y0 + k < a.Hp) {
        d_obs = f[k] ? (d_obs == kNone ? kNone : d_obs + 1) : 0;
        d_free = f[k] ? 0 : (d_free == kNone ? kNone : d_free + 1);
        seen |= f[k] ? 2u : 1u;
        w[(int64_t)(y0 + k) * a.Wp] = d_obs | (d_free << 16);
      }
    }
  }
  if (seen) atomicOr(a.flags + b, seen);
  // upward walk: combine with the nearest pixel BELOW
  d_obs = kNone; d_free = kNone;
  for (int y0 = a.Hp - 1; y0 >= 0; y0 -= 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (y0 - k >= 0) ? w[(int64_t)(y0 - k) * a.Wp] : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (y0 - k >= 0) {
        const uint32_t up_obs = v[k] & 0xFFFFu, up_free = v[k] >> 16;
        d_obs = (up_obs == 0) ? 0 : (d_obs == kNone ? kNone : d_obs + 1);
        d_free = (up_free == 0) ? 0 : (d_free == kNone ? kNone : d_free + 1);
        const uint32_t o = up_obs < d_obs ? up_obs : d_obs, fr = up_free < d_free ? up_free : d_free;
        w[(int64_t)(y0 - k) * a.Wp] = o | (fr << 16);
      }
    }
  }
}

// squared column distance of one kind, as the row pass stages it in LDS (kFar: no such pixel in the column; kFar + d^2 stays below 2^31)
constexpr int32_t kFar = 0x3FFFFFFF;
__device__ __forceinline__ int32_t sq_or_far(uint32_t g) { return g == kNone ? kFar : (int32_t)(g * g); }

// PAD: the two staged rows carry Wp "no such column" cells on either side, so the search reads x - d and x + d without clamping its indices
// (6 Wp cells of LDS: rows of up to kPadMaxW columns); otherwise indices are clamped to the row (2 Wp cells, any width the API accepts).
constexpr int kPadMaxW = 2560;                   // 6 x 2560 x 4 bytes = 60 KB
template <typename O, bool PAD>
__global__ void __launch_bounds__(256) edt_rows(const EdtArgs a) {
  extern __shared__ int32_t sq[];                // squared distance to the nearest obstacle of each column, then to the nearest free pixel
  const int y = blockIdx.x, b = blockIdx.y, Wp = a.Wp;
  const int span = PAD ? 3 * Wp : Wp, off = PAD ? Wp : 0;
  const uint32_t* w = a.words + ((int64_t)b * a.Hp + y) * Wp;
  const uint32_t have = a.flags[b];
  for (int x = threadIdx.x; x < Wp; x += 256) {
    const uint32_t v = w[x];
    sq[off + x] = sq_or_far(v & 0xFFFFu);
    sq[span + off + x] = sq_or_far(v >> 16);
    if (PAD) { sq[x] = kFar; sq[2 * Wp + x] = kFar; sq[span + x] = kFar; sq[span + 2 * Wp + x] = kFar; }
  }
  __syncthreads();
  O* out = (O*)a.out + ((int64_t)b * a.Hp + y) * Wp;
  for (int x = threadIdx.x; x < Wp; x += 256) {
    const bool free_px = sq[span + off + x] == 0;        // its own distance to the nearest free pixel is 0
    // a free pixel measures to the obstacles, an obstacle pixel to free space
    const int32_t* g2 = (free_px ? sq : sq + span) + off;
    int32_t best;
    if (!(have & (free_px ? 1u : 2u))) {
      best = (y + 1) * (y + 1) + x * x;          // no pixel of the other kind anywhere: scipy's reference point (-1, 0)
    } else {
      // outward search: columns x - d and x + d cost d^2 + g^2; once d^2 alone reaches the best candidate nothing further out can win.
      // Two offsets per trip: the second one may lie past the stopping radius -- a true candidate all the same, it cannot lower the minimum wrongly.
      best = g2[x];
      int32_t dd = 1, odd = 3;                   // dd = d^2, odd = 2 d + 1
      if (PAD) {
        const int reach = (x > Wp - 1 - x ? x : Wp - 1 - x) + 1;      // no column further out than this
        int32_t stop = reach * reach;
        stop = stop < best ? stop : best;
        const int32_t* pl = g2 + x - 1;
        const int32_t* pr = g2 + x + 1;
        while (dd < stop) {
          const int32_t a0 = pl[0], a1 = pl[-1], b0 = pr[0], b1 = pr[1];
          const int32_t m0 = (a0 < b0 ? a0 : b0) + dd;
          dd += odd; odd += 2;
          const int32_t m1 = (a1 < b1 ? a1 : b1) + dd;
          dd += odd; odd += 2;
          const int32_t m = m0 < m1 ? m0 : m1;
          best = m < best ? m : best;
          stop = m < stop ? m : stop;
          pl -= 2; pr += 2;
        }
      } else {
        // an index clamped to the row stands for a column that was already examined at its true, smaller offset: its candidate can only be larger
        for (int d = 1; dd < best; d += 2) {
          const int l0 = x - d < 0 ? 0 : x - d, r0 = x + d > Wp - 1 ? Wp - 1 : x + d;
          const int l1 = x - d - 1 < 0 ? 0 : x - d - 1, r1 = x + d + 1 > Wp - 1 ? Wp - 1 : x + d + 1;
          const int32_t a0 = g2[l0], b0 = g2[r0], a1 = g2[l1], b1 = g2[r1];
          const int32_t m0 = (a0 < b0 ? a0 : b0) + dd;
          dd += odd; odd += 2;
          const int32_t m1 = (a1 < b1 ? a1 : b1) + dd;
          dd += odd; odd += 2;
          const int32_t m = m0 < m1 ? m0 : m1;
          best = m < best ? m : best;
        }
      }
    }
    const double dist = sqrt((double)best);
    out[x] = (O)(free_px ? (dist - 0.0) * a.res : (0.0 - dist) * a.res);       // (im_dist - inv_im_dist) * res, sdf_utils.py:20
  }
}

size_t words_bytes(int64_t B, int64_t Hp, int64_t Wp) { return (size_t)(B * Hp * Wp) * sizeof(uint32_t); }
size_t flags_bytes(int64_t B) { return (size_t)((B * sizeof(uint32_t) + 255) / 256) * 256; }

}  // namespace

extern "C" {

size_t dgp_sdf_2d_workspace_bytes(int32_t batch, int32_t rows, int32_t cols, int32_t padlen) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || padlen < 0) return 0;
  return flags_bytes(batch) + words_bytes(batch, (int64_t)rows + 2 * padlen, (int64_t)cols + 2 * padlen);
}

int dgp_sdf_2d(const void* image, int32_t image_dtype, int32_t batch, int32_t rows, int32_t cols, int32_t padlen, double res,
               void* sdf_out, int32_t out_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!image || !sdf_out || !workspace) return fail(DGP_EINVAL, "dgp_sdf_2d: null image, output or workspace");
  if (batch <= 0 || rows <= 0 || cols <= 0 || padlen < 0) return fail(DGP_EINVAL, "dgp_sdf_2d: batch, rows, cols must be positive and padlen non-negative");
  if (image_dtype != DGP_F32 && image_dtype != DGP_F64 && image_dtype != DGP_U8) return fail(DGP_EINVAL, "dgp_sdf_2d: image_dtype %d", image_dtype);
  if (out_dtype != DGP_F32 && out_dtype != DGP_F64) return fail(DGP_EINVAL, "dgp_sdf_2d: out_dtype %d", out_dtype);
  const int64_t Hp = (int64_t)rows + 2 * padlen, Wp = (int64_t)cols + 2 * padlen;
  if (Hp > kMaxDim || Wp > kMaxDim || batch > 65535) return fail(DGP_EUNSUPPORTED, "dgp_sdf_2d: padded image %lld x %lld (limit %d) or batch %d (limit 65535)", (long long)Hp, (long long)Wp, kMaxDim, batch);
  if (workspace_bytes < dgp_sdf_2d_workspace_bytes(batch, rows, cols, padlen)) return fail(DGP_EINVAL, "dgp_sdf_2d: workspace of %zu bytes, %zu needed", workspace_bytes, dgp_sdf_2d_workspace_bytes(batch, rows, cols, padlen));
  if (((uintptr_t)workspace & 3u) != 0) return fail(DGP_EINVAL, "dgp_sdf_2d: workspace must be 4-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  EdtArgs a;
  a.image = image; a.out = sdf_out;
  a.flags = (uint32_t*)workspace;
  a.words = (uint32_t*)((char*)workspace + flags_bytes(batch));
  a.B = batch; a.rows = rows; a.cols = cols; a.pad = padlen; a.Hp = (int32_t)Hp; a.Wp = (int32_t)Wp; a.res = res;
  if (hipMemsetAsync(a.flags, 0, flags_bytes(batch), s) != hipSuccess) return fail(DGP_EHIP, "dgp_sdf_2d: hipMemsetAsync failed");
  const dim3 gc((unsigned)((Wp + 63) / 64), (unsigned)batch), gr((unsigned)Hp, (unsigned)batch);
  if (image_dtype == DGP_F32) hipLaunchKernelGGL(edt_columns<float>, gc, dim3(64), 0, s, a);
  else if (image_dtype == DGP_F64) hipLaunchKernelGGL(edt_columns<double>, gc, dim3(64), 0, s, a);
  else hipLaunchKernelGGL(edt_columns<uint8_t>, gc, dim3(64), 0, s, a);
  const bool pad = Wp <= kPadMaxW;
  const size_t lds = (pad ? 6 : 2) * (size_t)Wp * sizeof(int32_t);
  if (out_dtype == DGP_F32) { if (pad) hipLaunchKernelGGL((edt_rows<float, true>), gr, dim3(256), lds, s, a); else hipLaunchKernelGGL((edt_rows<float, false>), gr, dim3(256), lds, s, a); }
  else { if (pad) hipLaunchKernelGGL((edt_rows<double, true>), gr, dim3(256), lds, s, a); else hipLaunchKernelGGL((edt_rows<double, false>), gr, dim3(256), lds, s, a); }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_sdf_2d: launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

}  // extern "C"
