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
//      nearest OBSTACLE pixel and to the nearest FREE pixel of that column -- of which one is always 0, so a 16-bit word per pixel holds the class bit and the other distance (0x7FFF: none).
//      Adjacent lanes own adjacent columns, so every row step is one coalesced line; the image is read once and the downward values of a strip of
//      64 columns wait in LDS for the upward walk (images of up to 512 padded rows; taller ones make that round trip through the workspace).
//   2. edt_rows: one workgroup per image row; the row's words are staged in LDS and every lane resolves its pixels by an outward
//      search  D^2 = min_x' (x - x')^2 + g(x', y)^2  that stops as soon as (x - x')^2 alone reaches the best candidate -- the search
//      radius is the answer itself, a handful of pixels next to an obstacle.  A free pixel looks for the nearest obstacle and gets
//      +sqrt(D^2) res, an obstacle pixel for the nearest free pixel and gets -sqrt(D^2) res: of im_dist - inv_im_dist one term is 0.
// An image without any obstacle (or, unpadded, without any free pixel) has no nearest pixel of the other kind; scipy then measures
// from the pixel at (row -1, column 0), and so does this kernel (tests/test_sdf_edt.py pins that).
//
// HBM traffic per padded pixel: image in (once) + 2 x 2 bytes of the words (written by the column pass, read by
// the row pass) + the field out; no arithmetic to speak of -- an HBM-bound byte kernel, not MFMA work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dgp_host.h"

namespace {

using dgp_host::fail;

constexpr uint32_t kNone = 0x7FFFu;            // "no such pixel in this column"
constexpr uint32_t kFreeBit = 0x8000u;
constexpr int kMaxDim = 8192;                  // padded rows / columns: squared distances stay below 2^27 + 2^27, and the row pass keeps two rows of them in 64 KB of LDS
#ifndef DGP_EDT_LDS_STRIP
#define DGP_EDT_LDS_STRIP 0       // measured (profiles/r04_sdf_edt.txt): 33 KB of LDS per wavefront leaves four wavefronts per CU -- 3.0 against 1.2 ms at batch 4096; 92 against 98 us for one image
#endif
constexpr int kColsLdsRows = 512;              // tallest padded image whose 64-column strip (2 bytes per pixel) fits the 64 KB of LDS of the column pass

struct EdtArgs {
  const void* image;      // (B, rows, cols), contiguous
  void* out;              // (B, rows + 2 pad, cols + 2 pad)
  uint16_t* words;        // workspace: (B, Hp, Wp), per pixel: bit 15 = free space, bits 0-14 = vertical distance to the nearest pixel of the OTHER kind in its column
  uint32_t* flags;        // workspace: per image, bit 0: has an obstacle pixel, bit 1: has a free pixel
  int32_t B, rows, cols, pad, Hp, Wp;
  int32_t layout;         // DgpSdf::layout of the output: 0 row-major, 1 4 x 4 tiles
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

__device__ __forceinline__ uint32_t step_dist(uint32_t d) { return d == kNone ? kNone : d + 1; }

// One lane per image column, one wavefront per strip of 64 columns.  Downward walk: per pixel its class and the distance to the nearest pixel of the other
// kind ABOVE it; upward walk: the same from BELOW, the smaller of the two is the word.  LDS = true: the strip's downward values wait in LDS (2 bytes per pixel,
// Hp <= kColsLdsRows) and only the final words reach memory; otherwise they make the round trip through the workspace itself.
template <typename T, bool LDS>
__global__ void __launch_bounds__(64) edt_columns(const EdtArgs a) {
  extern __shared__ uint16_t strip[];            // [Hp][64] (LDS variant)
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int b = blockIdx.y;
  const bool on = x < a.Wp;
  uint16_t* w = a.words + (int64_t)b * a.Hp * a.Wp + (on ? x : 0);
  uint32_t d_obs = kNone, d_free = kNone, seen = 0;
  for (int y0 = 0; y0 < a.Hp; y0 += 16) {
    bool f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[k] = (on && y0 + k < a.Hp) ? is_free<T>(a, b, y0 + k, x) : true;      // (sixteen independent loads in flight)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (y0 + k < a.Hp) {
        d_obs = f[k] ? step_dist(d_obs) : 0;
        d_free = f[k] ? 0 : step_dist(d_free);
        seen |= f[k] ? 2u : 1u;
        const uint16_t v = (uint16_t)(f[k] ? (kFreeBit | d_obs) : d_free);
        if (LDS) strip[(y0 + k) * 64 + threadIdx.x] = v;
        else if (on) w[(int64_t)(y0 + k) * a.Wp] = v;
      }
    }
  }
  if (on && seen) atomicOr(a.flags + b, seen);
  d_obs = kNone; d_free = kNone;
  for (int y0 = a.Hp - 1; y0 >= 0; y0 -= 16) {
    uint16_t v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (y0 - k >= 0) ? (LDS ? strip[(y0 - k) * 64 + threadIdx.x] : (on ? w[(int64_t)(y0 - k) * a.Wp] : (uint16_t)0)) : (uint16_t)0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (y0 - k >= 0) {
        const bool fr = (v[k] & kFreeBit) != 0;
        const uint32_t up = v[k] & kNone;
        d_obs = fr ? step_dist(d_obs) : 0;
        d_free = fr ? 0 : step_dist(d_free);
        const uint32_t below = fr ? d_obs : d_free;
        const uint32_t m = up < below ? up : below;
        if (on) w[(int64_t)(y0 - k) * a.Wp] = (uint16_t)((fr ? kFreeBit : 0u) | m);
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
  const uint16_t* w = a.words + ((int64_t)b * a.Hp + y) * Wp;
  const uint32_t have = a.flags[b];
  for (int x = threadIdx.x; x < Wp; x += 256) {
    const uint32_t v = w[x];
    const bool fr = (v & kFreeBit) != 0;
    const int32_t d2 = sq_or_far(v & kNone);
    sq[off + x] = fr ? d2 : 0;                   // to the nearest obstacle (an obstacle pixel is its own)
    sq[span + off + x] = fr ? 0 : d2;            // to the nearest free pixel
    if (PAD) { sq[x] = kFar; sq[2 * Wp + x] = kFar; sq[span + x] = kFar; sq[span + 2 * Wp + x] = kFar; }
  }
  __syncthreads();
  // row-major: this row of the (B, Hp, Wp) output; 4 x 4 tiles: row (y % 4) of the tiles (y / 4, .) of image b's ceil(Hp / 4) x ceil(Wp / 4) tile grid
  const int Wt = (Wp + 3) >> 2;
  O* out = a.layout == 0 ? (O*)a.out + ((int64_t)b * a.Hp + y) * Wp
                         : (O*)a.out + (int64_t)b * ((a.Hp + 3) >> 2) * Wt * 16 + (int64_t)(y >> 2) * Wt * 16 + ((y & 3) << 2);
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
    out[a.layout == 0 ? x : ((x >> 2) << 4) + (x & 3)] = (O)(free_px ? (dist - 0.0) * a.res : (0.0 - dist) * a.res);       // (im_dist - inv_im_dist) * res, sdf_utils.py:20
  }
}

size_t words_bytes(int64_t B, int64_t Hp, int64_t Wp) { return ((size_t)(B * Hp * Wp) * sizeof(uint16_t) + 3) / 4 * 4; }
size_t flags_bytes(int64_t B) { return (size_t)((B * sizeof(uint32_t) + 255) / 256) * 256; }

}  // namespace

extern "C" {

size_t dgp_sdf_2d_workspace_bytes(int32_t batch, int32_t rows, int32_t cols, int32_t padlen) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || padlen < 0) return 0;
  return flags_bytes(batch) + words_bytes(batch, (int64_t)rows + 2 * padlen, (int64_t)cols + 2 * padlen);
}

int dgp_sdf_2d(const void* image, int32_t image_dtype, int32_t batch, int32_t rows, int32_t cols, int32_t padlen, double res,
               void* sdf_out, int32_t out_dtype, int32_t out_layout, void* workspace, size_t workspace_bytes, void* stream) {
  if (!image || !sdf_out || !workspace) return fail(DGP_EINVAL, "dgp_sdf_2d: null image, output or workspace");
  if (batch <= 0 || rows <= 0 || cols <= 0 || padlen < 0) return fail(DGP_EINVAL, "dgp_sdf_2d: batch, rows, cols must be positive and padlen non-negative");
  if (image_dtype != DGP_F32 && image_dtype != DGP_F64 && image_dtype != DGP_U8) return fail(DGP_EINVAL, "dgp_sdf_2d: image_dtype %d", image_dtype);
  if (out_dtype != DGP_F32 && out_dtype != DGP_F64) return fail(DGP_EINVAL, "dgp_sdf_2d: out_dtype %d", out_dtype);
  if (out_layout != DGP_SDF_ROWMAJOR && out_layout != DGP_SDF_TILED4) return fail(DGP_EINVAL, "dgp_sdf_2d: out_layout %d", out_layout);
  const int64_t Hp = (int64_t)rows + 2 * padlen, Wp = (int64_t)cols + 2 * padlen;
  if (Hp > kMaxDim || Wp > kMaxDim || batch > 65535) return fail(DGP_EUNSUPPORTED, "dgp_sdf_2d: padded image %lld x %lld (limit %d) or batch %d (limit 65535)", (long long)Hp, (long long)Wp, kMaxDim, batch);
  if (workspace_bytes < dgp_sdf_2d_workspace_bytes(batch, rows, cols, padlen)) return fail(DGP_EINVAL, "dgp_sdf_2d: workspace of %zu bytes, %zu needed", workspace_bytes, dgp_sdf_2d_workspace_bytes(batch, rows, cols, padlen));
  if (((uintptr_t)workspace & 3u) != 0) return fail(DGP_EINVAL, "dgp_sdf_2d: workspace must be 4-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  EdtArgs a;
  a.image = image; a.out = sdf_out;
  a.flags = (uint32_t*)workspace;
  a.words = (uint16_t*)((char*)workspace + flags_bytes(batch));
  a.B = batch; a.rows = rows; a.cols = cols; a.pad = padlen; a.Hp = (int32_t)Hp; a.Wp = (int32_t)Wp; a.res = res; a.layout = out_layout;
  if (hipMemsetAsync(a.flags, 0, flags_bytes(batch), s) != hipSuccess) return fail(DGP_EHIP, "dgp_sdf_2d: hipMemsetAsync failed");
  const dim3 gc((unsigned)((Wp + 63) / 64), (unsigned)batch), gr((unsigned)Hp, (unsigned)batch);
  const bool strip = DGP_EDT_LDS_STRIP != 0 && Hp <= kColsLdsRows;
  const size_t clds = strip ? (size_t)Hp * 64 * sizeof(uint16_t) : 0;
#define DGP_EDT_COLS(T)                                                                          \
  do {                                                                                           \
    if (strip) hipLaunchKernelGGL((edt_columns<T, true>), gc, dim3(64), clds, s, a);              \
    else hipLaunchKernelGGL((edt_columns<T, false>), gc, dim3(64), 0, s, a);                      \
  } while (0)
  if (image_dtype == DGP_F32) DGP_EDT_COLS(float);
  else if (image_dtype == DGP_F64) DGP_EDT_COLS(double);
  else DGP_EDT_COLS(uint8_t);
#undef DGP_EDT_COLS
  const bool pad = Wp <= kPadMaxW;
  const size_t lds = (pad ? 6 : 2) * (size_t)Wp * sizeof(int32_t);
  if (out_dtype == DGP_F32) { if (pad) hipLaunchKernelGGL((edt_rows<float, true>), gr, dim3(256), lds, s, a); else hipLaunchKernelGGL((edt_rows<float, false>), gr, dim3(256), lds, s, a); }
  else { if (pad) hipLaunchKernelGGL((edt_rows<double, true>), gr, dim3(256), lds, s, a); else hipLaunchKernelGGL((edt_rows<double, false>), gr, dim3(256), lds, s, a); }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_sdf_2d: launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

}  // extern "C"
