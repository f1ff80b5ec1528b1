// sdf_edt.hip -- dgp_sdf_2d: signed Euclidean distance fields of a batch of occupancy images, on the GPU (gfx950 / CDNA4).
//
// What it replaces: diff_gpmp2/utils/sdf_utils.py:6-21 (sdf_2d), i.e. two calls of scipy.ndimage.distance_transform_edt per image
// on the host -- 8 ms for a 256 x 256 image, 34 ms for 512 x 512 -- which is what produces the grids the obstacle factor reads
// (examples/diff_gpmp2_2d_step_example.py:39, datasets/generate_2d_dataset.py:211).  Here a batch of images becomes a batch of
// (B, H + 2 pad, W + 2 pad) fields in two launches, bit-identical to scipy's result (the squared distances are integers; the
// square root of an integer in fp64 is correctly rounded on both sides, and the sign / resolution are applied in sdf_2d's order).
//
// Exact two-pass transform (separable in the squared distance):
//   1. column pass, one lane per image column (adjacent lanes own adjacent columns, so every row step is one coalesced line): for every pixel the vertical distance to the
//      nearest pixel of the OTHER class in its column -- a 16-bit word per pixel holds the class bit and that distance (0x7FFF: none).  edt_columns_bits (round 6; images of up
//      to 1024 padded rows): the image is read once into bit planes in LDS, both walks run on the bits, every word is written once.  edt_columns (taller images): the walk
//      down writes its values to the workspace, the walk up reads them back and writes the words.
//   2. edt_rows: one workgroup per image row; the row's words are staged in LDS and every lane resolves its pixels by an outward
//      search  D^2 = min_x' (x - x')^2 + g(x', y)^2  that stops as soon as (x - x')^2 alone reaches the best candidate -- the search
//      radius is the answer itself, a handful of pixels next to an obstacle.  A free pixel looks for the nearest obstacle and gets
//      +sqrt(D^2) res, an obstacle pixel for the nearest free pixel and gets -sqrt(D^2) res: of im_dist - inv_im_dist one term is 0.
// An image without any obstacle (or, unpadded, without any free pixel) has no nearest pixel of the other kind; scipy then measures
// from the pixel at (row -1, column 0), and so does this kernel (tests/test_sdf_edt.py pins that).
//
// HBM traffic per padded pixel: image in (once) + 2 x 2 bytes of the words (written by the column pass, read by
// the row pass; + 4 more bytes in edt_columns) + the field out; no arithmetic to speak of -- an HBM-bound byte kernel, not MFMA work.
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

// Round 6: the column pass for images of up to kBitsMaxRows padded rows.  edt_columns above sends every pixel's downward value through memory (written, read back by the
// upward walk, written again: 6 bytes per pixel beside the image).  A pixel's class is ONE BIT, and both walks are functions of the class bits alone: this kernel reads the
// image once into bit planes (32 rows per word, per lane, in LDS), derives the state BELOW every 32-row chunk from the words in O(1) per chunk (count trailing zeros),
// then walks chunk by chunk -- down into 32 registers, up against them -- and writes each word exactly once (2 bytes per pixel beside the image).
// One counter per direction: d(y) = distance to the nearest pixel of the OTHER class above (below) y = 1 where the class flips, d(y -+ 1) + 1 where it does not.
constexpr int kBitsMaxRows = 1024;
constexpr uint32_t kNoneWide = 1u << 20;        // "none" inside the walks (stays above kNone after 8192 increments); clamped to kNone in the word
template <typename T>
__global__ void __launch_bounds__(64) edt_columns_bits(const EdtArgs a) {
  extern __shared__ uint32_t plane[];            // [nw][64] class bits (1 = free space), then [nw][64] the state at the top row of every chunk (walking upwards)
  const int lane = threadIdx.x, x = blockIdx.x * 64 + lane, b = blockIdx.y;
  const bool on = x < a.Wp;
  const int Hp = a.Hp, nw = (Hp + 31) >> 5;
  uint32_t* bits = plane + lane;                 // bits[c * 64]
  uint32_t* top = plane + nw * 64 + lane;        // top[c * 64]: distance from row 32 c to the nearest pixel of the other class BELOW it
  uint32_t seen = 0;
  for (int c = 0; c < nw; ++c) {
    uint32_t word = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool f[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {             // (sixteen independent loads in flight)
        const int y = c * 32 + h * 16 + k;
        f[k] = (on && y < Hp) ? is_free<T>(a, b, y, x) : true;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) word |= (f[k] ? 1u : 0u) << (h * 16 + k);
    }
    bits[c * 64] = word;
    const int valid = Hp - c * 32 >= 32 ? 32 : Hp - c * 32;
    const uint32_t mask = valid == 32 ? 0xFFFFFFFFu : ((1u << valid) - 1u);
    seen |= ((~word & mask) ? 1u : 0u) | ((word & mask) ? 2u : 0u);
  }
  if (on && seen) atomicOr(a.flags + b, seen);
  // the state at the top row of every chunk, walking upwards: the first row of the other class inside the chunk, else the row under the chunk, else that row's own state + the chunk
  {
    uint32_t below_d = kNoneWide, below_bit = 0;
    for (int c = nw - 1; c >= 0; --c) {
      const uint32_t word = bits[c * 64];
      const int valid = Hp - c * 32 >= 32 ? 32 : Hp - c * 32;
      const uint32_t mask = valid == 32 ? 0xFFFFFFFFu : ((1u << valid) - 1u);
      const uint32_t t = word & 1u;
      const uint32_t other = (t ? ~word : word) & mask;          // rows of the class the top row is not
      uint32_t d;
      if (other) d = (uint32_t)__builtin_ctz(other);
      else if (c == nw - 1) d = kNoneWide;
      else d = below_bit != t ? (uint32_t)valid : (below_d >= kNoneWide ? kNoneWide : below_d + (uint32_t)valid);
      top[c * 64] = d;
      below_d = d; below_bit = t;
    }
  }
  uint16_t* w = a.words + (int64_t)b * Hp * a.Wp + (on ? x : 0);
  uint32_t d = kNoneWide, above_bit = bits[0] & 1u;              // (row 0 has no row above: no flip, no distance)
  for (int c = 0; c < nw; ++c) {
    const uint32_t word = bits[c * 64];
    const int valid = Hp - c * 32 >= 32 ? 32 : Hp - c * 32;
    const bool last = c == nw - 1;
    // class flips against the row above / the row below (the row under the image: none)
    const uint32_t flip_dn = word ^ ((word << 1) | above_bit);
    const uint32_t under_bit = last ? ((word >> (valid - 1)) & 1u) : (bits[(c + 1) * 64] & 1u);
    uint32_t flip_up = word ^ ((word >> 1) | (under_bit << 31));
    if (valid < 32) flip_up = (flip_up & ~(1u << (valid - 1))) | ((((word >> (valid - 1)) & 1u) ^ under_bit) << (valid - 1));
    uint32_t down[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe(~flip_dn, k, 1);      // flip: 0, else all ones
      d = (d & keep) + 1u;
      down[k] = d;
    }
    above_bit = word >> 31;
    uint32_t u = last ? kNoneWide : top[(c + 1) * 64];
    if (valid == 32) {
#pragma unroll
      for (int k = 31; k >= 0; --k) {
        const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe(~flip_up, k, 1);
        u = (u & keep) + 1u;
        uint32_t m = down[k] < u ? down[k] : u;
        m = m < kNone ? m : kNone;
        if (on) w[(int64_t)(c * 32 + k) * a.Wp] = (uint16_t)((((word >> k) & 1u) << 15) | m);
      }
    } else {
      for (int k = valid - 1; k >= 0; --k) {
        const uint32_t keep = ((flip_up >> k) & 1u) ? 0u : ~0u;
        u = (u & keep) + 1u;
        uint32_t dk = down[0];
#pragma unroll
        for (int q = 1; q < 32; ++q) dk = q == k ? down[q] : dk;
        uint32_t m = dk < u ? dk : u;
        m = m < kNone ? m : kNone;
        if (on) w[(int64_t)(c * 32 + k) * a.Wp] = (uint16_t)((((word >> k) & 1u) << 15) | m);
      }
    }
  }
}

// squared column distance of one kind, as the row pass stages it in LDS (kFar: no such pixel in the column; kFar + d^2 stays below 2^31)
constexpr int32_t kFar = 0x3FFFFFFF;
__device__ __forceinline__ int32_t sq_or_far(uint32_t g) { return g == kNone ? kFar : (int32_t)(g * g); }

// PAD: the two staged rows carry Wp + kEdge "no such column" cells on either side, so the search reads x - d and x + d without clamping its indices
// (6 Wp + 4 kEdge cells of LDS: rows of up to kPadMaxW columns); otherwise indices are clamped to the row (2 Wp cells, any width the API accepts).
// The padded search takes DGP_EDT_UNROLL offsets per side and trip (rounds 4-5: 2; round 6: 8 -- ten VALU instructions per four candidates then, 24 per sixteen now, at
// which point the loop is bound by the LDS reads themselves: 4.62 -> 4.09 ms per 4096 x 256^2, profiles/r06_sdf_edt_ab.txt); the last trip may look up to
// DGP_EDT_UNROLL - 1 offsets past the stopping radius or the row's end: true candidates or "no such column" cells, neither can lower the minimum wrongly.
// Measured in the same A/B and NOT kept: a wavefront per TILE of 8 x 8 (4 x 16) pixels instead of 64 consecutive pixels of a row (the radii of a tile differ by at most
// 11 where those of a row segment differ by up to 63, and a 258-pixel row stops costing a fifth wavefront search for two pixels) -- 5.13 (4.20) against 4.17 ms: eight
// staged rows are 60 KB of LDS, two workgroups per CU, and the barriers around the search no longer overlap with other workgroups' searches.
#ifndef DGP_EDT_UNROLL
#define DGP_EDT_UNROLL 8
#endif
#ifndef DGP_EDT_BITS
#define DGP_EDT_BITS 1        // 0: the column pass of rounds 4-5 (edt_columns); 1: edt_columns_bits where the image has at most kBitsMaxRows padded rows
#endif
constexpr int kEdge = 8;                         // >= DGP_EDT_UNROLL
constexpr int kPadMaxW = 2560;                   // (6 x 2560 + 4 x 8) x 4 bytes = 60 KB
static_assert(DGP_EDT_UNROLL >= 2 && DGP_EDT_UNROLL <= kEdge && DGP_EDT_UNROLL % 2 == 0, "DGP_EDT_UNROLL");
// The padded outward search of one pixel: D^2 = min over columns x' of (x - x')^2 + g2[x'], g2 = the staged squared column distances of the kind the pixel looks for
// (Wp + kEdge "no such column" cells on either side).  Columns x - d and x + d cost d^2 + g^2; once d^2 alone reaches the best candidate nothing further out can win.
__device__ __forceinline__ int32_t padded_search(const int32_t* g2, int x, int Wp) {
  int32_t best = g2[x];
  int32_t dd = 1, odd = 3;                       // dd = d^2, odd = 2 d + 1
  const int reach = (x > Wp - 1 - x ? x : Wp - 1 - x) + 1;      // no column further out than this
  int32_t stop = reach * reach;
  stop = stop < best ? stop : best;
  const int32_t* pl = g2 + x - 1;
  const int32_t* pr = g2 + x + 1;
  while (dd < stop) {
    int32_t c[DGP_EDT_UNROLL];
#pragma unroll
    for (int k = 0; k < DGP_EDT_UNROLL; ++k) {
      const int32_t l = pl[-k], r = pr[k];
      c[k] = (l < r ? l : r) + dd;
      dd += odd; odd += 2;
    }
#pragma unroll
    for (int k = 0; k < DGP_EDT_UNROLL; k += 2) {
      const int32_t m = c[k] < c[k + 1] ? c[k] : c[k + 1];
      best = m < best ? m : best;                // (v_min3_i32)
    }
    stop = best < stop ? best : stop;
    pl -= DGP_EDT_UNROLL; pr += DGP_EDT_UNROLL;
  }
  return best;
}

// stages one row's words as the two squared-distance arrays of the padded search: row[0 .. span) to the nearest obstacle, row[span .. 2 span) to the nearest free pixel
template <int NT>
__device__ __forceinline__ void stage_padded_row(const uint16_t* w, int Wp, int32_t* row) {
  const int span = 3 * Wp + 2 * kEdge, off = Wp + kEdge;
  for (int i = threadIdx.x; i < span; i += NT) {
    const int x = i - off;
    int32_t to_obs = kFar, to_free = kFar;
    if (x >= 0 && x < Wp) {
      const uint32_t v = w[x];
      const bool fr = (v & kFreeBit) != 0;
      const int32_t d2 = sq_or_far(v & kNone);
      to_obs = fr ? d2 : 0;                      // to the nearest obstacle (an obstacle pixel is its own)
      to_free = fr ? 0 : d2;                     // to the nearest free pixel
    }
    row[i] = to_obs; row[span + i] = to_free;
  }
}

template <typename O>
__device__ __forceinline__ void store_field(const EdtArgs& a, int b, int y, int x, bool free_px, int32_t best) {
  // row-major: (B, Hp, Wp); 4 x 4 tiles: row (y % 4) of tile (y / 4, x / 4) of image b's ceil(Hp / 4) x ceil(Wp / 4) tile grid
  const int Wt = (a.Wp + 3) >> 2;
  O* out = a.layout == 0 ? (O*)a.out + ((int64_t)b * a.Hp + y) * a.Wp + x
                         : (O*)a.out + (int64_t)b * ((a.Hp + 3) >> 2) * Wt * 16 + (int64_t)(y >> 2) * Wt * 16 + ((y & 3) << 2) + ((x >> 2) << 4) + (x & 3);
  const double dist = sqrt((double)best);
  *out = (O)(free_px ? (dist - 0.0) * a.res : (0.0 - dist) * a.res);       // (im_dist - inv_im_dist) * res, sdf_utils.py:20
}

template <typename O, bool PAD>
__global__ void __launch_bounds__(256) edt_rows(const EdtArgs a) {
  extern __shared__ int32_t sq[];                // squared distance to the nearest obstacle of each column, then to the nearest free pixel
  const int y = blockIdx.x, b = blockIdx.y, Wp = a.Wp;
  const int span = PAD ? 3 * Wp + 2 * kEdge : Wp, off = PAD ? Wp + kEdge : 0;
  const uint16_t* w = a.words + ((int64_t)b * a.Hp + y) * Wp;
  const uint32_t have = a.flags[b];
  if (PAD) {
    stage_padded_row<256>(w, Wp, sq);
  } else {
    for (int x = threadIdx.x; x < Wp; x += 256) {
      const uint32_t v = w[x];
      const bool fr = (v & kFreeBit) != 0;
      const int32_t d2 = sq_or_far(v & kNone);
      sq[x] = fr ? d2 : 0;
      sq[span + x] = fr ? 0 : d2;
    }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < Wp; x += 256) {
    const bool free_px = sq[span + off + x] == 0;        // its own distance to the nearest free pixel is 0
    // a free pixel measures to the obstacles, an obstacle pixel to free space
    const int32_t* g2 = (free_px ? sq : sq + span) + off;
    int32_t best;
    if (!(have & (free_px ? 1u : 2u))) {
      best = (y + 1) * (y + 1) + x * x;          // no pixel of the other kind anywhere: scipy's reference point (-1, 0)
    } else if (PAD) {
      best = padded_search(g2, x, Wp);
    } else {
      // an index clamped to the row stands for a column that was already examined at its true, smaller offset: its candidate can only be larger
      best = g2[x];
      int32_t dd = 1, odd = 3;
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
    store_field<O>(a, b, y, x, free_px, best);
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
  const bool bitplanes = DGP_EDT_BITS != 0 && Hp <= kBitsMaxRows;
  const size_t blds = (size_t)((Hp + 31) / 32) * 64 * 2 * sizeof(uint32_t);
#define DGP_EDT_COLS(T)                                                                          \
  do {                                                                                           \
    if (bitplanes) hipLaunchKernelGGL((edt_columns_bits<T>), gc, dim3(64), blds, s, a);           \
    else if (strip) hipLaunchKernelGGL((edt_columns<T, true>), gc, dim3(64), clds, s, a);         \
    else hipLaunchKernelGGL((edt_columns<T, false>), gc, dim3(64), 0, s, a);                      \
  } while (0)
  if (image_dtype == DGP_F32) DGP_EDT_COLS(float);
  else if (image_dtype == DGP_F64) DGP_EDT_COLS(double);
  else DGP_EDT_COLS(uint8_t);
#undef DGP_EDT_COLS
  const bool pad = Wp <= kPadMaxW;
  const size_t lds = (pad ? 6 * (size_t)Wp + 4 * kEdge : 2 * (size_t)Wp) * sizeof(int32_t);
  if (out_dtype == DGP_F32) { if (pad) hipLaunchKernelGGL((edt_rows<float, true>), gr, dim3(256), lds, s, a); else hipLaunchKernelGGL((edt_rows<float, false>), gr, dim3(256), lds, s, a); }
  else { if (pad) hipLaunchKernelGGL((edt_rows<double, true>), gr, dim3(256), lds, s, a); else hipLaunchKernelGGL((edt_rows<double, false>), gr, dim3(256), lds, s, a); }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_sdf_2d: launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

}  // extern "C"
