// refign_amd/csrc/datastep.hip -- N4, second part: the pixel work of the SOURCE / TARGET sampling of the UDA iteration on
// the device -- rare-class sampling's crop re-draws (data_modules/datasets/cityscapes.py:139-158), RandomCrop with its
// category-ratio re-draws (data_modules/transforms.py:282-361), RandomHorizontalFlip (:363-390), ConvertImageDtype (:438-464)
// and Normalize (:467-495), writing straight into the sample's slot of the batch tensors that
// CombinedDataModule.on_before_batch_transfer (combined_data_module.py:263-310) would build with torch.cat.
//   crop_label_hist   the 256-bin label histograms of up to 16 candidate crop boxes of one uint8 label map in ONE launch: the
//                     reference evaluates `torch.unique(crop, return_counts=True)` per candidate (up to 11 per draw of the crop,
//                     up to 11 draws per sample) -- here the host draws the whole candidate chain of a RandomCrop call (its
//                     draws do not depend on the outcomes, only where the chain stops does), asks once, and rewinds its random
//                     stream to the stop (refign_amd/datastep.py).
//   crop_flip_norm    out[c, y, x] = (u8[c, top + y, left + (flip ? w - 1 - x : x)] / 255 - mean[c]) / std[c] in fp32 with true
//                     divisions (what torchvision's convert_image_dtype + normalize compute), label -> int64.
// Bandwidth-bound byte work: coalesced 16-byte reads where the crop's left edge allows, LDS histograms, no MFMA.
#include "common.h"

namespace rfn {

struct CropBoxes {
  int top[16], left[16], h[16], w[16];
};

// grid (row blocks, K); block 256.  A block histograms rows [r0, r0 + rows) of candidate blockIdx.y in LDS (one counter array
// per wave: 4 x 256 ints, so that same-label neighbours in different waves do not serialise on one LDS address) and adds its
// non-zero bins to hist[k][256] with global atomics.
__global__ __launch_bounds__(256) void crop_label_hist_kernel(const unsigned char* __restrict__ lbl, int H, int W, CropBoxes bx,
                                                              int rows_per_block, int* __restrict__ hist) {
  __shared__ int h[4][256];
  const int k = blockIdx.y, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const int top = bx.top[k], left = bx.left[k], ch = bx.h[k], cw = bx.w[k];
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, ch);
  for (int r = r0; r < r1; ++r) {
    const unsigned char* row = lbl + (size_t)(top + r) * W + left;
    for (int x = threadIdx.x; x < cw; x += 256) atomicAdd(&h[wave][row[x]], 1);
  }
  __syncthreads();
  const int v = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
  if (v != 0) atomicAdd(hist + k * 256 + threadIdx.x, v);
}

// grid (ceil(w / 256), h, images); one thread per output pixel, all channels
__global__ __launch_bounds__(256) void crop_flip_norm_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ lbl,
                                                             int C, int H, int W, int top, int left, int h, int w, int flip,
                                                             float m0, float m1, float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out_img, long* __restrict__ out_lbl) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int sx = left + (flip ? w - 1 - x : x), sy = top + y;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  if (img != nullptr) {
    for (int c = 0; c < C; ++c) {
      const float v = (float)img[((size_t)c * H + sy) * W + sx] / 255.0f;
      out_img[((size_t)c * h + y) * w + x] = (v - mean[c < 3 ? c : 2]) / sd[c < 3 ? c : 2];
    }
  }
  if (lbl != nullptr) out_lbl[(size_t)y * w + x] = (long)lbl[(size_t)sy * W + sx];
}

}  // namespace rfn

extern "C" {

int rfn_crop_label_hist_u8(const void* label, int H, int W, const int* boxes, int K, int* hist, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(label && boxes && hist, "rfn_crop_label_hist_u8: null pointer");
  RFN_REQUIRE(H > 0 && W > 0 && K > 0 && K <= 16, "rfn_crop_label_hist_u8: H, W > 0, 1 <= K <= 16");
  CropBoxes bx{};
  int maxh = 0;
  for (int k = 0; k < K; ++k) {
    bx.top[k] = boxes[4 * k], bx.left[k] = boxes[4 * k + 1], bx.h[k] = boxes[4 * k + 2], bx.w[k] = boxes[4 * k + 3];
    RFN_REQUIRE(bx.top[k] >= 0 && bx.left[k] >= 0 && bx.h[k] > 0 && bx.w[k] > 0 && bx.top[k] + bx.h[k] <= H &&
                    bx.left[k] + bx.w[k] <= W,
                "rfn_crop_label_hist_u8: box %d (%d, %d, %d, %d) outside the %d x %d map", k, bx.top[k], bx.left[k], bx.h[k], bx.w[k], H, W);
    maxh = bx.h[k] > maxh ? bx.h[k] : maxh;
  }
  hipStream_t st = (hipStream_t)stream;
  if (int rc = zero_async(hist, (size_t)K * 256 * sizeof(int), st)) return rc;
  const int rows = 16;
  hipLaunchKernelGGL(crop_label_hist_kernel, dim3((unsigned)cdiv(maxh, rows), (unsigned)K), dim3(256), 0, st,
                     (const unsigned char*)label, H, W, bx, rows, hist);
  return check_launch("crop_label_hist_kernel");
}

int rfn_crop_flip_norm_u8(const void* image, const void* label, int C, int H, int W, int top, int left, int h, int w, int flip,
                          const float* mean3, const float* std3, float* out_image, long* out_label, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE((image && out_image) || (label && out_label), "rfn_crop_flip_norm_u8: nothing to do");
  RFN_REQUIRE(!image || (mean3 && std3 && C >= 1 && C <= 3), "rfn_crop_flip_norm_u8: image needs mean / std and 1..3 channels");
  RFN_REQUIRE(H > 0 && W > 0 && top >= 0 && left >= 0 && h > 0 && w > 0 && top + h <= H && left + w <= W && h <= 65535,
              "rfn_crop_flip_norm_u8: crop (%d, %d, %d, %d) outside the %d x %d image", top, left, h, w, H, W);
  float m[3] = {0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
  if (image)
    for (int c = 0; c < C; ++c) m[c] = mean3[c], s[c] = std3[c];
  hipLaunchKernelGGL(crop_flip_norm_kernel, dim3((unsigned)cdiv(w, 256), (unsigned)h), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)image, (const unsigned char*)label, C, H, W, top, left, h, w, flip, m[0], m[1], m[2], s[0],
                     s[1], s[2], out_image, out_label);
  return check_launch("crop_flip_norm_kernel");
}

}  // extern "C"
