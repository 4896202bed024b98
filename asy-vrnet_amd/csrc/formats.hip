// Input formats of the training pipeline on the device: what YoloDataset.__getitem__ and
// yolo_dataset_collate (utils/dataloader.py:88-107, 440-457) do to a batch AFTER the host-side letterbox -- normalise the
// RGB image (utils_seg/utils.py:43-47 preprocess_input: /255, - mean, / std, in float64), HWC -> CHW, cast to float32;
// clamp the label map to the ignore class (:96-97) and expand it to one-hot with the extra channel (:103-105).  The host
// hands over the letterboxed batch as BYTES (image 3 B / pixel, label 1 B / pixel) instead of the float64 -> float32 tensors
// the reference's loader ships: (12 + 8 + 4 (nc + 1)) B per pixel become 4 B over PCIe (8.4 MB instead of 126 MB for a
// batch of 8 at 512 x 512 with 9 classes), and the one-hot expansion -- np.eye indexing on the host in the reference --
// is a store pattern here.  One thread per pixel; every output store is coalesced (CHW planes, label, one-hot rows).
// The image arithmetic is done in double and rounded once, as numpy does it: results are bit-identical to the reference's.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void batch_formats_kernel(const unsigned char* img, const unsigned char* png, long npix,
                                                            long HW, int ns, float* images, long long* png_out,
                                                            float* onehot) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= npix) return;
  if (img) {
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    const long b = e / HW, p = e - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = (double)img[e * 3 + c];
      v /= 255.0;
      v -= mean[c];
      v /= sd[c];
      images[(b * 3 + c) * HW + p] = (float)v;
    }
  }
  if (png) {
    int lab = png[e];
    if (lab >= ns) lab = ns;
    if (png_out) png_out[e] = lab;
    if (onehot) {
      float* row = onehot + e * (ns + 1);
      for (int c = 0; c <= ns; ++c) row[c] = c == lab ? 1.f : 0.f;
    }
  }
}

}  // namespace

extern "C" int vrnet_batch_formats_u8(const unsigned char* img, const unsigned char* png, int B, int H, int W,
                                      int num_classes_seg, float* images, long long* png_out, float* onehot, void* stream) {
  VR_CHECK_ARG(B > 0 && H > 0 && W > 0, "batch_formats: bad shape");
  VR_CHECK_ARG(img || png, "batch_formats: nothing to convert");
  VR_CHECK_ARG(!img || images, "batch_formats: image bytes without an output");
  VR_CHECK_ARG(!png || ((png_out || onehot) && num_classes_seg > 0 && num_classes_seg < 255),
               "batch_formats: label bytes need an output and 0 < num_classes_seg < 255");
  const long npix = (long)B * H * W;
  hipLaunchKernelGGL(batch_formats_kernel, dim3((unsigned)vr_cdiv(npix, 256)), dim3(256), 0, vr_stream(stream), img, png,
                     npix, (long)H * W, num_classes_seg, images, png_out, onehot);
  VR_LAUNCH_CHECK("batch_formats");
  return VR_OK;
}
