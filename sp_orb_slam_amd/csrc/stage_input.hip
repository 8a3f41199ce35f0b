// stage_input.hip — input staging on the GPU: the step BEFORE the path.
//
// The reference prepares every camera frame on the host with OpenCV:
//   cv::remap(mono, mono, m1, m2, cv::INTER_LINEAR)      orb_slam2/src/io/data_loader.cc:519-521
//       (m1, m2 = CV_32FC1 maps of cv::initUndistortRectifyMap, :485-486; BORDER_CONSTANT 0)
//   mono(cv::Rect(0, 0, camera::width, camera::height))  orb_slam2/src/system.cpp:160-161
//   cvtColor(..., CV_BGR2GRAY | CV_RGB2GRAY | CV_BGRA2GRAY | CV_RGBA2GRAY)
//                                                        orb_slam2/src/tracking/mono_tracker.cpp:18-28
// and SPExtractor then does convertTo(CV_32F, 1/255) (sp_extractor.cpp:388), which is already
// inside conv1a.  Here one kernel produces the cropped gray u8 frame conv1a reads, from the raw
// camera image resident in HBM: one lane per output pixel, a 4-tap gather per channel.
//
// Integer arithmetic of OpenCV 3.x, restated (and by oracle_stage_input):
//   remap, INTER_LINEAR, 8-bit: sx = cvRound(mx * 32), sy = cvRound(my * 32) (round half to even);
//   integer position (sx >> 5, sy >> 5) saturated to int16, 5-bit fractions fx, fy; weights =
//   the 15-bit table BilinearTab_i[fy][fx] = {(32-fy)(32-fx), (32-fy)fx, fy(32-fx), fy fx} * 32
//   — except the (0,0) entry, which OpenCV's table builder leaves as {32767, 0, 0, 1} (32768
//   saturates in int16 and the fix-up lands on the last tap); D = sat_u8((sum + 2^14) >> 15);
//   taps outside the source read the border value 0.
//   cvtColor 8-bit: gray = (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14.
#include "spfe_kernels.h"

namespace spfe {

__device__ __forceinline__ int cv_round_x32(float m) {
  // cvRound(m * INTER_TAB_SIZE): the product is exact (power of two), rint = round half to even
  return (int)__builtin_rintf(m * 32.0f);
}
__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

template <int CN>
__global__ __launch_bounds__(256) void stage_input_kernel(StageParams p) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= p.W || y >= p.H) return;
  const uint8_t *src = p.src + (size_t)b * p.src_frame_bytes;
  int ch[CN];
  if (p.map_x) {
    const size_t mi = (size_t)y * p.src_w + x;
    const int sx = cv_round_x32(p.map_x[mi]), sy = cv_round_x32(p.map_y[mi]);
    const int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
    const int fx = sx & 31, fy = sy & 31;
    int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
    if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
    const bool in_x0 = (unsigned)ix < (unsigned)p.src_w, in_x1 = (unsigned)(ix + 1) < (unsigned)p.src_w;
    const bool in_y0 = (unsigned)iy < (unsigned)p.src_h, in_y1 = (unsigned)(iy + 1) < (unsigned)p.src_h;
    const uint8_t *r0 = src + (size_t)(in_y0 ? iy : 0) * p.src_stride;
    const uint8_t *r1 = src + (size_t)(in_y1 ? iy + 1 : 0) * p.src_stride;
    const int c0 = (in_x0 ? ix : 0) * CN, c1 = (in_x1 ? ix + 1 : 0) * CN;
#pragma unroll
    for (int k = 0; k < CN; ++k) {
      const int v00 = (in_x0 && in_y0) ? r0[c0 + k] : 0, v01 = (in_x1 && in_y0) ? r0[c1 + k] : 0;
      const int v10 = (in_x0 && in_y1) ? r1[c0 + k] : 0, v11 = (in_x1 && in_y1) ? r1[c1 + k] : 0;
      const int v = (v00 * w0 + v01 * w1 + v10 * w2 + v11 * w3 + (1 << 14)) >> 15;
      ch[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
  } else {
    const uint8_t *q = src + (size_t)y * p.src_stride + (size_t)x * CN;
#pragma unroll
    for (int k = 0; k < CN; ++k) ch[k] = q[k];
  }
  int g;
  if (CN == 1) {
    g = ch[0];
  } else {
    const int bl = p.rgb ? ch[2] : ch[0], rd = p.rgb ? ch[0] : ch[2];
    g = (bl * 1868 + ch[1] * 9617 + rd * 4899 + (1 << 13)) >> 14;
  }
  p.gray[((size_t)b * p.H + y) * p.W + x] = (uint8_t)g;
}

hipError_t launch_stage_input(const StageParams &p, int channels, int n, hipStream_t s) {
  dim3 g((p.W + 63) / 64, (p.H + 3) / 4, n);
  if (channels == 1) hipLaunchKernelGGL(stage_input_kernel<1>, g, dim3(256), 0, s, p);
  else if (channels == 3) hipLaunchKernelGGL(stage_input_kernel<3>, g, dim3(256), 0, s, p);
  else if (channels == 4) hipLaunchKernelGGL(stage_input_kernel<4>, g, dim3(256), 0, s, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace spfe
