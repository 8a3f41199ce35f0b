// cov_host.cpp — computeCovariance on the host, for the host-facing calls.
//
// Follows /root/reference/orb_slam2/src/cv/sp_extractor.cpp:252-340: for each
// keypoint in emitted order, a FIFO breadth-first walk down the heat_inv hill
// (4-neighbours in the order left, up, right, down; a neighbour is taken when it
// has not been popped yet by ANY keypoint, its value is > 0 and strictly below
// the current pixel's), weighted second moments of the visited offsets, each
// clamped to >= 1, and their reciprocals.  The reference runs this stage on the
// host too; it stays sequential because every keypoint sees the pixels earlier
// keypoints consumed.
#include <cstddef>
#include <cstdint>
#include <vector>

namespace spfe {

void covariance_host(const float *heat_inv, int H, int W, const float *kp_xy, int K, float *cov2,
                     float *cov2_inv) {
  std::vector<uint8_t> taken((size_t)H * W, 0);
  std::vector<int> fifo;
  std::vector<float> val, ox2, oy2;
  for (int i = 0; i < K; ++i) {
    const int x0 = (int)kp_xy[2 * i], y0 = (int)kp_xy[2 * i + 1];
    fifo.clear(); val.clear(); ox2.clear(); oy2.clear();
    fifo.push_back(y0 * W + x0);
    for (size_t head = 0; head < fifo.size(); ++head) {
      const int id = fifo[head];
      const int x = id % W, y = id / W;
      taken[id] = 1;
      const float here = heat_inv[id];
      const float ddx = (float)x - (float)x0, ddy = (float)y - (float)y0;
      ox2.push_back(ddx * ddx);
      oy2.push_back(ddy * ddy);
      val.push_back(here);
      auto visit = [&](int nx, int ny) {
        const int nid = ny * W + nx;
        const float v = heat_inv[nid];
        if (!taken[nid] && v > 0.0f && v < here) fifo.push_back(nid);
      };
      if (x - 1 > 0) visit(x - 1, y);
      if (y - 1 > 0) visit(x, y - 1);
      if (x + 1 < W) visit(x + 1, y);
      if (y + 1 < H) visit(x, y + 1);
    }
    float total = 0.0f;
    for (float v : val) total += v;
    float cx = 0.0f, cy = 0.0f;
    for (size_t j = 0; j < val.size(); ++j) {
      const float wgt = val[j] / total;
      cx += wgt * ox2[j];
      cy += wgt * oy2[j];
    }
    cx = cx < 1.0f ? 1.0f : cx;
    cy = cy < 1.0f ? 1.0f : cy;
    cov2[2 * i] = cx;
    cov2[2 * i + 1] = cy;
    cov2_inv[2 * i] = 1.0f / cx;
    cov2_inv[2 * i + 1] = 1.0f / cy;
  }
}

}  // namespace spfe
