// dropin_latency — the drop-in call as the SLAM back-end makes it, timed: orbslam::SPExtractor (include/
// orbslam_sp_extractor.hpp; heat maps ON, the adaptor's default) constructed like tracker.cpp:131 and driven through the
// BaseExtractor pointer + dynamic_cast exactly as Frame::ExtractORB does (/root/reference/orb_slam2/src/type/frame.cpp:
// 296-311): operator(), getCov2Inv(), dense_dust_.clone(), heat_.clone(), occ_grid_.copyTo().  Host frame in, cv::KeyPoint /
// cv::Mat / Eigen out: PCIe inclusive (0.36 MB up; record 1.1 MB + 2 x 1.44 MB of heat maps down at 752x480).
// Built by __graft_entry__.build() against the interface stand-ins under tests/stubs (the GPU box has neither OpenCV nor the
// reference tree); bench.py runs it and puts the line into `dropin_operator_call_ms`.
// usage: dropin_latency <weights.spfw> <image.raw> <H> <W> <nfeatures> <calls> <warmup> [lazy | copy]   -> one JSON line
// (lazy: the opt-in form whose heat_inv_ stays on the device, orbslam_sp_extractor.hpp; default: the reference's post-call state;
// copy: the default with the maps through the library's buffers and deep copies into the members, setMapsInPlace(false))
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "orbslam_sp_extractor.hpp"

namespace orbslam {
namespace camera { int height = 0, width = 0; }
namespace common { std::string model_path; }
namespace tracking { int num_features = 0; }
SPExtractor::SPExtractor(int nfeatures) : SPExtractor(nfeatures, camera::height, camera::width, common::model_path) {}
}  // namespace orbslam

using namespace orbslam;

int main(int argc, char **argv) {
  if (argc != 8 && argc != 9) return 2;
  const bool lazy = argc == 9 && std::string(argv[8]) == "lazy";
  camera::height = atoi(argv[3]);
  camera::width = atoi(argv[4]);
  tracking::num_features = atoi(argv[5]);
  common::model_path = argv[1];
  const int calls = atoi(argv[6]), warm = atoi(argv[7]);
  if (calls < 1 || warm < 0) return 2;   // (the percentiles below index a non-empty list)
  const int H = camera::height, W = camera::width;
  std::vector<unsigned char> pix((size_t)H * W);
  FILE *f = fopen(argv[2], "rb");
  if (!f || fread(pix.data(), 1, pix.size(), f) != pix.size()) return 3;
  fclose(f);
  try {
    BaseExtractor *mpORBextractorLeft = lazy ? new SPExtractor(tracking::num_features, H, W, common::model_path, 0, true)
                                             : new SPExtractor(tracking::num_features);   // tracker.cpp:131
    if (argc == 9 && std::string(argv[8]) == "copy") dynamic_cast<SPExtractor *>(mpORBextractorLeft)->setMapsInPlace(false);
    cv::Mat im(H, W, CV_8UC1, pix.data());
    std::vector<double> ms, ms_op;   // the whole of Frame::ExtractORB's body / operator() alone
    size_t K = 0;
    for (int i = 0; i < calls + warm; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      // --- Frame::ExtractORB(0, im), frame.cpp:296-311 ---
      std::vector<cv::KeyPoint> mvKeys;
      cv::Mat mDescriptors, dust_, heat_, occ_grid;
      (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
      const auto t_op = std::chrono::steady_clock::now();
      std::vector<Eigen::Vector2f> cov2_inv_ = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->getCov2Inv();
      dust_ = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->dense_dust_.clone();
      heat_ = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->heat_.clone();
      dynamic_cast<SPExtractor *>(mpORBextractorLeft)->occ_grid_.copyTo(occ_grid);
      const auto t1 = std::chrono::steady_clock::now();
      if (i >= warm) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
      if (i >= warm) ms_op.push_back(std::chrono::duration<double, std::milli>(t_op - t0).count());
      K = mvKeys.size();
      if (cov2_inv_.size() != K || heat_.rows != H || occ_grid.cols != W / 8) return 5;
      if (dynamic_cast<SPExtractor *>(mpORBextractorLeft)->heat_inv_.empty() != lazy) return 6;
    }
    std::sort(ms.begin(), ms.end());
    std::sort(ms_op.begin(), ms_op.end());
    printf("{\"p50\": %.4f, \"p99\": %.4f, \"operator_call_p50\": %.4f, \"calls\": %d, \"K\": %zu, \"heat_maps\": true, \"heat_inv_after_call\": %s}\n",
           ms[ms.size() / 2], ms[std::max<size_t>(1, (size_t)(ms.size() * 0.99)) - 1], ms_op[ms_op.size() / 2], (int)ms.size(), K,
           lazy ? "false" : "true");
    delete mpORBextractorLeft;
  } catch (const std::exception &e) {
    fprintf(stderr, "dropin_latency: %s\n", e.what());
    return 1;
  }
  return 0;
}
