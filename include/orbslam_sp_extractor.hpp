// orbslam_sp_extractor.hpp — the drop-in `orbslam::SPExtractor` of sp_orb_slam on libspfe.so.
//
// Replaces the class of /root/reference/orb_slam2/include/orb_slam/cv/sp_extractor.h:49-88 (a libtorch
// module + CUDA device) by one that derives the reference's own abstract interface
// `orbslam::BaseExtractor` (include/orb_slam/cv/base_extractor.h:8-93, virtual operator() :54-56) and
// spfe::ExtractorCV (include/spfe_extractor.hpp, the C-ABI adaptor).  The caller
// (Frame::ExtractORB, src/type/frame.cpp:296-314) holds a `BaseExtractor*`, calls
//     (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
// and then `dynamic_cast<SPExtractor *>(...)` to read getCov2Inv(), dense_dust_, heat_, occ_grid_ —
// all of which keep their names and types here, so frame.cpp / tracker.cpp compile unchanged.
//
// Include path of the consumer: the reference's `orb_slam/cv/` directory (for "base_extractor.h"),
// OpenCV, Eigen, and this repo's include/.  Compiled and run by tests/cpp/dropin_main.cpp.
#pragma once

#include <string>
#include <vector>

#include <Eigen/Dense>

#include "base_extractor.h"     // the reference's header, unmodified
#include "spfe_extractor.hpp"

namespace orbslam {

class SPExtractor : public BaseExtractor, public spfe::ExtractorCV {
 public:
  // The reference constructor (sp_extractor.cpp:342-359) takes only nfeatures and reads
  // camera::height / camera::width / common::model_path from config globals; its two-line
  // definition on top of the explicit one below is in INTEGRATION.md §2b.
  SPExtractor(int nfeatures);
  SPExtractor(int nfeatures, int height, int width, const std::string &model_path, int device = 0)
      : BaseExtractor(nfeatures, 1.0f, 1, 1, 1),   // one level, scale 1: sp_extractor.cpp:343
        // heat_ is cloned by Frame::ExtractORB (frame.cpp:304); heat_inv_ is read by nobody outside computeCovariance
        // (sp_extractor.cpp:508), which runs on the device here: it stays there unless heatInv() asks for it
        spfe::ExtractorCV(nfeatures, height, width, model_path, device, /*with_heat=*/true, /*lazy_heat_inv=*/true) {}
  virtual ~SPExtractor() = default;

  void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                  cv::OutputArray descriptors) override {
    spfe::ExtractorCV::operator()(image, mask, keypoints, descriptors);
  }

  // sp_extractor.h:65-67 return std::vector<Eigen::Vector2f> by value
  const std::vector<Eigen::Vector2f> getCov() { return toEigen(spfe::ExtractorCV::getCov()); }
  const std::vector<Eigen::Vector2f> getCov2Inv() { return toEigen(spfe::ExtractorCV::getCov2Inv()); }
  // getMask(), getHeatMap(), semi_dust_, dense_dust_, mask_, heat_, heat_inv_, occ_grid_: inherited from
  // spfe::ExtractorCV with the reference's names and cv::Mat types (sp_extractor.h:61-73).  heat_inv_ is empty after
  // operator() (nobody reads it) and filled by heatInv() on demand.

 private:
  static std::vector<Eigen::Vector2f> toEigen(const std::vector<spfe::Vec2f> &v) {
    std::vector<Eigen::Vector2f> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = Eigen::Vector2f(v[i].x, v[i].y);
    return o;
  }
};

}  // namespace orbslam
