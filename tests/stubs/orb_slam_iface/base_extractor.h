// TEST-ONLY stand-in for the reference's abstract extractor interface, used where /root/reference does
// not exist (the GPU box): what SURVEY.md §8 row A10 / §8(b) describe of
// orb_slam2/include/orb_slam/cv/base_extractor.h — a polymorphic base with the ORBextractor call
// signature as a pure virtual and the pyramid getters Frame reads.  Where the reference tree IS present
// (the build container) tests compile tests/cpp/dropin_main.cpp against the reference's own header instead.
#pragma once
#include <vector>
#include <opencv2/opencv.hpp>

namespace orbslam {
class BaseExtractor {
 public:
  BaseExtractor(int n, float scale, int levels, int ini_fast, int min_fast)
      : nfeatures(n), scaleFactor(scale), nlevels(levels), iniThFAST(ini_fast), minThFAST(min_fast),
        mvScaleFactor(levels, 1.0f), mvInvScaleFactor(levels, 1.0f), mvLevelSigma2(levels, 1.0f), mvInvLevelSigma2(levels, 1.0f) {
    for (int l = 1; l < levels; ++l) {
      mvScaleFactor[l] = mvScaleFactor[l - 1] * scale;
      mvLevelSigma2[l] = mvScaleFactor[l] * mvScaleFactor[l];
      mvInvScaleFactor[l] = 1.0f / mvScaleFactor[l];
      mvInvLevelSigma2[l] = 1.0f / mvLevelSigma2[l];
    }
  }
  virtual ~BaseExtractor() = default;
  virtual void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                          cv::OutputArray descriptors) = 0;
  int GetLevels() { return nlevels; }
  float GetScaleFactor() { return (float)scaleFactor; }
  std::vector<float> GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

 protected:
  int nfeatures;
  double scaleFactor;
  int nlevels, iniThFAST, minThFAST;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};
}  // namespace orbslam
