// The compiled drop-in: orbslam::SPExtractor (include/orbslam_sp_extractor.hpp) deriving the reference's
// BaseExtractor, used exactly the way the SLAM back-end uses the reference class:
//   * constructed like tracker.cpp:131  `mpORBextractorLeft = new SPExtractor(tracking::num_features)`
//     (the 1-argument constructor reading config globals, defined below as INTEGRATION.md §2b shows);
//   * called through the BASE pointer and then down-cast, the body of Frame::ExtractORB
//     (/root/reference/orb_slam2/src/type/frame.cpp:296-311), copied here in shape: operator(), getCov2Inv(),
//     dense_dust_.clone(), heat_.clone(), occ_grid_.copyTo();
//   * the pyramid getters Frame's constructor reads (frame.cpp:211-217).
// usage: dropin_main <weights.spfw> <image.raw> <H> <W> <nfeatures> <out.bin>   (out.bin as adaptor_main's)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orbslam_sp_extractor.hpp"

// the reference's config globals (orb_slam/config.h) this constructor reads
namespace orbslam {
namespace camera { int height = 0, width = 0; }
namespace common { std::string model_path; }
namespace tracking { int num_features = 0; }

SPExtractor::SPExtractor(int nfeatures)   // == the whole of the new src/cv/sp_extractor.cpp
    : SPExtractor(nfeatures, camera::height, camera::width, common::model_path) {}
}  // namespace orbslam

using namespace orbslam;

int main(int argc, char **argv) {
  if (argc != 7) return 2;
  camera::height = atoi(argv[3]);
  camera::width = atoi(argv[4]);
  tracking::num_features = atoi(argv[5]);
  common::model_path = argv[1];
  const int H = camera::height, W = camera::width;
  std::vector<unsigned char> pix((size_t)H * W);
  FILE *f = fopen(argv[2], "rb");
  if (!f || fread(pix.data(), 1, pix.size(), f) != pix.size()) return 3;
  fclose(f);
  try {
    BaseExtractor *mpORBextractorLeft = new SPExtractor(tracking::num_features);   // tracker.cpp:131
    // Frame::Frame reads these (frame.cpp:211-217): one level, all factors 1
    if (mpORBextractorLeft->GetLevels() != 1 || mpORBextractorLeft->GetScaleFactor() != 1.0f) return 10;
    const std::vector<float> one{1.0f};
    if (mpORBextractorLeft->GetScaleFactors() != one || mpORBextractorLeft->GetInverseScaleFactors() != one ||
        mpORBextractorLeft->GetScaleSigmaSquares() != one || mpORBextractorLeft->GetInverseScaleSigmaSquares() != one)
      return 11;

    // --- Frame::ExtractORB(0, im), frame.cpp:296-311 ---
    cv::Mat im(H, W, CV_8UC1, pix.data());
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors, dust_, heat_, occ_grid;
    (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
    std::vector<Eigen::Vector2f> cov2_inv_ = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->getCov2Inv();
    dust_ = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->dense_dust_.clone();
    heat_ = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->heat_.clone();
    dynamic_cast<SPExtractor *>(mpORBextractorLeft)->occ_grid_.copyTo(occ_grid);
    const int grid_cols = occ_grid.cols, grid_rows = occ_grid.rows;
    if (grid_cols != W / 8 || grid_rows != H / 8) return 12;
    if (dynamic_cast<SPExtractor *>(mpORBextractorLeft) == nullptr) return 13;
    if ((int)cov2_inv_.size() != (int)mvKeys.size()) return 14;
    if (dynamic_cast<SPExtractor *>(mpORBextractorLeft)->getCov().size() != mvKeys.size()) return 15;
    // heat_inv_ (sp_extractor.h:73): filled by operator() as the reference fills it (Appendix A item 18) ...
    if (dynamic_cast<SPExtractor *>(mpORBextractorLeft)->heat_inv_.empty()) return 16;
    cv::Mat heat_inv = dynamic_cast<SPExtractor *>(mpORBextractorLeft)->heat_inv_.clone();
    if (heat_inv.rows != H || heat_inv.cols != W) return 17;
    {   // ... unless the integrator opts into the lazy form: empty after operator(), the same map from heatInv() on demand
      SPExtractor lazy(tracking::num_features, H, W, common::model_path, 0, /*lazy_heat_inv=*/true);
      std::vector<cv::KeyPoint> k2;
      cv::Mat d2;
      lazy(im, cv::Mat(), k2, d2);
      if (!lazy.heat_inv_.empty() || lazy.heat_.empty()) return 18;
      cv::Mat hi2 = lazy.heatInv().clone();
      if (lazy.heat_inv_.empty() || hi2.rows != H || hi2.cols != W || memcmp(hi2.data, heat_inv.data, (size_t)H * W * 4) != 0) return 19;
      if (k2.size() != mvKeys.size()) return 20;
    }

    {   // the members' storage is where the device writes the maps (aimMaps(): spfe_set_map_buffers) — as the reference's
        // `heat_ = ...` evaluates into the member's existing buffer: the same storage call after call, the same bits as the
        // copied form, a member the caller released is made (and aimed at) again
      SPExtractor *sp = dynamic_cast<SPExtractor *>(mpORBextractorLeft);
      const unsigned char *p_heat = sp->heat_.data, *p_inv = sp->heat_inv_.data;
      std::vector<cv::KeyPoint> k2;
      cv::Mat d2;
      (*mpORBextractorLeft)(im, cv::Mat(), k2, d2);
      if (sp->heat_.data != p_heat || sp->heat_inv_.data != p_inv) return 21;
      if (memcmp(sp->heat_.data, heat_.data, (size_t)H * W * 4) || memcmp(sp->heat_inv_.data, heat_inv.data, (size_t)H * W * 4)) return 22;
      SPExtractor copied(tracking::num_features, H, W, common::model_path);
      copied.setMapsInPlace(false);
      copied(im, cv::Mat(), k2, d2);
      if (memcmp(copied.heat_.data, heat_.data, (size_t)H * W * 4) || memcmp(copied.heat_inv_.data, heat_inv.data, (size_t)H * W * 4)) return 23;
      if (k2.size() != mvKeys.size() || (k2.size() && memcmp(d2.data, mDescriptors.data, k2.size() * 256 * 4))) return 24;
      cv::Mat kept = sp->heat_;    // a shallow holder (Frame could be one) ...
      sp->heat_ = cv::Mat();       // ... and the member released by the caller
      memset(sp->heat_inv_.data, 0, (size_t)H * W * 4);
      (*mpORBextractorLeft)(im, cv::Mat(), k2, d2);
      if (sp->heat_.empty() || sp->heat_.data == kept.data) return 25;
      if (memcmp(sp->heat_.data, heat_.data, (size_t)H * W * 4) || memcmp(sp->heat_inv_.data, heat_inv.data, (size_t)H * W * 4)) return 26;
      if (memcmp(kept.data, heat_.data, (size_t)H * W * 4)) return 27;   // (the holder's copy is untouched by the later call)
    }

    // the empty-image error of sp_extractor.cpp:364-365 through the base pointer
    bool threw = false;
    try {
      cv::Mat empty;
      (*mpORBextractorLeft)(empty, cv::Mat(), mvKeys, mDescriptors);
    } catch (const std::runtime_error &e) {
      threw = std::string(e.what()) == "input image is empty";
    }
    if (!threw) return 5;
    (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);   // the object is still usable

    FILE *o = fopen(argv[6], "wb");
    if (!o) return 4;
    const int K = (int)mvKeys.size();
    fwrite(&K, 4, 1, o);
    for (int i = 0; i < K; ++i) {
      const float rec[5] = {mvKeys[i].pt.x, mvKeys[i].pt.y, mvKeys[i].response, cov2_inv_[i](0), cov2_inv_[i](1)};
      fwrite(rec, 4, 5, o);
      if (mvKeys[i].size != 1.0f || mvKeys[i].angle != -1.0f || mvKeys[i].octave != 0) return 6;
    }
    if (K) fwrite(mDescriptors.data, 4, (size_t)K * 256, o);
    fwrite(occ_grid.data, 2, (size_t)grid_rows * grid_cols, o);
    fwrite(dust_.data, 4, (size_t)grid_rows * grid_cols, o);
    fwrite(heat_.data, 4, (size_t)H * W, o);
    fwrite(heat_inv.data, 4, (size_t)H * W, o);
    fclose(o);
    delete mpORBextractorLeft;   // virtual destructor of the base
  } catch (const std::exception &e) {
    fprintf(stderr, "dropin_main: %s\n", e.what());
    return 1;
  }
  return 0;
}
