// Drives include/spfe_extractor.hpp exactly the way Frame::ExtractORB drives the
// reference extractor (/root/reference/orb_slam2/src/type/frame.cpp:296-314):
//   (*extractor)(im, cv::Mat(), mvKeys, mDescriptors); then read getCov2Inv(),
//   dense_dust_, heat_, occ_grid_.
// usage: adaptor_main <weights.spfw> <image.raw> <H> <W> <nfeatures> <out.bin>
// out.bin: int32 K, then K x {x, y, response, cov2inv_x, cov2inv_y}, K x 256 desc,
//          occ_grid int16 [hc*wc], dense_dust f32 [hc*wc], heat f32 [H*W]
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "spfe_extractor.hpp"

int main(int argc, char **argv) {
  if (argc != 7) return 2;
  const int H = atoi(argv[3]), W = atoi(argv[4]), nf = atoi(argv[5]);
  std::vector<unsigned char> pix((size_t)H * W);
  FILE *f = fopen(argv[2], "rb");
  if (!f || fread(pix.data(), 1, pix.size(), f) != pix.size()) return 3;
  fclose(f);
  try {
    spfe::ExtractorCV extractor(nf, H, W, argv[1]);
    cv::Mat im(H, W, CV_8UC1, pix.data());
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors, noMask;
    extractor(im, noMask, mvKeys, mDescriptors);
    // empty image must throw like the reference (sp_extractor.cpp:364-365)
    bool threw = false;
    try {
      cv::Mat empty;
      extractor(empty, noMask, mvKeys, mDescriptors);
    } catch (const std::runtime_error &e) {
      threw = std::string(e.what()) == "input image is empty";
    }
    if (!threw) return 5;
    extractor(im, noMask, mvKeys, mDescriptors);
    const auto &cinv = extractor.getCov2Inv();
    FILE *o = fopen(argv[6], "wb");
    const int K = (int)mvKeys.size();
    fwrite(&K, 4, 1, o);
    for (int i = 0; i < K; ++i) {
      const float rec[5] = {mvKeys[i].pt.x, mvKeys[i].pt.y, mvKeys[i].response, cinv[i].x, cinv[i].y};
      fwrite(rec, 4, 5, o);
      if (mvKeys[i].size != 1.0f || mvKeys[i].angle != -1.0f || mvKeys[i].octave != 0) return 6;
    }
    if (K) fwrite(mDescriptors.data, 4, (size_t)K * 256, o);
    fwrite(extractor.occ_grid_.data, 2, (size_t)(H / 8) * (W / 8), o);
    fwrite(extractor.dense_dust_.data, 4, (size_t)(H / 8) * (W / 8), o);
    fwrite(extractor.heat_.data, 4, (size_t)H * W, o);
    fclose(o);
    // SearchByBruteForce's matcher (sp_matcher.cpp:1661-1668) through the adaptor: a frame
    // matched against itself pairs every keypoint with itself at distance 0
    std::vector<cv::DMatch> matches;
    extractor.matchBruteForce(mDescriptors, mDescriptors, matches);
    if ((int)matches.size() != K) return 7;
    for (int i = 0; i < K; ++i)
      if (matches[i].queryIdx != i || matches[i].trainIdx != i || matches[i].distance != 0.0f) return 7;
    // trackFrameDustKFLocal's association (tracker_dust.cpp:113-172) through the adaptor: map points that
    // ARE the frame's keypoints, projected to their own cells, each get their own keypoint back
    {
      std::vector<cv::Point2f> uv(K);
      for (int i = 0; i < K; ++i) { uv[i].x = (float)((int)mvKeys[i].pt.x / 8) + 0.5f; uv[i].y = (float)((int)mvKeys[i].pt.y / 8) + 0.5f; }
      std::vector<int> kp_idx;
      extractor.matchPatches(mDescriptors, uv, mDescriptors, kp_idx);
      if ((int)kp_idx.size() != K) return 9;
      for (int i = 0; i < K; ++i)
        if (kp_idx[i] != i) return 9;
    }
    // input staging through the adaptor: a 1-channel source without maps is the identity, so
    // extractRaw() must reproduce operator()'s result
    std::vector<cv::KeyPoint> k2;
    cv::Mat d2, noMap;
    extractor.setStaging(H, W, 1, false, noMap, noMap);
    extractor.extractRaw(im, k2, d2);
    if ((int)k2.size() != K) return 8;
    for (int i = 0; i < K; ++i)
      if (k2[i].pt.x != mvKeys[i].pt.x || k2[i].pt.y != mvKeys[i].pt.y) return 8;
    if (K && memcmp(d2.data, mDescriptors.data, (size_t)K * 256 * 4)) return 8;
  } catch (const std::exception &e) {
    fprintf(stderr, "adaptor_main: %s\n", e.what());
    return 4;
  }
  return 0;
}
