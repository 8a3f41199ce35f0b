// spfe_extractor.hpp — C++ host adaptor over the C ABI (include/spfe.h).
//
// Restores the reference's extractor call shape for the ORB-SLAM2 back-end:
//   virtual void operator()(cv::InputArray image, cv::InputArray mask,
//                           std::vector<cv::KeyPoint>& keypoints,
//                           cv::OutputArray descriptors)
// (/root/reference/orb_slam2/include/orb_slam/cv/base_extractor.h:54-56, overridden
// at sp_extractor.h:57-59) and the public side outputs Frame::ExtractORB reads
// right after the call (/root/reference/orb_slam2/src/type/frame.cpp:296-314):
// getCov2Inv(), dense_dust_, heat_, occ_grid_ (plus semi_dust_, heat_inv_, mask_,
// getCov(), getHeatMap(), getMask() — sp_extractor.h:61-73).
//
// Header-only; needs OpenCV core on the CONSUMER side only (libspfe.so itself has
// no OpenCV/Eigen/torch dependency).  The covariance getters return
// std::vector<spfe::Vec2f>; INTEGRATION.md shows the three-line conversion to
// the reference's std::vector<Eigen::Vector2f> and the drop-in
// `orbslam::SPExtractor` built on this class.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "spfe.h"

namespace spfe {

struct Vec2f {
  float x, y;
};

class ExtractorCV {
 public:
  // Reference ctor: SPExtractor(int nfeatures) reading camera::height/width and
  // common::model_path from globals (sp_extractor.cpp:342-359).
  // lazy_heat_inv: heat_inv_ — which no caller of the reference reads — is not copied back by operator();
  // heatInv() fetches it on demand (SPFE_FLAG_LAZY_HEAT_INV: 1.44 MB less D2H per 752x480 call)
  ExtractorCV(int nfeatures, int height, int width, const std::string &weights_path, int device = 0,
              bool with_heat = true, bool lazy_heat_inv = false)
      : height_(height), width_(width), with_heat_(with_heat), lazy_(lazy_heat_inv) {
    // this translation unit's view of spfe.h against the library's (struct strides, entry points)
    if (spfe_check_abi(SPFE_ABI_VERSION, sizeof(spfe_config), sizeof(spfe_result), sizeof(spfe_record_layout)) != SPFE_OK)
      throw std::runtime_error(std::string("libspfe: ") + spfe_last_error());
    spfe_config cfg{};
    cfg.height = height;
    cfg.width = width;
    cfg.num_features = nfeatures;
    cfg.max_batch = 1;
    cfg.device = device;
    cfg.precision = SPFE_PRECISION_F32;
    cfg.flags = with_heat ? (SPFE_FLAG_HEAT | (lazy_heat_inv ? SPFE_FLAG_LAZY_HEAT_INV : 0u)) : 0u;
    cfg.weights = nullptr;
    cfg.weights_path = weights_path.c_str();
    if (spfe_create(&cfg, &h_) != SPFE_OK) throw std::runtime_error(std::string("spfe_create: ") + spfe_last_error());
  }
  ExtractorCV(const ExtractorCV &) = delete;
  ExtractorCV &operator=(const ExtractorCV &) = delete;
  virtual ~ExtractorCV() { spfe_destroy(h_); }

  // Same signature and semantics as SPExtractor::operator() (sp_extractor.cpp:361-514).
  // `mask` is ignored, as in the reference.
  void operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                  cv::OutputArray _descriptors) {
    if (_image.empty()) throw std::runtime_error("input image is empty");  // :364-365
    cv::Mat image = _image.getMat();
    if (image.type() != CV_8UC1) throw std::runtime_error("input image must be CV_8UC1");  // assert :368
    if (image.rows != height_ || image.cols != width_)
      throw std::runtime_error("input image size differs from the configured extractor size");
    // The call in parts (spfe.h), so that this side's deep copies run while the device still works on the frame: the H x W maps
    // are in host memory before selection, sampling and covariance have finished, the descriptor rows before the covariance has.
    // With the members' storage as the maps' destination (aimMaps()) only the rows are left to copy.
    const bool in_place = aimMaps();
    const uint8_t *one[1] = {image.data};
    int rc = spfe_extract_begin(h_, one, static_cast<int>(image.step), 1);
    if (rc == SPFE_EEMPTY) throw std::runtime_error("input image is empty");
    if (rc != SPFE_OK) throw std::runtime_error(std::string("spfe_extract: ") + spfe_last_error());
    const float *heat = nullptr, *heat_inv = nullptr, *rows = nullptr;
    int K = 0;
    if (!in_place) {
      rc = spfe_extract_maps(h_, &heat, nullptr);          // heat is the first to arrive; heat_inv travels while it is copied
      if (rc == SPFE_OK && heat) {
        cv::Mat(height_, width_, CV_32FC1, const_cast<float *>(heat)).copyTo(heat_);
        rc = spfe_extract_maps(h_, nullptr, &heat_inv);
        if (rc == SPFE_OK && heat_inv) cv::Mat(height_, width_, CV_32FC1, const_cast<float *>(heat_inv)).copyTo(heat_inv_);
      }
    }
    if (rc == SPFE_OK) rc = spfe_extract_rows(h_, 0, &K, &rows);
    if (rc == SPFE_OK && rows) {
      _descriptors.create(K, SPFE_DESC_DIM, CV_32FC1);  // :512
      if (K > 0) cv::Mat(K, SPFE_DESC_DIM, CV_32FC1, const_cast<float *>(rows)).copyTo(_descriptors.getMat());
    }
    spfe_result r{};
    const int rcf = spfe_extract_finish(h_, &r);   // (always: it closes the call)
    if (rc != SPFE_OK || rcf != SPFE_OK) throw std::runtime_error(std::string("spfe_extract: ") + spfe_last_error());
    publish(r, _keypoints, _descriptors, heat != nullptr, heat_inv != nullptr, rows != nullptr && K == r.K);
  }

  // Input staging on the GPU.  setStaging() once, with the CV_32FC1 maps of
  // cv::initUndistortRectifyMap (data_loader.cc:485-486; empty Mats = no remap); then extractRaw()
  // takes the frame as cv::imread returned it and does what data_loader.cc:519-521 (cv::remap,
  // INTER_LINEAR), system.cpp:160-161 (crop to camera::width x height) and mono_tracker.cpp:18-28
  // (cvtColor to gray; rgb = mbRGB) did on the host, followed by operator()'s work.
  void setStaging(int src_height, int src_width, int channels, bool rgb, const cv::Mat &map_x,
                  const cv::Mat &map_y) {
    spfe_staging st{};
    st.src_height = src_height;
    st.src_width = src_width;
    st.channels = channels;
    st.rgb = rgb ? 1 : 0;
    if (!map_x.empty() || !map_y.empty()) {
      if (map_x.type() != CV_32FC1 || map_y.type() != CV_32FC1 || map_x.rows != src_height ||
          map_y.rows != src_height || map_x.cols != src_width || map_y.cols != src_width ||
          map_x.step != src_width * sizeof(float) || map_y.step != src_width * sizeof(float))
        throw std::runtime_error("setStaging: maps must be contiguous CV_32FC1 of the source size");
      st.map_x = reinterpret_cast<const float *>(map_x.data);
      st.map_y = reinterpret_cast<const float *>(map_y.data);
    }
    if (spfe_set_staging(h_, &st) != SPFE_OK) throw std::runtime_error(std::string("spfe_set_staging: ") + spfe_last_error());
    raw_row_bytes_ = static_cast<size_t>(src_width) * channels;
    raw_rows_ = src_height;
  }
  void extractRaw(const cv::Mat &raw, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors) {
    if (raw.empty()) throw std::runtime_error("input image is empty");  // :364-365
    if (raw.rows != raw_rows_ || raw.step < raw_row_bytes_)
      throw std::runtime_error("extractRaw: frame does not match setStaging()");
    spfe_result r{};
    aimMaps();
    const int rc = spfe_extract_staged(h_, raw.data, static_cast<int>(raw.step), &r);
    if (rc == SPFE_EEMPTY) throw std::runtime_error("input image is empty");
    if (rc != SPFE_OK) throw std::runtime_error(std::string("spfe_extract_staged: ") + spfe_last_error());
    publish(r, _keypoints, _descriptors);
  }

 protected:
  // heat_done / heat_inv_done / desc_done: that output of this call is in place already (operator() copied it beside the device's work)
  void publish(const spfe_result &r, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors,
               bool heat_done = false, bool heat_inv_done = false, bool desc_done = false) {
    const int hc = height_ / 8, wc = width_ / 8;
    _keypoints.resize(r.K);
    cov2_.resize(r.K);
    cov2_inv_.resize(r.K);
    for (int i = 0; i < r.K; ++i) {
      cv::KeyPoint kp(r.kp_xy[2 * i], r.kp_xy[2 * i + 1], 1.0f);  // :231-232
      kp.response = r.kp_response[i];                              // :271
      _keypoints[i] = kp;
      cov2_[i] = {r.cov2[2 * i], r.cov2[2 * i + 1]};
      cov2_inv_[i] = {r.cov2_inv[2 * i], r.cov2_inv[2 * i + 1]};
    }
    if (!desc_done) {
      _descriptors.create(r.K, SPFE_DESC_DIM, CV_32FC1);  // :512
      if (r.K > 0) cv::Mat(r.K, SPFE_DESC_DIM, CV_32FC1, const_cast<float *>(r.desc)).copyTo(_descriptors.getMat());
    }
    // side outputs: deep copies (the library buffers live until the next call)
    cv::Mat(hc, wc, CV_32FC1, const_cast<float *>(r.semi_dust)).copyTo(semi_dust_);
    cv::Mat(hc, wc, CV_32FC1, const_cast<float *>(r.dense_dust)).copyTo(dense_dust_);
    cv::Mat(hc, wc, CV_16SC1, const_cast<int16_t *>(r.occ_grid)).copyTo(occ_grid_);
    // (a map the device wrote into the member's own storage — aimMaps() — is in place already)
    if (r.heat && !heat_done && static_cast<const void *>(r.heat) != heat_.data)
      cv::Mat(height_, width_, CV_32FC1, const_cast<float *>(r.heat)).copyTo(heat_);
    if (r.heat_inv && !heat_inv_done && static_cast<const void *>(r.heat_inv) != heat_inv_.data)
      cv::Mat(height_, width_, CV_32FC1, const_cast<float *>(r.heat_inv)).copyTo(heat_inv_);
    else if (!r.heat_inv) heat_inv_ = cv::Mat();   // (lazy: heatInv() fetches this call's map; a stale one must not be mistaken for it)
    status_ = r.status;
  }

 public:
  // Replaces the body of SPMatcher::SearchByBruteForce's matcher (sp_matcher.cpp:1661-1668):
  //   auto matcher = cv::BFMatcher::create(cv::NORM_L2, true); matcher->add(desc_train);
  //   matcher->train(); matcher->match(desc_query, matches);
  // Both are K x 256 CV_32FC1 with contiguous rows.  Emits one cv::DMatch per matched query, in
  // query order, like BFMatcher::match.
  void matchBruteForce(const cv::Mat &desc_query, const cv::Mat &desc_train, std::vector<cv::DMatch> &matches,
                       bool cross_check = true) {
    matches.clear();
    if (desc_query.empty() || desc_train.empty()) return;
    if (desc_query.cols != 256 || desc_train.cols != 256 || desc_query.type() != CV_32FC1 ||
        desc_train.type() != CV_32FC1 || desc_query.step != 256 * sizeof(float) ||
        desc_train.step != 256 * sizeof(float))
      throw std::runtime_error("matchBruteForce: descriptors must be contiguous K x 256 CV_32FC1");
    std::vector<int32_t> idx(desc_query.rows);
    std::vector<float> dist(desc_query.rows);
    if (spfe_match(h_, reinterpret_cast<const float *>(desc_query.data), desc_query.rows,
                   reinterpret_cast<const float *>(desc_train.data), desc_train.rows, cross_check ? 1 : 0,
                   idx.data(), dist.data()) != SPFE_OK)
      throw std::runtime_error(spfe_last_error());
    for (int i = 0; i < desc_query.rows; ++i)
      if (idx[i] >= 0) matches.push_back(cv::DMatch(i, idx[i], 0, dist[i]));
  }

  // Replaces  flann->knnMatch(desc_query, matches, 2)  on the cv::FlannBasedMatcher that holds desc_train
  // (KeyFrame::matchMps keyframe.cpp:447-448 on the index of buildIndexesMps :421-445; SPMatcher sp_matcher.cpp:200,
  // :269) by the exact two nearest neighbours; the ratio test that follows (0.7, keyframe.cpp:462) is unchanged.
  // matches[i] holds 0, 1 or 2 DMatch for query i, nearest first, like OpenCV's knnMatch.
  void knnMatch2(const cv::Mat &desc_query, const cv::Mat &desc_train, std::vector<std::vector<cv::DMatch>> &matches) {
    matches.assign(static_cast<size_t>(desc_query.rows), std::vector<cv::DMatch>());
    if (desc_query.empty() || desc_train.empty()) return;
    if (desc_query.cols != 256 || desc_train.cols != 256 || desc_query.type() != CV_32FC1 ||
        desc_train.type() != CV_32FC1 || desc_query.step != 256 * sizeof(float) ||
        desc_train.step != 256 * sizeof(float))
      throw std::runtime_error("knnMatch2: descriptors must be contiguous K x 256 CV_32FC1");
    std::vector<int32_t> idx(2 * static_cast<size_t>(desc_query.rows));
    std::vector<float> dist(2 * static_cast<size_t>(desc_query.rows));
    if (spfe_match_knn2(h_, reinterpret_cast<const float *>(desc_query.data), desc_query.rows,
                        reinterpret_cast<const float *>(desc_train.data), desc_train.rows, idx.data(),
                        dist.data()) != SPFE_OK)
      throw std::runtime_error(spfe_last_error());
    for (int i = 0; i < desc_query.rows; ++i)
      for (int k = 0; k < 2; ++k)
        if (idx[2 * i + k] >= 0) matches[i].push_back(cv::DMatch(i, idx[2 * i + k], 0, dist[2 * i + k]));
  }

  // Direct "dust" alignment: Optimizer::PoseOptimizationDust(pFrame, mps, is_visible) (optimizer_dust.cpp:170-294) on
  // the dust map of the frame extracted last.  Tcw: CV_32F 4x4 (Frame::mTcw), updated in place (SetPose, :287);
  // points: N x 3 CV_32F world positions (MapPoint::GetWorldPos); fx..cy: Frame::fx.. at full resolution.
  // Returns n_inlier; inlier[i] = is_visible / in_view, proj_uv[i] = (dust_proj_u, dust_proj_v).
  int alignDust(cv::Mat &Tcw, const cv::Mat &points, float fx, float fy, float cx, float cy, std::vector<bool> &inlier,
                std::vector<cv::Point2f> &proj_uv, int max_iterations = 40) {
    if (Tcw.rows != 4 || Tcw.cols != 4 || Tcw.type() != CV_32FC1 || Tcw.step != 4 * sizeof(float))
      throw std::runtime_error("alignDust: Tcw must be a contiguous 4 x 4 CV_32F matrix");
    const int n = points.rows;
    if (n > 0 && (points.cols != 3 || points.type() != CV_32FC1 || points.step != 3 * sizeof(float)))
      throw std::runtime_error("alignDust: points must be contiguous N x 3 CV_32F");
    if (dense_dust_.empty()) throw std::runtime_error("alignDust: no frame has been extracted yet");
    spfe_dust_params prm{fx, fy, cx, cy, max_iterations, 0.9, 0.9};
    std::vector<uint8_t> inl(n > 0 ? n : 1);
    std::vector<float> uv(2 * static_cast<size_t>(n > 0 ? n : 1));
    float out[16];
    int n_inlier = 0, iters = 0;
    if (spfe_align_dust(h_, reinterpret_cast<const float *>(dense_dust_.data),
                        n > 0 ? reinterpret_cast<const float *>(points.data) : nullptr, n,
                        reinterpret_cast<const float *>(Tcw.data), &prm, out, inl.data(), uv.data(), &n_inlier,
                        &iters) != SPFE_OK)
      throw std::runtime_error(spfe_last_error());
    for (int k = 0; k < 16; ++k) reinterpret_cast<float *>(Tcw.data)[k] = out[k];
    inlier.assign(n, false);
    proj_uv.assign(n, cv::Point2f());
    for (int i = 0; i < n; ++i) {
      inlier[i] = inl[i] != 0;
      proj_uv[i].x = uv[2 * i];
      proj_uv[i].y = uv[2 * i + 1];
    }
    return n_inlier;
  }

  // The association loop of Tracker::trackFrameDustKFLocal (tracker_dust.cpp:113-172) against the frame
  // extracted last: map point i (descriptor row i, projected dust-map position uv[i] in cells —
  // dust_proj_u / dust_proj_v) takes the nearest keypoint of its 2 x 2 cells below max_dist, earlier map
  // points first; kp_idx[i] = index into the keypoints / descriptors of that frame, or -1.
  void matchPatches(const cv::Mat &mp_desc, const std::vector<cv::Point2f> &uv, const cv::Mat &frame_desc,
                    std::vector<int> &kp_idx, float max_dist = 0.75f) {
    const int m = static_cast<int>(uv.size());
    kp_idx.assign(m, -1);
    if (m == 0 || frame_desc.empty()) return;
    if (mp_desc.rows != m || mp_desc.cols != 256 || mp_desc.type() != CV_32FC1 || mp_desc.step != 256 * sizeof(float) ||
        frame_desc.cols != 256 || frame_desc.type() != CV_32FC1 || frame_desc.step != 256 * sizeof(float))
      throw std::runtime_error("matchPatches: descriptors must be contiguous rows of 256 CV_32F");
    std::vector<float> puv(2 * static_cast<size_t>(m));
    for (int i = 0; i < m; ++i) { puv[2 * i] = uv[i].x; puv[2 * i + 1] = uv[i].y; }
    std::vector<int32_t> out(m);
    if (spfe_match_patches(h_, reinterpret_cast<const float *>(mp_desc.data), puv.data(), m,
                           reinterpret_cast<const int16_t *>(occ_grid_.data),
                           reinterpret_cast<const float *>(frame_desc.data), frame_desc.rows, max_dist,
                           out.data()) != SPFE_OK)
      throw std::runtime_error(spfe_last_error());
    for (int i = 0; i < m; ++i) kp_idx[i] = out[i];
  }

  cv::Mat getMask() { return mask_; }
  cv::Mat getHeatMap() { return heat_; }
  // heat_inv_ of the last call (sp_extractor.cpp:468): the member when operator() brought it back, else fetched now
  const cv::Mat &heatInv() {
    if (heat_inv_.empty()) {
      const float *p = nullptr;
      if (spfe_fetch_heat_inv(h_, 0, &p) != SPFE_OK) throw std::runtime_error(std::string("spfe_fetch_heat_inv: ") + spfe_last_error());
      cv::Mat(height_, width_, CV_32FC1, const_cast<float *>(p)).copyTo(heat_inv_);
    }
    return heat_inv_;
  }
  // The members' own storage as the destination of the maps' D2H (spfe_set_map_buffers: page-locked by the library while set).
  // The reference re-fills heat_ / heat_inv_ with every call — `heat_ = (img - min) / (max - min)` evaluates into the member's
  // existing buffer when size and type match (cv::Mat::create returns early), whoever else shares it — and so does this: the
  // buffers are created once, re-created (and re-aimed) only when the caller released or replaced a member.  false: the library's
  // buffers and deep copies, as before (no heat maps, or page-locking refused).
  bool aimMaps() {
    if (!with_heat_ || aim_refused_) return false;
    auto fits = [&](const cv::Mat &m) {
      return !m.empty() && m.type() == CV_32FC1 && m.rows == height_ && m.cols == width_ && m.step == width_ * sizeof(float);
    };
    if (!fits(heat_)) heat_.create(height_, width_, CV_32FC1);
    if (!lazy_ && !fits(heat_inv_)) heat_inv_.create(height_, width_, CV_32FC1);
    float *a = reinterpret_cast<float *>(heat_.data), *b = lazy_ ? nullptr : reinterpret_cast<float *>(heat_inv_.data);
    if (a == aimed_heat_ && b == aimed_inv_) return true;
    if (spfe_set_map_buffers(h_, a, b) != SPFE_OK) {
      (void)spfe_set_map_buffers(h_, nullptr, nullptr);
      aimed_heat_ = aimed_inv_ = nullptr;
      aim_refused_ = true;
      return false;
    }
    aimed_heat_ = a;
    aimed_inv_ = b;
    return true;
  }
  // off: the maps through the library's buffers and deep copies into the members (what a refused page-lock falls back to)
  void setMapsInPlace(bool on) {
    if (!on && !aim_refused_) { (void)spfe_set_map_buffers(h_, nullptr, nullptr); aimed_heat_ = aimed_inv_ = nullptr; }
    aim_refused_ = !on;
  }
  const std::vector<Vec2f> &getCov() const { return cov2_; }
  const std::vector<Vec2f> &getCov2Inv() const { return cov2_inv_; }
  int status() const { return status_; }
  spfe_handle handle() const { return h_; }

  cv::Mat semi_dust_, dense_dust_;
  cv::Mat mask_, heat_, heat_inv_;  // mask_ is never written (nor by the reference)
  cv::Mat occ_grid_;

 protected:
  std::vector<Vec2f> cov2_, cov2_inv_;
  spfe_handle h_ = nullptr;
  int height_, width_, status_ = 0;
  bool with_heat_ = true, lazy_ = false, aim_refused_ = false;
  float *aimed_heat_ = nullptr, *aimed_inv_ = nullptr;   // what spfe_set_map_buffers was last given
  size_t raw_row_bytes_ = 0;
  int raw_rows_ = -1;
};

}  // namespace spfe
