// spfe_widen.hip — C ABI of the rows SURVEY.md §8f widens into, on the handle's buffers and streams: direct "dust"
// alignment (optimizer_dust.cpp:170-294), input staging (data_loader.cc:485-521), descriptor matching and patch-wise
// association (sp_matcher.cpp:1636-1674, tracker_dust.cpp:113-172).
#include "spfe_host.h"
using namespace spfe_host;

extern "C" {

// ---- direct "dust" alignment (optimizer_dust.cpp:170-294) -----------------
namespace {
int dust_check(spfe_handle h, int n, const spfe_dust_params *prm) {
  if (n < 0 || n > SPFE_DUST_MAX_POINTS) return fail(SPFE_EINVAL, "n_points %d not in [0, %d]", n, SPFE_DUST_MAX_POINTS);
  if (prm->max_iterations < 0 || prm->max_iterations > 1000) return fail(SPFE_EINVAL, "max_iterations %d", prm->max_iterations);
  if (!(prm->huber_delta > 0)) return fail(SPFE_EINVAL, "huber_delta must be positive");
  if (spfe::dust_lds_bytes(h->hc, h->wc) > 160 * 1024) return fail(SPFE_EINVAL, "dust map %dx%d too large for LDS", h->wc, h->hc);
  return SPFE_OK;
}
int dust_launch(spfe_handle h, const float *d_dust, const float *d_pts, int n, const float *d_T,
                const spfe_dust_params *prm, uint8_t *d_out, hipStream_t s, int nframes = 1, size_t dust_stride = 0,
                const int *d_n = nullptr) {
  spfe::DustArgs a{};
  a.nframes = nframes; a.dust_stride = dust_stride; a.pts_stride = (size_t)SPFE_DUST_MAX_POINTS * 12; a.pose_stride = 64;
  a.out_stride = SPFE_DUST_OUT_BYTES; a.n_dev = d_n;
  a.dust = d_dust; a.hc = h->hc; a.wc = h->wc; a.pts = d_pts; a.n = n; a.Tcw_in = d_T;
  a.fx = prm->fx; a.fy = prm->fy; a.cx = prm->cx; a.cy = prm->cy;
  a.max_iterations = prm->max_iterations; a.delta = prm->huber_delta; a.inlier_chi2 = prm->inlier_chi2;
  a.Tcw_out = reinterpret_cast<float *>(d_out);
  a.counts = reinterpret_cast<int *>(d_out + 64);
  a.uv = reinterpret_cast<float *>(d_out + SPFE_DUST_OFF_UV);
  a.inlier = d_out + SPFE_DUST_OFF_INLIER;
  HIP_TRY(spfe::launch_dust_align(a, s));
  return SPFE_OK;
}
}  // namespace

int spfe_align_dust_record_device(spfe_handle h, const void *d_record, const void *d_points_xyz, int n,
                                  const void *d_Tcw, const spfe_dust_params *prm, void *d_out, void *stream) {
  if (!h || !d_record || !d_Tcw || !prm || !d_out || (n > 0 && !d_points_xyz)) return fail(SPFE_EINVAL, "null argument");
  int rc = dust_check(h, n, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const float *d_dust = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(d_record) + h->rl.off_dd);
  return dust_launch(h, d_dust, reinterpret_cast<const float *>(d_points_xyz), n, reinterpret_cast<const float *>(d_Tcw),
                     prm, reinterpret_cast<uint8_t *>(d_out), s);
}

int spfe_align_dust_batch_device(spfe_handle h, const void *d_records, int n_frames, const void *d_points_xyz,
                                 const void *d_n_points, const void *d_Tcw, const spfe_dust_params *prm, void *d_out,
                                 void *stream) {
  if (!h || !d_records || !d_Tcw || !prm || !d_out || !d_points_xyz || !d_n_points) return fail(SPFE_EINVAL, "null argument");
  if (n_frames < 1 || n_frames > 65535) return fail(SPFE_EINVAL, "n_frames %d", n_frames);
  int rc = dust_check(h, 0, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const float *d_dust = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(d_records) + h->rl.off_dd);
  return dust_launch(h, d_dust, reinterpret_cast<const float *>(d_points_xyz), 0, reinterpret_cast<const float *>(d_Tcw), prm,
                     reinterpret_cast<uint8_t *>(d_out), s, n_frames, h->rl.bytes, reinterpret_cast<const int *>(d_n_points));
}

int spfe_align_dust(spfe_handle h, const float *dense_dust, const float *points_xyz, int n, const float *Tcw,
                    const spfe_dust_params *prm, float *Tcw_out, uint8_t *inlier, float *proj_uv, int *n_inlier,
                    int *iterations) {
  if (!h || !dense_dust || !Tcw || !prm || !Tcw_out || (n > 0 && !points_xyz)) return fail(SPFE_EINVAL, "null argument");
  int rc = dust_check(h, n, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t map_b = (size_t)h->C * 4, pts_b = (size_t)SPFE_DUST_MAX_POINTS * 12, out_off = map_b + pts_b + 64;
  if (!h->dust_scratch) {
    if ((rc = dev_alloc(h, &h->dust_scratch, out_off + SPFE_DUST_OUT_BYTES))) return rc;
    if ((rc = host_alloc(h, &h->dust_host, (size_t)SPFE_DUST_OUT_BYTES))) return rc;
  }
  hipStream_t s = h->stream;
  uint8_t *d = h->dust_scratch;
  HIP_TRY(hipMemcpyAsync(d, dense_dust, map_b, hipMemcpyHostToDevice, s));
  if (n > 0) HIP_TRY(hipMemcpyAsync(d + map_b, points_xyz, (size_t)n * 12, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + map_b + pts_b, Tcw, 64, hipMemcpyHostToDevice, s));
  rc = dust_launch(h, reinterpret_cast<const float *>(d), reinterpret_cast<const float *>(d + map_b), n,
                   reinterpret_cast<const float *>(d + map_b + pts_b), prm, d + out_off, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->dust_host, d + out_off, SPFE_DUST_OUT_BYTES, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(Tcw_out, h->dust_host, 64);
  const int *cnt = reinterpret_cast<const int *>(h->dust_host + 64);
  if (n_inlier) *n_inlier = cnt[0];
  if (iterations) *iterations = cnt[1];
  if (proj_uv && n > 0) memcpy(proj_uv, h->dust_host + SPFE_DUST_OFF_UV, (size_t)n * 8);
  if (inlier && n > 0) memcpy(inlier, h->dust_host + SPFE_DUST_OFF_INLIER, (size_t)n);
  return SPFE_OK;
}

// ---- pipelined host path ------------------------------------------------------------------------
// The host boundary of SPExtractor::operator() (upload sp_extractor.cpp:379-390, six synchronous D2H copies

// ---- input staging ------------------------------------------------------
int spfe_set_staging(spfe_handle h, const spfe_staging *st) {
  if (!h || !st) return fail(SPFE_EINVAL, "null argument");
  if (st->channels != 1 && st->channels != 3 && st->channels != 4)
    return fail(SPFE_EINVAL, "staging: %d channels unsupported (1, 3, 4)", st->channels);
  if (st->src_height < h->H || st->src_width < h->W)
    return fail(SPFE_EINVAL, "staging: source %dx%d smaller than the extractor's %dx%d (system.cpp:160 crop)",
                st->src_width, st->src_height, h->W, h->H);
  if (st->src_height > 32767 || st->src_width > 32767) return fail(SPFE_EINVAL, "staging: source too large");
  if ((st->map_x == nullptr) != (st->map_y == nullptr)) return fail(SPFE_EINVAL, "staging: one map is null");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  for (void **p : {(void **)&h->d_map_x, (void **)&h->d_map_y, (void **)&h->d_raw})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->h_raw) { (void)hipHostFree(h->h_raw); h->h_raw = nullptr; }
  h->st_set = false;
  const size_t npx = (size_t)st->src_height * st->src_width;
  if (st->map_x) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_map_x), npx * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_map_y), npx * 4));
    HIP_TRY(hipMemcpy(h->d_map_x, st->map_x, npx * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_map_y, st->map_y, npx * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_raw), (size_t)h->B * npx * st->channels));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->h_raw), (size_t)h->B * npx * st->channels,
                        hipHostMallocDefault));
  h->st = *st;
  h->st.map_x = h->st.map_y = nullptr;  // the caller's arrays are not kept
  h->st_set = true;
  return SPFE_OK;
}

namespace {
int enqueue_stage(spfe_handle h, const uint8_t *d_src, int n, uint8_t *d_gray, hipStream_t s) {
  spfe::StageParams p{};
  p.src = d_src;
  p.src_stride = h->st.src_width * h->st.channels;
  p.src_frame_bytes = (size_t)h->st.src_height * p.src_stride;
  p.src_h = h->st.src_height;
  p.src_w = h->st.src_width;
  p.map_x = h->d_map_x;
  p.map_y = h->d_map_y;
  p.rgb = h->st.rgb;
  p.gray = d_gray;
  p.H = h->H;
  p.W = h->W;
  HIP_TRY(spfe::launch_stage_input(p, h->st.channels, n, s));
  return SPFE_OK;
}
}  // namespace

int spfe_stage_batch_device(spfe_handle h, const void *d_src, int n, void *d_gray, void *stream) {
  if (!h || !d_gray) return fail(SPFE_EINVAL, "null argument");
  if (!h->st_set) return fail(SPFE_EINVAL, "spfe_set_staging has not been called");
  if (!d_src) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  return enqueue_stage(h, reinterpret_cast<const uint8_t *>(d_src), n, reinterpret_cast<uint8_t *>(d_gray), s);
}

int spfe_extract_batch_staged(spfe_handle h, const uint8_t *const *srcs, int stride, int n, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  if (!h->st_set) return fail(SPFE_EINVAL, "spfe_set_staging has not been called");
  if (!srcs) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  const int row = h->st.src_width * h->st.channels;
  if (stride < row) return fail(SPFE_EINVAL, "stride %d smaller than a source row (%d bytes)", stride, row);
  const size_t frame = (size_t)h->st.src_height * row;
  for (int i = 0; i < n; ++i) {
    if (!srcs[i]) return fail(SPFE_EEMPTY, "input image is empty");  // sp_extractor.cpp:364-365
    for (int y = 0; y < h->st.src_height; ++y)
      memcpy(h->h_raw + i * frame + (size_t)y * row, srcs[i] + (size_t)y * stride, row);
  }
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_raw, h->h_raw, n * frame, hipMemcpyHostToDevice, s));
  int rc = enqueue_stage(h, h->d_raw, n, h->d_img, s);
  if (rc) return rc;
  rc = enqueue(h, h->d_img, n, h->d_records, s);
  if (rc) return rc;
  return finish_host(h, n, outs);
}

int spfe_extract_staged(spfe_handle h, const uint8_t *src, int stride, spfe_result *out) {
  if (!src) return fail(SPFE_EEMPTY, "input image is empty");
  const uint8_t *one[1] = {src};
  return spfe_extract_batch_staged(h, one, stride, 1, out);
}

// ---- patch-wise association (tracker_dust.cpp:113-172) -------------------------------------------
namespace {
constexpr int kPatchMax = 4096;
int patch_scratch(spfe_handle h) {
  if (h->p_cidx) return SPFE_OK;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->p_cidx), (size_t)kPatchMax * 4 * sizeof(int)));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->p_cdist), (size_t)kPatchMax * 4 * sizeof(float)));
  return SPFE_OK;
}
}  // namespace

int spfe_match_patches_record_device(spfe_handle h, const void *d_mp_desc, const void *d_mp_uv, int n_points,
                                     const void *d_record, float max_dist, void *d_kp_idx, void *stream) {
  if (!h || !d_record || !d_kp_idx) return fail(SPFE_EINVAL, "null argument");
  if (n_points < 0 || n_points > kPatchMax) return fail(SPFE_EINVAL, "n_points %d not in [0, %d]", n_points, kPatchMax);
  if (n_points == 0) return SPFE_OK;
  if (!d_mp_desc || !d_mp_uv) return fail(SPFE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = patch_scratch(h);
  if (rc) return rc;
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const uint8_t *rec = reinterpret_cast<const uint8_t *>(d_record);
  spfe::PatchArgs a{};
  a.mp_desc = reinterpret_cast<const float *>(d_mp_desc);
  a.mp_uv = reinterpret_cast<const float *>(d_mp_uv);
  a.n_points = n_points;
  a.occ = reinterpret_cast<const int16_t *>(rec + h->rl.off_occ);
  a.hc = h->hc; a.wc = h->wc;
  a.kp_desc = reinterpret_cast<const float *>(rec + h->rl.off_desc);
  a.kp_desc_bf16 = h->rl.desc_bf16;
  a.k_ptr = reinterpret_cast<const int *>(rec + h->rl.off_hdr);
  a.k_imm = 0;
  HIP_TRY(spfe::launch_match_patches(a, h->kmax, max_dist, h->p_cidx, h->p_cdist,
                                     reinterpret_cast<int32_t *>(d_kp_idx), s));
  return SPFE_OK;
}

int spfe_track_dust_record_device(spfe_handle h, const void *d_record, const void *d_points_xyz, const void *d_mp_desc, int n,
                                  const void *d_Tcw, const spfe_dust_params *prm, int min_inliers, float max_dist,
                                  void *d_dust_out, void *d_kp_idx, void *stream) {
  if (!h || !d_record || !d_Tcw || !prm || !d_dust_out || !d_kp_idx || (n > 0 && (!d_points_xyz || !d_mp_desc)))
    return fail(SPFE_EINVAL, "null argument");
  int rc = dust_check(h, n, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if ((rc = patch_scratch(h))) return rc;
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const uint8_t *rec = reinterpret_cast<const uint8_t *>(d_record);
  uint8_t *dout = reinterpret_cast<uint8_t *>(d_dust_out);
  // PoseOptimizationDust(&mCurrentFrame, mps_for_track, is_visible)   tracker_dust.cpp:92-94
  rc = dust_launch(h, reinterpret_cast<const float *>(rec + h->rl.off_dd), reinterpret_cast<const float *>(d_points_xyz), n,
                   reinterpret_cast<const float *>(d_Tcw), prm, dout, s);
  if (rc || n == 0) return rc;
  // the patch-wise association of the in_view points at their dust_proj_u / v   :113-172, on the same stream: the
  // projections, the flags and n_inlier are read where the alignment left them
  spfe::PatchArgs a{};
  a.mp_desc = reinterpret_cast<const float *>(d_mp_desc);
  a.mp_uv = reinterpret_cast<const float *>(dout + SPFE_DUST_OFF_UV);
  a.n_points = n;
  a.occ = reinterpret_cast<const int16_t *>(rec + h->rl.off_occ);
  a.hc = h->hc; a.wc = h->wc;
  a.kp_desc = reinterpret_cast<const float *>(rec + h->rl.off_desc);
  a.kp_desc_bf16 = h->rl.desc_bf16;
  a.k_ptr = reinterpret_cast<const int *>(rec + h->rl.off_hdr);
  a.k_imm = 0;
  a.in_view = dout + SPFE_DUST_OFF_INLIER;
  a.gate_ptr = reinterpret_cast<const int *>(dout + 64);
  a.gate_min = min_inliers;
  HIP_TRY(spfe::launch_match_patches(a, h->kmax, max_dist, h->p_cidx, h->p_cdist, reinterpret_cast<int32_t *>(d_kp_idx), s));
  return SPFE_OK;
}

int spfe_match_patches(spfe_handle h, const float *mp_desc, const float *mp_uv, int n_points,
                       const int16_t *occ_grid, const float *kp_desc, int n_keypoints, float max_dist,
                       int32_t *kp_idx) {
  if (!h || !kp_idx) return fail(SPFE_EINVAL, "null argument");
  if (n_points < 0 || n_points > kPatchMax) return fail(SPFE_EINVAL, "n_points %d not in [0, %d]", n_points, kPatchMax);
  if (n_keypoints < 0 || n_keypoints > 32767) return fail(SPFE_EINVAL, "n_keypoints %d out of range", n_keypoints);
  for (int i = 0; i < n_points; ++i) kp_idx[i] = -1;
  if (n_points == 0 || n_keypoints == 0) return SPFE_OK;
  if (!mp_desc || !mp_uv || !occ_grid || !kp_desc) return fail(SPFE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = patch_scratch(h);
  if (rc) return rc;
  const size_t cells = (size_t)h->hc * h->wc;
  const size_t o_mp = 0, o_uv = o_mp + (size_t)n_points * 1024, o_occ = align_up(o_uv + (size_t)n_points * 8, 16),
               o_kp = align_up(o_occ + cells * 2, 16), o_out = o_kp + (size_t)n_keypoints * 1024,
               total = o_out + (size_t)n_points * 4;
  if (total > h->p_stage_bytes) {
    HIP_TRY(hipDeviceSynchronize());
    if (h->p_stage) (void)hipFree(h->p_stage);
    h->p_stage = nullptr;
    h->p_stage_bytes = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->p_stage), total));
    h->p_stage_bytes = total;
  }
  hipStream_t s = h->stream;
  uint8_t *d = h->p_stage;
  HIP_TRY(hipMemcpyAsync(d + o_mp, mp_desc, (size_t)n_points * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + o_uv, mp_uv, (size_t)n_points * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + o_occ, occ_grid, cells * 2, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + o_kp, kp_desc, (size_t)n_keypoints * 1024, hipMemcpyHostToDevice, s));
  spfe::PatchArgs a{};
  a.mp_desc = reinterpret_cast<const float *>(d + o_mp);
  a.mp_uv = reinterpret_cast<const float *>(d + o_uv);
  a.n_points = n_points;
  a.occ = reinterpret_cast<const int16_t *>(d + o_occ);
  a.hc = h->hc; a.wc = h->wc;
  a.kp_desc = reinterpret_cast<const float *>(d + o_kp);
  a.k_ptr = nullptr;
  a.k_imm = n_keypoints;
  HIP_TRY(spfe::launch_match_patches(a, n_keypoints, max_dist, h->p_cidx, h->p_cdist,
                                     reinterpret_cast<int32_t *>(d + o_out), s));
  HIP_TRY(hipMemcpyAsync(kp_idx, d + o_out, (size_t)n_points * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SPFE_OK;
}

// ---- descriptor matching ------------------------------------------------
namespace {
int match_scratch(spfe_handle h, int pairs, int cap) {
  if (pairs <= h->m_pairs && cap <= h->m_cap) return SPFE_OK;
  pairs = std::max(pairs, h->m_pairs);
  cap = std::max(cap, h->m_cap);
  HIP_TRY(hipDeviceSynchronize());
  if (h->m_best_t) (void)hipFree(h->m_best_t);
  if (h->m_best_q) (void)hipFree(h->m_best_q);
  h->m_best_t = h->m_best_q = nullptr;
  h->m_pairs = h->m_cap = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_best_t), (size_t)pairs * cap * 8));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_best_q), (size_t)pairs * cap * 8));
  h->m_pairs = pairs;
  h->m_cap = cap;
  return SPFE_OK;
}
constexpr size_t kMatchHdr = 16;  // staging block of the host API: int32 count, pad, then rows
}  // namespace

size_t spfe_match_out_bytes(spfe_handle h) { return h ? (size_t)h->kmax * 8 : 0; }

int spfe_match_records_device(spfe_handle h, const void *d_query_records, const void *d_train_records, int n_pairs,
                              int cross_check, void *d_out, void *stream) {
  if (!h || !d_query_records || !d_train_records || !d_out) return fail(SPFE_EINVAL, "null argument");
  if (n_pairs < 1) return fail(SPFE_EINVAL, "n_pairs %d must be >= 1", n_pairs);
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = match_scratch(h, n_pairs, h->kmax);
  if (rc) return rc;
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  spfe::MatchSide q{reinterpret_cast<const uint8_t *>(d_query_records), h->rl.bytes, h->rl.off_hdr, h->rl.off_desc,
                    h->kmax};
  spfe::MatchSide t{reinterpret_cast<const uint8_t *>(d_train_records), h->rl.bytes, h->rl.off_hdr, h->rl.off_desc,
                    h->kmax};
  q.desc_bf16 = t.desc_bf16 = h->rl.desc_bf16;   // (records made with SPFE_FLAG_DESC_BF16: bf16 rows, widened on load)
  HIP_TRY(spfe::launch_match(q, t, n_pairs, cross_check != 0, h->m_best_t, h->m_best_q,
                             reinterpret_cast<uint8_t *>(d_out), (size_t)h->kmax * 8, s));
  return SPFE_OK;
}

int spfe_match(spfe_handle h, const float *query, int n_query, const float *train, int n_train, int cross_check,
               int32_t *train_idx, float *distance) {
  if (!h || !train_idx || !distance) return fail(SPFE_EINVAL, "null argument");
  if (n_query < 0 || n_train < 0) return fail(SPFE_EINVAL, "negative descriptor count");
  if ((n_query && !query) || (n_train && !train)) return fail(SPFE_EINVAL, "null descriptor array");
  for (int i = 0; i < n_query; ++i) { train_idx[i] = -1; distance[i] = FLT_MAX; }
  if (n_query == 0 || n_train == 0) return SPFE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int cap = std::max(n_query, n_train);
  if (cap > h->m_host_cap) {
    HIP_TRY(hipDeviceSynchronize());
    for (uint8_t **p : {&h->m_stage_q, &h->m_stage_t, &h->m_out, &h->m_out2})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->m_host_cap = 0;
    const int want = std::max(cap, h->kmax);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_q), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_t), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_out), (size_t)want * 8));
    h->m_host_cap = want;
  }
  int rc = match_scratch(h, 1, std::max(cap, h->kmax));
  if (rc) return rc;
  hipStream_t s = h->stream;
  const int32_t hq[4] = {n_query, 0, 0, 0}, ht[4] = {n_train, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h->m_stage_q, hq, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t, ht, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_q + kMatchHdr, query, (size_t)n_query * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t + kMatchHdr, train, (size_t)n_train * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // hq / ht live on this frame
  spfe::MatchSide q{h->m_stage_q, 0, 0, kMatchHdr, n_query};
  spfe::MatchSide t{h->m_stage_t, 0, 0, kMatchHdr, n_train};
  HIP_TRY(spfe::launch_match(q, t, 1, cross_check != 0, h->m_best_t, h->m_best_q, h->m_out, 0, s));
  HIP_TRY(hipMemcpyAsync(train_idx, h->m_out, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(distance, h->m_out + (size_t)n_query * 4, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SPFE_OK;
}

// knnMatch(query, matches, 2): the two nearest train rows of every query, exactly (the FLANN kd-tree the
// reference builds for this is approximate and randomised)
int spfe_match_knn2(spfe_handle h, const float *query, int n_query, const float *train, int n_train,
                    int32_t *train_idx, float *distance) {
  if (!h || !train_idx || !distance) return fail(SPFE_EINVAL, "null argument");
  if (n_query < 0 || n_train < 0) return fail(SPFE_EINVAL, "negative descriptor count");
  if ((n_query && !query) || (n_train && !train)) return fail(SPFE_EINVAL, "null descriptor array");
  for (int i = 0; i < 2 * n_query; ++i) { train_idx[i] = -1; distance[i] = FLT_MAX; }
  if (n_query == 0 || n_train == 0) return SPFE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int cap = std::max(n_query, n_train);
  if (cap > h->m_host_cap || !h->m_out2) {
    HIP_TRY(hipDeviceSynchronize());
    for (uint8_t **p : {&h->m_stage_q, &h->m_stage_t, &h->m_out, &h->m_out2})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->m_host_cap = 0;
    const int want = std::max(cap, h->kmax);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_q), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_t), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_out), (size_t)want * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_out2), (size_t)want * 16));
    h->m_host_cap = want;
  }
  int rc = match_scratch(h, 1, std::max(cap, h->kmax));
  if (rc) return rc;
  hipStream_t s = h->stream;
  const int32_t hq[4] = {n_query, 0, 0, 0}, ht[4] = {n_train, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h->m_stage_q, hq, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t, ht, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_q + kMatchHdr, query, (size_t)n_query * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t + kMatchHdr, train, (size_t)n_train * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // hq / ht live on this frame
  spfe::MatchSide q{h->m_stage_q, 0, 0, kMatchHdr, n_query};
  spfe::MatchSide t{h->m_stage_t, 0, 0, kMatchHdr, n_train};
  // scratch: best_q holds the first neighbours, best_t (>= cap entries) the second
  HIP_TRY(spfe::launch_match_knn2(q, t, 1, h->m_best_q, h->m_best_t, h->m_out2, 0, s));
  // device layout idx1 | dist1 | idx2 | dist2 -> host layout [n_query][2]
  std::vector<int32_t> hi(2 * (size_t)n_query);
  std::vector<float> hd(2 * (size_t)n_query);
  HIP_TRY(hipMemcpyAsync(hi.data(), h->m_out2, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hd.data(), h->m_out2 + (size_t)n_query * 4, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hi.data() + n_query, h->m_out2 + (size_t)n_query * 8, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hd.data() + n_query, h->m_out2 + (size_t)n_query * 12, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n_query; ++i) {
    train_idx[2 * i] = hi[i]; train_idx[2 * i + 1] = hi[n_query + i];
    distance[2 * i] = hd[i]; distance[2 * i + 1] = hd[n_query + i];
  }
  return SPFE_OK;
}

}  // extern "C"
