// spfe_api.hip — the C ABI of the path itself (include/spfe.h): create / destroy, the extract calls, the pipelined host
// path (submit / collect), record views, debug reads and stage timing.  The handle's construction is spfe_pack.hip, the
// launch sequence spfe_schedule.hip, the collective spfe_comm.hip, the widened rows spfe_widen.hip.
#include "spfe_host.h"

namespace spfe_host {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

const char *const kStageNames[NSTAGE] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b",
                                         "conv4a", "conv4b", "convPaDa", "convPb", "convDb", "tail",
                                         "select", "post_side", "total"};  // post_side = select + heat_norm + desc + cov (side stream)

void view_record(const spfe_handle h, const uint8_t *rec, const float *heat, const float *heat_inv,
                 spfe_result *out) {
  const spfe::RecordLayout &r = h->rl;
  const int *hdr = reinterpret_cast<const int *>(rec + r.off_hdr);
  out->K = hdr[0];
  out->n_candidates = hdr[1];
  out->status = hdr[2];
  out->kp_xy = reinterpret_cast<const float *>(rec + r.off_xy);
  out->kp_response = reinterpret_cast<const float *>(rec + r.off_resp);
  out->desc = r.desc_bf16 ? nullptr : reinterpret_cast<const float *>(rec + r.off_desc);
  out->desc_bf16 = r.desc_bf16 ? reinterpret_cast<const uint16_t *>(rec + r.off_desc) : nullptr;
  out->cov2 = reinterpret_cast<const float *>(rec + r.off_cov);
  out->cov2_inv = reinterpret_cast<const float *>(rec + r.off_cinv);
  out->occ_grid = reinterpret_cast<const int16_t *>(rec + r.off_occ);
  out->dense_dust = reinterpret_cast<const float *>(rec + r.off_dd);
  out->semi_dust = reinterpret_cast<const float *>(rec + r.off_sd);
  out->heat = heat;
  out->heat_inv = heat_inv;
}

}  // namespace spfe_host
using namespace spfe_host;

extern "C" {

const char *spfe_last_error(void) { return g_err.c_str(); }
const char *spfe_version(void) { return "spfe 0.5 abi 5 (gfx950, f32-mfma + bf16-mfma)"; }
int spfe_abi_version(void) { return SPFE_ABI_VERSION; }
int spfe_check_abi(int abi_version, size_t sizeof_config, size_t sizeof_result, size_t sizeof_record_layout) {
  if (abi_version != SPFE_ABI_VERSION)
    return fail(SPFE_EINVAL, "ABI mismatch: the caller was built against spfe.h ABI %d, this library is ABI %d", abi_version, SPFE_ABI_VERSION);
  if (sizeof_config != sizeof(spfe_config) || sizeof_result != sizeof(spfe_result) || sizeof_record_layout != sizeof(spfe_record_layout))
    return fail(SPFE_EINVAL, "ABI mismatch: struct sizes config %zu / result %zu / record_layout %zu, library %zu / %zu / %zu",
                sizeof_config, sizeof_result, sizeof_record_layout, sizeof(spfe_config), sizeof(spfe_result), sizeof(spfe_record_layout));
  return SPFE_OK;
}
const char *spfe_stage_name(int i) { return (i >= 0 && i < NSTAGE) ? kStageNames[i] : ""; }

// The twin of a handle whose pipelined calls are to alternate between two sets of buffers (spfe_host.h: two side chains in
// flight) — where the workload (or SPFE_TWO_CHAINS) says so.  spfe_create for SPFE_FLAG_ASYNC_COV handles, the first
// spfe_submit_batch for the others.
// The twin is an optimisation: when it cannot be built (a second full set of buffers — several GB for bf16 3840x2160) the
// handle carries on with one side chain; spfe_debug_read("two_chains") says 0, ("twin_failed") 1, and the error text stays
// readable through spfe_last_error() until the next failing call.  Built from the handle's OWN copy of the weight blob and
// its own switches (spfe_host.h: blob), never from the caller's pointers.
static int make_twin(spfe_handle h) {
  if (h->twin || h->is_twin || h->twin_failed) return SPFE_OK;
  // By workload: where a call's side chain (bounded below by its frame's longest component chain, whatever the batch) outlasts
  // the next call's convolutions — bf16 frames of >= 10,000 cells in SHORT calls (2+ frames, fewer than 50,000 cells a call:
  // 1280x720 x 2: +32 %) and frames beyond select_kernel (3840x2160 x 1: +3.7 %).  Longer calls hide one chain completely and a
  // second set of buffers only costs them cache (round 6: 1280x720 x 4 / 6 / 8: -2.2 / -0.7 / -0.8 %, 1920x1080 x 1: -7 %).
  const bool by_workload = h->bf16 && h->C >= 10000 && (h->C > 65535 || (h->B >= 2 && (long)h->B * h->C < 50000));
  if (!(h->two_chains_env >= 0 ? h->two_chains_env > 0 : by_workload)) { h->blob = std::vector<float>(); return SPFE_OK; }
  h->twin = new spfe_handle_s();
  h->twin->is_twin = true;
  // (SPFE_TWO_CHAINS=99, tests: a twin is wanted and its build fails — the failure path without exhausting 288 GB)
  const int rc = h->two_chains_env == 99 ? fail(SPFE_EHIP, "twin build made to fail (SPFE_TWO_CHAINS=99)") : build(h->twin, &h->cfg, h);
  if (rc) {
    std::string keep = g_err;
    spfe_destroy(h->twin);
    h->twin = nullptr;
    h->twin_failed = true;
    (void)hipGetLastError();   // (an out-of-memory stays sticky otherwise)
    g_err = "two side chains unavailable, continuing with one: " + keep;
  }
  h->blob = std::vector<float>();   // (no further use: a handle has at most one twin)
  return SPFE_OK;
}

int spfe_create(const spfe_config *cfg, spfe_handle *out) {
  if (!cfg || !out) return fail(SPFE_EINVAL, "spfe_create: null argument");
  *out = nullptr;
  if (cfg->height <= 0 || cfg->width <= 0 || cfg->height % 8 || cfg->width % 8)
    return fail(SPFE_EINVAL, "image size %dx%d must be positive multiples of 8 (sp_extractor.cpp:70)",
                cfg->width, cfg->height);
  if (cfg->height < 16 || cfg->width < 16)
    return fail(SPFE_EINVAL, "image size %dx%d too small", cfg->width, cfg->height);
  if (cfg->num_features < 1 || cfg->num_features > 10000)
    return fail(SPFE_EINVAL, "num_features %d out of range (1..10000: the covariance link stage keeps 16 bytes per "
                             "keypoint in one workgroup's 160 KB of LDS)", cfg->num_features);
  if (cfg->max_batch < 1) return fail(SPFE_EINVAL, "max_batch must be >= 1");
  if (cfg->precision != SPFE_PRECISION_F32 && cfg->precision != SPFE_PRECISION_BF16)
    return fail(SPFE_EINVAL, "unsupported precision %d", cfg->precision);
  {
    // Frame size limits.  The convolutions address a frame's activations with 32-bit byte offsets below 2^31 (0x80000000 and
    // above mean "outside the image": conv_f32.hip SPFE_OOB; the bf16 kernels alike), and the largest activation is conv1a's
    // output, 64 channels a pixel: H W 256 bytes in f32 mode (3840x2160 = 2,123,366,400 fits; 4096x2304 does not), H W 128 in
    // bf16 mode.  The selection handles 65,535 cells as select_kernel and 262,143 as select_huge_kernel (tested against the
    // oracle at 3840x2160 in both precisions; bf16: logits within tolerance, everything behind them exact given them).
    const size_t cells = (size_t)(cfg->height / 8) * (cfg->width / 8);
    const size_t act0_bytes = (size_t)cfg->height * cfg->width * 64 * (cfg->precision == SPFE_PRECISION_BF16 ? 2 : 4);
    const size_t max_cells = spfe::select_huge_max_cells();
    if (cells > max_cells || act0_bytes >= 0x80000000ull ||
        (cells <= spfe::select_max_cells() ? spfe::select_lds_bytes(cfg->height, cfg->width) : spfe::select_huge_lds_bytes(cfg->height, cfg->width)) > 160 * 1024)
      return fail(SPFE_EINVAL, "image %dx%d is too large: %zu cells (limit %zu in this precision) / %zu bytes of first-layer "
                               "activations per frame (limit 2^31: 32-bit buffer offsets); 3840x2160 fits",
                  cfg->width, cfg->height, cells, max_cells, act0_bytes);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SPFE_EHIP, "no HIP device available (libspfe has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(SPFE_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
  spfe_handle h = new spfe_handle_s();
  int rc = build(h, cfg);
  if (!rc && (cfg->flags & SPFE_FLAG_ASYNC_COV)) rc = make_twin(h);   // two side chains in flight (spfe_host.h)
  if (rc) {
    std::string keep = g_err;
    spfe_destroy(h);
    g_err = keep;
    return rc;
  }
  *out = h;
  return SPFE_OK;
}

void spfe_destroy(spfe_handle h) {
  if (!h) return;
  if (h->twin) { spfe_destroy(h->twin); h->twin = nullptr; }
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->side) (void)hipStreamSynchronize(h->side);
  for (hipStream_t c : h->conv2_pool) (void)hipStreamSynchronize(c);
  for (int i = 0; i < spfe_handle_s::NTICKET; ++i) {
    if (h->ev_post[i]) (void)hipEventDestroy(h->ev_post[i]);
    if (h->ev_cov[i]) (void)hipEventDestroy(h->ev_cov[i]);
  }
  if (h->ev_desc) (void)hipEventDestroy(h->ev_desc);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (hipStream_t c : h->conv2_pool) (void)hipStreamDestroy(c);
  if (h->ev_db) (void)hipEventDestroy(h->ev_db);
  if (h->ev_sel) (void)hipEventDestroy(h->ev_sel);
  for (int i = 0; i < 2; ++i) if (h->ev_dbs[i]) (void)hipEventDestroy(h->ev_dbs[i]);
  (void)spfe_comm_destroy(h);
  for (auto &ps : h->pipe) {
    if (ps.ev_h2d) (void)hipEventDestroy(ps.ev_h2d);
    if (ps.ev_done) (void)hipEventDestroy(ps.ev_done);
  }
  if (h->s_heat) { (void)hipStreamSynchronize(h->s_heat); (void)hipStreamDestroy(h->s_heat); }
  if (h->usr_heat) (void)hipHostUnregister(h->usr_heat);
  if (h->usr_heat_inv) (void)hipHostUnregister(h->usr_heat_inv);
  if (h->ev_heat) (void)hipEventDestroy(h->ev_heat);
  if (h->ev_heat_copied) (void)hipEventDestroy(h->ev_heat_copied);
  if (h->ev_heat_copied1) (void)hipEventDestroy(h->ev_heat_copied1);
  if (h->s_h2d) { (void)hipStreamSynchronize(h->s_h2d); (void)hipStreamDestroy(h->s_h2d); }
  if (h->s_d2h) { (void)hipStreamSynchronize(h->s_d2h); (void)hipStreamDestroy(h->s_d2h); }
  if (h->side) (void)hipStreamDestroy(h->side);
  for (void *p : {(void *)h->d_map_x, (void *)h->d_map_y, (void *)h->d_raw})
    if (p) (void)hipFree(p);
  if (h->h_raw) (void)hipHostFree(h->h_raw);
  for (void *p : {(void *)h->p_cidx, (void *)h->p_cdist, (void *)h->p_stage})
    if (p) (void)hipFree(p);
  for (void *p : {(void *)h->m_best_t, (void *)h->m_best_q, (void *)h->m_stage_q, (void *)h->m_stage_t,
                  (void *)h->m_out, (void *)h->m_out2})
    if (p) (void)hipFree(p);
  for (void *p : h->dev_allocs) (void)hipFree(p);
  for (void *p : h->host_allocs) (void)hipHostFree(p);
  for (auto &e : h->evpool)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int spfe_get_record_layout(spfe_handle h, spfe_record_layout *o) {
  if (!h || !o) return fail(SPFE_EINVAL, "null argument");
  const spfe::RecordLayout &r = h->rl;
  o->bytes = r.bytes; o->kmax = r.kmax; o->off_hdr = r.off_hdr; o->off_xy = r.off_xy;
  o->off_resp = r.off_resp; o->off_cov = r.off_cov; o->off_cinv = r.off_cinv;
  o->off_desc = r.off_desc; o->off_occ = r.off_occ; o->off_dd = r.off_dd; o->off_sd = r.off_sd;
  o->desc_elem_bytes = r.desc_bf16 ? 2 : 4;
  return SPFE_OK;
}

size_t spfe_record_bytes(spfe_handle h) { return h ? h->rl.bytes : 0; }

int spfe_extract_batch_device(spfe_handle h, const void *d_images, int n, void *d_records, void *stream) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (h->open_n) return fail(SPFE_EINVAL, "a call begun by spfe_extract_begin is open: spfe_extract_finish first");
  if (!d_images) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  uint8_t *rec = d_records ? reinterpret_cast<uint8_t *>(d_records) : h->d_records;
  if (h->twin) {   // two side chains in flight: even tickets on this handle, odd ones on its twin (own buffers, own side stream)
    // (a handle whose twin was made by the pipelined host path but whose device calls are synchronous stays on its own buffers)
    spfe_handle t = (h->cfg.flags & SPFE_FLAG_ASYNC_COV) && (h->g_ticket & 1) ? h->twin : h;
    {   // the same record buffer twice in a row: tail_waits() orders a call behind the previous chain of ITS handle only, and the
        // previous call ran on the other of the pair — whose chain (its side stream) may still be writing this buffer
      spfe_handle o = t == h ? h->twin : h;
      if (o->cov_inflight && o->ticket > 0) {
        const int prev = (int)((o->ticket - 1) % spfe_handle_s::NTICKET);
        if (o->rec_of[prev] == rec) HIP_TRY(wait_if_pending(s, o->ev_cov[prev]));
      }
    }
    const int rc = enqueue(t, reinterpret_cast<const uint8_t *>(d_images), n, rec, s);
    if (rc) return rc;
    h->tmap[h->g_ticket % 8] = {t, t->ticket - 1};
    h->g_ticket++;
    return SPFE_OK;
  }
  return enqueue(h, reinterpret_cast<const uint8_t *>(d_images), n, rec, s);
}

int finish_host(spfe_handle h, int n, spfe_result *outs);

int spfe_postprocess(spfe_handle h, const float *semi, const float *coarse, int n, spfe_result *outs) {
  if (!h || !outs || !semi || !coarse) return fail(SPFE_EINVAL, "null argument");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  { const int rcj = settle_join(h, s); if (rcj) return rcj; }
  if (h->desc_recorded) HIP_TRY(hipStreamWaitEvent(s, h->ev_desc, 0));  // previous call's reader of d_coarse
  HIP_TRY(hipMemcpyAsync(h->d_semi, semi, (size_t)n * h->C * SPFE_SEMI_CH * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_coarse, coarse, (size_t)n * h->C * SPFE_DESC_DIM * 4, hipMemcpyHostToDevice, s));
  if (h->timing) h->ev = h->evpool.data() + (size_t)(h->calls % spfe_handle_s::EVSETS) * (NSTAGE + 1);
  if (h->timing) for (int i = 0; i <= 11; ++i) HIP_TRY(hipEventRecord(h->ev[i], s));
  h->calls++;
  h->host_sync_call = true;
  int rc = enqueue_post(h, n, h->d_records, s);
  h->host_sync_call = false;
  if (rc) return rc;
  return finish_host(h, n, outs);
}

int spfe_extract_begin(spfe_handle h, const uint8_t *const *images, int stride, int n) {
  if (!h) return fail(SPFE_EINVAL, "null argument");
  if (h->open_n) return fail(SPFE_EINVAL, "a call of %d frames is open: spfe_extract_finish first", h->open_n);
  if (!images) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  const int H = h->H, W = h->W;
  if (stride < W) return fail(SPFE_EINVAL, "stride %d smaller than width %d", stride, W);
  for (int i = 0; i < n; ++i) {
    if (!images[i]) return fail(SPFE_EEMPTY, "input image is empty");  // sp_extractor.cpp:364-365
    if (stride == W) memcpy(h->h_img + (size_t)i * H * W, images[i], (size_t)H * W);   // (a continuous cv::Mat: one copy)
    else
      for (int y = 0; y < H; ++y)
        memcpy(h->h_img + ((size_t)i * H + y) * W, images[i] + (size_t)y * stride, W);
  }
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_img, h->h_img, (size_t)n * H * W, hipMemcpyHostToDevice, s));
  h->host_sync_call = true;   // (the heat maps may leave ahead of the record: spfe_host.h, s_heat)
  int rc = enqueue(h, h->d_img, n, h->d_records, s);
  h->host_sync_call = false;
  if (rc) return rc;
  h->open_n = n;
  return SPFE_OK;
}

int spfe_extract_maps(spfe_handle h, const float **heat, const float **heat_inv) {
  if (!h || (!heat && !heat_inv)) return fail(SPFE_EINVAL, "null argument");
  if (!h->open_n) return fail(SPFE_EINVAL, "no open call: spfe_extract_begin first");
  if (heat) *heat = nullptr;
  if (heat_inv) *heat_inv = nullptr;
  if (!h->heat_early) return SPFE_OK;   // the maps come with the record (spfe_extract_finish)
  HIP_TRY(hipSetDevice(h->cfg.device));
  const bool inv = heat_inv && !(h->cfg.flags & SPFE_FLAG_LAZY_HEAT_INV);
  HIP_TRY(hipEventSynchronize(inv ? h->ev_heat_copied : h->ev_heat_copied1));   // (one copy stream: heat first, heat_inv behind it)
  if (heat) *heat = h->h_heat;
  if (inv) *heat_inv = h->h_heat_inv;
  return SPFE_OK;
}

int spfe_extract_rows(spfe_handle h, int frame, int *K, const float **desc) {
  if (!h || !K || !desc) return fail(SPFE_EINVAL, "null argument");
  if (!h->open_n) return fail(SPFE_EINVAL, "no open call: spfe_extract_begin first");
  if (frame < 0 || frame >= h->open_n) return fail(SPFE_EINVAL, "frame %d is not one of the open call's %d", frame, h->open_n);
  *K = 0;
  *desc = nullptr;
  if (!h->desc_early || (h->cfg.flags & SPFE_FLAG_DESC_BF16)) return SPFE_OK;   // the rows come with the record
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipEventSynchronize(h->ev_desc));   // (side stream: sampling, then the rows' and the headers' D2H)
  const uint8_t *rec = h->h_records + (size_t)frame * h->rl.bytes;
  *K = reinterpret_cast<const int *>(rec + h->rl.off_hdr)[0];
  *desc = reinterpret_cast<const float *>(rec + h->rl.off_desc);
  return SPFE_OK;
}

int spfe_extract_finish(spfe_handle h, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  if (!h->open_n) return fail(SPFE_EINVAL, "no open call: spfe_extract_begin first");
  const int n = h->open_n;
  h->open_n = 0;
  HIP_TRY(hipSetDevice(h->cfg.device));
  return finish_host(h, n, outs);
}

int spfe_extract_batch(spfe_handle h, const uint8_t *const *images, int stride, int n, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  const int rc = spfe_extract_begin(h, images, stride, n);
  if (rc) return rc;
  return spfe_extract_finish(h, outs);
}

// D2H of the records (+ maps), sync, host views.
int finish_host(spfe_handle h, int n, spfe_result *outs) {
  const int H = h->H, W = h->W;
  hipStream_t s = h->stream;
  if (h->cov_inflight) {
    const int prev = (int)((h->ticket + spfe_handle_s::NTICKET - 1) % spfe_handle_s::NTICKET);
    HIP_TRY(hipStreamWaitEvent(s, h->ev_cov[prev], 0));
    h->cov_inflight = false;
  }
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  // (the synchronous path keeps the runtime's copy: a copy kernel as in spfe_submit_batch measured +4 % in f32 and -4 % in
  // bf16 mode here, nothing for a single frame)
  if (h->desc_early) {   // the descriptor rows left behind the sampling (spfe_host.h): the record's two ends remain
    HIP_TRY(hipMemcpy2DAsync(h->h_records, h->rl.bytes, h->d_records, h->rl.bytes, h->rl.off_desc, (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpy2DAsync(h->h_records + h->rl.off_occ, h->rl.bytes, h->d_records + h->rl.off_occ, h->rl.bytes,
                             h->rl.bytes - h->rl.off_occ, (size_t)n, hipMemcpyDeviceToHost, s));
    h->desc_early = false;
  } else {
    HIP_TRY(hipMemcpyAsync(h->h_records, h->d_records, (size_t)n * h->rl.bytes, hipMemcpyDeviceToHost, s));
  }
  const bool want_inv = want && !(h->cfg.flags & SPFE_FLAG_LAZY_HEAT_INV);   // (lazy: spfe_fetch_heat_inv on demand)
  if (want && !h->heat_early) {
    if (want_inv) HIP_TRY(hipMemcpyAsync(h->h_heat_inv, h->d_heat_inv, (size_t)n * H * W * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h->h_heat, h->d_heat, (size_t)n * H * W * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (h->heat_early) {   // the maps left behind the heat normalisation, on their own copy stream (spfe_host.h)
    HIP_TRY(hipEventSynchronize(h->ev_heat_copied));
    h->heat_early = false;
  }
  for (int i = 0; i < n; ++i) {
    uint8_t *rec = h->h_records + (size_t)i * h->rl.bytes;
    float *hinv = h->h_heat_inv + (size_t)i * H * W;
    view_record(h, rec, want ? h->h_heat + (size_t)i * H * W : nullptr, want_inv ? hinv : nullptr, &outs[i]);
  }
  h->host_sync_n = n;
  return SPFE_OK;
}

int spfe_fetch_heat_inv(spfe_handle h, int frame, const float **out) {
  if (!h || !out) return fail(SPFE_EINVAL, "null argument");
  if (!(h->cfg.flags & SPFE_FLAG_HEAT)) return fail(SPFE_EINVAL, "the handle was created without SPFE_FLAG_HEAT");
  if (frame < 0 || frame >= h->host_sync_n) return fail(SPFE_EINVAL, "frame %d was not part of the last synchronous host call (%d frames)", frame, h->host_sync_n);
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t HW = (size_t)h->H * h->W;
  // (the stream of the call: its covariance — the only device reader / the writer's successor — has finished, finish_host synchronised)
  HIP_TRY(hipMemcpyAsync(h->h_heat_inv + (size_t)frame * HW, h->d_heat_inv + (size_t)frame * HW, HW * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *out = h->h_heat_inv + (size_t)frame * HW;
  return SPFE_OK;
}

int spfe_set_map_buffers(spfe_handle h, float *heat, float *heat_inv) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (!(h->cfg.flags & SPFE_FLAG_HEAT)) return fail(SPFE_EINVAL, "the handle was created without SPFE_FLAG_HEAT");
  if (h->open_n) return fail(SPFE_EINVAL, "a call begun by spfe_extract_begin is open: spfe_extract_finish first");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipStreamSynchronize(h->stream));   // (no copy into the buffers about to be replaced is in flight)
  if (h->s_heat) HIP_TRY(hipStreamSynchronize(h->s_heat));
  if (!h->own_heat) { h->own_heat = h->h_heat; h->own_heat_inv = h->h_heat_inv; }
  const size_t bytes = (size_t)h->B * h->H * h->W * 4;
  struct Slot { float **cur; float *own; float **usr; float *want; };
  for (Slot q : {Slot{&h->h_heat, h->own_heat, &h->usr_heat, heat}, Slot{&h->h_heat_inv, h->own_heat_inv, &h->usr_heat_inv, heat_inv}}) {
    if (*q.usr == q.want) continue;
    if (*q.usr) { (void)hipHostUnregister(*q.usr); *q.usr = nullptr; *q.cur = q.own; }
    if (q.want) {
      HIP_TRY(hipHostRegister(q.want, bytes, hipHostRegisterDefault));
      *q.usr = *q.cur = q.want;
    }
  }
  h->host_sync_n = 0;   // (spfe_fetch_heat_inv: the last call's frames are no longer what the buffers describe)
  return SPFE_OK;
}

int spfe_extract(spfe_handle h, const uint8_t *image, int stride, spfe_result *out) {
  if (!image) return fail(SPFE_EEMPTY, "input image is empty");
  const uint8_t *imgs[1] = {image};
  return spfe_extract_batch(h, imgs, stride, 1, out);
}

long spfe_last_ticket(spfe_handle h) { return h ? api_tickets(h) - 1 : -1; }

int spfe_wait_records(spfe_handle h, long ticket, void *stream) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (ticket < 0 || ticket >= api_tickets(h) || ticket + spfe_handle_s::NTICKET <= api_tickets(h))
    return fail(SPFE_EINVAL, "ticket %ld is not one of the last %d calls", ticket, spfe_handle_s::NTICKET);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const spfe_handle_s::TicketRef r = ticket_ref(h, ticket);   // (a handle with a twin: whichever of the two ran that call)
  HIP_TRY(hipStreamWaitEvent(s, r.who->ev_cov[r.local % spfe_handle_s::NTICKET], 0));
  if (h->twin && ticket > 0) {   // one side stream used to make "ticket t is done" mean "and every earlier one": keep that — the
    const spfe_handle_s::TicketRef p = ticket_ref(h, ticket - 1);   // call before ran on the other of the pair (each side stream is serial)
    if (p.who) HIP_TRY(hipStreamWaitEvent(s, p.who->ev_cov[p.local % spfe_handle_s::NTICKET], 0));
  }
  return SPFE_OK;
}

int spfe_view_record(spfe_handle h, const void *host_record, spfe_result *out) {
  if (!h || !host_record || !out) return fail(SPFE_EINVAL, "null argument");
  view_record(h, reinterpret_cast<const uint8_t *>(host_record), nullptr, nullptr, out);
  return SPFE_OK;
}

long spfe_debug_read(spfe_handle h, const char *name, int frame, void *dst, size_t cap) {
  if (!h || !name || !dst) return fail(SPFE_EINVAL, "null argument");
  if (std::string(name) == "two_chains" || std::string(name) == "twin_failed") {   // 1: pipelined device calls alternate between this handle and its twin / 1: the twin could not be built
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer '%s' needs 4 bytes", name);
    *reinterpret_cast<int *>(dst) = std::string(name) == "two_chains" ? (h->twin ? 1 : 0) : (h->twin_failed ? 1 : 0);
    return (long)sizeof(int);
  }
  if (h->twin) {   // the intermediates of the last call live in whichever of the two ran it; neither has work in flight afterwards
    if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(SPFE_EHIP, "hipSetDevice failed");
    for (spfe_handle q : {h, h->twin})
      if (hipStreamSynchronize(q->stream) != hipSuccess || hipStreamSynchronize(q->side) != hipSuccess) return fail(SPFE_EHIP, "debug read of '%s' failed", name);
    h = last_caller(h);
  }
  if (frame < 0 || frame >= h->B) return fail(SPFE_EINVAL, "frame %d out of range", frame);
  const size_t C = h->C, HW = (size_t)h->H * h->W;
  if (std::string(name) == "conv1b_split_rows") {
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'conv1b_split_rows' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->conv1b_split_rows;
    return (long)sizeof(int);
  }
  if (std::string(name) == "conv1b_tile_rows") {   // f32: which conv1b instantiation the last call launched (8 or 16 rows per tile)
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'conv1b_tile_rows' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->conv1b_tile_rows;
    return (long)sizeof(int);
  }
  if (std::string(name) == "select_huge") {   // 1: this handle's selection runs as select_huge_kernel (> 65,535 cells, or SPFE_SELECT_HUGE=1)
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'select_huge' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->select_huge ? 1 : 0;
    return (long)sizeof(int);
  }
  if (std::string(name) == "split_streams") {   // [2] int: outcome of the queue probe (see spfe_handle_s::split_probe), and whether the last call ran as two half batches
    if (cap < 2 * sizeof(int)) return fail(SPFE_EINVAL, "buffer 'split_streams' needs 8 bytes");
    reinterpret_cast<int *>(dst)[0] = h->split_probe;
    reinterpret_cast<int *>(dst)[1] = h->split_last ? 1 : 0;
    return (long)(2 * sizeof(int));
  }
  if (std::string(name) == "da_gathered") {   // 1: the last call ran convDa on the listed cells only (host-side flag)
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'da_gathered' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->sparse_last && h->sparse_da_call ? 1 : 0;
    return (long)sizeof(int);
  }
  const void *src = nullptr;
  size_t bytes = 0;
  bool bf16_src = false;
  std::string nm(name);
  if (nm == "semi") { src = h->d_semi + frame * C * SPFE_SEMI_CH; bytes = C * SPFE_SEMI_CH * 4; }
  else if (nm == "coarse" || nm == "coarse_sparse") {
    src = h->d_coarse + frame * C * SPFE_DESC_DIM; bytes = C * SPFE_DESC_DIM * 4;
    if (nm == "coarse" && h->sparse_last) {   // the last call wrote only the rows its keypoints read: complete the map
      if (hipSetDevice(h->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(SPFE_EHIP, "debug read of 'coarse' failed");
      const int rc = launch_db_dense(h, h->last_n, h->stream);
      if (rc) return rc;
      h->sparse_last = false;
    }
  }
  else if (nm == "db_total" && h->d_db_total) { src = h->d_db_total; bytes = 4; }
  else if (nm == "db_list" && h->d_db_list) { src = h->d_db_list; bytes = (size_t)h->B * h->db_cap * 4; }
  else if (nm == "head") {
    if (h->bf16) return fail(SPFE_EINVAL, "'head' is f32 only: the bf16 mode keeps ReLU(convPa) | ReLU(convDa) as bf16");
    src = h->d_head + frame * C * 512; bytes = C * 512 * 4;
  }
  else if (nm == "heat_log") { src = h->d_heat_log[(h->ticket + 1) & 1] + frame * HW; bytes = HW * 4; }
  else if (nm == "heat_inv") {
    if (!(h->cfg.flags & SPFE_FLAG_HEAT)) return fail(SPFE_EINVAL, "'heat_inv' is materialised with SPFE_FLAG_HEAT only");
    src = h->d_heat_inv + frame * HW; bytes = HW * 4;
  }
  else if (nm == "heat" && h->d_heat) { src = h->d_heat + frame * HW; bytes = HW * 4; }
  else if (nm == "image") { src = h->d_img + frame * HW; bytes = HW; }
  else if (nm == "cell_score") { src = h->d_cell_score[(h->ticket + 1) & 1] + frame * C; bytes = C * 4; }
  else if (nm == "cov_counters") { src = h->cov.counters + frame * 4; bytes = 16; }
  else if (nm == "cov_nxt") { src = h->cov.nxt + (size_t)frame * h->kmax; bytes = (size_t)h->kmax * 4; }
  else if (nm == "cov_workers") { src = h->cov.workers + (size_t)frame * h->kmax; bytes = (size_t)h->kmax * 4; }
  else if (nm == "cov_npop") { src = h->cov.npop + (size_t)frame * h->kmax; bytes = (size_t)h->kmax * 4; }
  else if (nm == "feat") { src = (h->feat_cur ? h->feat_cur : h->act[7]) + frame * C * 128; bytes = C * 128 * 4; bf16_src = h->bf16; }
  else if (nm.size() == 4 && nm.compare(0, 3, "act") == 0 && nm[3] >= '0' && nm[3] <= '7') {
    const int i = nm[3] - '0';
    if (i == 0 && h->act0_missing)
      return fail(SPFE_EINVAL, "act0 is not materialised: conv1a was fused into conv1b in the last call");
    const int lh[8] = {1, 2, 2, 4, 4, 8, 8, 8};
    const int lc[8] = {64, 64, 64, 64, 128, 128, 128, 128};
    const size_t per = (size_t)(h->H / lh[i]) * (h->W / lh[i]) * lc[i];
    // (conv4b's output exists twice, by ticket parity, when convDa runs gathered: "act7" is the last call's, like "feat")
    src = (i == 7 && h->feat_cur ? h->feat_cur : h->act[i]) + frame * per; bytes = per * 4; bf16_src = h->bf16;
  } else return fail(SPFE_EINVAL, "unknown debug buffer '%s'", name);
  if (bytes > cap) return fail(SPFE_EINVAL, "buffer '%s' needs %zu bytes, cap %zu", name, bytes, cap);
  // both streams: with SPFE_FLAG_ASYNC_COV heat / heat_inv are written on the side stream
  if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
      hipStreamSynchronize(h->side) != hipSuccess)
    return fail(SPFE_EHIP, "debug read of '%s' failed", name);
  if (bf16_src) {
    // bf16 mode keeps the conv stack's activations as bf16 NHWC (half the elements' bytes, the frame offset
    // in bf16 elements): read them as such and widen to the f32 the caller expects
    const size_t n = bytes / 4;
    std::vector<unsigned short> tmp(n);
    const unsigned short *bsrc = reinterpret_cast<const unsigned short *>(
        nm == "feat" || nm == "act7" ? (const void *)(h->feat_cur ? h->feat_cur : h->act[7]) : (const void *)h->act[nm[3] - '0']) + (size_t)frame * n;
    if (hipMemcpy(tmp.data(), bsrc, n * 2, hipMemcpyDeviceToHost) != hipSuccess)
      return fail(SPFE_EHIP, "debug read of '%s' failed", name);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
    for (size_t k = 0; k < n; ++k) d32[k] = (uint32_t)tmp[k] << 16;
    return (long)bytes;
  }
  if (hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(SPFE_EHIP, "debug read of '%s' failed", name);
  return (long)bytes;
}

int spfe_stage_reset(spfe_handle h) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  h->calls_at_reset = h->calls;
  if (h->twin) h->twin->calls_at_reset = h->twin->calls;
  return SPFE_OK;
}

// Average per-stage GPU time (ms) over the calls since spfe_stage_reset (at most
// the last EVSETS calls).  Needs SPFE_STAGE_TIMING=1 at spfe_create.
static int stage_times_one(spfe_handle h, float *ms, int cap);
int spfe_stage_times(spfe_handle h, float *ms, int cap) {
  if (!h || !ms) return fail(SPFE_EINVAL, "null argument");
  if (!h->twin) return stage_times_one(h, ms, cap);
  // the calls since the reset ran on both of the pair: the call-weighted mean of the two
  float a[NSTAGE] = {}, b[NSTAGE] = {};
  const int na = stage_times_one(h, a, NSTAGE), nb = stage_times_one(h->twin, b, NSTAGE);
  if (na < 0 || nb < 0) return na < 0 ? na : nb;
  const double ca = na ? (double)(h->calls - h->calls_at_reset) : 0.0, cb = nb ? (double)(h->twin->calls - h->twin->calls_at_reset) : 0.0;
  if (ca + cb == 0.0) return 0;
  const int nst = cap < NSTAGE ? cap : NSTAGE;
  for (int i = 0; i < nst; ++i) ms[i] = (float)((a[i] * ca + b[i] * cb) / (ca + cb));
  return nst;
}
static int stage_times_one(spfe_handle h, float *ms, int cap) {
  if (!h->timing || h->calls == h->calls_at_reset) return 0;
  if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(SPFE_EHIP, "hipSetDevice failed");
  long first = h->calls_at_reset;
  if (h->calls - first > spfe_handle_s::EVSETS) first = h->calls - spfe_handle_s::EVSETS;
  const int nst = cap < NSTAGE ? cap : NSTAGE;
  std::vector<double> acc(NSTAGE, 0.0);
  for (long c = first; c < h->calls; ++c) {
    hipEvent_t *ev = h->evpool.data() + (size_t)(c % spfe_handle_s::EVSETS) * (NSTAGE + 1);
    if (!h->timing_all) {  // only the bracket of the dominant kernel was recorded
      float t = 0;
      if (hipEventSynchronize(ev[2]) != hipSuccess) return fail(SPFE_EHIP, "event sync failed");
      (void)hipEventElapsedTime(&t, ev[1], ev[2]);
      acc[1] += t;
      continue;
    }
    if (hipEventSynchronize(ev[NSTAGE - 1]) != hipSuccess) return fail(SPFE_EHIP, "event sync failed");
    for (int i = 0; i < NSTAGE - 1; ++i) {
      float t = 0;
      (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
      acc[i] += t;
    }
    float t = 0;
    (void)hipEventElapsedTime(&t, ev[0], ev[NSTAGE - 1]);
    acc[NSTAGE - 1] += t;
  }
  for (int i = 0; i < nst; ++i) ms[i] = (float)(acc[i] / (double)(h->calls - first));
  return nst;
}

// :427-433) as a depth-NPIPE pipeline: pinned staging, H2D of batch i + 1 and D2H of batch i - 1 on copy
// streams beside the compute of batch i, covariance on the side stream.
namespace {
int pipe_setup(spfe_handle h) {
  if (h->pipe_ready) return SPFE_OK;
  { const int rct = make_twin(h); if (rct) return rct; }
  const size_t img = (size_t)h->B * h->H * h->W, rec = (size_t)h->B * h->rl.bytes;
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  int rc;
  HIP_TRY(hipStreamCreateWithFlags(&h->s_h2d, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&h->s_d2h, hipStreamNonBlocking));
  for (auto &ps : h->pipe) {
    if ((rc = host_alloc(h, &ps.h_img, img))) return rc;
    if ((rc = host_alloc(h, &ps.h_rec, rec))) return rc;
    if ((rc = dev_alloc(h, &ps.d_img, img))) return rc;
    if ((rc = dev_alloc(h, &ps.d_rec, rec))) return rc;
    if (want) {
      if ((rc = host_alloc(h, &ps.h_heat, img))) return rc;
      if ((rc = host_alloc(h, &ps.h_heat_inv, img))) return rc;
    }
    HIP_TRY(hipEventCreateWithFlags(&ps.ev_h2d, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ps.ev_done, hipEventDisableTiming));
  }
  h->pipe_ready = true;
  return SPFE_OK;
}
}  // namespace

int spfe_submit_batch(spfe_handle h, const uint8_t *const *images, int stride, int n, long *ticket) {
  if (!h || !ticket) return fail(SPFE_EINVAL, "null argument");
  if (h->open_n) return fail(SPFE_EINVAL, "a call begun by spfe_extract_begin is open: spfe_extract_finish first");
  if (!images) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  const int H = h->H, W = h->W;
  if (stride < W) return fail(SPFE_EINVAL, "stride %d smaller than width %d", stride, W);
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = pipe_setup(h);
  if (rc) return rc;
  spfe_handle_s::PipeSlot &ps = h->pipe[h->pipe_submitted % spfe_handle_s::NPIPE];
  if (ps.ticket >= 0)
    return fail(SPFE_EINVAL, "pipeline full: %d batches in flight, collect ticket %ld first", spfe_handle_s::NPIPE, ps.ticket);
  for (int i = 0; i < n; ++i) {
    if (!images[i]) return fail(SPFE_EEMPTY, "input image is empty");  // sp_extractor.cpp:364-365
    if (stride == W) memcpy(ps.h_img + (size_t)i * H * W, images[i], (size_t)H * W);
    else
      for (int y = 0; y < H; ++y) memcpy(ps.h_img + ((size_t)i * H + y) * W, images[i] + (size_t)y * stride, W);
  }
  HIP_TRY(hipMemcpyAsync(ps.d_img, ps.h_img, (size_t)n * H * W, hipMemcpyHostToDevice, h->s_h2d));
  HIP_TRY(hipEventRecord(ps.ev_h2d, h->s_h2d));
  hipStream_t s = h->stream;
  HIP_TRY(hipStreamWaitEvent(s, ps.ev_h2d, 0));
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  // two side chains in flight: even submissions on this handle, odd ones on its twin (own buffers, own side stream)
  spfe_handle hc = h->twin && (h->g_ticket & 1) ? h->twin : h;
  if (want)
    // the heat maps are single buffers (per handle of the pair): this batch's heat_norm (side stream) must not overwrite them
    // before the copy of the batch that went through the same buffers has left
    for (const auto &pp : h->pipe)
      if (&pp != &ps && pp.ticket >= 0 && pp.who == hc) HIP_TRY(hipStreamWaitEvent(hc->side, pp.ev_done, 0));
  hc->pipe_mode = true;
  rc = enqueue(hc, ps.d_img, n, ps.d_rec, s);
  hc->pipe_mode = false;
  if (rc) return rc;
  long t = h->ticket - 1;
  if (h->twin) {
    h->tmap[h->g_ticket % 8] = {hc, hc->ticket - 1};
    t = h->g_ticket++;
  }
  // D2H on the SIDE stream, behind the covariance kernels it has to follow anyway.  (A copy stream of its own, waiting
  // for the covariance event, looked cleaner and cost half the throughput in bf16 mode: HIP maps streams onto a few
  // hardware queues, the waiting copy stream shared one with the compute stream, and its barrier packet held the NEXT
  // batch's convolutions until the previous batch's covariance had finished — tools/microbench/run_hosttrace.sh.)
  hipStream_t sc = hc->side;
  {
    // f32 mode: a copy kernel of our own (2038 against 2000 ... 2028 frames/s with the runtime's copy at 752x480 x 8).  bf16
    // mode: the runtime's copy engine — the kernel's 64 workgroups sit on the chip for the 0.2 ms the PCIe transfer takes,
    // beside convolutions that are 4x shorter than the f32 ones: 6895 against 7990 frames/s at 1280x720 x 8 (= the
    // device-resident rate)
    const int copy_mode = h->pipe_copy_kernel >= 0 ? h->pipe_copy_kernel : (h->bf16 && h->C >= 10000 ? 0 : 1);   // (bf16 752x480: kernel 11,560, engine 11,250)
    if (copy_mode == 1) {          // a copy kernel of our own writing the pinned buffer
      const size_t n16 = ((size_t)n * h->rl.bytes + 15) / 16;
      hipLaunchKernelGGL(spfe::copy_records_kernel, dim3(64), dim3(256), 0, sc, reinterpret_cast<uint4 *>(ps.h_rec),
                         reinterpret_cast<const uint4 *>(ps.d_rec), n16);
      HIP_TRY(hipGetLastError());
    } else if (copy_mode != 2) {   // (2: no copy at all, timing probe)
      HIP_TRY(hipMemcpyAsync(ps.h_rec, ps.d_rec, (size_t)n * h->rl.bytes, hipMemcpyDeviceToHost, sc));
    }
  }
  if (want) {
    const size_t m16 = (size_t)n * H * W * 4 / 16;   // (H, W multiples of 8)
    if (!(h->cfg.flags & SPFE_FLAG_LAZY_HEAT_INV))
      hipLaunchKernelGGL(spfe::copy_records_kernel, dim3(64), dim3(256), 0, sc, reinterpret_cast<uint4 *>(ps.h_heat_inv),
                         reinterpret_cast<const uint4 *>(hc->d_heat_inv), m16);
    hipLaunchKernelGGL(spfe::copy_records_kernel, dim3(64), dim3(256), 0, sc, reinterpret_cast<uint4 *>(ps.h_heat),
                       reinterpret_cast<const uint4 *>(hc->d_heat), m16);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(ps.ev_done, sc));
  ps.ticket = t;
  ps.n = n;
  ps.who = hc;
  h->pipe_submitted++;
  *ticket = t;
  return SPFE_OK;
}

int spfe_collect_batch(spfe_handle h, long ticket, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  spfe_handle_s::PipeSlot *ps = nullptr;
  for (auto &c : h->pipe)
    if (c.ticket == ticket && ticket >= 0) ps = &c;
  if (!ps) return fail(SPFE_EINVAL, "ticket %ld is not in flight", ticket);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipEventSynchronize(ps->ev_done));
  const int H = h->H, W = h->W;
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  for (int i = 0; i < ps->n; ++i) {
    uint8_t *rec = ps->h_rec + (size_t)i * h->rl.bytes;
    view_record(h, rec, want ? ps->h_heat + (size_t)i * H * W : nullptr,
                want && !(h->cfg.flags & SPFE_FLAG_LAZY_HEAT_INV) ? ps->h_heat_inv + (size_t)i * H * W : nullptr, &outs[i]);
  }
  ps->ticket = -1;   // the views stay valid until NPIPE further submits reuse the slot
  return SPFE_OK;
}

// test hook: run the exact-math device functions on n floats (host buffers)
int spfe_math_probe(const float *in, float *out_exp, float *out_log, int n) {
  float *d_in = nullptr, *d_e = nullptr, *d_l = nullptr;
  HIP_TRY(hipMalloc(&d_in, n * 4));
  HIP_TRY(hipMalloc(&d_e, n * 4));
  HIP_TRY(hipMalloc(&d_l, n * 4));
  HIP_TRY(hipMemcpy(d_in, in, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(spfe::launch_math_probe(d_in, d_e, d_l, n, nullptr));
  HIP_TRY(hipMemcpy(out_exp, d_e, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_log, d_l, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_e); (void)hipFree(d_l);
  return SPFE_OK;
}

}  // extern "C"
