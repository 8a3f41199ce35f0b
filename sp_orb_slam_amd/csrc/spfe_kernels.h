// spfe_kernels.h — launch interface between the host pipeline (spfe_schedule.hip) and
// the gfx950 kernels.  Everything here is internal to libspfe.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spfe {

// ---------------------------------------------------------------------------
// f32 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (conv_f32.hip)
// Activations are NHWC f32: element (b,y,x,c) at ((b*H+y)*W+x)*stride + choff + c.
// ---------------------------------------------------------------------------
struct ConvParams {
  const float *in;
  int in_stride, in_choff;
  const float *wpack;  // [nblk][chunk][n-tile(2)][tap][KC][32], K order of spfe_exact_math.h
  const float *bias;   // [nblk*64]
  float *out;
  int out_stride, out_choff, cout_real;
  int B, H, W;  // conv input == conv output size (before the optional 2x2 pool)
  int tiles_x, tiles_y, nblk;
  int num_cus;  // persistent grid size (multiProcessorCount)
  // layer_tag 2 (conv1a fused into conv1b): the u8 frames [B][H][W] and conv1a's taps [9][64] / bias [64];
  // `in` is then unused
  const uint8_t *img;
  const float *w1a, *b1a;
  // conv_bf16_ws.hip: nblk * 8 tile counters (one per XCD and 64-channel block), zero before the launch
  int *tile_ctr;
  // conv_f32.hip: the launch walks work items [item_lo, item_hi) of the list (frame, tile row, tile column, 64-channel
  // block); item_hi = 0: the whole list.  (conv1b's list is cut in two launches with different tile heights, spfe_schedule.hip.)
  int item_lo = 0, item_hi = 0;
};

// cin: 64/128/256; ksize: 3 or 1; pool/relu: fused epilogue; small_tile: 4-row
// tiles (more workgroups for the low-resolution layers).
// tile_mode: 0 = 8-row tiles (4 waves), 1 = 4-row tiles, 2 = 16-row tiles (8 waves, two per SIMD), 3 = 2-row tiles,
// 4 = 16-row tiles of 4 waves x 4 rows (conv1b)
hipError_t launch_conv_f32(const ConvParams &p, int cin, int ksize, bool pool, bool relu,
                           int tile_mode, int layer_tag, hipStream_t s);
// 2x2 / 2 max-pool of an NHWC f32 activation [B][H][W][C] -> [B][H/2][W/2][C] (H, W even, C % 4 == 0)
hipError_t launch_pool2x2_f32(const float *in, float *out, int B, int H, int W, int C, hipStream_t s);
int conv_kc(int ksize);        // K-chunk the kernel stages per barrier (16 for 3x3, 64 for 1x1)
int conv_tile_rows(int tile_mode);

// bf16 3x3 convolutions (conv_bf16.hip): in/out NHWC bf16 (out f32 when out_f32), strides in
// elements; wpack = bf16 slabs of conv_bf16_slab_bytes() each, [nblk][chunk of 32 ch][tap][64][80 B]
hipError_t launch_conv_bf16(const ConvParams &p, int cin, bool pool, bool out_f32, hipStream_t s, int tile_rows = 8);
size_t conv_bf16_slab_bytes();
// wave-specialised variant for the Cin = 64 layers (conv_bf16_ws.hip): wpack = conv_bf16_ws_weight_bytes() per
// 64-channel block, [nblk][tap][cout 64][8 x 16-byte pieces, piece g in slot g ^ ((cout >> 1) & 7)]
hipError_t launch_conv_bf16_ws(const ConvParams &p, bool pool, int layer_tag /* 1 = conv1b, 2 = conv1b with conv1a fused in (p.img / p.w1a / p.b1a) */, hipStream_t s);
size_t conv_bf16_ws_weight_bytes();
// register-resident-weights variant for the Cin = 128 layers (conv_bf16_rw.hip): p.nblk = 128-channel output groups,
// wpack = conv_bf16_rw_weight_bytes() per group in fragment order (conv_bf16_rw_pack_weights from [cout][128][9] bf16),
// bias f32 in channel order; tile_rows 4 | 2; p.tile_ctr: nblk * 8 zeroed counters.  Bit-identical to launch_conv_bf16.
hipError_t launch_conv_bf16_rw(const ConvParams &p, bool pool, int tile_rows, hipStream_t s);
size_t conv_bf16_rw_weight_bytes();
void conv_bf16_rw_pack_weights(const unsigned short *Wb, int cout, unsigned char *dst);
// conv1a of the bf16 mode (conv1a_mfma.h): wtab = the bf16 operand table [2][64][8] (conv1a_bf16_table_bytes()), b64 = bias
hipError_t launch_conv1a_bf16(const uint8_t *img, const void *wtab, const float *b64, void *out, int B, int H,
                              int W, hipStream_t s);
inline size_t conv1a_bf16_table_bytes() { return 2 * 64 * 8 * 2; }

// bf16 mode, the 1x1 heads (head_bf16.hip): out[npix][cout] f32 = in[npix][512] bf16 (channels 0..255 for the
// detector head, cout = 65; 256..511 for the descriptor head, cout = 256) x W^T + bias;
// wpack: head_bf16_weight_bytes(cout) bytes in MFMA fragment order, made by head_bf16_pack_weights from Wb = [cout][256] bf16
hipError_t launch_head1x1_bf16(const void *in_bf16, const void *wpack, const float *bias, float *out, int npix, int cout,
                               hipStream_t s);
// the descriptor head (cout = 256) on the *total (<= max_total) rows of in_bf16 / out that `list` names (device memory,
// select_kernel's FrameBufs::db_list / db_total): "sparse convDb"
hipError_t launch_head1x1_bf16_gather(const void *in_bf16, const void *wpack, const float *bias, float *out, int npix,
                                      const int *list, const int *total, int max_total, int tiles_per_wg, hipStream_t s);
// convDa (3x3, 128 -> 256, ReLU) on the listed cells (da_gather_bf16.hip): feat = conv4b's output [B][hc][wc][128] bf16,
// wpack = convPa|Da in conv_bf16_rw_pack_weights order, out = head activations [B * hc * wc][512] bf16 (channels 256..511)
hipError_t launch_da_gather_bf16(const void *feat, const void *wpack, const float *bias, void *out, const int *list,
                                 const int *total, int max_total, int B, int hc, int wc, int num_cus, hipStream_t s);
size_t head_bf16_weight_bytes(int cout);
void head_bf16_pack_weights(const unsigned short *Wb, int cout, unsigned char *dst);

// f32 mode, the 1x1 heads (head_f32.hip): same shapes, in / out f32, bit-identical to conv_f32.hip's 1x1 path
hipError_t launch_head1x1_f32(const float *in, const float *wpack, const float *bias, float *out, int npix, int cout,
                              hipStream_t s);
// f32 mode, convDa (3x3, 128 -> 256, ReLU) on the listed cells (da_gather_f32.hip), bit-identical to conv_f32.hip's rows:
// feat = conv4b's output [B][hc][wc][128], wpack = da_gather_f32_pack_weights(convDa's OIHW weights), bias = convDa's,
// out = head activations [B * hc * wc][512] (channels 256..511 of the listed rows)
hipError_t launch_da_gather_f32(const float *feat, const float *wpack, const float *bias, float *out, const int *list,
                                const int *total, int max_total, int B, int hc, int wc, int num_cus, hipStream_t s);
size_t da_gather_f32_weight_bytes();
void da_gather_f32_pack_weights(const float *W, float *dst);
hipError_t launch_head1x1_f32_gather(const float *in, const float *wpack, const float *bias, float *out, int npix,
                                     const int *list, const int *total, int max_total, int tiles_per_wg, hipStream_t s);
size_t head_f32_weight_bytes(int cout);
void head_f32_pack_weights(const float *W, int cout, float *dst);

// conv1a: u8 image -> (x * 1/255) -> 3x3 conv 1->64 + bias + relu, NHWC out.
// w: [9][64] (tap-major), b: [64]
hipError_t launch_conv1a(const uint8_t *img, const float *w9x64, const float *b64, float *out, int B,
                         int H, int W, hipStream_t s);

// ---------------------------------------------------------------------------
// detector tail, selection, descriptors (tail_select.hip)
// ---------------------------------------------------------------------------
struct FrameBufs {
  // per-frame strides are implied by H, W; all arrays are [B][...]
  const float *semi;    // [B][C][65]
  const float *coarse;  // [B][C][256]
  float *heat_log;      // [B][H][W]
  float *heat;          // [B][H][W] or null
  float *heat_inv;      // [B][H][W]
  uint32_t *minmax;     // [B][tail_parts][2] float partials of min/max of heat_log
  float *cell_score;    // [B][C]  0 = no candidate
  uint8_t *cell_k;      // [B][C]  arg-max channel
  uint8_t *cell_mask;   // [B][C]  which of the 8 neighbouring cells' candidates can suppress this one (nms_mask_kernel)
  int *kp_cell;         // [B][kmax] cell index of emitted keypoint
  int *sel_slot;        // [B][C] frames of more than 16,384 cells: select_kernel's per-cell slot / index hand-off (else null)
  uint16_t *sel_list;   // [B][C] ... and its tie / layout list
  uint8_t *sel_state;   // [B][C] frames of more than 65,535 cells (select_huge_kernel): the cell states ...
  int *sel_list32;      // [B][C] ... and the tie / layout list with 32-bit cell indices (else null)
  int sel_huge;         // 1: the selection runs as select_huge_kernel
  int *db_list;         // [B * min(4 kmax, C)] global cell indices (b * C + cell) some emitted keypoint's descriptor taps read,
  int *db_total;        // [1] ... and how many: written by select_kernel for the gathered descriptor head (or both null)
  uint8_t *records;     // [B][record_bytes]
  float *heat_consts;   // [B][4] a_heat, b_heat, a_inv, b_inv
};

struct RecordLayout {
  size_t bytes;
  int kmax;
  size_t off_hdr, off_xy, off_resp, off_cov, off_cinv, off_desc, off_occ, off_dd, off_sd;
  int desc_bf16;   // SPFE_FLAG_DESC_BF16: descriptor rows are 256 bf16 (RNE of the f32 descriptor), not 256 f32
};

// scratch of the covariance kernels (cov.hip)
struct CovScratch {
  int *claim;     // [B][H*W] lowest keypoint index whose lone walk popped the pixel
  int *done;      // [B][H*W] lowest FINAL keypoint index that popped the pixel
  int *queue;     // [B][kmax][qcap] per-keypoint pop list (pixel index)
  float *qval;    // [B][kmax][qcap] heat_inv value of each popped pixel
  int *npop;      // [B][kmax]
  int *dirty;     // [B][kmax] keypoints whose lone region meets a lower keypoint's
  int *nxt;       // [B][kmax] next dirty member of the same component (ascending) or -1
  float *nxy;     // [B][kmax][2] keypoint position of nxt
  int *workers;   // [B][kmax] lowest dirty member of each component
  int *counters;  // [B][4] number of dirty keypoints, number of components, overflow slots taken, claim edges listed
  int *edges;     // [B][ecap] int2 (lower claimant, dirty keypoint): the union-find's input, written by the classification (or null)
  int ecap;
  int qcap;
  // walks that outgrow qcap redo themselves in one of ovf_slots per-frame slots of ovf_cap entries
  int *ovf_slot;  // [B][kmax] slot of the keypoint's pop list, or -1
  int *ovf_q;     // [B][ovf_slots][ovf_cap]
  float *ovf_v;
  int ovf_slots, ovf_cap;
  // last resort (cov_fallback_kernel): ONE pop list of fb_cap entries for the whole batch; a flagged frame is redone
  // sequentially, literally, on the device
  int *fb_q;
  float *fb_v;
  int fb_cap;
  // claim / done entries carry the batch's GENERATION in their upper 16 bits (gen = code << 16, the code counts DOWN from
  // 32,766 per chain): an entry of an earlier batch reads as "nobody" and loses every atomicMin against this batch's, so the
  // maps are not cleared per batch (2 x 4 H W bytes per frame: half of what the heat normalisation moved) — only before a
  // frame's first use and when the code wraps (reset_maps; COV_RESET, code 32,767, is never a batch's)
  int gen;
  int reset_maps;
};
#define COV_RESET 0x7fffffff
size_t cov_link_lds(int kmax);
// with_desc: the descriptor sampling (launch_desc) as extra wavefronts of the replay launch, behind `before_replay` if given
// replay_waves: components per replay workgroup, 2 (default) or 8 (bf16 pipelined calls: see cov.hip)
hipError_t launch_cov(const FrameBufs &f, const RecordLayout &r, const CovScratch &cs, int B, int H, int W,
                      hipStream_t s, bool with_desc = false, hipEvent_t before_replay = nullptr, int replay_waves = 2,
                      bool defer_moments = false);   // defer_moments: synchronous calls (cov.hip, cov_replay_kernel)

int tail_parts(int H, int W);  // min/max partials per frame written by the tail kernel
// f32 mode: convPb (1x1, 256 -> 65) + the detector tail in one launch (pbtail_f32.hip), bit-identical to convPb through
// conv_f32.hip followed by launch_tail.  head = [B * C][512] f32 (channels 0..255 = ReLU(convPa)); wpack =
// head_f32_pack_weights(convPb's weights, 65); wdust = convPb's row 64 [256]; bias [>= 65]; semi [B][C][65] is written too
hipError_t launch_pbtail_f32(const float *head, const float *wpack, const float *wdust, const float *bias, float *semi,
                             const FrameBufs &f, const RecordLayout &r, int B, int H, int W, hipStream_t s, int b0 = 0);
// bf16 mode: the same fusion (pbtail_bf16.hip), bit-identical to launch_head1x1_bf16(.., 65, ..) followed by launch_tail.
// head = [B * C][512] bf16; wpack = head_bf16_pack_weights(convPb, 65); zero_ints / nzero as launch_tail's
hipError_t launch_pbtail_bf16(const void *head, const void *wpack, const float *bias, float *semi, const FrameBufs &f,
                              const RecordLayout &r, int B, int H, int W, hipStream_t s, int b0 = 0, int *zero_ints = nullptr,
                              int nzero = 0, int zstride = 32, int force = 0);   // (zero_ints: nzero ints in runs of 32, zstride apart; force: 2 | 4 wavefronts, 0 = by launch size)   // frames [b0, b0 + B) of the batch-wide buffers
// zero_ints / nzero: ints the kernel also clears (the bf16 convolutions' tile-queue counters, for the next call)
hipError_t launch_tail(const FrameBufs &f, const RecordLayout &r, int B, int H, int W, hipStream_t s, int *zero_ints = nullptr, int nzero = 0);
// with_heat_norm: the heat normalisation (launch_heat_norm) rides in the neighbour-mask launch in front of the selection
// lean: the form that keeps only the cell states and neighbour masks in LDS (2 bytes a cell; the per-cell private data in
// FrameBufs::sel_slot / sel_list) also on frames that would fit the all-in-LDS form (9 bytes a cell: 64 KB at 752x480, 143 KB
// at 1280x720 — a whole CU): a pipelined call's selection then starts beside a convolution workgroup instead of waiting for a CU
hipError_t launch_select(const FrameBufs &f, const RecordLayout &r, int B, int H, int W,
                         int num_features, hipStream_t s, const CovScratch *with_heat_norm = nullptr, int kmax_hn = 0,
                         bool lean = false, hipEvent_t done = nullptr, hipEvent_t heat_done = nullptr);
// (heat_done: an event that fires when the heat normalisation — the launch in FRONT of the selection — has completed: the heat
// maps are final there, ~150 us before a single-frame call's record is, and a synchronous host call starts their D2H behind it)
// (also resets the covariance scratch: launch_cov must follow it)
hipError_t launch_heat_norm(const FrameBufs &f, const CovScratch &cs, int kmax, int B, int H, int W, hipStream_t s);
hipError_t launch_desc(const FrameBufs &f, const RecordLayout &r, int B, int H, int W, hipStream_t s);
size_t select_lds_bytes(int H, int W, bool lean = false);
bool select_big(int H, int W);     // more than 16,384 cells: per-cell private data in global scratch (FrameBufs::sel_slot / sel_list)
size_t select_max_cells();         // 65,535: select_kernel
size_t select_huge_max_cells();    // 262,143: select_huge_kernel (frames beyond select_kernel's)
size_t select_huge_lds_bytes(int H, int W);

// ---------------------------------------------------------------------------
// brute-force descriptor matching (match.hip)
// One "side" = `pairs` blocks `stride` bytes apart, each holding an int32 count at off_cnt
// and count x 256 f32 descriptors at off_desc (a record of the extraction path has this
// shape; so has the staging block of the host API).  cap bounds the count.
// ---------------------------------------------------------------------------
struct MatchSide {
  const uint8_t *base;
  size_t stride, off_cnt, off_desc;
  int cap;
  int desc_bf16 = 0;   // rows are 256 bf16 (records made with SPFE_FLAG_DESC_BF16): widened on load, distances in f32 on those values
};
// out: per pair `out_stride` bytes: int32 train_idx[query.cap] (-1 = none), float dist[query.cap].
// best_t: [pairs][train.cap], best_q: [pairs][query.cap] scratch.
hipError_t launch_match(const MatchSide &query, const MatchSide &train, int pairs, bool cross_check,
                        unsigned long long *best_t, unsigned long long *best_q, uint8_t *out, size_t out_stride,
                        hipStream_t s);

// the two nearest train rows per query (cv::DescriptorMatcher::knnMatch(query, matches, 2)); out per pair:
// idx1[query.cap] | dist1[query.cap] | idx2[query.cap] | dist2[query.cap]
hipError_t launch_match_knn2(const MatchSide &query, const MatchSide &train, int pairs, unsigned long long *best1,
                             unsigned long long *best2, uint8_t *out, size_t out_stride, hipStream_t s);

// ---------------------------------------------------------------------------
// input staging (stage_input.hip): raw camera frames -> cropped gray u8 frames
// ---------------------------------------------------------------------------
struct StageParams {
  const uint8_t *src;        // [n] frames, src_frame_bytes apart, rows src_stride bytes apart
  size_t src_frame_bytes;
  int src_stride, src_h, src_w;
  const float *map_x, *map_y;  // [src_h][src_w] or null (no remap)
  int rgb;                     // 1: R first
  uint8_t *gray;               // [n][H][W]
  int H, W;
};
hipError_t launch_stage_input(const StageParams &p, int channels, int n, hipStream_t s);

// patch-wise association of projected map points (match.hip; tracker_dust.cpp:113-172)
struct PatchArgs {
  const float *mp_desc;  // [n_points][256]
  const float *mp_uv;    // [n_points][2] projected dust-map position (cells)
  int n_points;
  const int16_t *occ;    // [hc][wc] keypoint index per cell, -1 = empty
  int hc, wc;
  const float *kp_desc;  // [K][256]
  int kp_desc_bf16 = 0;  // ... as bf16 rows (a record made with SPFE_FLAG_DESC_BF16)
  const int *k_ptr;      // K on the device (a record header), or null -> k_imm
  int k_imm;
  // chained form (spfe_track_dust_record_device): map points whose in_view flag is 0 are skipped
  // (tracker_dust.cpp:114-116), and all of them when *gate_ptr < gate_min (the tracker gives up at :97-102)
  const uint8_t *in_view;  // [n_points] or null
  const int *gate_ptr;     // n_inlier of the alignment on the device, or null
  int gate_min;
};
// cand_idx / cand_dist: [n_points][4] scratch; out: [n_points] keypoint index or -1; kcap >= K
hipError_t launch_match_patches(const PatchArgs &a, int kcap, float max_dist, int *cand_idx, float *cand_dist,
                                int32_t *out, hipStream_t s);

// direct "dust" alignment (dust.hip; optimizer_dust.cpp:170-294): one workgroup, Levenberg-Marquardt over n points
constexpr int DUST_MAX_POINTS = 512;
struct DustArgs {
  const float *dust;   // [hc][wc] dense_dust (device)
  int hc, wc;
  const float *pts;    // [n][3] world positions (device)
  int n;
  const float *Tcw_in; // [16] row-major 4x4 (device)
  float fx, fy, cx, cy;   // full-resolution intrinsics
  int max_iterations;
  double delta, inlier_chi2;
  float *Tcw_out;      // [16]
  unsigned char *inlier;  // [n]
  float *uv;           // [n][2]
  int *counts;         // n_inlier, iterations
  // batched form (one workgroup per frame f = blockIdx.x): byte strides added to dust / pts+Tcw_in / the four outputs;
  // n_dev (or null) = per-frame point counts on the device
  int nframes;
  size_t dust_stride, pts_stride, pose_stride, out_stride;
  const int *n_dev;
  int map_in_lds;   // set by launch_dust_align: the dust map fits in LDS beside the per-point arrays
};
hipError_t launch_dust_align(const DustArgs &a, hipStream_t s);
size_t dust_lds_bytes(int hc, int wc);

// exact-math probe kernels for tests (device bits vs host bits)
hipError_t launch_math_probe(const float *in, float *out_exp, float *out_log, int n, hipStream_t s);

}  // namespace spfe
