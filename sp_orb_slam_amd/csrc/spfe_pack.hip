// spfe_pack.hip — construction of a handle: the weight blob (register_module order, sp_extractor.cpp:46-62) packed into each
// kernel family's device tables, the activation / scratch / record buffers, streams and events (SPExtractor's constructor,
// /root/reference/orb_slam2/src/cv/sp_extractor.cpp:342-359).
#include "spfe_host.h"

namespace spfe_host {

void make_layout(int kmax, int C, bool desc_bf16, spfe::RecordLayout *r) {
  size_t o = 0;
  r->kmax = kmax;
  r->desc_bf16 = desc_bf16 ? 1 : 0;
  r->off_hdr = o; o += 16;
  r->off_xy = o; o = align_up(o + (size_t)kmax * 2 * 4, 16);
  r->off_resp = o; o = align_up(o + (size_t)kmax * 4, 16);
  r->off_cov = o; o = align_up(o + (size_t)kmax * 2 * 4, 16);
  r->off_cinv = o; o = align_up(o + (size_t)kmax * 2 * 4, 16);
  r->off_desc = o; o = align_up(o + (size_t)kmax * SPFE_DESC_DIM * (desc_bf16 ? 2 : 4), 16);
  r->off_occ = o; o = align_up(o + (size_t)C * 2, 16);
  r->off_dd = o; o = align_up(o + (size_t)C * 4, 16);
  r->off_sd = o; o = align_up(o + (size_t)C * 4, 16);
  r->bytes = align_up(o, 256);
}

// offsets into the flat blob (register_module order, sp_extractor.cpp:46-62)
size_t blob_weight_offset(int l) {
  size_t off = 0;
  for (int i = 0; i < l; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[i];
    off += (size_t)L.cout * L.cin * L.ksize * L.ksize + L.cout;
  }
  return off;
}

// pack OIHW weights of one or two layers (concatenated along cout) into slabs
// [nblk][chunk][n-tile(2)][tap][KC][32] (K order of spfe_exact_math.h) + padded bias
int pack_layer(spfe_handle h, const float *blob, const int *lids, int nl, ConvLayer *out) {
  const spfe_layer_t &L0 = SPFE_LAYERS[lids[0]];
  const int cin = L0.cin, ks = L0.ksize, taps = ks * ks;
  int cout = 0;
  for (int i = 0; i < nl; ++i) cout += SPFE_LAYERS[lids[i]].cout;
  const int kc = spfe::conv_kc(ks), nchunk = cin / kc, nblk = (cout + 63) / 64;
  std::vector<float> w((size_t)nblk * nchunk * taps * kc * 64, 0.0f), bia((size_t)nblk * 64, 0.0f);
  int co_base = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    const float *W = blob + blob_weight_offset(lids[i]);
    const float *Bv = W + (size_t)L.cout * L.cin * taps;
    for (int co = 0; co < L.cout; ++co) {
      const int g = co_base + co, nb = g / 64, j = g % 64;
      bia[g] = Bv[co];
      for (int ci = 0; ci < cin; ++ci) {
        const int ch = ci / kc, c = ci % kc;
        for (int t = 0; t < taps; ++t)
          w[(((((size_t)nb * nchunk + ch) * 2 + j / 32) * taps + t) * kc + c) * 32 + j % 32] =
              W[((size_t)co * cin + ci) * taps + t];
      }
    }
    co_base += L.cout;
  }
  int rc;
  if ((rc = dev_alloc(h, &out->d_w, w.size()))) return rc;
  if ((rc = dev_alloc(h, &out->d_b, bia.size()))) return rc;
  HIP_TRY(hipMemcpy(out->d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(out->d_b, bia.data(), bia.size() * 4, hipMemcpyHostToDevice));
  out->cin = cin;
  out->cout_real = cout;
  out->nblk = nblk;
  out->ks = ks;
  return SPFE_OK;
}

unsigned short host_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// bf16 slabs for conv_bf16.hip: [nblk][chunk of 32 channels][tap][n 64][80-byte row: 32 bf16 + pad],
// each slab padded to conv_bf16_slab_bytes(); bias stays f32
int pack_layer_bf16(spfe_handle h, const float *blob, const int *lids, int nl, ConvLayer *out) {
  const spfe_layer_t &L0 = SPFE_LAYERS[lids[0]];
  const int cin = L0.cin, taps = 9;
  int cout = 0;
  for (int i = 0; i < nl; ++i) cout += SPFE_LAYERS[lids[i]].cout;
  const int nchunk = cin / 32, nblk = (cout + 63) / 64;
  const size_t slab = spfe::conv_bf16_slab_bytes();
  std::vector<unsigned char> w((size_t)nblk * nchunk * slab, 0);
  std::vector<float> bia((size_t)nblk * 64, 0.0f);
  int co_base = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    const float *W = blob + blob_weight_offset(lids[i]);
    const float *Bv = W + (size_t)L.cout * L.cin * taps;
    for (int co = 0; co < L.cout; ++co) {
      // row of the 64-channel block: even channels fill accumulator tile 0, odd ones tile 1 (the kernels pack a lane's
      // channel pair into one dword store); the bias stays in channel order
      const int g = co_base + co, nb = g / 64, c64 = g % 64, j = (c64 & 1) * 32 + (c64 >> 1);
      bia[g] = Bv[co];
      for (int ci = 0; ci < cin; ++ci) {
        const int ch = ci / 32, c = ci % 32;
        for (int t = 0; t < taps; ++t) {
          const unsigned short v = host_bf16_rne(W[((size_t)co * cin + ci) * taps + t]);
          // row (tap, cout) = 64 B: 4 pieces of 8 channels, piece g in slot g ^ ((cout >> 2) & 3) (conv_bf16.hip's LDS layout)
          memcpy(&w[((size_t)nb * nchunk + ch) * slab + ((size_t)t * 64 + j) * 64 + (((c / 8) ^ ((j >> 2) & 3)) * 16) + (c % 8) * 2], &v, 2);
        }
      }
    }
    co_base += L.cout;
  }
  int rc;
  unsigned char *dw = nullptr;
  if ((rc = dev_alloc(h, &dw, w.size()))) return rc;
  if ((rc = dev_alloc(h, &out->d_b, bia.size()))) return rc;
  HIP_TRY(hipMemcpy(dw, w.data(), w.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(out->d_b, bia.data(), bia.size() * 4, hipMemcpyHostToDevice));
  out->d_w = reinterpret_cast<float *>(dw);
  out->cin = cin;
  out->cout_real = cout;
  out->nblk = nblk;
  out->ks = 3;
  return SPFE_OK;
}

// conv_bf16_ws.hip layout: [nblk][tap][cout 64][8 pieces of 8 cin, piece g in slot g ^ ((cout >> 1) & 7)]
int pack_layer_bf16_ws(spfe_handle h, const float *blob, int lid, unsigned char **out) {
  const spfe_layer_t &L = SPFE_LAYERS[lid];
  if (L.cin != 64 || L.ksize != 3 || L.cout % 64) return fail(SPFE_EINVAL, "internal: layer %d is not a Cin = 64 3x3 layer", lid);
  const size_t blk = spfe::conv_bf16_ws_weight_bytes();
  std::vector<unsigned char> w((size_t)(L.cout / 64) * blk, 0);
  const float *W = blob + blob_weight_offset(lid);
  for (int co = 0; co < L.cout; ++co)
    for (int ci = 0; ci < 64; ++ci)
      for (int t = 0; t < 9; ++t) {
        const unsigned short v = host_bf16_rne(W[((size_t)co * 64 + ci) * 9 + t]);
        // row of the block: even channels fill accumulator tile 0, odd ones tile 1 (conv_bf16_ws.hip's epilogue
        // packs a lane's channel pair into one dword store)
        const int c64 = co % 64, j = (c64 & 1) * 32 + (c64 >> 1), slot = (ci / 8) ^ ((j >> 1) & 7);
        memcpy(&w[(size_t)(co / 64) * blk + ((size_t)t * 64 + j) * 128 + slot * 16 + (ci % 8) * 2], &v, 2);
      }
  int rc;
  if ((rc = dev_alloc(h, out, w.size()))) return rc;
  HIP_TRY(hipMemcpy(*out, w.data(), w.size(), hipMemcpyHostToDevice));
  return SPFE_OK;
}

// conv_bf16_rw.hip layout for a Cin = 128 layer (or two concatenated ones: convPa | convDa)
int pack_layer_bf16_rw(spfe_handle h, const float *blob, const int *lids, int nl, unsigned char **out) {
  int cout = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    if (L.cin != 128 || L.ksize != 3) return fail(SPFE_EINVAL, "internal: layer %d is not a Cin = 128 3x3 layer", lids[i]);
    cout += L.cout;
  }
  if (cout % 128) return fail(SPFE_EINVAL, "internal: %d output channels are not whole 128-channel groups", cout);
  std::vector<unsigned short> wb((size_t)cout * 128 * 9);
  size_t o = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    const float *W = blob + blob_weight_offset(lids[i]);
    for (size_t k = 0; k < (size_t)L.cout * 128 * 9; ++k) wb[o++] = host_bf16_rne(W[k]);
  }
  std::vector<unsigned char> w((size_t)(cout / 128) * spfe::conv_bf16_rw_weight_bytes());
  spfe::conv_bf16_rw_pack_weights(wb.data(), cout, w.data());
  int rc;
  if ((rc = dev_alloc(h, out, w.size()))) return rc;
  HIP_TRY(hipMemcpy(*out, w.data(), w.size(), hipMemcpyHostToDevice));
  return SPFE_OK;
}

int load_blob(const spfe_config *cfg, std::vector<float> *blob) {
  blob->resize(SPFE_NUM_PARAMS);
  if (cfg->weights) {
    memcpy(blob->data(), cfg->weights, (size_t)SPFE_NUM_PARAMS * 4);
    return SPFE_OK;
  }
  if (!cfg->weights_path) return fail(SPFE_EWEIGHTS, "no weights: both weights and weights_path are NULL");
  FILE *f = fopen(cfg->weights_path, "rb");
  if (!f) return fail(SPFE_EWEIGHTS, "cannot open weight file %s", cfg->weights_path);
  unsigned char head[16];
  uint32_t ver = 0;
  uint64_t n = 0;
  bool ok = fread(head, 1, 16, f) == 16 && memcmp(head, "SPFW", 4) == 0;
  if (ok) {
    memcpy(&ver, head + 4, 4);
    memcpy(&n, head + 8, 8);
    ok = ver == 1 && n == SPFE_NUM_PARAMS && fread(blob->data(), 4, n, f) == n;
  }
  fclose(f);
  if (!ok) return fail(SPFE_EWEIGHTS, "%s is not a valid SPFW v1 file with %d params", cfg->weights_path, SPFE_NUM_PARAMS);
  return SPFE_OK;
}


// Every environment switch libspfe.so reads — ONCE per handle, here, at spfe_create (README.md "Environment switches" lists
// them with their tests).  A switch either selects between product paths that give the same bits (so that a test can put
// two of them side by side), or sizes a capacity so that a test can reach an overflow path.  Nothing else in the library
// calls getenv, and no value is cached across handles.
static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? (int)strtol(e, nullptr, 0) : dflt;
}
// "a,b,c": field k as an int, dflt where the field is missing or empty
static int env_field(const char *name, int k, int dflt) {
  const char *e = getenv(name);
  if (!e) return dflt;
  for (int i = 0; i < k; ++i) {
    e = strchr(e, ',');
    if (!e) return dflt;
    ++e;
  }
  return (*e && *e != ',') ? (int)strtol(e, nullptr, 0) : dflt;
}
static void read_switches(spfe_handle h) {
  // schedule
  const int st = env_int("SPFE_STAGE_TIMING", 0);            // 1: events around every stage, 2: around the dominant kernel only
  h->timing = st != 0;
  h->timing_all = st != 2;
  h->split_mode = env_int("SPFE_SPLIT", -1);                 // layers behind conv1b as two half batches: -1 by workload, 0 never, 1 always
  h->inline_chain = env_int("SPFE_INLINE_CHAIN", 1) != 0;    // synchronous calls: detector chain on the launch stream
  h->defer_join = env_int("SPFE_DEFER_JOIN", 1) != 0;        // pipelined two-half steps: the join in front of the NEXT conv1b
  h->tail_per_half = env_int("SPFE_TAIL_PER_HALF", 1) != 0;  // ... each half's tail right behind its convPa
  h->two_chains_env = env_int("SPFE_TWO_CHAINS", -1);        // a twin handle, pipelined device calls alternate: -1 by workload, 0 never, 1 always, 99 wanted + its build fails (tests)
  h->sel_ext_event = env_int("SPFE_SEL_EXT_EVENT", 1) != 0;  // the selection's completion signal as the descriptor branch's event
  h->replay_waves = env_int("SPFE_REPLAY_WAVES", 0);         // components per replay workgroup: 0 by workload, 2 | 8
  h->zero_in_tail = env_int("SPFE_ZERO_IN_TAIL", 1) != 0;    // bf16 tile-queue counters cleared by the previous call's tail
  // which branch
  h->sparse_db_env = env_int("SPFE_SPARSE_DB", -1);          // gathered descriptor head: 0 never, 1 every call, 2 synchronous calls
  h->sparse_da_env = env_int("SPFE_SPARSE_DA", -1);          // gathered convDa: 0 never, 1 synchronous calls, 2 pipelined too
  h->pbtail_env = env_int("SPFE_PBTAIL", 1);                 // convPb inside the tail's launch: 0 off, 1 on, 2 | 4: on, bf16 form on that many wavefronts
  h->f32_heads = env_int("SPFE_F32_HEADS", 0) != 0;          // dense f32 convPb / convDb on head_f32.hip
  h->fuse1a_env = env_int("SPFE_FUSE_CONV1A", -1);           // conv1a inside conv1b: f32 default 0, bf16 default 1
  // host path
  h->early_heat_copy = env_int("SPFE_EARLY_HEAT_COPY", 1) != 0;   // synchronous host calls: the heat maps' D2H behind the normalisation, beside the chain
  h->pipe_copy_kernel = env_int("SPFE_PIPE_COPY_KERNEL", -1);   // D2H of a pipelined batch: -1 by precision, 0 runtime copy, 1 copy kernel
  // (SPFE_COMM_OWN_STREAM — the all-gather on a stream of its own instead of the side stream — belongs to the communicator: read
  // by spfe_comm_init, spfe_comm.hip, each time one is made)
  // f32 tile shapes (bit-identical: the tile shape does not touch an output's K order)
  h->tile16x4 = env_int("SPFE_TILE16X4", 1);                 // conv1b 16-row tiles: 0 never, 1 cost model (+ cut), 2 always, 3 model without the cut
  {
    const int m = env_int("SPFE_TILE2_MASK", -1);            // 2-row tiles: unset = cost model, 0 = never, else the layers forced onto them
    h->tile2_auto = m < 0;
    h->tile2_mask = m > 0 ? (unsigned)m : 0u;
  }
  h->select_huge_env = env_int("SPFE_SELECT_HUGE", 0);       // 1: the selection of frames of any size on select_huge_kernel (the form for > 65,535 cells)
  h->pool_split = env_int("SPFE_POOL_SPLIT", -1);            // pooled layer as un-pooled 2-row tiles + pool pass: -1 model, 0 never, 1 wherever possible
  // bf16 kernel selection (bit-identical kernels; which one takes a launch is a size decision)
  h->ws_mask = (unsigned)env_field("SPFE_BF16_WS", 0, 15) & 0xfu;   // "mask[,min_items]": Cin = 64 layers that may take conv_bf16_ws.hip
  if (getenv("SPFE_BF16_WS") && strchr(getenv("SPFE_BF16_WS"), ','))
    h->ws_min_items = h->ws_min_items_sync = env_field("SPFE_BF16_WS", 1, h->ws_min_items);
  h->bf16_dyn = env_int("SPFE_BF16_DYN_QUEUE", 1) != 0;      // conv_bf16.hip: work items in queue order
  h->tile_rows_big = env_field("SPFE_BF16_TILE_ROWS", 0, 12);       // "rows[,min_items]": conv_bf16.hip's taller tiles (12 | 16), 0 items = never
  h->tile16_min_items = env_field("SPFE_BF16_TILE_ROWS", 1, 3);
  h->bf16_rw = env_field("SPFE_BF16_RW", 0, 1) != 0;         // "on[,min4,min2,rows3]": conv_bf16_rw.hip for the Cin = 128 layers
  h->rw_min4 = env_field("SPFE_BF16_RW", 1, 3);
  h->rw_min2 = env_field("SPFE_BF16_RW", 2, 2);
  h->rw_rows3 = env_field("SPFE_BF16_RW", 3, 1);
  // capacities of the covariance stage (tests reach the overflow paths): "qcap,ovf_slots,ovf_cap,fallback_cap,ecap"
  h->cov.qcap = std::max(16, env_field("SPFE_COV_CAPS", 0, 1024));
  h->cov.ovf_slots = std::max(0, env_field("SPFE_COV_CAPS", 1, 16));
  h->cov.ovf_cap = std::max(h->cov.qcap, env_field("SPFE_COV_CAPS", 2, 16384));
  h->cov.fb_cap = std::max(1024, env_field("SPFE_COV_CAPS", 3, 1 << 22));
  h->cov_gen_start = std::min(32766, std::max(2, env_field("SPFE_COV_CAPS", 5, 32766)));   // first generation code of the claim / done maps (counts down; a small one reaches the wrap)
  h->cov_ecap_env = env_field("SPFE_COV_CAPS", 4, -1);       // -1: 32 x kmax; 0: no edge list (the link kernel walks the pop lists)
}

// A twin runs its sibling's switches: they were read once, at the sibling's spfe_create (the environment may have changed
// since — the twin of a handle that pipelines through spfe_submit_batch is built at the first submission).  Same list as
// read_switches().
static void copy_switches(spfe_handle h, const spfe_handle_s *o) {
  h->timing = o->timing; h->timing_all = o->timing_all; h->split_mode = o->split_mode; h->inline_chain = o->inline_chain;
  h->defer_join = o->defer_join; h->tail_per_half = o->tail_per_half; h->two_chains_env = o->two_chains_env;
  h->sel_ext_event = o->sel_ext_event; h->replay_waves = o->replay_waves; h->zero_in_tail = o->zero_in_tail;
  h->sparse_db_env = o->sparse_db_env; h->sparse_da_env = o->sparse_da_env; h->pbtail_env = o->pbtail_env;
  h->f32_heads = o->f32_heads; h->fuse1a_env = o->fuse1a_env; h->pipe_copy_kernel = o->pipe_copy_kernel;
  h->early_heat_copy = o->early_heat_copy;
  h->tile16x4 = o->tile16x4; h->tile2_auto = o->tile2_auto; h->tile2_mask = o->tile2_mask;
  h->select_huge_env = o->select_huge_env; h->pool_split = o->pool_split;
  h->ws_mask = o->ws_mask; h->ws_min_items = o->ws_min_items; h->ws_min_items_sync = o->ws_min_items_sync;
  h->bf16_dyn = o->bf16_dyn; h->tile_rows_big = o->tile_rows_big; h->tile16_min_items = o->tile16_min_items;
  h->bf16_rw = o->bf16_rw; h->rw_min4 = o->rw_min4; h->rw_min2 = o->rw_min2; h->rw_rows3 = o->rw_rows3;
  h->cov.qcap = o->cov.qcap; h->cov.ovf_slots = o->cov.ovf_slots; h->cov.ovf_cap = o->cov.ovf_cap; h->cov.fb_cap = o->cov.fb_cap;
  h->cov_gen_start = o->cov_gen_start; h->cov_ecap_env = o->cov_ecap_env;
}

int build(spfe_handle h, const spfe_config *cfg, spfe_handle sibling) {
  h->cfg = *cfg;
  // the caller's weight memory is read HERE and never again: the stored configuration does not point into it
  h->cfg.weights = nullptr;
  h->cfg.weights_path = nullptr;
  if (sibling) copy_switches(h, sibling);
  else read_switches(h);
  h->H = cfg->height; h->W = cfg->width;
  h->hc = h->H / 8; h->wc = h->W / 8; h->C = h->hc * h->wc;
  h->kmax = cfg->num_features + 1;
  h->B = cfg->max_batch;
  h->bf16 = cfg->precision == SPFE_PRECISION_BF16;
  const int H = h->H, W = h->W, B = h->B, C = h->C;
  HIP_TRY(hipSetDevice(cfg->device));
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    h->num_cus = prop.multiProcessorCount;
  }
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  // (measured and not kept: the side stream at another priority; confined to N CUs by hipExtStreamCreateWithCUMask — 64 CUs
  // -15 %, 128 -2.5 %, 160 -0.8 % at bf16 1280x720: HISTORY.md "Round 4")
  HIP_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
  for (int i = 0; i < spfe_handle_s::NTICKET; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&h->ev_post[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_cov[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&h->ev_desc, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_db, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_sel, hipEventDisableTiming));
  for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&h->ev_dbs[i], hipEventDisableTiming));
  // Measured, pipelined, 8 frames per call (same-box A/B): bf16 1280x720 +2.5 ... 3.8 % (7590 -> 7780, 7322 -> 7604 frames/s),
  // f32 752x480 +0.4 ... 0.7 %, bf16 752x480 -2 ... 3 %: there the launch stream runs as two half batches on two streams, the
  // dense head (HBM-bound) hid completely beside the other half's convolutions (removing it altogether gains nothing), and
  // the gathered launch is pure extra work for the chip.  So: f32, and bf16 frames of >= 10,000 cells (= no two-stream split).
  // SYNCHRONOUS calls of those small bf16 frames take the gathered branch all the same (round 4): there is no other half
  // batch to hide the dense head beside, and the gathered form brings the inline chain with it (enqueue_post) — 752x480:
  // a single frame's p50 0.287 -> 0.264 ... 0.274 ms, 8 frames per synchronous call +0.5 ... 0.8 %; 640x480: 0.311 -> 0.288
  // ms, +2.4 %.  SPFE_SPARSE_DB = 0 never, 1 every call, 2 synchronous calls only
  h->sparse_db = true;
  h->sparse_db_sync_only = h->bf16 && h->C < 10000;
  h->db_tiles_per_wg = h->bf16 ? 4 : 1;
  if (h->sparse_db_env >= 0) { h->sparse_db = h->sparse_db_env != 0; h->sparse_db_sync_only = h->sparse_db_env == 2; }
  // the gathered kernels form row byte offsets in 32 bits (the head activations' rows are 2048 / 1024 bytes, 0x80000000 is their
  // out-of-range marker): batches beyond that take the dense head (the launchers refuse them as well)
  if ((size_t)cfg->max_batch * h->C * (h->bf16 ? 1024 : 2048) >= ((size_t)1 << 31)) h->sparse_db = false;
  h->fuse1a = !h->bf16 && h->fuse1a_env > 0;                      // f32: opt-in (measured perf-neutral)
  h->fuse1a_bf16 = h->bf16 && h->fuse1a_env != 0;                 // bf16: conv1b's producer waves compute conv1a
  // bf16, Cin = 64 layers, the wave-specialised kernel's bar.  Synchronous calls (latency): it wins from ~5 items per workgroup
  // (batch 1 at 752x480: 0.43 -> 0.385 ms, conv1a fused).  Pipelined calls (SPFE_FLAG_ASYNC_COV): it holds all of a CU's LDS, the
  // side-stream kernels of the previous batch cannot start beside it, and at 752x480 x 8 (0.65 ms steps) their chain becomes
  // the critical path when the quarter-resolution layers take it too (12,450 -> 12,050 frames/s): those keep the higher bar
  // (the bar is picked per call: spfe_submit_batch pipelines on a handle created without the flag)
  if (h->timing) {
    h->evpool.resize((size_t)spfe_handle_s::EVSETS * (NSTAGE + 1), nullptr);
    for (auto &e : h->evpool) HIP_TRY(hipEventCreate(&e));
  }

  int rc;
  if (!sibling) {
    if ((rc = load_blob(cfg, &h->blob))) return rc;
  } else if (sibling->blob.size() != (size_t)SPFE_NUM_PARAMS) return fail(SPFE_EWEIGHTS, "internal: the sibling's weight blob is gone");
  const std::vector<float> &blob = sibling ? sibling->blob : h->blob;

  // conv1a weights: [tap][64]
  {
    const float *Wt = blob.data() + blob_weight_offset(0);
    std::vector<float> w(9 * 64), bv(64);
    for (int co = 0; co < 64; ++co) {
      for (int t = 0; t < 9; ++t) w[t * 64 + co] = Wt[co * 9 + t];
      bv[co] = Wt[64 * 9 + co];
    }
    if ((rc = dev_alloc(h, &h->d_w1a, w.size()))) return rc;
    if ((rc = dev_alloc(h, &h->d_b1a, bv.size()))) return rc;
    HIP_TRY(hipMemcpy(h->d_w1a, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_b1a, bv.data(), bv.size() * 4, hipMemcpyHostToDevice));
    if (h->bf16) {
      // conv1a_mfma.h: [j 2][lane 64][e 8] = bf16(w[channel 32 j + row_channel(lane & 31)][tap 8 (lane >> 5) + e]), 0 for taps >= 9
      // (row_channel: bits 2 and 3 of the row swapped, so that a lane of the product holds whole 16-byte pieces)
      std::vector<unsigned short> tab(2 * 64 * 8, 0);
      for (int j = 0; j < 2; ++j)
        for (int ln = 0; ln < 64; ++ln)
          for (int e = 0; e < 8; ++e) {
            const int t = 8 * (ln >> 5) + e, m = ln & 31, co = 32 * j + ((m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1));
            if (t < 9) tab[(j * 64 + ln) * 8 + e] = host_bf16_rne(Wt[co * 9 + t] * (1.0f / 255.0f));   // the 1/255 of convertTo folded into the weight before its rounding (conv1a_mfma.h)
          }
      if ((rc = dev_alloc(h, &h->d_w1a_tab, tab.size()))) return rc;
      HIP_TRY(hipMemcpy(h->d_w1a_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    }
  }

  // activations (NHWC f32).  act[0]=conv1a .. act[7]=conv4b
  const int lh[8] = {H, H / 2, H / 2, H / 4, H / 4, H / 8, H / 8, H / 8};
  const int lw[8] = {W, W / 2, W / 2, W / 4, W / 4, W / 8, W / 8, W / 8};
  const int lc[8] = {64, 64, 64, 64, 128, 128, 128, 128};
  for (int i = 0; i < 8; ++i)
    if ((rc = dev_alloc(h, &h->act[i], (size_t)B * lh[i] * lw[i] * lc[i]))) return rc;
  if ((rc = dev_alloc(h, &h->d_img, (size_t)B * H * W))) return rc;
  if (!h->bf16 && h->pool_split != 0 && !(H & 15) && !(W & 15))   // (the un-pooled output of the largest pooled layer this serves: conv2b / conv3b of <= 2 frames)
    if ((rc = dev_alloc(h, &h->d_unpooled, (size_t)std::min(B, 2) * (H / 2) * (W / 2) * 64))) return rc;
  if ((rc = dev_alloc(h, &h->d_head, (size_t)B * C * 512))) return rc;
  if ((rc = dev_alloc(h, &h->d_semi, (size_t)B * C * SPFE_SEMI_CH))) return rc;
  if ((rc = dev_alloc(h, &h->d_coarse, (size_t)B * C * SPFE_DESC_DIM))) return rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = dev_alloc(h, &h->d_heat_log[k], (size_t)B * H * W))) return rc;
    if ((rc = dev_alloc(h, &h->d_minmax[k], (size_t)B * spfe::tail_parts(h->H, h->W) * 2))) return rc;
    if ((rc = dev_alloc(h, &h->d_cell_score[k], (size_t)B * C))) return rc;
    if ((rc = dev_alloc(h, &h->d_cell_k[k], (size_t)B * C))) return rc;
  }
  if ((rc = dev_alloc(h, &h->d_heat_inv, (size_t)B * H * W))) return rc;
  if (cfg->flags & SPFE_FLAG_HEAT)
    if ((rc = dev_alloc(h, &h->d_heat, (size_t)B * H * W))) return rc;
  if ((rc = dev_alloc(h, &h->d_heat_consts, (size_t)B * 4))) return rc;
  if ((rc = dev_alloc(h, &h->d_cell_mask, (size_t)B * C))) return rc;
  if ((rc = dev_alloc(h, &h->d_kp_cell, (size_t)B * h->kmax))) return rc;
  if (h->sparse_db) {
    h->db_cap = (int)std::min<size_t>((size_t)4 * h->kmax, (size_t)C);
    if ((rc = dev_alloc(h, &h->d_db_list, (size_t)B * h->db_cap))) return rc;
    if ((rc = dev_alloc(h, &h->d_db_total, 16))) return rc;
    HIP_TRY(hipMemset(h->d_db_total, 0, 16 * sizeof(int)));
  }
  {   // select_kernel's global scratch: frames of more than 16,384 cells
    if ((rc = dev_alloc(h, &h->d_sel_slot, (size_t)B * C))) return rc;
    if ((rc = dev_alloc(h, &h->d_sel_list, (size_t)B * C))) return rc;
  }
  h->select_huge = (size_t)C > spfe::select_max_cells() || h->select_huge_env != 0;
  if (h->select_huge) {   // select_huge_kernel's: states and a 32-bit list
    if ((rc = dev_alloc(h, &h->d_sel_state, (size_t)B * C))) return rc;
    if ((rc = dev_alloc(h, &h->d_sel_list32, (size_t)B * C))) return rc;
  }
  {
    if ((rc = dev_alloc(h, &h->cov.claim, (size_t)B * H * W))) return rc;
    if ((rc = dev_alloc(h, &h->cov.done, (size_t)B * H * W))) return rc;
    if ((rc = dev_alloc(h, &h->cov.queue, (size_t)B * h->kmax * h->cov.qcap))) return rc;
    if ((rc = dev_alloc(h, &h->cov.qval, (size_t)B * h->kmax * h->cov.qcap))) return rc;
    if ((rc = dev_alloc(h, &h->cov.npop, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.dirty, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.nxt, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.nxy, (size_t)B * h->kmax * 2))) return rc;
    if ((rc = dev_alloc(h, &h->cov.workers, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.counters, (size_t)B * 4))) return rc;
    h->cov.ecap = h->cov_ecap_env >= 0 ? h->cov_ecap_env : 32 * h->kmax;   // (~24 pops per keypoint on the dense synthetic detector, a quarter of the keypoints dirty)
    if (h->cov.ecap && (rc = dev_alloc(h, &h->cov.edges, (size_t)B * h->cov.ecap * 2))) return rc;
    if ((rc = dev_alloc(h, &h->cov.ovf_slot, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.ovf_q, (size_t)B * h->cov.ovf_slots * h->cov.ovf_cap + 1))) return rc;
    if ((rc = dev_alloc(h, &h->cov.ovf_v, (size_t)B * h->cov.ovf_slots * h->cov.ovf_cap + 1))) return rc;
    // the device-side last resort (cov.hip, cov_fallback_kernel): one list for the batch, 4 M pops by default (48 MB)
    if ((rc = dev_alloc(h, &h->cov.fb_q, (size_t)h->cov.fb_cap))) return rc;
    if ((rc = dev_alloc(h, &h->cov.fb_v, (size_t)h->cov.fb_cap))) return rc;
  }
  make_layout(h->kmax, C, (cfg->flags & SPFE_FLAG_DESC_BF16) != 0, &h->rl);
  if ((rc = dev_alloc(h, &h->d_records, (size_t)B * h->rl.bytes))) return rc;
  HIP_TRY(hipMemset(h->d_records, 0, (size_t)B * h->rl.bytes));

  // the MFMA conv chain
  struct Spec { int nl, l0, l1, src, dst; bool pool; };
  // src/dst index into act[]; -1 = head buffer
  const Spec specs[8] = {{1, 1, 0, 0, 1, true},  {1, 2, 0, 1, 2, false}, {1, 3, 0, 2, 3, true},
                         {1, 4, 0, 3, 4, false}, {1, 5, 0, 4, 5, true},  {1, 6, 0, 5, 6, false},
                         {1, 7, 0, 6, 7, false}, {2, 8, 10, 7, -1, false}};
  const int small_maxh = -1;   // (the 4- / 8-row choice is enqueue()'s, per call)
  h->small_maxh = small_maxh;
  for (int i = 0; i < 8; ++i) {
    ConvLayer &L = h->layers[i];
    const int lids[2] = {specs[i].l0, specs[i].l1};
    if (h->bf16) rc = pack_layer_bf16(h, blob.data(), lids, specs[i].nl, &L);
    else rc = pack_layer(h, blob.data(), lids, specs[i].nl, &L);
    if (rc) return rc;
    L.pool = specs[i].pool;
    L.relu = true;
    L.H = lh[specs[i].src];
    L.W = lw[specs[i].src];
    L.small_tile = L.H <= small_maxh;
    L.in = h->act[specs[i].src];
    L.in_stride = lc[specs[i].src];
    L.in_choff = 0;
    if (specs[i].dst >= 0) { L.out = h->act[specs[i].dst]; L.out_stride = lc[specs[i].dst]; }
    else { L.out = h->d_head; L.out_stride = 512; }
    L.out_choff = 0;
  }
  {  // convPb: head[0:256] -> semi (65)
    ConvLayer &L = h->layers[8];
    const int lids[1] = {9};
    if ((rc = pack_layer(h, blob.data(), lids, 1, &L))) return rc;
    L.pool = false; L.relu = false; L.small_tile = true; L.H = H / 8; L.W = W / 8;
    L.in = h->d_head; L.in_stride = 512; L.in_choff = 0;
    L.out = h->d_semi; L.out_stride = SPFE_SEMI_CH; L.out_choff = 0;
  }
  {  // convDb: head[256:512] -> coarse (256)
    ConvLayer &L = h->layers[9];
    const int lids[1] = {11};
    if ((rc = pack_layer(h, blob.data(), lids, 1, &L))) return rc;
    L.pool = false; L.relu = false; L.small_tile = true; L.H = H / 8; L.W = W / 8;
    L.in = h->d_head; L.in_stride = 512; L.in_choff = 256;
    L.out = h->d_coarse; L.out_stride = SPFE_DESC_DIM; L.out_choff = 0;
  }
  if (h->bf16) {  // Cin = 64 layers selected for the wave-specialised kernel (conv1b by default)
    for (int i = 0; i < 4; ++i)
      if ((h->ws_mask >> i) & 1)
        if ((rc = pack_layer_bf16_ws(h, blob.data(), specs[i].l0, &h->d_wws[i]))) return rc;
    if (h->bf16_rw)
      for (int i = 4; i < 8; ++i) {
        const int lids2[2] = {specs[i].l0, specs[i].l1};   // (convPa | convDa for the last one)
        if ((rc = pack_layer_bf16_rw(h, blob.data(), lids2, specs[i].nl, &h->d_wrw[i - 4]))) return rc;
      }
    if ((rc = dev_alloc(h, &h->d_tile_ctr, 8 * 64))) return rc;   // [layer][part of the batch][32]
    h->sparse_da = h->sparse_db && h->d_wrw[3] && (size_t)B * C * 1024 < ((size_t)1 << 31);
    // Measured at 1280x720 x 8 (da_gather_bf16.hip): 25 us alone against the 43 us the dense launch loses without convDa, a
    // single-frame call's p50 0.357 -> 0.352 ms; but pipelined 7640 -> 7500 frames/s — a workgroup needs a whole CU (148 KB
    // of LDS, 380 registers), so beside the next batch's convolutions it only starts where one of theirs has ended, and
    // then holds that CU for its ~6 tiles.  So: synchronous calls only — until round 5: with the replay workers listed
    // longest chain first and the lone walks at 6 KB of LDS (cov.hip) the same A/B reads 7796 / 7830 -> 8159 / 8180 frames/s
    // (+4.5 %): pipelined calls take it too (mode 2; it only ever applies where the gathered convDb does: frames of >= 10,000 cells)
    h->sparse_da_mode = 2;
    if (h->sparse_da_env >= 0) h->sparse_da_mode = h->sparse_da_env;
    h->sparse_da = h->sparse_da && h->sparse_da_mode != 0;
    if (h->sparse_da && (rc = dev_alloc(h, &h->act7_alt, (size_t)B * C * 128))) return rc;
  }
  if (!h->bf16) {  // f32 heads with register-resident weights (head_f32.hip), bit-identical to the generic kernel — opt-in:
    // measured 63 + 38.5 us per eight 752x480 frames against 72 + 35.5 for the generic kernel (matrix-bound: 47 us at the peak)
    h->pbtail = h->pbtail_env != 0;
    if (h->f32_heads) h->pbtail = false;
    if (h->pbtail) {
      const float *Wp = blob.data() + blob_weight_offset(9);   // layer 9 = convPb, [65][256]
      if ((rc = dev_alloc(h, &h->d_wpb_dust, 256))) return rc;
      HIP_TRY(hipMemcpy(h->d_wpb_dust, Wp + (size_t)64 * 256, 256 * 4, hipMemcpyHostToDevice));
    }
    for (int which = 0; which < 2; ++which) {
      if (!h->f32_heads && !(which == 0 && h->sparse_db) && !(which == 1 && h->pbtail)) continue;   // (the gathered descriptor head is head_f32.hip's kernel; pbtail_f32.hip reads convPb's table)
      const int lid = which ? 9 : 11;
      const spfe_layer_t &Ld = SPFE_LAYERS[lid];
      std::vector<float> w(spfe::head_f32_weight_bytes(Ld.cout) / 4, 0.0f);
      spfe::head_f32_pack_weights(blob.data() + blob_weight_offset(lid), Ld.cout, w.data());
      float **dst = which ? &h->d_wpb32 : &h->d_wdb32;
      if ((rc = dev_alloc(h, dst, w.size()))) return rc;
      HIP_TRY(hipMemcpy(*dst, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    }
    // convDa gathered as well (da_gather_f32.hip), pipelined calls included: 752x480 x 8, 19 k of 45 k cells listed: 122 us
    // (two workgroups per CU) against the 210 us the dense launch loses without convDa — pipelined 2035 -> 2057 ... 2075
    // frames/s, a single-frame call's p50 -1.4 %
    h->sparse_da_mode = 2;
    h->sparse_da = h->sparse_db && (size_t)B * C * 2048 < ((size_t)1 << 31);
    if (h->sparse_da_env >= 0) h->sparse_da_mode = h->sparse_da_env;
    h->sparse_da = h->sparse_da && h->sparse_da_mode != 0;
    if (h->sparse_da) {
      std::vector<float> w(spfe::da_gather_f32_weight_bytes() / 4);
      spfe::da_gather_f32_pack_weights(blob.data() + blob_weight_offset(10), w.data());   // layer 10 = convDa
      if ((rc = dev_alloc(h, &h->d_wda32, w.size()))) return rc;
      HIP_TRY(hipMemcpy(h->d_wda32, w.data(), w.size() * 4, hipMemcpyHostToDevice));
      if ((rc = dev_alloc(h, &h->act7_alt, (size_t)B * C * 128))) return rc;
    }
  }
  if (h->bf16) {  // both heads in bf16: convPa | convDa write bf16, convPb and convDb are head_bf16.hip's GEMMs
    h->pbtail = h->pbtail_env != 0;   // (convPb inside the tail's launch: pbtail_bf16.hip)
    if ((rc = dev_alloc(h, &h->d_hd, (size_t)B * C * 512))) return rc;
    for (int which = 0; which < 2; ++which) {
      const int lid = which ? 9 : 11;
      const spfe_layer_t &Ld = SPFE_LAYERS[lid];
      const float *Wd = blob.data() + blob_weight_offset(lid);
      std::vector<unsigned char> w(spfe::head_bf16_weight_bytes(Ld.cout), 0);
      std::vector<unsigned short> wb((size_t)Ld.cout * Ld.cin);
      for (size_t k = 0; k < wb.size(); ++k) wb[k] = host_bf16_rne(Wd[k]);
      spfe::head_bf16_pack_weights(wb.data(), Ld.cout, w.data());
      unsigned char **dst = which ? &h->d_wpb : &h->d_wdb;
      if ((rc = dev_alloc(h, dst, w.size()))) return rc;
      HIP_TRY(hipMemcpy(*dst, w.data(), w.size(), hipMemcpyHostToDevice));
    }
  }

  // pinned host mirrors for the host-facing calls
  if ((rc = host_alloc(h, &h->h_img, (size_t)B * H * W))) return rc;
  if ((rc = host_alloc(h, &h->h_records, (size_t)B * h->rl.bytes))) return rc;
  if ((rc = host_alloc(h, &h->h_heat_inv, (size_t)B * H * W))) return rc;
  if (cfg->flags & SPFE_FLAG_HEAT)
    if ((rc = host_alloc(h, &h->h_heat, (size_t)B * H * W))) return rc;
  HIP_TRY(hipDeviceSynchronize());
  return SPFE_OK;
}

}  // namespace spfe_host
