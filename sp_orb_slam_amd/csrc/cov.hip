// cov.hip — computeCovariance on the GPU, exact.
//
// Reference: /root/reference/orb_slam2/src/cv/sp_extractor.cpp:252-340.  For each
// keypoint IN EMITTED ORDER a FIFO breadth-first walk runs down the heat_inv hill
// (neighbours left, up, right, down; taken iff not yet popped by ANY keypoint,
// value > 0, value < current), then cov = sum (s_i / sum s) * delta_i^2 clamped to
// >= 1.  The shared visited mask makes the loop sequential over keypoints.
//
// Parallel form used here (every walk itself stays a sequential FIFO, so pop
// order, duplicate pops and float accumulation order are the reference's):
//   A. cov_walk_kernel: every keypoint j walks ALONE (sees only its own visits)
//      and claims each popped pixel with atomicMin(claim[p], j).  -> region iso(j)
//   B. cov_classify_kernel: j is CLEAN when no pixel of iso(j) except its start
//      was claimed by a lower index.  Blocking only ever shrinks a walk, so
//      seq(i) is a subset of iso(i): nothing an earlier keypoint really visited can
//      touch iso(j), and the lone walk IS the sequential result.  Clean keypoints
//      are final and stamp done[p] = min(done[p], j); the others go on the
//      frame's dirty list.
//   C. cov_link_kernel + cov_replay_kernel: two keypoints interact only if their
//      lone regions share a pixel, and every pixel links all its claimants to its
//      lowest claimant, so the connected components of {(claim[p], j) : p in
//      iso(j)} are closed under interaction and own disjoint pixel sets.  Each
//      component is replayed by ONE wavefront in ascending keypoint order against
//      `done` (blocked iff a lower FINAL keypoint or an earlier member of the
//      component popped the pixel) — exactly the sequential loop restricted to
//      that component.  No rounds, no barriers; components run side by side.  For
//      a trained detector (regions of a few pixels) there are no dirty keypoints.
// The result equals the sequential algorithm exactly (same pixels, same
// multiplicities, same order), not approximately.
//
// Latency design: a walk is a chain of dependent lookups, and every global round
// trip in it costs ~1-2 us (the maps were last written by other XCDs, so they
// miss this XCD's L2).  So ONE WAVEFRONT serves one walk: its 64 lanes stage the
// 32x32-pixel window around the keypoint (heat_inv, and `done` for replays) into
// LDS with coalesced loads — one round trip — and the FIFO then runs entirely out
// of LDS (values, own-visited bitmap, FIFO), four lanes checking the four
// neighbours of each pop side by side.  Pixels outside the window fall back to
// global loads.  Claims, stamps, the list flush and the products of the moment
// sums are done by all lanes in parallel after the walk.
#include "spfe_kernels.h"
#include "desc_body.h"

namespace spfe {

#define COV_INF 0x7f7f7f7f
// map entries: this batch's generation code in the upper half, the keypoint index in the lower (CovScratch::gen)
__device__ __forceinline__ int cov_untag(int gen, int v) { return (v & (int)0xffff0000) == gen ? (v & 0xffff) : COV_INF; }
#define COV_LCAP 128     // FIFO entries kept in LDS per wavefront
#define COV_WIN 16       // the window covers dx,dy in [-16, 15] around the keypoint
#define COV_WAVES 2      // wavefronts (= walks) per workgroup: 12.5 KB of LDS for lone walks (21 KB for replays, which stage `done` too)
#define COV_OW 256       // popped pixels OUTSIDE the window a walk can remember (its visited set out there)

struct WaveMem {         // LDS of one wavefront
  float hv[32 * 32];     // heat_inv window
  int lq[COV_LCAP];
  float lqv[COV_LCAP];
  uint32_t bm[32];       // own-visited bitmap
  int ow[COV_OW];        // popped pixels outside the window (visited set there), now = count
  int dn[32 * 32];       // done window — REPLAY ONLY, and behind everything the lone walks use: their workgroups allocate the struct up to here
};                       // (6.1 KB a walk instead of 10.1: 24 walks fit a CU instead of 14, and three walk workgroups fit beside an
                         // f32 convolution workgroup's 120 KB instead of one)
constexpr size_t COV_WALK_LDS = offsetof(WaveMem, dn);
static_assert(COV_WALK_LDS % 16 == 0, "the float4 reads of hv / lqv need 16-byte aligned wave blocks");

struct Walk {
  WaveMem *m;
  const float *hinv;     // frame's heat_log: heat_inv(p) = hinv[p] * ha + hb, formed where it is read (hinv_at)
  const int *done;       // frame's done map (replay) or null
  int *gq;               // global pop list of this keypoint [qcap]
  float *gqv;
  int qcap, W, H, x0, y0, j;
  unsigned wmagic;       // floor((2^32 - 1) / W): row_of()
  int gen = 0;           // CovScratch::gen (replay: the done map's entries are tagged)
  float ha = 1.0f, hb = 0.0f;   // to_heat's scale / shift of heat_inv for this frame (FrameBufs::heat_consts[2..3])
};
// heat_inv of a pixel from the log heat map: one float multiply, then one float add (sp_extractor.cpp:461-474 as the oracle
// fixes it; -ffp-contract=off) — the bits mask_and_heat_norm_kernel writes when the map is an output (SPFE_FLAG_HEAT).  Without
// the flag the map is never materialised: 2 x 4 H W bytes per frame less for the normalisation to move beside conv1b.
__device__ __forceinline__ float hinv_of(float L, float ha, float hb) { const float t = L * ha; return t + hb; }

// id / W for any id < 2^32 without the 40-instruction division (it sits on the walk's dependent chain): the
// multiply-high with floor(2^32 / W) is the quotient or one less
__device__ __forceinline__ unsigned w_magic(int W) { return 0xffffffffu / (unsigned)W; }
__device__ __forceinline__ int row_of(int id, int W, unsigned magic) {
  int q = (int)__umulhi((unsigned)id, magic);
  if (id - q * W >= W) ++q;
  return q;
}

// The slow paths (FIFO entries beyond the LDS part, pixels outside the window) are
// real calls: written as `cond ? lds[i] : global[i]` the compiler speculates the
// global load on every iteration and the walk pays a memory round trip per lookup.
__device__ __attribute__((noinline)) int slow_ld_i(const int *p, int i) { return p[i]; }
__device__ __attribute__((noinline)) float slow_ld_f(const float *p, int i) { return p[i]; }
__device__ __forceinline__ int fifo_id(const Walk &w, int i) {
  if (__builtin_expect(i < COV_LCAP, 1)) return w.m->lq[i];
  return slow_ld_i(w.gq, i);
}
__device__ __forceinline__ float fifo_val(const Walk &w, int i) {
  if (__builtin_expect(i < COV_LCAP, 1)) return w.m->lqv[i];
  return slow_ld_f(w.gqv, i);
}

// all 64 lanes: stage the window (zero outside the image: never read there).  Two halves so that a caller can
// have the loads of the NEXT window in flight while it walks the current one.
struct WinRegs {
  float hv[16];
  int dn[16];
};
template <bool REPLAY>
__device__ __forceinline__ void load_window(const float *hinv, const int *done, int W, int H, int x0, int y0, int lane,
                                            WinRegs &r, int gen, float ha, float hb) {
  const int wx0 = x0 - COV_WIN, wy0 = y0 - COV_WIN;
#pragma unroll
  for (int k = 0; k < 16; ++k) {  // all 16 (32) loads in flight together: one round trip
    const int i = lane + 64 * k;
    const int dy = i >> 5, dx = i & 31;
    const int x = wx0 + dx, y = wy0 + dy;
    const bool in = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
    const size_t g = in ? (size_t)y * W + x : 0;
    r.hv[k] = hinv_of(hinv[g], ha, hb);
    if (REPLAY) r.dn[k] = cov_untag(gen, done[g]);
    if (!in) { r.hv[k] = 0.0f; r.dn[k] = COV_INF; }
  }
}
template <bool REPLAY>
__device__ __forceinline__ void store_window(WaveMem *m, int lane, const WinRegs &r) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    m->hv[lane + 64 * k] = r.hv[k];
    if (REPLAY) m->dn[lane + 64 * k] = r.dn[k];
  }
  if (lane < 32) m->bm[lane] = 0;
}
template <bool REPLAY>
__device__ __forceinline__ void stage_window(const Walk &w, int lane) {
  WinRegs r;
  load_window<REPLAY>(w.hinv, w.done, w.W, w.H, w.x0, w.y0, lane, r, w.gen, w.ha, w.hb);
  store_window<REPLAY>(w.m, lane, r);
}

__device__ __attribute__((noinline)) bool ow_seen(const WaveMem *m, int now, int id) {
  for (int u = 0; u < now; ++u)
    if (m->ow[u] == id) return true;
  return false;
}

// The FIFO walk, run by the whole wavefront in lock step, up to 16 pops at a time.  A group is the next
// G = min(16, tail - head) FIFO entries; lane = 4 * g + t examines neighbour t (left, up, right, down —
// :302-313) of the group's g-th entry, and the survivors are appended in lane order with a ballot + prefix
// count, i.e. in the reference's push order (entry by entry, neighbour by neighbour).  This is the
// sequential loop exactly: "visited" means POPPED (:285), so entry g must see the pops of entries
// 0..g-1 of its own group — its candidates are compared with their ids — and must NOT see later ones
// (the bitmap is updated after the group's lookups); entries pushed by the group are popped by later groups,
// as a FIFO would.  A walk of n pops takes ~log-ish many LDS round trips (the BFS frontier) instead of n.
// REPLAY additionally blocks pixels with done[p] < j.  Returns the number of pops, or -1 when the FIFO
// outgrew qcap, -2 when more than COV_OW distinct pixels outside the staged window were popped (a hill wider
// than the window: reported, not guessed).  (head, tail and the result are wave-uniform.)
template <bool REPLAY>
__device__ int walk(const Walk &w, int lane) {
  WaveMem *m = w.m;
  const int W = w.W, H = w.H, x0 = w.x0, y0 = w.y0;
  if (lane == 0) {
    m->lq[0] = y0 * W + x0;
    m->lqv[0] = m->hv[COV_WIN * 32 + COV_WIN];
  }
  int head = 0, tail = 1, now = 0;
  const int t = lane & 3, gi = lane >> 2;
  const int ox = t == 0 ? -1 : (t == 2 ? 1 : 0), oy = t == 1 ? -1 : (t == 3 ? 1 : 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  while (head < tail) {
    // (head and tail are wave-uniform by construction — ballots and popcounts — but the compiler carried them in vector
    // registers and turned every test on them into an exec-mask branch: pinned to scalar registers here, the loop's control
    // flow is s_cmp / s_cbranch and ~15 VALU instructions shorter per step)
    head = __builtin_amdgcn_readfirstlane(head);
    tail = __builtin_amdgcn_readfirstlane(tail);
    const int G = tail - head < 16 ? tail - head : 16;
    const bool act = gi < G;
    const int e = head + (act ? gi : 0);
    int id;
    float here;
    if (__builtin_expect(head + G <= COV_LCAP, 1)) {   // (uniform) the group's entries are in the LDS part of the list
      id = m->lq[e];
      here = m->lqv[e];
    } else {
      id = fifo_id(w, e);
      here = fifo_val(w, e);
    }
    const int y = row_of(id, W, w.wmagic), x = id - y * W;
    const int cdx = x - x0 + COV_WIN, cdy = y - y0 + COV_WIN;
    const bool pin = ((unsigned)cdx < 32u) & ((unsigned)cdy < 32u);   // the popped pixel is inside the window
    const int nx = x + ox, ny = y + oy;
    // bounds as in the reference: xx > 0, yy > 0, xx < w, yy < h   (selects, not branches: `&` on purpose)
    const int cc = (t & 1) ? ny : nx, lim = (t & 1) ? H : W;
    const bool inb = act & ((t < 2) ? (cc > 0) : (cc < lim));
    const int nid = id + oy * W + ox;
    const int dx = cdx + ox, dy = cdy + oy;
    const bool inwin = ((unsigned)dx < 32u) & ((unsigned)dy < 32u);
    // common case: the three lookups are independent LDS reads issued together (index 0 for the lanes they do not concern)
    const int wi = inwin ? dy * 32 + dx : 0;
    const float hv = m->hv[wi];
    const int dstamp = REPLAY ? m->dn[wi] : COV_INF;
    const uint32_t row = m->bm[inwin ? dy : 0];
    float v = hv;
    bool take = inb & inwin & (hv > 0.0f) & (hv < here) & !(REPLAY & (dstamp < w.j)) & !((row >> dx) & 1u);
    if (__builtin_expect(__ballot(inb & !inwin) != 0ull, 0)) {   // (uniform, rare) a neighbour outside the staged window:
      if (inb && !inwin) {                                       // global lookups, search of the outside list
        v = hinv_of(slow_ld_f(w.hinv, nid), w.ha, w.hb);
        take = v > 0.0f && v < here;
        if (take && REPLAY) take = !(cov_untag(w.gen, slow_ld_i(w.done, nid)) < w.j);
        if (take) take = !ow_seen(m, now, nid);
      }
    }
    // popped earlier in this very group: entry k < gi with the same pixel.  Unrolled over constant lanes in three blocks
    // (a lane past the group holds entry 0's id, and no active lane has gi > k there: harmless) — as a loop over k < G - 1
    // with a variable lane index this was more instructions than the rest of the step
#define COV_CHK(k) take &= !((gi > (k)) & (nid == __builtin_amdgcn_readlane(id, 4 * (k))))
    if (G > 1) { COV_CHK(0); COV_CHK(1); COV_CHK(2); }
    if (G > 4) { COV_CHK(3); COV_CHK(4); COV_CHK(5); COV_CHK(6); }
    if (G > 8) { COV_CHK(7); COV_CHK(8); COV_CHK(9); COV_CHK(10); COV_CHK(11); COV_CHK(12); COV_CHK(13); COV_CHK(14); }
#undef COV_CHK
    const unsigned long long mask = __ballot(take);
    const int pos = tail + __popcll(mask & below);
    const int ntail = tail + __popcll(mask);
    if (ntail > w.qcap) return -1;
    if (take & (pos < COV_LCAP)) { m->lq[pos] = nid; m->lqv[pos] = v; }
    if (__builtin_expect(ntail > COV_LCAP, 0)) {                  // (uniform) the list has outgrown its LDS part
      if (take && pos >= COV_LCAP) { w.gq[pos] = nid; w.gqv[pos] = v; }
    }
    // visited at POP (:285): a bitmap inside the window, a short list outside it
    if (act & (t == 0) & pin) atomicOr(&m->bm[cdy], 1u << cdx);
    if (__builtin_expect(__ballot(act & !pin) != 0ull, 0)) {   // rare: some popped pixel lies outside the window
      for (int k = 0; k < G; ++k) {
        const int idk = __builtin_amdgcn_readlane(id, 4 * k);
        const int pk = __builtin_amdgcn_readlane(pin ? 1 : 0, 4 * k);
        if (!pk && !ow_seen(m, now, idk)) {
          if (now >= COV_OW) return -2;
          if (lane == 0) m->ow[now] = idk;
          ++now;
        }
      }
    }
    head += G;
    tail = ntail;
  }
  return tail;
}

// second moments over the popped sequence, in pop order (:316-333): the running sums are accumulated
// one entry at a time exactly like the reference's loops.
__device__ void moments(const Walk &w, int n, int lane, float *cov2, float *cov2_inv) {
  // The running sums are sequential float adds in pop order (:316-333): every lane runs them redundantly on values it
  // reads from LDS with wave-uniform addresses (broadcast reads, four values a read, issued ahead of the dependent add
  // chain).  The first version passed the terms from lane to lane with v_readlane — ~70 cycles per term, a third of a
  // replayed member's time.
  WaveMem *m = w.m;
  float sum = 0.0f;
  {
    const int nl = n < COV_LCAP ? n : COV_LCAP;
    const float4 *v4 = reinterpret_cast<const float4 *>(m->lqv);
    int i = 0;
    for (; i + 16 <= nl; i += 16) {   // four reads in flight ahead of sixteen dependent adds
      const float4 a = v4[(i >> 2)], b = v4[(i >> 2) + 1], c = v4[(i >> 2) + 2], d = v4[(i >> 2) + 3];
      sum += a.x; sum += a.y; sum += a.z; sum += a.w; sum += b.x; sum += b.y; sum += b.z; sum += b.w;
      sum += c.x; sum += c.y; sum += c.z; sum += c.w; sum += d.x; sum += d.y; sum += d.z; sum += d.w;
    }
    for (; i + 4 <= nl; i += 4) {
      const float4 v = v4[i >> 2];
      sum += v.x; sum += v.y; sum += v.z; sum += v.w;
    }
    for (; i < nl; ++i) sum += m->lqv[i];
    for (; i < n; ++i) sum += slow_ld_f(w.gqv, i);   // beyond the LDS part of the list (rare)
  }
  // the weighted squares are independent per entry (all lanes), then staged in the window buffer — the walk is over,
  // nothing reads hv any more — for the two ordered sums
  float cx = 0.0f, cy = 0.0f;
  float *sx = m->hv, *sy = m->hv + 64;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    float tx = 0.0f, ty = 0.0f;
    if (i < n) {
      const int id = fifo_id(w, i);
      const int y = row_of(id, w.W, w.wmagic), x = id - y * w.W;
      const float wgt = fifo_val(w, i) / sum;
      const float dx = (float)x - (float)w.x0, dy = (float)y - (float)w.y0;
      tx = wgt * (dx * dx);
      ty = wgt * (dy * dy);
    }
    sx[lane] = tx;
    sy[lane] = ty;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int cnt = n - base < 64 ? n - base : 64;
    const float4 *x4 = reinterpret_cast<const float4 *>(sx), *y4 = reinterpret_cast<const float4 *>(sy);
    int u = 0;
    for (; u + 8 <= cnt; u += 8) {    // four reads in flight ahead of the two add chains
      const float4 a = x4[u >> 2], b = y4[u >> 2], c = x4[(u >> 2) + 1], d = y4[(u >> 2) + 1];
      cx += a.x; cy += b.x; cx += a.y; cy += b.y; cx += a.z; cy += b.z; cx += a.w; cy += b.w;
      cx += c.x; cy += d.x; cx += c.y; cy += d.y; cx += c.z; cy += d.z; cx += c.w; cy += d.w;
    }
    for (; u + 4 <= cnt; u += 4) {
      const float4 a = x4[u >> 2], b = y4[u >> 2];
      cx += a.x; cy += b.x; cx += a.y; cy += b.y; cx += a.z; cy += b.z; cx += a.w; cy += b.w;
    }
    for (; u < cnt; ++u) { cx += sx[u]; cy += sy[u]; }
    __builtin_amdgcn_wave_barrier();   // (the next chunk overwrites the staging area)
  }
  if (lane == 0) {
    cx = cx < 1.0f ? 1.0f : cx;
    cy = cy < 1.0f ? 1.0f : cy;
    cov2[0] = cx;
    cov2[1] = cy;
    cov2_inv[0] = 1.0f / cx;
    cov2_inv[1] = 1.0f / cy;
  }
}

struct CovFrame {
  const float *kp_xy;
  float *cov2, *cinv;
  int *hdr;
  const float *hinv;   // the frame's heat_log (see Walk::hinv)
  float ha, hb;
  int *claim, *done, *queues, *npop, *dirty, *ndirty, *nxt, *workers, *nworkers;
  float *qvals;
  int *ovf_slot, *novf, *ovf_q;   // overflow slots: pop lists of the walks that outgrew qcap
  float *ovf_v;
  float *nxy;                     // [kmax][2] position of nxt[j] (so that one load yields the next member AND its window)
  int *nedges;                    // claim edges (lower claimant, dirty keypoint) found by the classification: count ...
  int2 *edges;                    // ... and list [ecap] (null: the link kernel walks the pop lists itself)
  int K;
};

__device__ __forceinline__ CovFrame cov_frame(const FrameBufs &f, const RecordLayout &rl, const CovScratch &cs,
                                              int b, int H, int W) {
  CovFrame c;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  c.hdr = reinterpret_cast<int *>(rec + rl.off_hdr);
  c.K = c.hdr[0];
  c.kp_xy = reinterpret_cast<const float *>(rec + rl.off_xy);
  c.cov2 = reinterpret_cast<float *>(rec + rl.off_cov);
  c.cinv = reinterpret_cast<float *>(rec + rl.off_cinv);
  c.hinv = f.heat_log + (size_t)b * H * W;
  c.ha = f.heat_consts[(size_t)b * 4 + 2];
  c.hb = f.heat_consts[(size_t)b * 4 + 3];
  c.claim = cs.claim + (size_t)b * H * W;
  c.done = cs.done + (size_t)b * H * W;
  c.queues = cs.queue + (size_t)b * rl.kmax * cs.qcap;
  c.qvals = cs.qval + (size_t)b * rl.kmax * cs.qcap;
  c.npop = cs.npop + (size_t)b * rl.kmax;
  c.dirty = cs.dirty + (size_t)b * rl.kmax;
  c.nxt = cs.nxt + (size_t)b * rl.kmax;
  c.workers = cs.workers + (size_t)b * rl.kmax;
  c.ndirty = cs.counters + 4 * b;
  c.nworkers = cs.counters + 4 * b + 1;
  c.novf = cs.counters + 4 * b + 2;
  c.ovf_slot = cs.ovf_slot + (size_t)b * rl.kmax;
  c.ovf_q = cs.ovf_q + (size_t)b * cs.ovf_slots * cs.ovf_cap;
  c.ovf_v = cs.ovf_v + (size_t)b * cs.ovf_slots * cs.ovf_cap;
  c.nxy = cs.nxy + (size_t)b * rl.kmax * 2;
  c.nedges = cs.counters + 4 * b + 3;
  c.edges = cs.edges ? reinterpret_cast<int2 *>(cs.edges) + (size_t)b * cs.ecap : nullptr;
  return c;
}

// where keypoint j's pop list lives: its regular row, or the overflow slot its lone walk took
__device__ __forceinline__ void pop_list(const CovFrame &c, const CovScratch &cs, int j, int *&q, float *&qv, int &cap) {
  const int s = c.ovf_slot[j];
  if (__builtin_expect(s < 0, 1)) {
    q = c.queues + (size_t)j * cs.qcap; qv = c.qvals + (size_t)j * cs.qcap; cap = cs.qcap;
  } else {
    q = c.ovf_q + (size_t)s * cs.ovf_cap; qv = c.ovf_v + (size_t)s * cs.ovf_cap; cap = cs.ovf_cap;
  }
}

// ---- A: lone walks, one wavefront per keypoint ----
__global__ __launch_bounds__(64 * COV_WAVES) void cov_walk_kernel(FrameBufs f, RecordLayout rl, CovScratch cs,
                                                                  int H, int W) {
  __shared__ __attribute__((aligned(16))) char s_raw[COV_WAVES * COV_WALK_LDS];   // (WaveMem without its `dn` tail)
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = blockIdx.x * COV_WAVES + wv;
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  if (j >= c.K) return;
  Walk w{reinterpret_cast<WaveMem *>(s_raw + wv * COV_WALK_LDS), c.hinv, nullptr, c.queues + (size_t)j * cs.qcap, c.qvals + (size_t)j * cs.qcap, cs.qcap, W, H,
         (int)c.kp_xy[2 * j], (int)c.kp_xy[2 * j + 1], j, w_magic(W), 0, c.ha, c.hb};
  stage_window<false>(w, lane);
  int n = walk<false>(w, lane);
  if (n == -1) {
    // The region outgrew the regular FIFO (the reference's visited-at-pop rule multiplies pops on smooth
    // hills): take one of the frame's overflow slots and walk again from the start, still on the device.
    int slot = 0;
    if (lane == 0) slot = atomicAdd(c.novf, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot < cs.ovf_slots) {
      if (lane == 0) c.ovf_slot[j] = slot;
      w.gq = c.ovf_q + (size_t)slot * cs.ovf_cap;
      w.gqv = c.ovf_v + (size_t)slot * cs.ovf_cap;
      w.qcap = cs.ovf_cap;
      if (lane < 32) w.m->bm[lane] = 0;
      n = walk<false>(w, lane);
    }
  }
  if (lane == 0) {
    c.npop[j] = n;
    if (n < 0) atomicOr(&c.hdr[2], 1);  // beyond the overflow capacity too: report, do not guess
  }
  if (n < 0) return;
  moments(w, n, lane, c.cov2 + 2 * j, c.cinv + 2 * j);  // final if the keypoint turns out clean
  const int k = n < COV_LCAP ? n : COV_LCAP;
  for (int i = lane; i < k; i += 64) { w.gq[i] = w.m->lq[i]; w.gqv[i] = w.m->lqv[i]; }  // publish the pop list
  for (int i = lane; i < n; i += 64) atomicMin(&c.claim[fifo_id(w, i)], cs.gen | j);
}

// ---- B: clean keypoints are final (their moments are already in the record);
// the rest go on the frame's dirty list ----
__global__ __launch_bounds__(64 * COV_WAVES) void cov_classify_kernel(FrameBufs f, RecordLayout rl,
                                                                      CovScratch cs, int H, int W) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * COV_WAVES + (threadIdx.x >> 6);
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  if (j >= c.K || (c.hdr[2] & 1)) return;
  int *q; float *qv; int cap;
  pop_list(c, cs, j, q, qv, cap);
  const int n = c.npop[j];
  int bad = 0;
  for (int i0 = 1; i0 < n; i0 += 64) {   // claims were made by the previous kernel  (uniform trip count: ballots inside)
    const int i = i0 + lane;
    const int a = i < n ? cov_untag(cs.gen, c.claim[q[i]]) : COV_INF;
    bad |= a < j;
    // the link kernel's input, while the claim is in a register: (lower claimant, this keypoint).  One entry per pixel
    // (duplicates are harmless for a union); a frame with more edges than the list holds makes the link kernel walk the pop
    // lists itself, as it used to (the count keeps counting)
    if (c.edges) {   // (uniform) one atomic per wavefront pass: the edge lanes take consecutive entries
      const unsigned long long em = __ballot(a < j);
      if (em) {
        int base = 0;
        if (lane == __ffsll((long long)em) - 1) base = atomicAdd(c.nedges, __popcll(em));
        base = __shfl(base, __ffsll((long long)em) - 1, 64);
        const int e = base + __popcll(em & ((1ull << lane) - 1ull));
        if (a < j && e < cs.ecap) c.edges[e] = make_int2(a, j);
      }
    }
  }
  if (__ballot(bad) == 0) {
    for (int i = lane; i < n; i += 64) atomicMin(&c.done[q[i]], cs.gen | j);
  } else if (lane == 0) {
    c.dirty[atomicAdd(c.ndirty, 1)] = j;
  }
}

// ---- C1: link.  One workgroup per frame: union-find over the dirty keypoints'
// claim edges, then per component the ascending chain of its dirty members. ----
#define LINK_THREADS 1024   // 16 wavefronts: the union phase is a chain of dependent global loads per dirty keypoint

__device__ __forceinline__ int uf_find(volatile int *parent, int x) {
  while (true) {
    const int p = parent[x];
    if (p == x) return x;
    x = p;
  }
}

__global__ __launch_bounds__(LINK_THREADS) void cov_link_kernel(FrameBufs f, RecordLayout rl, CovScratch cs,
                                                                int H, int W) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  const int nd = *c.ndirty;
  const int K = c.K;
  if (nd == 0 || (c.hdr[2] & 1)) return;
  extern __shared__ __attribute__((aligned(16))) int smem_i[];
  int *parent = smem_i;      // [K] union-find forest over keypoint indices
  int *leader = smem_i + K;  // [K] lowest DIRTY member of the component rooted here
  for (int j = tid; j < K; j += LINK_THREADS) { parent[j] = j; leader[j] = COV_INF; }
  __syncthreads();
  // every pixel links its claimants to its lowest claimant: union(claim[p], j).
  // (one wavefront per dirty keypoint, lanes over its pixels)
  // (a quarter-wavefront per dirty keypoint, its 16 lanes over the pixels: the phase is two dependent global round trips
  // per keypoint — pop list, then the claims of its pixels — and 64 groups keep four times as many of them in flight
  // as 16 wavefronts did)
  auto unite = [&](int a, int bb) {   // hook the larger root under the smaller one
    while (true) {
      a = uf_find(parent, a);
      bb = uf_find(parent, bb);
      if (a == bb) break;
      const int hi = a > bb ? a : bb, lo = a > bb ? bb : a;
      const int old = atomicMin(&parent[hi], lo);
      if (old == hi) break;
      a = old;
      bb = lo;
    }
  };
  const int ne = c.edges ? *c.nedges : -1;
  const bool from_edges = ne >= 0 && ne <= cs.ecap;   // the classification listed every claim edge: ONE global round trip
  if (from_edges)
    for (int e = tid; e < ne; e += LINK_THREADS) {
      const int2 ed = c.edges[e];
      unite(ed.x, ed.y);
    }
  constexpr int GL = 16;
  const int gl = tid & (GL - 1), grp = tid / GL;
  for (int d = grp; d < (from_edges ? 0 : nd); d += LINK_THREADS / GL) {
    const int j = c.dirty[d];
    int *q; float *qv; int cap;
    pop_list(c, cs, j, q, qv, cap);
    const int n = c.npop[j];
    for (int i = 1 + gl; i < n; i += GL) {
      int a = cov_untag(cs.gen, c.claim[q[i]]);
      int bb = j;
      if (a >= j) continue;
      while (true) {  // hook the larger root under the smaller one
        a = uf_find(parent, a);
        bb = uf_find(parent, bb);
        if (a == bb) break;
        const int hi = a > bb ? a : bb, lo = a > bb ? bb : a;
        const int old = atomicMin(&parent[hi], lo);
        if (old == hi) break;
        a = old;
        bb = lo;
      }
    }
  }
  __syncthreads();
  for (int j = tid; j < K; j += LINK_THREADS) parent[j] = uf_find(parent, j);  // flatten (roots are fixed now)
  __syncthreads();
  // chain: the dirty members of a component in ascending index.  Sort the dirty list by (root, index)
  // — a bitonic sort in LDS — and the successor / the component's first member (the replay kernel's
  // worker) are the neighbours in the sorted order.
  int *key = smem_i + 2 * K;   // [P], P = next power of two >= nd (<= 2 K)
  if (nd <= LINK_THREADS) {
    // few dirty keypoints (a frame has a few hundred at most on the dense synthetic detector, none on a trained one): every
    // thread owns one and scans the others' (root, index) keys in LDS — broadcast reads, no barrier — for its successor in
    // the component and for "am I the first": the 36 barrier-separated passes of the bitonic sort below were most of this
    // kernel's 25 us on a single frame
    int myj = -1, myroot = -1;
    if (tid < nd) {
      myj = c.dirty[tid];
      myroot = parent[myj];
      key[tid] = (myroot << 15) | myj;
    }
    __syncthreads();
    // The workers go on the list LONGEST CHAIN FIRST (ties: lower keypoint first): a replay workgroup takes WV consecutive
    // workers and lives as long as its longest chain, so with chains of similar length side by side most workgroups are gone
    // after a member or two and only the first few — launched first — stay for the long chains.  (In arrival order of an
    // atomic counter nearly every fat workgroup of a 1280x720 frame held one long chain, and in pipelined bf16 calls each of
    // them keeps a register-resident-weights convolution workgroup off its CU for that long.)  The order changes no result:
    // components are independent.
    int *wkey = key + nd;                 // [<= nd] (chain length << 15 | 0x7fff - first member) of the workers
    __shared__ int s_nw;
    if (tid == 0) s_nw = 0;
    int jn = 0x7fff, first = 1, len = 0;
    if (tid < nd) {
      for (int e = 0; e < nd; ++e) {
        const int ke = key[e], je = ke & 0x7fff;
        const bool same = (ke >> 15) == myroot;
        jn = same && je > myj && je < jn ? je : jn;
        first &= !(same && je < myj);
        len += same;
      }
      jn = jn == 0x7fff ? -1 : jn;
      c.nxt[myj] = jn;
      if (jn >= 0) { c.nxy[2 * myj] = c.kp_xy[2 * jn]; c.nxy[2 * myj + 1] = c.kp_xy[2 * jn + 1]; }
    }
    __syncthreads();
    const int mykey = (len << 15) | (0x7fff - myj);
    if (tid < nd && first) wkey[atomicAdd(&s_nw, 1)] = mykey;
    __syncthreads();
    const int nw = s_nw;
    if (tid < nd && first) {
      int rank = 0;
      for (int e = 0; e < nw; ++e) rank += wkey[e] > mykey;
      c.workers[rank] = myj;
    }
    if (tid == 0) *c.nworkers = nw;
    return;
  }
  int P = 1;
  while (P < nd) P <<= 1;
  for (int d = tid; d < P; d += LINK_THREADS) {
    if (d < nd) {
      const int j = c.dirty[d];
      key[d] = (parent[j] << 15) | j;   // K <= 10001 < 2^15: spfe_create bounds num_features to 10000
    } else {
      key[d] = COV_INF;
    }
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      for (int i = tid; i < P; i += LINK_THREADS) {
        const int ixj = i ^ jj;
        if (ixj > i) {
          const int a = key[i], bq = key[ixj];
          const bool up = (i & k) == 0;
          if ((a > bq) == up) { key[i] = bq; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int d = tid; d < nd; d += LINK_THREADS) {
    const int kd = key[d], j = kd & 0x7fff, root = kd >> 15;
    const int kn = d + 1 < nd ? key[d + 1] : COV_INF;
    const int jn = (kn != COV_INF && (kn >> 15) == root) ? (kn & 0x7fff) : -1;
    c.nxt[j] = jn;
    if (jn >= 0) { c.nxy[2 * j] = c.kp_xy[2 * jn]; c.nxy[2 * j + 1] = c.kp_xy[2 * jn + 1]; }
    if (d == 0 || (key[d - 1] >> 15) != root) c.workers[atomicAdd(c.nworkers, 1)] = j;
  }
}

// ---- C2: replay.  One wavefront per component, members in ascending order. ----
// DEFERRED MOMENTS (round 6).  What one member hands to the next is its POP SET (stamps + the patch of the next window); its
// second moments depend on nothing behind it and nothing behind it depends on them.  `-DSPFE_REPLAY_PROBE` puts a member at
// 16.4 k cycles, 3.2 k of them moments() — a serial sum, a round of divisions, two more serial sums, two reciprocals, one
// wavefront's instruction latency each.  So a chain's members park their pop lists (pixel, value) in LDS and the moments of
// ALL parked members are formed at the end, one LANE per member, every lane running the reference's loops (:316-333) over its
// own list: the same IEEE operations in the same order per member — same bits — in the time of the longest list instead of
// the sum of all.  Single-member components (most of a frame's) and members whose list spilled past the LDS FIFO take
// moments() as before; a full park area is drained where it fills.
#define COV_DM 512        // parked pops per wavefront (a member of the dense synthetic detector pops ~35)
#define COV_DM_MEMBERS 64 // parked members per drain (one lane each)
struct ReplayMem {
  WaveMem m;
  int dm_xy[COV_DM];              // (y << 16) | x of a parked pop
  float dm_v[COV_DM];             // its heat_inv value
  int4 dm_meta[COV_DM_MEMBERS];   // {first parked pop, pops, keypoint, (y0 << 16) | x0}
};

// the moments of the parked members, lane m <-> member m: the loops of moments(), per lane
__device__ void drain_moments(ReplayMem *rm, int nmem, int lane, float *cov2_base, float *cinv_base) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the parks were written by other lanes of this wavefront)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < nmem) {
    const int4 me = rm->dm_meta[lane];
    const int off = me.x, n = me.y, j = me.z;
    const float x0 = (float)(me.w & 0xffff), y0 = (float)(me.w >> 16);
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum += rm->dm_v[off + i];
    float cx = 0.0f, cy = 0.0f;
    for (int i = 0; i < n; ++i) {
      const int xy = rm->dm_xy[off + i];
      const float wgt = rm->dm_v[off + i] / sum;
      const float dx = (float)(xy & 0xffff) - x0, dy = (float)(xy >> 16) - y0;
      const float tx = wgt * (dx * dx), ty = wgt * (dy * dy);
      cx += tx;
      cy += ty;
    }
    cx = cx < 1.0f ? 1.0f : cx;
    cy = cy < 1.0f ? 1.0f : cy;
    cov2_base[2 * j] = cx;
    cov2_base[2 * j + 1] = cy;
    cinv_base[2 * j] = 1.0f / cx;
    cinv_base[2 * j + 1] = 1.0f / cy;
  }
}

// desc_first >= 0: blocks from that index on are not replay workers but the descriptor sampling of the frame's keypoints, one
// wavefront each (desc_body.h) — synchronous calls: the sampling is needed by the finished record only, so it runs beside the
// longest kernel of the chain instead of in front of the chain.
// WV wavefronts (= components) per workgroup.  2 (21 KB of LDS) fits beside an f32 convolution workgroup; 8 (83 KB) packs a
// frame's ~120 live workers into ~15 workgroups, so that in bf16 pipelined calls — where a register-resident-weights
// convolution workgroup needs every register of its CU and cannot start on a CU that hosts ONE side-chain wavefront — the
// replay holds ~120 CUs' worth of nothing instead of a wavefront on nearly every CU (launch_cov).
// DEFER: the moments of a chain's members at the end of the chain (ReplayMem, above) — synchronous calls, where the replay is on
// a single call's critical path (a 9-member chain 147.7 k -> 128.2 k cycles, 21 members 437.7 k -> 349.9 k; a single frame's call
// -5 us).  Pipelined calls keep the inline form: the park area is 5 KB more LDS per wavefront beside the next batch's
// convolutions, and measured -0.6 % on the f32 headline for nothing (the replay is off their critical path).
template <int WV, bool DEFER>
__global__ __launch_bounds__(64 * WV) void cov_replay_kernel(FrameBufs f, RecordLayout rl, CovScratch cs,
                                                              int H, int W, int desc_first) {
  extern __shared__ __attribute__((aligned(16))) char s_replay_raw[];
  constexpr size_t STRIDE = DEFER ? sizeof(ReplayMem) : sizeof(WaveMem);   // (ReplayMem begins with its WaveMem)
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (desc_first >= 0 && (int)blockIdx.x >= desc_first) {
    desc_keypoint(f, rl, H, W, b, ((int)blockIdx.x - desc_first) * WV + wv, lane);
    return;
  }
  const int widx = blockIdx.x * WV + wv;
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  if ((c.hdr[2] & 1) || widx >= *c.nworkers) return;
  int j = c.workers[widx];
  ReplayMem *const rm = reinterpret_cast<ReplayMem *>(s_replay_raw + (size_t)wv * STRIDE);   // (its park arrays exist when DEFER)
  WaveMem *m = &rm->m;
  int dm_n = 0, dm_members = 0;   // parked pops / members (wave-uniform)
  const int *prev_q = nullptr;
  float fx = c.kp_xy[2 * j], fy = c.kp_xy[2 * j + 1];
  int jn = c.nxt[j];
  float nfx = c.nxy[2 * j], nfy = c.nxy[2 * j + 1];   // (garbage when jn < 0: never used)
  WinRegs win;
  load_window<true>(c.hinv, c.done, W, H, (int)fx, (int)fy, lane, win, cs.gen, c.ha, c.hb);
  int n_prev = 0, j_prev = -1;
  const unsigned wmagic = w_magic(W);
#ifdef SPFE_REPLAY_PROBE   // phase cycles of the long chains (printf from chains of >= 8 members; tools/microbench/README.md)
  unsigned long long tp0 = 0, tp_store = 0, tp_walk = 0, tp_mom = 0, tp_stamp = 0, tp_all = __builtin_readcyclecounter();
  int members = 0, pops = 0;
#define RP(acc) do { const unsigned long long t_ = __builtin_readcyclecounter(); acc += t_ - tp0; tp0 = t_; } while (0)
#else
#define RP(acc) do { } while (0)
#endif
  while (j >= 0) {
#ifdef SPFE_REPLAY_PROBE
    tp0 = __builtin_readcyclecounter();
#endif
    const int x0 = (int)fx, y0 = (int)fy;
    // this member's window goes to LDS; the pixels the previous member just stamped were loaded before its
    // stamps existed: patch them from its pop list, which still sits in the LDS FIFO (no other wavefront
    // writes this component's pixels)
    store_window<true>(m, lane, win);
    for (int i = lane; i < n_prev; i += 64) {
      const int id = i < COV_LCAP ? m->lq[i] : slow_ld_i(prev_q, i);
      const int py = row_of(id, W, wmagic), px = id - py * W;
      const int dx = px - x0 + COV_WIN, dy = py - y0 + COV_WIN;
      if ((unsigned)dx < 32u && (unsigned)dy < 32u) m->dn[dy * 32 + dx] = j_prev;
    }
    int *q; float *qv; int cap;
    pop_list(c, cs, j, q, qv, cap);
    Walk w{m, c.hinv, c.done, q, qv, cap, W, H, x0, y0, j, wmagic, cs.gen, c.ha, c.hb};
    // the next member's window and the member after it (index + position): all addresses are known, so the
    // requests go out now and their round trips pass under this member's walk
    int jnn = -1;
    float nnfx = 0.0f, nnfy = 0.0f;
    if (jn >= 0) {
      load_window<true>(c.hinv, c.done, W, H, (int)nfx, (int)nfy, lane, win, cs.gen, c.ha, c.hb);
      jnn = c.nxt[jn];
      nnfx = c.nxy[2 * jn];
      nnfy = c.nxy[2 * jn + 1];
    }
    RP(tp_store);
    // a replay's pop list is a subsequence of the lone walk's: it cannot overflow
    const int n = walk<true>(w, lane);
    RP(tp_walk);
    if (n < 0) { if (lane == 0) atomicOr(&c.hdr[2], 1); return; }
    if (!DEFER || (j_prev < 0 && jn < 0) || n > COV_LCAP || n > COV_DM) {
      moments(w, n, lane, c.cov2 + 2 * j, c.cinv + 2 * j);   // a lone member, or a list that left the LDS FIFO: as before
    } else {
      if (dm_n + n > COV_DM || dm_members == COV_DM_MEMBERS) {   // (uniform) the park area is full: drain it
        drain_moments(rm, dm_members, lane, c.cov2, c.cinv);
        dm_n = 0; dm_members = 0;
      }
      for (int i = lane; i < n; i += 64) {
        const int id = m->lq[i];
        const int py = row_of(id, W, wmagic), px = id - py * W;
        rm->dm_xy[dm_n + i] = (py << 16) | px;
        rm->dm_v[dm_n + i] = m->lqv[i];
      }
      if (lane == 0) rm->dm_meta[dm_members] = make_int4(dm_n, n, j, (y0 << 16) | x0);
      dm_n += n;
      ++dm_members;
    }
    RP(tp_mom);
    // stamp before the next member starts: this wavefront is the only writer and
    // the only reader of these pixels during the kernel
    for (int i = lane; i < n; i += 64) c.done[fifo_id(w, i)] = cs.gen | j;  // popped => its stamp was >= j
    __threadfence_block();  // same wavefront, same CU: L1 is coherent for it   (round 6: patching from the last TWO members' lists
                            // instead of this wait measured 132.0 k against 128.2 k cycles on the 9-member chain: the wait is not what a member costs)
    RP(tp_stamp);
#ifdef SPFE_REPLAY_PROBE
    ++members; pops += n;
#endif
    n_prev = n; j_prev = j; prev_q = q;
    j = jn; fx = nfx; fy = nfy;
    jn = jnn; nfx = nnfx; nfy = nnfy;
  }
  if (DEFER && dm_members) drain_moments(rm, dm_members, lane, c.cov2, c.cinv);   // (LDS operations of a wavefront are in order: the parks are visible)
#ifdef SPFE_REPLAY_PROBE
  if (lane == 0 && members >= 8)
    printf("REPLAY chain %d members %d pops | store+patch %llu walk %llu moments %llu stamp %llu | total %llu\n", members, pops,
           tp_store, tp_walk, tp_mom, tp_stamp, __builtin_readcyclecounter() - tp_all);
#endif
}

// ---- D: the last resort, on the device.  A frame whose record carries SPFE_STATUS_COV_OVERFLOW — more walks outgrew their
// lists than there are overflow slots, a region has more pops than a slot holds, a hill left the staged window by more than
// COV_OW pixels — is redone here from scratch, literally as the reference's loop (:252-340): keypoints in emitted order, ONE
// visited mask for the whole frame (the claim map, reused), every lookup from global memory, the pop list in one large
// list shared by the batch (fb_cap entries; one workgroup walks the flagged frames one after the other).  Same group-parallel
// FIFO step as walk(), so pop order, duplicates and the float sums are the sequential loop's.  Slow (a chain of global round
// trips per FIFO group) and never taken by benchmark, golden or sequence inputs; what it buys is that records are complete
// on the device — the all-gathered ones too — and that the library has no host compute routine.  Only a region with more
// than fb_cap pops (4 M by default: the reference itself would spend ~0.1 s in that one BFS) leaves the status bit set.
__device__ int walk_fallback(const Walk &w, int *visited, int lane) {
  WaveMem *m = w.m;
  const int W = w.W, H = w.H;
  if (lane == 0) {
    const int id0 = w.y0 * W + w.x0;
    const float v0 = hinv_of(w.hinv[id0], w.ha, w.hb);
    m->lq[0] = id0; m->lqv[0] = v0;
    w.gq[0] = id0; w.gqv[0] = v0;
  }
  __threadfence_block();
  int head = 0, tail = 1;
  const int t = lane & 3, gi = lane >> 2;
  const int ox = t == 0 ? -1 : (t == 2 ? 1 : 0), oy = t == 1 ? -1 : (t == 3 ? 1 : 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  while (head < tail) {
    const int G = tail - head < 16 ? tail - head : 16;
    const bool act = gi < G;
    const int e = head + (act ? gi : 0);
    const int id = w.gq[e];
    const float here = w.gqv[e];
    const int y = row_of(id, W, w.wmagic), x = id - y * W;
    const int nx = x + ox, ny = y + oy;
    const int cc = (t & 1) ? ny : nx, lim = (t & 1) ? H : W;
    const bool inb = act & ((t < 2) ? (cc > 0) : (cc < lim));          // xx > 0, yy > 0, xx < w, yy < h
    const int nid = id + oy * W + ox;
    float v = 0.0f;
    bool take = false;
    if (inb) {
      v = hinv_of(w.hinv[nid], w.ha, w.hb);
      take = v > 0.0f && v < here && visited[nid] == 0;
    }
#define COV_CHK(k) take &= !((gi > (k)) & (nid == __builtin_amdgcn_readlane(id, 4 * (k))))
    if (G > 1) { COV_CHK(0); COV_CHK(1); COV_CHK(2); }
    if (G > 4) { COV_CHK(3); COV_CHK(4); COV_CHK(5); COV_CHK(6); }
    if (G > 8) { COV_CHK(7); COV_CHK(8); COV_CHK(9); COV_CHK(10); COV_CHK(11); COV_CHK(12); COV_CHK(13); COV_CHK(14); }
#undef COV_CHK
    const unsigned long long mask = __ballot(take);
    const int pos = tail + __popcll(mask & below);
    const int ntail = tail + __popcll(mask);
    if (ntail > w.qcap) return -1;
    if (take) {
      w.gq[pos] = nid; w.gqv[pos] = v;
      if (pos < COV_LCAP) { m->lq[pos] = nid; m->lqv[pos] = v; }
    }
    if (act & (t == 0)) visited[id] = 1;      // visited at POP (:285)
    __threadfence_block();                    // the next group reads the list and the mask this one wrote (same wavefront)
    head += G;
    tail = ntail;
  }
  return tail;
}

__global__ __launch_bounds__(256) void cov_fallback_kernel(FrameBufs f, RecordLayout rl, CovScratch cs, int B, int H, int W) {
  __shared__ WaveMem s_mem;
  const int tid = threadIdx.x, lane = tid & 63;
  {  // the common case — no record carries the bit — is one parallel look at the headers and out
    int any = 0;
    for (int b = tid; b < B; b += 256) any |= reinterpret_cast<const int *>(f.records + (size_t)b * rl.bytes + rl.off_hdr)[2] & 1;
    if (!__syncthreads_or(any)) return;
  }
  for (int b = 0; b < B; ++b) {
    const CovFrame c = cov_frame(f, rl, cs, b, H, W);
    if (!(c.hdr[2] & 1)) continue;            // (uniform: written by earlier kernels of this stream)
    for (int i = tid; i < H * W; i += 256) c.claim[i] = 0;
    __threadfence();
    __syncthreads();
    if (tid < 64) {
      bool ok = true;
      for (int j = 0; j < c.K && ok; ++j) {
        Walk w{&s_mem, c.hinv, nullptr, cs.fb_q, cs.fb_v, cs.fb_cap, W, H, (int)c.kp_xy[2 * j], (int)c.kp_xy[2 * j + 1], j,
               w_magic(W), 0, c.ha, c.hb};
        const int n = walk_fallback(w, c.claim, lane);
        if (n < 0) { ok = false; break; }
        moments(w, n, lane, c.cov2 + 2 * j, c.cinv + 2 * j);
        __builtin_amdgcn_wave_barrier();
      }
      if (ok && lane == 0) atomicAnd(&c.hdr[2], ~1);
    }
    __syncthreads();
    // the claim map served as the visited mask (0 / 1): back to "nobody" — entries below every generation's would win all of
    // the next batches' atomicMin
    for (int i = tid; i < H * W; i += 256) c.claim[i] = COV_RESET;
    __syncthreads();
  }
}

size_t cov_link_lds(int kmax) { return (size_t)kmax * 4 * sizeof(int); }   // parent, leader, 2 K sort keys

template <int WV, bool DEFER>
static hipError_t launch_replay(const FrameBufs &f, const RecordLayout &r, const CovScratch &cs, int B, int H, int W, hipStream_t s,
                                bool with_desc) {
  const size_t lds = (DEFER ? sizeof(ReplayMem) : sizeof(WaveMem)) * WV;
  auto k = cov_replay_kernel<WV, DEFER>;
  if (lds > 64 * 1024) {
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  const int nb = (r.kmax + WV - 1) / WV;
  hipLaunchKernelGGL(k, dim3(with_desc ? 2 * nb : nb, B), dim3(64 * WV), lds, s, f, r, cs, H, W, with_desc ? nb : -1);
  return hipGetLastError();
}

hipError_t launch_cov(const FrameBufs &f, const RecordLayout &r, const CovScratch &cs, int B, int H, int W,
                      hipStream_t s, bool with_desc, hipEvent_t before_replay, int replay_waves, bool defer_moments) {
  // claim / done / counters / ovf_slot were reset by heat_norm_kernel (the kernel in front of this stage)
  hipError_t e = hipSuccess;
  const dim3 grid((r.kmax + COV_WAVES - 1) / COV_WAVES, B), block(64 * COV_WAVES);
  hipLaunchKernelGGL(cov_walk_kernel, grid, block, 0, s, f, r, cs, H, W);
  hipLaunchKernelGGL(cov_classify_kernel, grid, block, 0, s, f, r, cs, H, W);
  const size_t lds = cov_link_lds(r.kmax);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(cov_link_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(cov_link_kernel, dim3(B), dim3(LINK_THREADS), lds, s, f, r, cs, H, W);
  if (before_replay) {   // (the descriptor head, when it was launched behind the detector tail: the sampling reads its output)
    e = hipStreamWaitEvent(s, before_replay, 0);
    if (e != hipSuccess) return e;
  }
  if (defer_moments) e = replay_waves >= 8 ? launch_replay<8, true>(f, r, cs, B, H, W, s, with_desc) : launch_replay<COV_WAVES, true>(f, r, cs, B, H, W, s, with_desc);
  else e = replay_waves >= 8 ? launch_replay<8, false>(f, r, cs, B, H, W, s, with_desc) : launch_replay<COV_WAVES, false>(f, r, cs, B, H, W, s, with_desc);
  if (e != hipSuccess) return e;
  // (one workgroup that returns at once unless a record carries the overflow bit: ~2 us at the end of the chain)
  if (cs.fb_q) hipLaunchKernelGGL(cov_fallback_kernel, dim3(1), dim3(256), 0, s, f, r, cs, B, H, W);
  return hipGetLastError();
}

}  // namespace spfe
