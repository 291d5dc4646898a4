// api.hip -- C-ABI entry points of include/trase_rast.h: argument validation, workspace carving,
// kernel sequencing on the caller's stream.  No torch types, no persistent allocations.
#include <stdarg.h>
#include <string.h>

#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace trase {

// ---- error reporting ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return TRASE_ERR_HIP;
}

// ---- opt-in profiler (the only global state; off by default) -----------------------------------------
struct ProfRec { std::string name; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static int g_prof_mode = 0;   // 1: every kernel, 2: only the compositing kernels (render_*)
static std::vector<ProfRec*> g_prof_recs;

bool prof_on() { return g_prof_on; }

ProfScope::ProfScope(const char* n, hipStream_t s) : name(n), stream(s), rec(nullptr) {
  if (!g_prof_on) return;
  if (g_prof_mode == 2 && strncmp(n, "render_", 7) != 0) return;
  ProfRec* r = new ProfRec();
  r->name = n;
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) { delete r; return; }
  hipEventRecord(r->e0, s);
  rec = r;
}

ProfScope::~ProfScope() {
  if (!rec) return;
  ProfRec* r = (ProfRec*)rec;
  hipEventRecord(r->e1, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(r);
}

// ---- workspace layout ---------------------------------------------------------------------------------
#ifndef TRASE_RS_ITEMS
#define TRASE_RS_ITEMS 8
#endif
static inline int rs_blocks(size_t n) { return (int)((n + 256 * TRASE_RS_ITEMS - 1) / (256 * TRASE_RS_ITEMS)); }   // binning.hip RS_TILE

size_t geom_bytes(int P) {
  const size_t p = (size_t)P;
  return align_up(sizeof(uint32_t) * HDR_WORDS) + align_up(sizeof(float2) * p) + align_up(sizeof(float4) * p) * 2 +
         align_up(sizeof(uint32_t) * p) * 2 + align_up(sizeof(float4) * 4 * p) + align_up(sizeof(uint32_t) * 32 * p);
}
GeomBuf carve_geom(void* ptr, int P) {
  const size_t p = (size_t)P;
  char* c = (char*)ptr;
  GeomBuf g;
  g.hdr = (uint32_t*)c; c += align_up(sizeof(uint32_t) * HDR_WORDS);
  g.xy = (float2*)c; c += align_up(sizeof(float2) * p);
  g.conic_o = (float4*)c; c += align_up(sizeof(float4) * p);
  g.rgbd = (float4*)c; c += align_up(sizeof(float4) * p);
  g.tiles = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p);
  g.clamped = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p);
  g.geo = (float4*)c; c += align_up(sizeof(float4) * 4 * p);
  g.ftab = (uint32_t*)c;
  return g;
}
// ranges holds T sub-tiles + 1 sentinel ("trash") entry
size_t bin_bytes(int64_t cap, int T) { return align_up(sizeof(uint32_t) * (size_t)cap) * 2 + align_up(sizeof(uint2) * ((size_t)T + 1)); }
BinBuf carve_bin(void* ptr, int64_t cap, int T) {
  char* c = (char*)ptr;
  BinBuf b;
  b.point_list = (uint32_t*)c; c += align_up(sizeof(uint32_t) * (size_t)cap);
  b.pair_slot = (uint32_t*)c; c += align_up(sizeof(uint32_t) * (size_t)cap);
  b.ranges = (uint2*)c;
  (void)T;
  return b;
}
size_t img_bytes(int W, int H) { return align_up(sizeof(float) * (size_t)W * H) * 2; }
ImgBuf carve_img(void* ptr, int W, int H) {
  char* c = (char*)ptr;
  ImgBuf i;
  i.final_T = (float*)c; c += align_up(sizeof(float) * (size_t)W * H);
  i.n_contrib = (uint32_t*)c;
  return i;
}
static size_t sort_bytes_common(size_t n, int digit_bits = 8) {   // hist + digit_total
  const size_t nd = (size_t)1 << digit_bits;
  return align_up(sizeof(uint32_t) * nd * (size_t)rs_blocks(n) * rs_hist_copies(rs_blocks(n))) + align_up(sizeof(uint32_t) * nd * 8);
}
// bits of a packed list value left for the pair index (HDR_PACK); 0 = the variant's kernels need emit-order slots
static int list_pack_bits(const TraseRastSettings* s, int P) {
  if (s->variant & TRASE_VARIANT_SLOT_LISTS) return 0;
  int lg = 0;
  while ((1ll << lg) < (long long)P) ++lg;
  if (lg >= 28) return 0;
  return lg < 1 ? 31 : 32 - lg;       // never 32: the kernels evaluate (1u << jb) and v >> jb
}
size_t pre_bytes(int P) {
  const size_t p = (size_t)(P > 0 ? P : 1);
  return align_up(sizeof(uint32_t) * p) * 7 + align_up(sizeof(uint32_t) * (p / 1024 + 2) * 3) + sort_bytes_common(p, DEPTH_MAX_DIGIT_BITS);
}
PreBuf carve_pre(void* ptr, int P) {
  const size_t p = (size_t)(P > 0 ? P : 1);
  char* c = (char*)ptr;
  PreBuf t;
  for (int i = 0; i < 2; ++i) { t.sort.keys[i] = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p); }
  for (int i = 0; i < 2; ++i) { t.sort.vals[i] = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p); }
  t.offsets = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p);
  t.id_end = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p);
  t.live_ids = (uint32_t*)c; c += align_up(sizeof(uint32_t) * p);
  t.block_sums = (uint32_t*)c; c += align_up(sizeof(uint32_t) * (p / 1024 + 2) * 3);
  t.sort.hist = (uint32_t*)c; c += align_up(sizeof(uint32_t) * ((size_t)1 << DEPTH_MAX_DIGIT_BITS) * (size_t)rs_blocks(p) * rs_hist_copies(rs_blocks(p)));
  t.sort.digit_total = (uint32_t*)c;
  t.sort.nb_max = rs_blocks(p);
  t.sort.hist_copies = rs_hist_copies(rs_blocks(p));
  return t;
}
size_t tmp_bytes(int64_t cap) {
  const size_t n = (size_t)cap;
  return align_up(sizeof(uint32_t) * n) * 4 + sort_bytes_common(n);
}
PairBuf carve_tmp(void* ptr, int64_t cap) {
  const size_t n = (size_t)cap;
  char* c = (char*)ptr;
  PairBuf t;
  for (int i = 0; i < 2; ++i) { t.sort.keys[i] = (uint32_t*)c; c += align_up(sizeof(uint32_t) * n); }
  t.spare_vals = (uint32_t*)c; c += align_up(sizeof(uint32_t) * n);
  t.pair_gauss = (uint32_t*)c; c += align_up(sizeof(uint32_t) * n);
  t.sort.vals[0] = t.sort.vals[1] = nullptr;
  t.sort.hist = (uint32_t*)c; c += align_up(sizeof(uint32_t) * 256 * (size_t)rs_blocks(n) * rs_hist_copies(rs_blocks(n)));
  t.sort.digit_total = (uint32_t*)c;
  t.sort.nb_max = rs_blocks(n);
  t.sort.hist_copies = rs_hist_copies(rs_blocks(n));
  return t;
}
size_t bwd_tmp_bytes(int P, int F, int64_t cap) {
  return align_up(sizeof(float) * BWD_ACC * (size_t)P) + align_up((size_t)cap) + align_up(sizeof(float) * bwd_row_stride(F) * (size_t)cap);
}

static int validate(const TraseRastSettings* s, const TraseRastInputs* in) {
  if (!s || !in) { set_error("null settings/inputs"); return TRASE_ERR_INVALID; }
  if (in->P < 0 || s->image_width <= 0 || s->image_height <= 0) { set_error("bad sizes"); return TRASE_ERR_INVALID; }
  if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos) { set_error("bg/viewmatrix/projmatrix/campos are required"); return TRASE_ERR_INVALID; }
  if (in->F != 0 && in->F != 16 && in->F != 32) { set_error("feature width %d not compiled in (0,16,32)", in->F); return TRASE_ERR_UNSUPPORTED; }
  if (in->P == 0) return TRASE_OK;   // nothing to validate: empty tensors carry null pointers
  if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) {
    set_error("Please provide excatly one of either SHs or precomputed colors!");   // wording of the lineage's wrapper
    return TRASE_ERR_INVALID;
  }
  const bool sr = in->scales != nullptr || in->rotations != nullptr;
  if ((sr && in->cov3D_precomp) || (!sr && !in->cov3D_precomp) || (sr && (!in->scales || !in->rotations))) {
    set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    return TRASE_ERR_INVALID;
  }
  if (!in->means3D || !in->opacities) { set_error("means3D/opacities are required"); return TRASE_ERR_INVALID; }
  if (s->sh_degree < 0 || s->sh_degree > 3) { set_error("sh_degree %d outside 0..3", s->sh_degree); return TRASE_ERR_INVALID; }
  if (in->shs && (in->M < (s->sh_degree + 1) * (s->sh_degree + 1) || in->M > 16)) {
    set_error("shs holds %d coefficients, degree %d needs %d (max 16)", in->M, s->sh_degree, (s->sh_degree + 1) * (s->sh_degree + 1));
    return TRASE_ERR_INVALID;
  }
  if (in->F > 0 && !in->sh_objs) { set_error("F>0 but sh_objs is null"); return TRASE_ERR_INVALID; }
  return TRASE_OK;
}

// ---- launch-graph replay ----------------------------------------------------------------------------------------
// key = every byte of the argument records of a call; value = the instantiated graph of its launch sequence
struct GraphEntry { std::string key; hipGraphExec_t exec; };
static std::mutex g_graph_mu;
// 0 = off, 1 = every sync-free call, 2 = auto (the default): calls with at most g_graph_auto_p Gaussians -- the sizes whose ~45 launches
// per direction cost more host time than the kernels run (BASELINE configs 1 and 2); TRASE_GRAPH=0|1|auto, TRASE_GRAPH_AUTO_P=<n>
static int g_graph_mode = [] { const char* e = getenv("TRASE_GRAPH"); return !e || !strcmp(e, "auto") ? 2 : (atoi(e) != 0 ? 1 : 0); }();
static int g_graph_auto_p = [] { const char* e = getenv("TRASE_GRAPH_AUTO_P"); return e ? atoi(e) : 200000; }();
static std::map<uint64_t, std::vector<GraphEntry>> g_graphs;
static size_t g_graph_count = 0;
static int64_t g_graph_hits = 0, g_graph_misses = 0;
static std::map<int, hipStream_t> g_capture_streams;      // one capture stream per device (created with that device current)

static uint64_t fnv1a(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

static void graph_clear_locked() {
  for (auto& kv : g_graphs) for (auto& e : kv.second) hipGraphExecDestroy(e.exec);
  g_graphs.clear();
  g_graph_count = 0;
}

// Runs body(stream) either directly, or -- graph mode, profiler off, no per-kernel debug sync -- as a cached graph.
template <class Body>
static int run_maybe_graphed(int entry_id, std::initializer_list<std::pair<const void*, size_t>> parts, int debug, int device,
                             hipStream_t stream, int P, Body&& body) {
  if (!g_graph_mode || debug || prof_on() || (g_graph_mode == 2 && P > g_graph_auto_p)) return body(stream);
  {   // the CALLER is capturing (a whole training iteration inside torch.cuda.graph): its graph takes the launches as they are --
      // a replay of our own cached graph cannot be enqueued on a capturing stream
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return body(stream);
    (void)hipGetLastError();
  }
  std::string key((const char*)&entry_id, sizeof(entry_id));
  for (auto& p : parts) if (p.first) key.append((const char*)p.first, p.second); else key.append(p.second, '\0');
  const uint64_t h = fnv1a(key);
  std::lock_guard<std::mutex> lk(g_graph_mu);
  auto it = g_graphs.find(h);
  if (it != g_graphs.end()) {
    for (auto& e : it->second) {
      if (e.key == key) {
        ++g_graph_hits;
        return check_hip(hipGraphLaunch(e.exec, stream), "hipGraphLaunch");
      }
    }
  }
  ++g_graph_misses;
  // records that never repeat (an allocator handing out new blocks every iteration): give up, a capture per call is a loss
  if (g_graph_misses > 512 && g_graph_hits < g_graph_misses) { g_graph_mode = 0; graph_clear_locked(); return body(stream); }
  if (g_graph_count >= 256) graph_clear_locked();
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return body(stream); }
  hipStream_t& cap = g_capture_streams[device];
  if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) { cap = nullptr; (void)hipGetLastError(); return body(stream); }
  if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return body(stream); }
  const int rc = body(cap);
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(cap, &graph);
  if (rc != TRASE_OK || ec != hipSuccess || !graph) {
    // whatever went wrong while capturing: run the sequence directly (a genuine argument error fails again, with its message)
    if (graph) hipGraphDestroy(graph);
    (void)hipGetLastError();
    return body(stream);
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) { (void)hipGetLastError(); return body(stream); }
  g_graphs[h].push_back(GraphEntry{key, exec});
  ++g_graph_count;
  return check_hip(hipGraphLaunch(exec, stream), "hipGraphLaunch");
}

// A tile-row strip (second axis) leaves most Gaussians without a pair: their ids are compacted away before the
// depth sort (binning.hip launch_compact_live), so that the per-Gaussian stages cost what the strip holds, not what the scene does
static inline bool strip_mode(const TraseRastSettings* s) { return s->tile_row_begin != 0 || s->tile_row_end != 0; }
// the packed-FP32 cross-check backward knows no scopes: the features-only scope is a property of the MFMA backward and wins
static inline bool valu_backward(const TraseRastSettings* s) {
  return (s->variant & TRASE_VARIANT_VALU_BACKWARD) != 0 && (s->variant & TRASE_VARIANT_FEATURES_ONLY_BWD) == 0;
}

// depth order of the Gaussians (stable sort of the float32 depth bits; ties keep ascending Gaussian index) + the scan of their
// sub-tile counts in that order.  `keys` = where the preprocess kernel left the keys (strip mode: the sort's SECOND buffer)
static int depth_order(const LaunchCtx& c, const TraseRastSettings* s, const GeomBuf& g, const PreBuf& t, int P, const int32_t* radii,
                       int pack_bits) {
  // the input sits in buffer cfg.start (odd pass counts start from buffer 1), so that the sorted ids land in vals[0], where
  // stage 2 reads them; the 27-bit sort watches for a saturated key in its first pass (common.h)
  const DepthSortCfg cfg = depth_sort_cfg(s->variant);
  uint32_t* const flag = cfg.key_bits == 27 ? g.hdr + HDR_OVERFLOW : nullptr;
  int rc, idx = 0;
  if (strip_mode(s)) {
    rc = launch_compact_live(c, g, P, t, t.sort.keys[1 - cfg.start], t.sort.keys[cfg.start], t.sort.vals[cfg.start]);
    if (rc) return rc;
    rc = radix_sort_pairs(c, t.sort, g.hdr + (HDR_WORDS - 1), (uint32_t)P, 0, cfg.key_bits, false, &idx, cfg.digit_bits, cfg.start, DEPTH27_SAT, flag);
  } else {
    rc = radix_sort_pairs(c, t.sort, g.hdr + (HDR_WORDS - 1), (uint32_t)P, 0, cfg.key_bits, true, &idx, cfg.digit_bits, cfg.start, DEPTH27_SAT, flag);   // ids generated on the fly
  }
  if (rc) return rc;
  if (idx != 0) {
    set_error("internal: depth sort ended in buffer %d", idx);
    return TRASE_ERR_INVALID;
  }
  // the pair count is compared against the capacity later (stage 2 knows it); 0xffffffff = no limit yet
  return launch_scan_tiles(c, g, t.sort.vals[0], P, t, 0xffffffffu, radii, (s->image_width + TILE - 1) / TILE,
                           (s->image_height + TILE - 1) / TILE, pack_bits);
}

enum { WS_GEOM = 1, WS_PRE = 2, WS_BIN = 4, WS_IMG = 8, WS_TMP = 16 };
static int check_ws(const TraseRastInputs* in, const TraseRastSettings* s, const TraseRastWorkspace* ws, int need) {
  if (!ws) { set_error("null workspace"); return TRASE_ERR_WORKSPACE; }
  const int gx = (s->image_width + SUB - 1) / SUB, gy = (s->image_height + SUB - 1) / SUB;
  if ((need & WS_GEOM) && (!ws->geom || ws->geom_bytes < geom_bytes(in->P))) { set_error("geom workspace missing/too small"); return TRASE_ERR_WORKSPACE; }
  if ((need & WS_PRE) && (!ws->pre || ws->pre_bytes < pre_bytes(in->P))) { set_error("pre workspace missing/too small"); return TRASE_ERR_WORKSPACE; }
  if (need & (WS_BIN | WS_TMP)) {
    if (ws->capacity < 1 || ws->capacity > 0xfffffff0ll) { set_error("capacity out of range"); return TRASE_ERR_WORKSPACE; }
  }
  if ((need & WS_BIN) && (!ws->bin || ws->bin_bytes < bin_bytes(ws->capacity, gx * gy))) { set_error("bin workspace missing/too small"); return TRASE_ERR_WORKSPACE; }
  if ((need & WS_IMG) && (!ws->img || ws->img_bytes < img_bytes(s->image_width, s->image_height))) { set_error("img workspace missing/too small"); return TRASE_ERR_WORKSPACE; }
  if ((need & WS_TMP) && (!ws->tmp || ws->tmp_bytes < tmp_bytes(ws->capacity))) { set_error("tmp workspace missing/too small"); return TRASE_ERR_WORKSPACE; }
  return TRASE_OK;
}

}  // namespace trase

using namespace trase;

extern "C" {

const char* trase_last_error(void) { return g_err; }
const char* trase_version(void) { return "trase_amd 0.1 (gfx950)"; }

int trase_rast_sizes(int32_t P, int32_t W, int32_t H, int32_t F, int64_t capacity, TraseRastSizes* out) {
  (void)F;
  if (!out || P < 0 || W <= 0 || H <= 0 || capacity < 1) { set_error("trase_rast_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  const int gx = (W + SUB - 1) / SUB, gy = (H + SUB - 1) / SUB;   // lists are per 8x8 sub-tile
  out->geom_bytes = geom_bytes(P);
  out->bin_bytes = bin_bytes(capacity, gx * gy);
  out->img_bytes = img_bytes(W, H);
  out->pre_bytes = pre_bytes(P);
  out->tmp_bytes = tmp_bytes(capacity);
  out->bwd_tmp_bytes = bwd_tmp_bytes(P, F, capacity);
  return TRASE_OK;
}

int trase_rast_geom_layout(int32_t P, int64_t off[6]) {
  if (P < 0 || !off) { set_error("trase_rast_geom_layout: bad arguments"); return TRASE_ERR_INVALID; }
  const GeomBuf g = carve_geom(nullptr, P);
  off[0] = (int64_t)((char*)g.hdr - (char*)nullptr); off[1] = (int64_t)((char*)g.xy - (char*)nullptr);
  off[2] = (int64_t)((char*)g.conic_o - (char*)nullptr); off[3] = (int64_t)((char*)g.rgbd - (char*)nullptr);
  off[4] = (int64_t)((char*)g.tiles - (char*)nullptr); off[5] = (int64_t)((char*)g.clamped - (char*)nullptr);
  return TRASE_OK;
}

int trase_rast_preprocess(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                          const TraseRastWorkspace* ws, trase_stream_t stream_) {
  int rc = validate(s, in);
  if (rc) return rc;
  if (!out || (in->P > 0 && !out->radii)) { set_error("radii output required"); return TRASE_ERR_INVALID; }
  rc = check_ws(in, s, ws, WS_GEOM | WS_PRE);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(s->device));
  LaunchCtx c{stream, s->debug, s->variant};
  GeomBuf g = carve_geom(ws->geom, in->P);
  PreBuf t = carve_pre(ws->pre, in->P);
  if (in->P == 0) return launch_zero_bytes(g.hdr, sizeof(uint32_t) * HDR_WORDS, stream);   // (P > 0: the preprocess kernel clears it; a kernel, not a memset node: common.h)
  const DepthSortCfg cfg = depth_sort_cfg(s->variant);
  rc = launch_preprocess_fwd(c, *s, *in, out->radii, g, t.sort.keys[strip_mode(s) ? 1 - cfg.start : cfg.start], cfg.key_bits == 27);
  if (rc) return rc;
  return depth_order(c, s, g, t, in->P, out->radii, list_pack_bits(s, in->P));
}

int trase_rast_status(const TraseRastWorkspace* ws, int64_t status[3], trase_stream_t stream_) {
  if (!ws || !ws->geom || !status) { set_error("trase_rast_status: bad arguments"); return TRASE_ERR_INVALID; }
  uint32_t h[32];
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipMemcpyAsync(h, ws->geom, sizeof(h), hipMemcpyDeviceToHost, stream));
  TRASE_CHECK(hipStreamSynchronize(stream));
  status[0] = h[HDR_R]; status[1] = h[HDR_OVERFLOW]; status[2] = h[HDR_R_EFF];
  if (h[16] || h[20]) {   // internal consistency guards of the binning stage (hdr words 16..22)
    set_error("binning guard tripped: key flag %u (i=%u key=%u), slot flag %u (i=%u slot=%u)", h[16], h[17], h[18], h[20], h[21], h[22]);
    return TRASE_ERR_HIP;
  }
  return TRASE_OK;
}

int trase_rast_render(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                      const TraseRastWorkspace* ws, trase_stream_t stream_) {
  int rc = validate(s, in);
  if (rc) return rc;
  if (!out || !out->image || !out->depth || (in->P > 0 && !out->radii) || (in->F > 0 && !out->feats)) { set_error("null output"); return TRASE_ERR_INVALID; }
  rc = check_ws(in, s, ws, WS_GEOM | WS_PRE | WS_BIN | WS_IMG | WS_TMP);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(s->device));
  LaunchCtx c{stream, s->debug, s->variant};
  const int gx = (s->image_width + SUB - 1) / SUB, gy = (s->image_height + SUB - 1) / SUB;
  const int T = gx * gy;
  GeomBuf g = carve_geom(ws->geom, in->P);
  BinBuf b = carve_bin(ws->bin, ws->capacity, T);
  ImgBuf im = carve_img(ws->img, s->image_width, s->image_height);
  PreBuf pre = carve_pre(ws->pre, in->P);
  b.id_end = pre.id_end;       // the forward turns packed list values into row slots (HDR_PACK)
  PairBuf t = carve_tmp(ws->tmp, ws->capacity);
  const uint32_t cap = (uint32_t)ws->capacity;
  if (in->P > 0) {
    int bits = 1;
    while ((1 << bits) < T + 1) ++bits;          // keys 0..T (T = sentinel sub-tile)
    // the sort carries each pair's emit-order slot (generated on the fly in the first pass); arrange
    // the value ping-pong so that the last pass lands in the saved pair_slot array
    const int passes = radix_passes(0, bits);
    const int final_idx = passes & 1;
    t.sort.vals[final_idx] = b.pair_slot;
    t.sort.vals[final_idx ^ 1] = t.spare_vals;
    rc = launch_emit_pairs(c, *s, g, out->radii, pre.sort.vals[0], in->P, pre, t.sort.keys[0], t.pair_gauss, cap, b.ranges,
                           t.sort.vals[0]);
    if (rc) return rc;
    int idx = 0;
    rc = radix_sort_pairs(c, t.sort, g.hdr + HDR_R_EFF, cap, 0, bits, false, &idx);
    if (rc) return rc;
    if (idx != final_idx) { set_error("internal: tile sort ended in buffer %d", idx); return TRASE_ERR_INVALID; }
    // ranges only; with emit-order slots as list values the forward's staging step translates slot -> id
    rc = launch_tile_ranges(c, t.sort.keys[idx], g.hdr + HDR_R_EFF, cap, b.ranges, T + 1, g.hdr + 16, false);
    if (rc) return rc;
    return launch_render_fwd(c, *s, *in, *out, g, b, im, t.pair_gauss, (uint32_t)cap);
  } else {
    launch_zero_bytes(b.ranges, sizeof(uint2) * ((size_t)T + 1), stream);
  }
  return launch_render_fwd(c, *s, *in, *out, g, b, im);
}

int trase_rast_forward(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                       const TraseRastWorkspace* ws, trase_stream_t stream) {
  if (!s || !in || !out || !ws) { set_error("null argument"); return TRASE_ERR_INVALID; }
  return run_maybe_graphed(1, {{s, sizeof(*s)}, {in, sizeof(*in)}, {out, sizeof(*out)}, {ws, sizeof(*ws)}}, s->debug, s->device, (hipStream_t)stream, in->P,
                           [&](hipStream_t st) {
                             int rc = trase_rast_preprocess(s, in, out, ws, st);
                             if (rc) return rc;
                             return trase_rast_render(s, in, out, ws, st);
                           });
}

int trase_rast_graph_mode(int mode) {
  std::lock_guard<std::mutex> lk(g_graph_mu);
  g_graph_mode = mode;
  g_graph_hits = g_graph_misses = 0;
  if (!mode) graph_clear_locked();
  return TRASE_OK;
}

int trase_rast_graph_stats(int64_t stats[4]) {
  if (!stats) return TRASE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(g_graph_mu);
  stats[0] = g_graph_hits; stats[1] = g_graph_misses; stats[2] = (int64_t)g_graph_count; stats[3] = g_graph_mode;
  return TRASE_OK;
}

static int backward_impl(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                         const TraseRastWorkspace* ws, const TraseRastGrads* gr, trase_stream_t stream_);

int trase_rast_backward(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                        const TraseRastWorkspace* ws, const TraseRastGrads* gr, trase_stream_t stream_) {
  if (!s || !in || !out || !ws || !gr) { set_error("null argument"); return TRASE_ERR_INVALID; }
  return run_maybe_graphed(2, {{s, sizeof(*s)}, {in, sizeof(*in)}, {out, sizeof(*out)}, {ws, sizeof(*ws)}, {gr, sizeof(*gr)}}, s->debug, s->device,
                           (hipStream_t)stream_, in->P, [&](hipStream_t st) { return backward_impl(s, in, out, ws, gr, st); });
}

static int backward_impl(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                         const TraseRastWorkspace* ws, const TraseRastGrads* gr, trase_stream_t stream_) {
  int rc = validate(s, in);
  if (rc) return rc;
  if (!gr || !out || (in->P > 0 && !out->radii)) { set_error("null grads/outputs"); return TRASE_ERR_INVALID; }
  rc = check_ws(in, s, ws, WS_GEOM | WS_PRE | WS_BIN | WS_IMG);
  if (rc) return rc;
  if (!ws->tmp || ws->tmp_bytes < bwd_tmp_bytes(in->P, in->F, ws->capacity)) { set_error("backward tmp workspace too small"); return TRASE_ERR_WORKSPACE; }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(s->device));
  LaunchCtx c{stream, s->debug, s->variant};
  const int gx = (s->image_width + SUB - 1) / SUB, gy = (s->image_height + SUB - 1) / SUB;
  GeomBuf g = carve_geom(ws->geom, in->P);
  BinBuf b = carve_bin(ws->bin, ws->capacity, gx * gy);
  ImgBuf im = carve_img(ws->img, s->image_width, s->image_height);
  PreBuf pre = carve_pre(ws->pre, in->P);
  float* acc = (float*)ws->tmp;
  uint8_t* row_flags = (uint8_t*)ws->tmp + align_up(sizeof(float) * BWD_ACC * (size_t)in->P);
  float* rows = (float*)(row_flags + align_up((size_t)ws->capacity));
  if (in->P == 0) return TRASE_OK;
  TraseRastGrads g2 = *gr;
  if (!(s->variant & TRASE_VARIANT_DEPTH_GRAD)) g2.dL_ddepth = nullptr;   // lineage: depth carries no gradient
  TraseRastInputs in2 = *in;
  bool zero_feats = false;
  if (!g2.dL_dfeats) {
    // feature map unused by the loss (GAUSSIAN state): its channels contribute nothing
    in2.F = 0;
    zero_feats = g2.dL_dsh_objs != nullptr && in->F > 0;
    g2.dL_dsh_objs = nullptr;
  }
  if (zero_feats) launch_zero_bytes(gr->dL_dsh_objs, sizeof(float) * (size_t)in->F * in->P, stream);
  // phase 1: one gradient row per (sub-tile, Gaussian) pair, written to the pair's emit-order slot;
  // phase 2: every Gaussian sums its contiguous rows.  No atomics, bit-reproducible.
  // F == 0 here also means "no feature cotangent" (GAUSSIAN-state iterations): the MFMA kernel's image-only scope
  if ((in2.F == 32 || in2.F == 0) && !valu_backward(s)) {
    rc = launch_render_bwd_hw(c, *s, in2, g, b, im, g2, rows, row_flags, align_up((size_t)ws->capacity), out->depth);
  } else {
    launch_zero_bytes(row_flags, (size_t)ws->capacity, stream);
    rc = launch_render_bwd_gs(c, *s, in2, g, b, im, g2, rows, row_flags, out->depth);
  }
  if (rc) return rc;
  // a tile-row strip: the forward compacted the ids of the Gaussians with a pair in the strip in front of the depth order and
  // sorted only those -- the ranks behind them hold nothing.  The reduction walks the live ranks only (round 5: it used to walk
  // all P and read ids out of the unsorted tail: a memory fault for strips through THIS entry point; the raw entry point
  // always did this); the rows it does not write -- Gaussians without a pair: zero gradient -- start at zero.
  const int live_only = strip_mode(s) ? 1 : 0;
  if (live_only) {
    launch_zero_bytes(acc, sizeof(float) * BWD_ACC * (size_t)in->P, stream);
    if (g2.dL_dsh_objs) launch_zero_bytes(g2.dL_dsh_objs, sizeof(float) * (size_t)in2.F * in->P, stream);
  }
  rc = launch_reduce_rows(c, g, pre, in->P, in2.F, rows, row_flags, acc, g2.dL_dsh_objs, nullptr, 0, -1, -1, live_only);
  if (rc) return rc;
  return launch_preprocess_bwd(c, *s, *in, out->radii, g, acc, *gr);
}

// ---- render() with A1 fused in (raw parameters) ---------------------------------------------------------
static int validate_raw(const TraseRastSettings* s, const TraseRastRawInputs* r) {
  if (!s || !r) { set_error("null settings/inputs"); return TRASE_ERR_INVALID; }
  if (r->P < 0 || s->image_width <= 0 || s->image_height <= 0) { set_error("bad sizes"); return TRASE_ERR_INVALID; }
  if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos) { set_error("bg/viewmatrix/projmatrix/campos are required"); return TRASE_ERR_INVALID; }
  if (r->F != 0 && r->F != 16 && r->F != 32) { set_error("feature width %d not compiled in (0,16,32)", r->F); return TRASE_ERR_UNSUPPORTED; }
  if (s->sh_degree < 0 || s->sh_degree > 3) { set_error("sh_degree %d outside 0..3", s->sh_degree); return TRASE_ERR_INVALID; }
  if (r->P == 0) return TRASE_OK;
  if (!r->xyz || !r->opacity || !r->scaling || !r->rotation || (!r->colors_precomp && (!r->features_dc || !r->features_rest))) {
    set_error("raw inputs: xyz/opacity/scaling/rotation and either features_dc + features_rest or colors_precomp are required"); return TRASE_ERR_INVALID;
  }
  if (r->colors_precomp && r->sh_dir_undeformed) { set_error("raw inputs: colors_precomp and sh_dir_undeformed exclude each other"); return TRASE_ERR_INVALID; }
  if (r->d_xyz_se3 && r->d_xyz) { set_error("raw inputs: d_xyz_se3 (is_6dof) replaces d_xyz"); return TRASE_ERR_INVALID; }
  if (r->F > 0 && (!r->gaussian_features || !r->featn)) { set_error("raw inputs: gaussian_features/featn required when F > 0"); return TRASE_ERR_INVALID; }
  return TRASE_OK;
}

static TraseRastInputs raw_as_inputs(const TraseRastRawInputs* r) {
  TraseRastInputs in;
  memset(&in, 0, sizeof(in));
  in.P = r->P; in.M = 16; in.F = r->F; in.sh_objs = r->featn;
  return in;
}

int trase_rast_preprocess_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                              const TraseRastWorkspace* ws, trase_stream_t stream_) {
  int rc = validate_raw(s, raw);
  if (rc) return rc;
  if (!out || (raw->P > 0 && !out->radii)) { set_error("radii output required"); return TRASE_ERR_INVALID; }
  const TraseRastInputs in = raw_as_inputs(raw);
  rc = check_ws(&in, s, ws, WS_GEOM | WS_PRE);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(s->device));
  LaunchCtx c{stream, s->debug, s->variant};
  GeomBuf g = carve_geom(ws->geom, in.P);
  PreBuf t = carve_pre(ws->pre, in.P);
  if (in.P == 0) return launch_zero_bytes(g.hdr, sizeof(uint32_t) * HDR_WORDS, stream);   // (P > 0: the preprocess kernel clears it; a kernel, not a memset node: common.h)
  const DepthSortCfg cfg = depth_sort_cfg(s->variant);
  rc = launch_preprocess_fwd_raw(c, *s, *raw, out->radii, g, t.sort.keys[strip_mode(s) ? 1 - cfg.start : cfg.start], cfg.key_bits == 27, 0u,
                                 depth_dead_key(cfg.key_bits == 27));
  if (rc) return rc;
  return depth_order(c, s, g, t, in.P, out->radii, list_pack_bits(s, in.P));
}

// ---- two views per launch sequence (reduced to what is latency-bound) ---------------------------------
// The depth sort of one view is twelve launches of ~5 us over 1.2 MB of keys: the chip idles between them.  Two views' keys in ONE
// sort cost the same twelve launches: a live Gaussian's key is a positive float (z > 0.2), so its sign bit is free to carry the view
// index -- view 0's keys sort in front of view 1's, each in its own depth order, ties by ascending id as before.  Everything after
// the sort (scan, emit, sub-tile sort, compositing) and the whole backward run per view, unchanged, on the per-view workspaces; the
// forward results are bit-identical to two single-view calls.
static size_t pair_ws_bytes(int P) {
  const size_t n = 2 * (size_t)(P > 0 ? P : 1);
  return align_up(sizeof(uint32_t) * n) * 4 + sort_bytes_common(n) + align_up(sizeof(uint32_t) * 4);
}
int trase_rast_pair_sizes(int32_t P, size_t* bytes) {
  if (!bytes || P < 0) { set_error("trase_rast_pair_sizes: bad arguments"); return TRASE_ERR_INVALID; }
  *bytes = pair_ws_bytes(P);
  return TRASE_OK;
}

static int forward_raw_pair_impl(const TraseRastSettings* const s[2], const TraseRastRawInputs* const raw[2], const TraseRastOutputs* const out[2],
                                 const TraseRastWorkspace* const ws[2], void* pair_ws, hipStream_t stream) {
  const int P = raw[0]->P;
  TRASE_CHECK(hipSetDevice(s[0]->device));
  char* cp = (char*)pair_ws;
  const size_t n = 2 * (size_t)P;
  SortBufs cs;
  for (int i = 0; i < 2; ++i) { cs.keys[i] = (uint32_t*)cp; cp += align_up(sizeof(uint32_t) * n); }
  for (int i = 0; i < 2; ++i) { cs.vals[i] = (uint32_t*)cp; cp += align_up(sizeof(uint32_t) * n); }
  cs.hist = (uint32_t*)cp; cp += align_up(sizeof(uint32_t) * 256 * (size_t)rs_blocks(n) * rs_hist_copies(rs_blocks(n)));
  cs.hist_copies = rs_hist_copies(rs_blocks(n));
  cs.digit_total = (uint32_t*)cp; cp += align_up(sizeof(uint32_t) * 256 * 8);
  cs.nb_max = rs_blocks(n);
  uint32_t* n_word = (uint32_t*)cp;
  launch_fill_u32(n_word, (uint32_t)n, stream);
  GeomBuf g[2];
  PreBuf t[2];
  for (int v = 0; v < 2; ++v) {
    LaunchCtx c{stream, s[v]->debug, s[v]->variant};
    g[v] = carve_geom(ws[v]->geom, P);
    t[v] = carve_pre(ws[v]->pre, P);
    const int rc = launch_preprocess_fwd_raw(c, *s[v], *raw[v], out[v]->radii, g[v], cs.keys[0] + (size_t)v * P, false,
                                             v ? 0x80000000u : 0u, v ? 0xffffffffu : 0x7fffffffu);
    if (rc) return rc;
  }
  LaunchCtx c0{stream, s[0]->debug, s[0]->variant};
  int idx = 0;
  int rc = radix_sort_pairs(c0, cs, n_word, (uint32_t)n, 0, 32, true, &idx, 8, 0);      // ids generated on the fly: 0 .. 2 P - 1
  if (rc) return rc;
  if (idx != 0) { set_error("internal: pair depth sort ended in buffer %d", idx); return TRASE_ERR_INVALID; }
  launch_split_pair_ids(c0, cs.vals[0], P, t[0].sort.vals[0], t[1].sort.vals[0]);
  for (int v = 0; v < 2; ++v) {
    LaunchCtx c{stream, s[v]->debug, s[v]->variant};
    rc = launch_scan_tiles(c, g[v], t[v].sort.vals[0], P, t[v], 0xffffffffu, out[v]->radii, (s[v]->image_width + TILE - 1) / TILE,
                           (s[v]->image_height + TILE - 1) / TILE, list_pack_bits(s[v], P));
    if (rc) return rc;
    rc = trase_rast_render_raw(s[v], raw[v], out[v], ws[v], (trase_stream_t)stream);
    if (rc) return rc;
  }
  return TRASE_OK;
}

int trase_rast_forward_raw_pair(const TraseRastSettings* s0, const TraseRastRawInputs* raw0, const TraseRastOutputs* out0,
                                const TraseRastWorkspace* ws0, const TraseRastSettings* s1, const TraseRastRawInputs* raw1,
                                const TraseRastOutputs* out1, const TraseRastWorkspace* ws1, void* pair_ws, size_t pair_bytes,
                                trase_stream_t stream) {
  if (!s0 || !raw0 || !out0 || !ws0 || !s1 || !raw1 || !out1 || !ws1) { set_error("null argument"); return TRASE_ERR_INVALID; }
  const TraseRastSettings* s[2] = {s0, s1};
  const TraseRastRawInputs* raw[2] = {raw0, raw1};
  const TraseRastOutputs* out[2] = {out0, out1};
  const TraseRastWorkspace* ws[2] = {ws0, ws1};
  for (int v = 0; v < 2; ++v) {
    int rc = validate_raw(s[v], raw[v]);
    if (rc) return rc;
    if (raw[v]->P > 0 && !out[v]->radii) { set_error("radii output required"); return TRASE_ERR_INVALID; }
    const TraseRastInputs in = raw_as_inputs(raw[v]);
    rc = check_ws(&in, s[v], ws[v], WS_GEOM | WS_PRE);
    if (rc) return rc;
    if (strip_mode(s[v])) { set_error("trase_rast_forward_raw_pair: tile-row strips are rendered one view at a time"); return TRASE_ERR_UNSUPPORTED; }
  }
  if (raw0->P != raw1->P || s0->device != s1->device) { set_error("trase_rast_forward_raw_pair: both views render the same P Gaussians on one device"); return TRASE_ERR_INVALID; }
  if (raw0->P == 0) {      // nothing to sort: the single-view path handles the empty scene
    int rc = trase_rast_forward_raw(s0, raw0, out0, ws0, stream);
    return rc ? rc : trase_rast_forward_raw(s1, raw1, out1, ws1, stream);
  }
  if (!pair_ws || pair_bytes < pair_ws_bytes(raw0->P)) { set_error("trase_rast_forward_raw_pair: pair workspace too small"); return TRASE_ERR_WORKSPACE; }
  struct { const void* p; size_t b; } extra = {pair_ws, pair_bytes};
  return run_maybe_graphed(5, {{s0, sizeof(*s0)}, {raw0, sizeof(*raw0)}, {out0, sizeof(*out0)}, {ws0, sizeof(*ws0)}, {s1, sizeof(*s1)},
                               {raw1, sizeof(*raw1)}, {out1, sizeof(*out1)}, {ws1, sizeof(*ws1)}, {&extra, sizeof(extra)}},
                           s0->debug | s1->debug, s0->device, (hipStream_t)stream, raw0->P,
                           [&](hipStream_t st) { return forward_raw_pair_impl(s, raw, out, ws, pair_ws, st); });
}

int trase_rast_render_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                          const TraseRastWorkspace* ws, trase_stream_t stream) {
  int rc = validate_raw(s, raw);
  if (rc) return rc;
  // stage 2 only needs P, F and the feature rows: reuse the regular entry point on a synthetic input record
  TraseRastInputs in = raw_as_inputs(raw);
  static const float dummy = 0.f;
  in.means3D = &dummy; in.opacities = &dummy; in.shs = &dummy; in.scales = &dummy; in.rotations = &dummy;   // never dereferenced in stage 2
  if (in.F > 0 && !in.sh_objs) { set_error("featn required"); return TRASE_ERR_INVALID; }
  return trase_rast_render(s, &in, out, ws, stream);
}

// The raw backward in two phases (see include/trase_rast.h): phase & 1 = compositing backward over every sub-tile (one
// gradient row per pair), phase & 2 = per-Gaussian tail over the ids [p_begin, p_end).
static int backward_raw_phases(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                               const TraseRastWorkspace* ws, const TraseRastRawGrads* gr, trase_stream_t stream_, int phase,
                               int p_begin, int p_end) {
  int rc = validate_raw(s, raw);
  if (rc) return rc;
  if (!gr || !out || (raw->P > 0 && !out->radii)) { set_error("null grads/outputs"); return TRASE_ERR_INVALID; }
  TraseRastInputs in = raw_as_inputs(raw);
  rc = check_ws(&in, s, ws, WS_GEOM | WS_PRE | WS_BIN | WS_IMG);
  if (rc) return rc;
  if (!ws->tmp || ws->tmp_bytes < bwd_tmp_bytes(in.P, in.F, ws->capacity)) { set_error("backward tmp workspace too small"); return TRASE_ERR_WORKSPACE; }
  if (s->variant & TRASE_VARIANT_FORWARD_ONLY) {
    set_error("backward on a forward that ran with TRASE_VARIANT_FORWARD_ONLY: its backward-only state was never stored"); return TRASE_ERR_INVALID;
  }
  const bool ranged = p_begin >= 0;
  if (ranged && (p_begin % 64 != 0 || p_end > raw->P || p_end < p_begin || (p_end != raw->P && p_end % 64 != 0))) {
    set_error("backward_raw_gaussians: [%d, %d) must start on a multiple of 64 and end on one or at P = %d", p_begin, p_end, raw->P);
    return TRASE_ERR_INVALID;
  }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(s->device));
  LaunchCtx c{stream, s->debug, s->variant};
  const int gx = (s->image_width + SUB - 1) / SUB, gy = (s->image_height + SUB - 1) / SUB;
  GeomBuf g = carve_geom(ws->geom, in.P);
  BinBuf b = carve_bin(ws->bin, ws->capacity, gx * gy);
  ImgBuf im = carve_img(ws->img, s->image_width, s->image_height);
  PreBuf pre = carve_pre(ws->pre, in.P);
  float* acc = (float*)ws->tmp;
  uint8_t* row_flags = (uint8_t*)ws->tmp + align_up(sizeof(float) * BWD_ACC * (size_t)in.P);
  float* rows = (float*)(row_flags + align_up((size_t)ws->capacity));
  if (in.P == 0) return TRASE_OK;
  TraseRastGrads g2;
  memset(&g2, 0, sizeof(g2));
  g2.dL_dimage = gr->dL_dimage; g2.dL_dfeats = gr->dL_dfeats;
  g2.dL_ddepth = (s->variant & TRASE_VARIANT_DEPTH_GRAD) ? gr->dL_ddepth : nullptr;
  float* d_feats = gr->dL_dgaussian_features;
  const bool no_feat_cotangent = !g2.dL_dfeats;
  if (no_feat_cotangent) {
    in.F = 0;
    d_feats = nullptr;
  }
  // sparse strip gradients: only the rows of the Gaussians with a pair in the strip are written (the caller keeps the others zero)
  const bool sparse = strip_mode(s) && !ranged && (s->variant & TRASE_VARIANT_SPARSE_STRIP_GRADS) != 0;
  if (phase & 1) {
    if (no_feat_cotangent && gr->dL_dgaussian_features && raw->F > 0 && !sparse)
      launch_zero_bytes(gr->dL_dgaussian_features, sizeof(float) * (size_t)raw->F * raw->P, stream);
    if ((in.F == 32 || in.F == 0) && !valu_backward(s)) {
      rc = launch_render_bwd_hw(c, *s, in, g, b, im, g2, rows, row_flags, align_up((size_t)ws->capacity), out->depth);
    } else {
      launch_zero_bytes(row_flags, (size_t)ws->capacity, stream);
      rc = launch_render_bwd_gs(c, *s, in, g, b, im, g2, rows, row_flags, out->depth);
    }
    if (rc) return rc;
  }
  if (phase & 2) {
    // a tile-row strip: the row reduction walks the Gaussians that have a pair only (their ids were compacted in front of the
    // depth order); the feature-gradient rows of the others are zeroed by the per-Gaussian kernel, which writes their other
    // gradients as zeros anyway
    const int live_only = (strip_mode(s) && !ranged) ? 1 : 0;
    rc = launch_reduce_rows(c, g, pre, in.P, in.F, rows, row_flags, acc, d_feats, raw->gaussian_features, raw->norm_features,
                            ranged ? p_begin : -1, ranged ? p_end : -1, live_only);
    if (rc) return rc;
    rc = launch_preprocess_bwd_raw(c, *s, *raw, out->radii, g, acc, *gr, ranged ? p_begin : 0, ranged ? p_end : raw->P,
                                   (live_only && d_feats && !sparse) ? 1 : 0, sparse ? pre.live_ids : nullptr);
  }
  return rc;
}

int trase_rast_backward_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                            const TraseRastWorkspace* ws, const TraseRastRawGrads* gr, trase_stream_t stream) {
  if (!s || !raw || !out || !ws || !gr) { set_error("null argument"); return TRASE_ERR_INVALID; }
  return run_maybe_graphed(4, {{s, sizeof(*s)}, {raw, sizeof(*raw)}, {out, sizeof(*out)}, {ws, sizeof(*ws)}, {gr, sizeof(*gr)}}, s->debug, s->device,
                           (hipStream_t)stream, raw->P, [&](hipStream_t st) { return backward_raw_phases(s, raw, out, ws, gr, st, 3, -1, -1); });
}

int trase_rast_forward_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                           const TraseRastWorkspace* ws, trase_stream_t stream) {
  if (!s || !raw || !out || !ws) { set_error("null argument"); return TRASE_ERR_INVALID; }
  return run_maybe_graphed(3, {{s, sizeof(*s)}, {raw, sizeof(*raw)}, {out, sizeof(*out)}, {ws, sizeof(*ws)}}, s->debug, s->device, (hipStream_t)stream, raw->P,
                           [&](hipStream_t st) {
                             int rc = trase_rast_preprocess_raw(s, raw, out, ws, st);
                             if (rc) return rc;
                             return trase_rast_render_raw(s, raw, out, ws, st);
                           });
}

int trase_rast_zero_live_rows(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastWorkspace* ws,
                              const TraseRastRawGrads* gr, trase_stream_t stream_) {
  if (!s || !raw || !ws || !gr) { set_error("null argument"); return TRASE_ERR_INVALID; }
  if (!ws->geom || ws->geom_bytes < geom_bytes(raw->P) || !ws->pre || ws->pre_bytes < pre_bytes(raw->P)) {
    set_error("zero_live_rows: geom / pre workspaces of the forward whose live rows are to be cleared are required");
    return TRASE_ERR_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  TRASE_CHECK(hipSetDevice(s->device));
  LaunchCtx c{stream, s->debug, s->variant};
  return launch_zero_live_rows(c, carve_geom(ws->geom, raw->P), carve_pre(ws->pre, raw->P), raw->P, raw->F, *gr);
}

int trase_rast_backward_raw_compose(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                                    const TraseRastWorkspace* ws, const TraseRastRawGrads* gr, trase_stream_t stream) {
  return backward_raw_phases(s, raw, out, ws, gr, stream, 1, -1, -1);
}

int trase_rast_backward_raw_gaussians(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                                      const TraseRastWorkspace* ws, const TraseRastRawGrads* gr, int32_t p_begin,
                                      int32_t p_end, trase_stream_t stream) {
  if (p_begin < 0) { set_error("backward_raw_gaussians: negative range start"); return TRASE_ERR_INVALID; }
  return backward_raw_phases(s, raw, out, ws, gr, stream, 2, p_begin, p_end);
}

int trase_prof_enable(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  g_prof_mode = enable;
  if (enable) {
    for (ProfRec* r : g_prof_recs) { hipEventDestroy(r->e0); hipEventDestroy(r->e1); delete r; }
    g_prof_recs.clear();
  }
  return TRASE_OK;
}

int trase_prof_report(char* buf, size_t buf_bytes) {
  if (!buf || buf_bytes < 4) return TRASE_ERR_INVALID;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, std::pair<double, int>> agg;
  for (ProfRec* r : g_prof_recs) {
    if (hipEventSynchronize(r->e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r->e0, r->e1) != hipSuccess) continue;
    auto& a = agg[r->name];
    a.first += ms; a.second += 1;
  }
  std::string js = "{";
  bool first = true;
  for (auto& kv : agg) {
    char tmp[256];
    snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"ms\": %.6f, \"n\": %d}", first ? "" : ", ", kv.first.c_str(),
             kv.second.first / (kv.second.second > 0 ? kv.second.second : 1), kv.second.second);
    js += tmp;
    first = false;
  }
  js += "}";
  for (ProfRec* r : g_prof_recs) { hipEventDestroy(r->e0); hipEventDestroy(r->e1); delete r; }
  g_prof_recs.clear();
  if (js.size() + 1 > buf_bytes) { set_error("profile report buffer too small"); return TRASE_ERR_WORKSPACE; }
  memcpy(buf, js.c_str(), js.size() + 1);
  return TRASE_OK;
}

}  // extern "C"
