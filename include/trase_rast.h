/*
 * trase_rast.h -- C ABI of the MI355X-native TRASE rasterizer path.
 *
 * Drop-in boundary for the one hot path of yunjinli/TRASE: the native operator
 * behind gaussian_renderer.render().  The reference binds this path through a
 * pybind11/torch CUDA extension (un-vendored submodule, .gitmodules:4-6):
 *
 *   diff_gaussian_rasterization.GaussianRasterizationSettings   gaussian_renderer/__init__.py:58-71
 *   diff_gaussian_rasterization.GaussianRasterizer.forward      gaussian_renderer/__init__.py:137-146
 *   (its autograd backward, triggered by loss.backward())       train.py:299
 *   simple_knn._C.distCUDA2                                     scene/gaussian_model.py:237
 *
 * This header is the C-level equivalent: plain pointers and sizes, no torch
 * types.  All pointers named "device" are HIP device pointers valid on
 * `settings.device`; every call enqueues work on the caller's `stream` and
 * returns without synchronising unless stated.  The library allocates nothing
 * persistent; every call works on the caller's workspaces only and the entry
 * points are re-entrant (PyTorch calls the backward from an autograd worker
 * thread).  Process-wide state is limited to the last-error string (thread-local)
 * and the opt-in profiler (trase_prof_enable; mutex-guarded event records).  The
 * Python wrapper's policy (capacity, variant, strip: trase_amd/rasterizer.py) is
 * per process; a backward always uses what its forward captured.
 *
 * Return codes: 0 ok; <0 invalid argument / HIP error (see trase_strerror);
 * >0 is never returned by enqueue calls (capacity overflow is reported through
 * trase_rast_status, because it is only known on the device).
 */
#ifndef TRASE_RAST_H
#define TRASE_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* trase_stream_t; /* hipStream_t */

enum {
  TRASE_OK = 0,
  TRASE_ERR_INVALID = -1,     /* bad argument combination (mirrors the reference's ValueError cases) */
  TRASE_ERR_UNSUPPORTED = -2, /* e.g. feature width not compiled in */
  TRASE_ERR_WORKSPACE = -3,   /* workspace too small */
  TRASE_ERR_HIP = -4          /* a HIP call failed; message via trase_last_error() */
};

/* GaussianRasterizationSettings (12 fields, gaussian_renderer/__init__.py:58-71).
 * The tensors of the reference record (bg, viewmatrix, projmatrix, campos) stay
 * on the device and are read by the kernels -- no D2H copy.  Matrices are the
 * transposed (row-vector) forms of scene/cameras.py:76-78: flat[4*col+row]. */
typedef struct TraseRastSettings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  const float* bg;         /* device, 3 floats  */
  float scale_modifier;
  const float* viewmatrix; /* device, 16 floats */
  const float* projmatrix; /* device, 16 floats */
  int32_t sh_degree;       /* active degree 0..3 */
  const float* campos;     /* device, 3 floats  */
  int32_t prefiltered;
  int32_t debug;           /* !=0: synchronise + check after every kernel */
  int32_t device;          /* HIP device ordinal of all pointers and of `stream` */
  int32_t variant;         /* 0 = default; a bit set of TraseVariant (below): lineage switches, backward scope, the
                            * VALU cross-check kernels, the list-value form */
  /* Tile-row strip (second multi-GPU axis, SURVEY.md 8e: one view sharded over ranks by rows of 16x16 tiles).  Rows
   * [tile_row_begin, tile_row_end) are binned, composited and differentiated; pixels outside the strip are not written,
   * per-Gaussian gradients are this strip's partial sums (the ranks' strips add up to the full gradient), radii stay
   * whole-image.  begin == end == 0 means the whole image. */
  int32_t tile_row_begin;
  int32_t tile_row_end;
  float feat_bg;           /* background value of every feature channel; read only when variant & TRASE_VARIANT_FEATS_BG */
  int32_t reserved0;
} TraseRastSettings;

/* Bits of TraseRastSettings.variant.  Every other bit is ignored. */
typedef enum TraseVariant {
  /* -- lineage switches (the three places the absent fork of the CUDA extension is most likely to
   *    differ from the public lineage; each flips the HIP kernels AND oracle/raster_oracle.py's OracleOptions) -- */
  TRASE_VARIANT_DEPTH_GRAD = 0x100,          /* dL_ddepth is honoured (lineage: the depth output carries no gradient) */
  TRASE_VARIANT_FEATS_BG = 0x10000,          /* feats[c] += T_final * feat_bg */
  TRASE_VARIANT_DEPTH_NORM = 0x20000,        /* depth = sum(w z) / (1 - T_final) */
  /* -- backward scope -- */
  TRASE_VARIANT_FEATURES_ONLY_BWD = 0x400,   /* F = 32: only dL/dsh_objs is produced (FEATURE state once densification has
                                              * ended); every other gradient is written as zeros */
  /* -- cross-check formulations (same results to fp32 rounding; the tests compare the MFMA kernels against them) -- */
  TRASE_VARIANT_VALU_BACKWARD = 0x40,        /* packed-FP32 compositing backward (render_bwd_gs.hip) also for F = 32 (ignored under
                                              * TRASE_VARIANT_FEATURES_ONLY_BWD: that scope exists in the MFMA backward only) */
  TRASE_VARIANT_VALU_FORWARD = 0x2000,       /* packed-FP32 compositing forward (render.hip) also for F = 32 */
  TRASE_VARIANT_SLOT_LISTS = 0x100000,       /* sub-tile lists always carry emit-order slots (default: packed (id, pair index)
                                              * values whenever every Gaussian has few enough pairs -- decided on the device,
                                              * results identical) */
  /* -- tile-row strips (tile_row_begin / tile_row_end set) -- */
  TRASE_VARIANT_DEPTH32 = 0x400000,          /* depth sort on the raw float32 depth bits (four 8-bit passes) instead of the default 27-bit
                                              * key (three 9-bit passes, exact for view depth < 13 107): selected by the caller after a
                                              * forward has raised bit 1 of the header's overflow word (a saturated depth key) */
  TRASE_VARIANT_FORWARD_ONLY = 0x800000,     /* a forward nobody will differentiate (torch.no_grad()): what only the backward reads -- per-Gaussian
                                              * colour / clamp arrays, per-pixel final transmittance and contributor count -- is not stored.  A
                                              * backward on such a forward's workspaces is refused */
  TRASE_VARIANT_SPARSE_STRIP_GRADS = 0x200000, /* trase_rast_backward_raw writes ONLY the gradient rows of the Gaussians that have a
                                              * pair in the strip (~1 / world of them); the caller guarantees that every other row
                                              * of the gradient tensors is zero -- persistent tensors, zero-filled once, whose
                                              * previously written rows trase_rast_zero_live_rows clears.  Default: dense (every
                                              * row written, zeros included) */
  /* -- diagnostics: compiled only into `make AB=1` builds of the library (-DTRASE_AB), ignored otherwise -- */
  TRASE_VARIANT_AB_TIMING = 0x1000,          /* phase cycle counters of the MFMA backward (geom header words 32..39) */
  TRASE_VARIANT_AB_COUNT = 0x8000,           /* lane-utilisation counters of the MFMA backward (header words 40..47) */
  TRASE_VARIANT_AB_ORDER_IMAGE = 0x40000,    /* compositing kernels visit the sub-tiles in image order ... */
  TRASE_VARIANT_AB_ORDER_8 = 0x80000         /* ... in 8x8 blocks (default: 16x16 blocks) */
} TraseVariant;

/* Inputs of GaussianRasterizer.forward (gaussian_renderer/__init__.py:137-146).
 * Exactly one of shs/colors_precomp and one of (scales,rotations)/cov3D_precomp
 * is non-NULL, as in the reference.  All contiguous float32. */
typedef struct TraseRastInputs {
  int32_t P;                   /* number of Gaussians */
  int32_t M;                   /* SH coefficients stored per Gaussian (16 for degree 3); 0 if shs NULL */
  int32_t F;                   /* feature channels of sh_objs (32 in TRASE); 0 if sh_objs NULL */
  const float* means3D;        /* (P,3)   */
  const float* shs;            /* (P,M,3) coefficient-major */
  const float* sh_objs;        /* (P,1,F) */
  const float* colors_precomp; /* (P,3)   */
  const float* opacities;      /* (P,1)   */
  const float* scales;         /* (P,3)   */
  const float* rotations;      /* (P,4) (r,x,y,z), NOT assumed unit length */
  const float* cov3D_precomp;  /* (P,6)   */
} TraseRastInputs;

/* Outputs: the reference's 4-tuple (image, radii, feats, depth), channel-planar. */
typedef struct TraseRastOutputs {
  float* image;   /* (3,H,W) */
  int32_t* radii; /* (P,)    */
  float* feats;   /* (F,H,W) or NULL when F == 0 */
  float* depth;   /* (1,H,W) */
} TraseRastOutputs;

/* Caller-owned workspaces.  geom/bin/img are saved by the caller between
 * forward and backward (the reference saves geomBuffer/binningBuffer/imgBuffer
 * the same way).  pre is stage-1 scratch that must survive until stage 2 (its
 * size depends on P only); tmp is stage-2 scratch (size depends on capacity) and
 * doubles as the backward's scratch (bwd_tmp_bytes).  Only geom/bin/img sizes
 * depend on nothing else, so a caller may run stage 1, read the pair count with
 * trase_rast_status and only then allocate bin/tmp. */
typedef struct TraseRastWorkspace {
  void* geom; size_t geom_bytes;
  void* bin;  size_t bin_bytes;
  void* img;  size_t img_bytes;
  void* pre;  size_t pre_bytes;
  void* tmp;  size_t tmp_bytes;
  int64_t capacity;            /* max (tile,Gaussian) pairs `bin`/`tmp` were sized for */
} TraseRastWorkspace;

typedef struct TraseRastSizes {
  size_t geom_bytes, bin_bytes, img_bytes, pre_bytes, tmp_bytes, bwd_tmp_bytes;
} TraseRastSizes;

/* Cotangents in, gradients out (A4).  NULL cotangent == that output was unused
 * (ctx.set_materialize_grads(False) semantics); NULL gradient == not needed. */
typedef struct TraseRastGrads {
  const float* dL_dimage;   /* (3,H,W) or NULL */
  const float* dL_dfeats;   /* (F,H,W) or NULL */
  const float* dL_ddepth;   /* (1,H,W) or NULL (lineage: ignored unless settings.variant bit says so) */
  float* dL_dmeans3D;       /* (P,3)   */
  float* dL_dmeans2D;       /* (P,3) x,y = NDC-scaled screen gradient, z = 0 */
  float* dL_dshs;           /* (P,M,3) */
  float* dL_dsh_objs;       /* (P,1,F) */
  float* dL_dcolors;        /* (P,3)   */
  float* dL_dopacities;     /* (P,1)   */
  float* dL_dscales;        /* (P,3)   */
  float* dL_drotations;     /* (P,4)   */
  float* dL_dcov3D;         /* (P,6)   */
} TraseRastGrads;

/* Sizes of every workspace for P Gaussians, a W x H image, F feature channels
 * and room for `capacity` (tile,Gaussian) pairs. */
int trase_rast_sizes(int32_t P, int32_t W, int32_t H, int32_t F, int64_t capacity, TraseRastSizes* out);

/* Stage 1: per-Gaussian projection / EWA / colour + tile counting + depth sort.
 * Writes out->radii and the geom workspace (incl. the device-side pair count). */
int trase_rast_preprocess(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                          const TraseRastWorkspace* ws, trase_stream_t stream);

/* Blocking read of {num_rendered, overflow, num_rendered_culled} from the geom
 * workspace (the reference's forward returns num_rendered the same way). */
int trase_rast_status(const TraseRastWorkspace* ws, int64_t status[3], trase_stream_t stream);

/* Byte offsets of the per-Gaussian arrays inside the geom workspace sized for P Gaussians (pure host
 * arithmetic): off[0] header (64 u32: [0] lineage pair count, [1] overflow flag, [2] binned pair count),
 * off[1] xy float2[P] (pixel centre), off[2] conic_opacity float4[P], off[3] rgb_depth float4[P] (.w = view-space
 * depth, the blend-order key), off[4] tiles u32[P], off[5] clamp bits u32[P].  Introspection for parity tests and
 * debugging -- the counterpart of reading the reference's geomBuffer; entries of culled Gaussians are undefined. */
int trase_rast_geom_layout(int32_t P, int64_t off[6]);

/* Stage 2: binning (tile lists in depth order) + alpha compositing. */
int trase_rast_render(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                      const TraseRastWorkspace* ws, trase_stream_t stream);

/* Stage 1 + 2 back to back, no host synchronisation. */
int trase_rast_forward(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                       const TraseRastWorkspace* ws, trase_stream_t stream);

/* Backward of trase_rast_forward; ws->tmp must hold bwd_tmp_bytes. */
int trase_rast_backward(const TraseRastSettings* s, const TraseRastInputs* in, const TraseRastOutputs* out,
                        const TraseRastWorkspace* ws, const TraseRastGrads* g, trase_stream_t stream);

/* ---- render() with the reference's A1 "prep" fused in --------------------------
 * Takes the RAW GaussianModel parameters (scene/gaussian_model.py:56-63) and the per-view deformation and
 * applies gaussian_renderer/__init__.py:82-121 inside the per-Gaussian kernels: means3D = _xyz + d_xyz,
 * scales = exp(_scaling) + d_scaling, rotations = normalize(_rotation) + d_rotation, opacity =
 * sigmoid(_opacity), shs = cat(_features_dc, _features_rest), sh_objs = f / (||f|| + 1e-9).  d_* may be
 * NULL (warm-up passes the float 0.0, train.py:192-193).  SH layout: features_dc (P,1,3), features_rest
 * (P,15,3).  featn is a caller-owned (P,F) buffer that must live until the backward. */
typedef struct TraseRastRawInputs {
  int32_t P;
  int32_t F;
  int32_t norm_features;            /* norm_gaussian_features flag of render() */
  const float* xyz;                 /* (P,3)    */
  const float* d_xyz;               /* (P,3) or NULL */
  const float* features_dc;         /* (P,1,3)  */
  const float* features_rest;       /* (P,15,3) */
  const float* opacity;             /* (P,1) logits */
  const float* scaling;             /* (P,3) log-scales */
  const float* d_scaling;           /* (P,3) or NULL */
  const float* rotation;            /* (P,4) raw quaternion */
  const float* d_rotation;          /* (P,4) or NULL */
  const float* gaussian_features;   /* (P,1,F) raw, or NULL when F == 0 */
  float* featn;                     /* (P,F) work buffer */
  /* render()'s other call patterns, fused as well (round 6; every pointer may be NULL = not used):
   *   colors_precomp   override_color= (gaussian_renderer/__init__.py:112-113; render.py:240,296,344, gui.py): (P,3) colours taken as
   *                    they are -- no SH evaluation, features_dc / features_rest are not read and may be NULL;
   *   mask             mask= (:123-135): (P) bytes, 0 removes the Gaussian from the view (the reference indexes every input by the
   *                    mask).  out->radii and every gradient stay FULL size (P rows; rows of removed Gaussians: radii 0, gradients
   *                    0 -- what the index's backward scatters); the caller compacts radii to the subset for render()'s dict;
   *   d_xyz_se3        is_6dof (:75-80): (P,4,4) row-major transforms, means3D = from_homogenous(M [xyz, 1]); replaces d_xyz;
   *   sh_dir_undeformed  != 0: pipe.convert_SHs_python (:103-108) -- the SH is evaluated in the direction of the UNDEFORMED xyz
   *                    (the Python fallback uses pc.get_xyz there), colour clamp as in the rasterizer. */
  const float* colors_precomp;
  const uint8_t* mask;
  const float* d_xyz_se3;
  int32_t sh_dir_undeformed;
  int32_t reserved1;
} TraseRastRawInputs;

typedef struct TraseRastRawGrads {
  const float* dL_dimage; const float* dL_dfeats; const float* dL_ddepth;   /* cotangents, NULL if unused */
  float* dL_dxyz; float* dL_dd_xyz; float* dL_dmeans2D;
  float* dL_dfeatures_dc; float* dL_dfeatures_rest; float* dL_dopacity;
  float* dL_dscaling; float* dL_dd_scaling; float* dL_drotation; float* dL_dd_rotation;
  float* dL_dgaussian_features;
  float* dL_dcolors_precomp;        /* (P,3), with colors_precomp */
  float* dL_dd_xyz_se3;             /* (P,4,4), with d_xyz_se3 */
} TraseRastRawGrads;

int trase_rast_preprocess_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                              const TraseRastWorkspace* ws, trase_stream_t stream);
int trase_rast_render_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                          const TraseRastWorkspace* ws, trase_stream_t stream);
int trase_rast_backward_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                            const TraseRastWorkspace* ws, const TraseRastRawGrads* g, trase_stream_t stream);
/* trase_rast_preprocess_raw + trase_rast_render_raw in one call, for callers that know the capacity beforehand (the
 * sync-free policy): one boundary crossing per direction. */
int trase_rast_forward_raw(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                           const TraseRastWorkspace* ws, trase_stream_t stream);
/* Two views of the SAME Gaussians in one launch sequence (round 5; no counterpart in the reference, whose loop renders one view
 * per iteration, train.py:180).  Identical results to two trase_rast_forward_raw calls -- each view keeps its own settings,
 * per-view deformation (d_*), outputs and workspaces, and its backward is the ordinary trase_rast_backward_raw on those -- but the
 * two depth sorts, the latency-bound part of a view (twelve launches over 1.2 MB of keys at 300k Gaussians), are ONE sort of 2 P
 * keys with the view index in the sign bit of the float32 depth key.  `pair_ws`: scratch of trase_rast_pair_sizes(P) bytes, free
 * again when the call's work has run.  Whole-image views only (no tile-row strips); the sync-free capacity policy (both
 * workspaces sized by the caller beforehand). */
int trase_rast_pair_sizes(int32_t P, size_t* bytes);
int trase_rast_forward_raw_pair(const TraseRastSettings* s0, const TraseRastRawInputs* raw0, const TraseRastOutputs* out0,
                                const TraseRastWorkspace* ws0, const TraseRastSettings* s1, const TraseRastRawInputs* raw1,
                                const TraseRastOutputs* out1, const TraseRastWorkspace* ws1, void* pair_ws, size_t pair_bytes,
                                trase_stream_t stream);

/* Launch-graph replay (hipGraph) for small workloads, where the ~45 kernel launches of a direction cost more host time
 * than the kernels run (BASELINE configs 1 and 2).  With mode != 0, trase_rast_forward / _forward_raw / _backward /
 * _backward_raw capture their launch sequence once per distinct argument record (every scalar and every pointer of the
 * settings, inputs, outputs, workspace and gradient structs) on an internal stream and replay the instantiated graph on
 * the caller's stream when the same record comes back -- which it does in a training loop whose allocator hands out the
 * same blocks every iteration.  Only the sync-free entry points are graphed (nothing in them reads back to the host).
 * The cache holds up to 256 graphs and switches itself off when records stop repeating (a miss costs a capture).
 * mode: 0 = off, 1 = every call, 2 = auto -- calls of at most 200 000 Gaussians (environment TRASE_GRAPH_AUTO_P); the library
 * starts in mode 2 (environment TRASE_GRAPH = 0 | 1 | auto).  trase_rast_graph_stats: {hits, misses, cached, mode}. */
int trase_rast_graph_mode(int mode);
int trase_rast_graph_stats(int64_t stats[4]);

/* The same backward in two phases, for a view-parallel caller that starts exchanging the gradients of the first Gaussians
 * while the last ones are still being reduced (trase_amd/dp.py; the reference is single-process, train.py:303):
 *   _compose    the compositing backward over every sub-tile -- one gradient row per (sub-tile, Gaussian) pair in the
 *               workspace, no per-Gaussian output yet (only dL_dgaussian_features is zero-filled when dL_dfeats is NULL);
 *   _gaussians  the per-Gaussian tail for the ids [p_begin, p_end): sums their pair rows and writes THEIR entries of every
 *               gradient tensor in `g` (same pointers as for the whole call: the tensors' bases).  p_begin must be a
 *               multiple of 64, p_end a multiple of 64 or P.
 * _compose followed by _gaussians over a partition of [0, P) equals trase_rast_backward_raw bit for bit. */
int trase_rast_backward_raw_compose(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                                    const TraseRastWorkspace* ws, const TraseRastRawGrads* g, trase_stream_t stream);
/* Sparse strip gradients (TRASE_VARIANT_SPARSE_STRIP_GRADS): sets the rows of every non-NULL gradient tensor in `g` to zero for
 * the Gaussians that had a pair in the strip of the forward whose `geom` and `pre` workspaces are given -- i.e. exactly the rows
 * that forward's backward wrote.  Cost: what the strip holds, not what the scene does. */
int trase_rast_zero_live_rows(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastWorkspace* ws,
                              const TraseRastRawGrads* g, trase_stream_t stream);
int trase_rast_backward_raw_gaussians(const TraseRastSettings* s, const TraseRastRawInputs* raw, const TraseRastOutputs* out,
                                      const TraseRastWorkspace* ws, const TraseRastRawGrads* g, int32_t p_begin, int32_t p_end,
                                      trase_stream_t stream);

/* simple_knn._C.distCUDA2 (scene/gaussian_model.py:237): mean squared distance
 * to the 3 nearest neighbours.  ws_bytes from trase_knn_sizes. */
int trase_knn_sizes(int32_t N, size_t* ws_bytes);
int trase_knn_dist2(const float* points, int32_t N, float* out, void* ws, size_t ws_bytes, int32_t device,
                    trase_stream_t stream);

/* pytorch3d.ops.knn_points replacement (scene/gaussian_model.py:88-92 K=16 self-KNN for feature
 * smoothing; render.py:222, gui.py:1048, utils/loss_utils.py:141,192 cross-KNN): for every point of
 * p1 the K (<= 16) nearest points of p2, ascending; idx int64 (N1,K), squared distances (N1,K).
 * Workspace: trase_knn_sizes(N2). */
int trase_knn_points(const float* p1, int32_t N1, const float* p2, int32_t N2, int32_t K, int64_t* idx,
                     float* dists, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);

/* Deformation MLP (utils/time_utils.py:60-131 DeformNetwork.forward, reached through
 * scene/deform_model.py:34-35 DeformModel.step at train.py:202-204, render.py:195, gui.py:965).
 * Weights are the reference's nn.Linear parameters as they are (fp32, [out][in] row-major, device
 * pointers); they are re-packed to bf16 on every call.  trase_mlp_forward serves the no_grad call sites; the training pair is
 * declared further down. */
typedef struct TraseMlpWeights {
  int32_t D;             /* hidden layers (8) */
  int32_t W;             /* hidden width (256) */
  int32_t xyz_multires;  /* 10 -> 63 input channels */
  int32_t t_multires;    /* 10 -> 21 input channels; 6 with is_blender */
  int32_t is_blender;    /* 0: cat(PE(x), PE(t)), 84 inputs.  1 (D-NeRF, utils/time_utils.py:74-86): cat(PE(x),
                          * timenet(PE(t))), 93 inputs -- the caller evaluates the tiny timenet and passes its 30 outputs
                          * as `t` with t_stride 0 (train.py:190-202 feeds the same time to every row when is_blender) */
  int32_t is_6dof;       /* 1: the 3-vector head is a screw-axis branch (branch_w / branch_v); trase_amd/deform.py runs the network twice */
  int32_t variant;       /* reserved, must be 0 (the superseded kernel organisations it used to select are gone) */
  int32_t reserved;
  const float* weight[8];/* linear.{i}.weight: (256, 84|93) / (256, 256) / (256, 340|349) for the skip layer i = 5 */
  const float* bias[8];  /* linear.{i}.bias  : (256,) */
  const float* w_warp;     const float* b_warp;      /* gaussian_warp     (3,256), (3,) */
  const float* w_rotation; const float* b_rotation;  /* gaussian_rotation (4,256), (4,) */
  const float* w_scaling;  const float* b_scaling;   /* gaussian_scaling  (3,256), (3,) */
} TraseMlpWeights;

int trase_mlp_sizes(size_t* ws_bytes);
/* x (N,3); t: one float per row at stride t_stride floats (0 = the same scalar for every row, as the
 * reference's expand() produces at train.py:196; is_blender: the 30 timenet outputs, t_stride 0); outputs d_xyz (N,3), d_rotation (N,4), d_scaling (N,3). */
int trase_mlp_forward(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                      float* d_xyz, float* d_rotation, float* d_scaling, void* ws, size_t ws_bytes, int32_t device,
                      trase_stream_t stream);

/* Training pair (train.py:202-204 runs the MLP with gradients in the GAUSSIAN state; loss.backward() at
 * train.py:299 reaches utils/time_utils.py:106-131 through autograd).
 *   trase_mlp_forward_train: the same fused forward; additionally fills the opaque `saved` buffer (bf16
 *     activations and encoding as transposed images, the ReLU gates as bits -- 4.2 KB per Gaussian).
 *   trase_mlp_backward: from the cotangents of the three outputs (any may be NULL = zero) and `saved`, the
 *     gradients of every parameter (fp32, overwritten; NULL entries are skipped).  Fused data chain on the
 *     matrix cores, then one split-N MFMA GEMM per layer input and a partial-sum reduction.  x and t are
 *     detached at the reference's call site, so nothing is propagated into the encoding. */
typedef struct TraseMlpGrads {
  float* weight[8];        /* same shapes as TraseMlpWeights */
  float* bias[8];
  float* w_warp;     float* b_warp;
  float* w_rotation; float* b_rotation;
  float* w_scaling;  float* b_scaling;
} TraseMlpGrads;

int trase_mlp_train_sizes(int32_t N, size_t* fwd_ws_bytes, size_t* saved_bytes, size_t* bwd_ws_bytes);
int trase_mlp_forward_train(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                            float* d_xyz, float* d_rotation, float* d_scaling, void* saved, size_t saved_bytes,
                            void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_mlp_backward(const TraseMlpWeights* w, int32_t N, const float* dL_dd_xyz, const float* dL_dd_rotation,
                       const float* dL_dd_scaling, const void* saved, size_t saved_bytes, const TraseMlpGrads* grads,
                       void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
/* The same pair with a ROW ORDER: batch row r evaluates Gaussian row_order[r] (a permutation of 0..N-1, int32, device;
 * NULL = identity = the two entry points above).  Inputs are gathered and outputs / cotangents addressed through it, so the
 * caller sees the reference's row order on both sides; the saved state is in batch order and the same row_order must be
 * handed to the backward.  Why: a Gaussian the view culled (radii == 0, gaussian_renderer/__init__.py:152) sends back an
 * exactly-zero cotangent and contributes nothing to any parameter gradient of utils/time_utils.py:106-131; the backward skips
 * every 32-row tile whose cotangents are all zero (with or without a row order), and culling is spatially coherent, so an
 * order that follows a space-filling curve turns the culled quarter of an orbit view into whole dead tiles. */
int trase_mlp_forward_train_rows(const TraseMlpWeights* w, const float* x, const float* t, int32_t t_stride, int32_t N,
                                 const int32_t* row_order, float* d_xyz, float* d_rotation, float* d_scaling, void* saved,
                                 size_t saved_bytes, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_mlp_backward_rows(const TraseMlpWeights* w, int32_t N, const int32_t* row_order, const float* dL_dd_xyz,
                            const float* dL_dd_rotation, const float* dL_dd_scaling, const void* saved, size_t saved_bytes,
                            const TraseMlpGrads* grads, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
/* Introspection: number of live 32-row tiles the backward that last used `bwd_ws` (N rows) worked on, copied
 * (stream-ordered, device to device) into one int32 of the caller. */
int trase_mlp_live_tiles(const void* bwd_ws, size_t ws_bytes, int32_t N, int32_t* n_live_device, trase_stream_t stream);

/* ---- KNN feature smoothing of the FEATURE state ------------------------------------
 * GaussianModel.get_smoothed_gaussian_features (scene/gaussian_model.py:79-104; gaussian_renderer/__init__.py:118,
 * train.py:274-275):  out[i] = mean_s normalize(features[knn_idx[i][select[s]]]),  F.normalize eps = 1e-12.
 *   features (P,32), knn_idx (P,K) int64 (pytorch3d.ops.knn_points(...).idx), select: S distinct slots of 0..K-1
 *   (the reference's torch.randperm(K)[:int(K*dropout)]), out (P,32).  inv_norm (P floats) is written by the
 *   forward and read by the backward.
 * The backward gathers over the REVERSE adjacency instead of scattering with atomics: rev_src holds the flat
 * positions i*K+k of knn_idx sorted (stably) by the Gaussian they point at, rev_ptr (P+1) the CSR offsets; the host
 * builds both once per KNN map.  select_mask has bit k set for every selected slot. */
int trase_smooth_forward(const float* features, const int64_t* knn_idx, int32_t P, int32_t F, int32_t K,
                         const int32_t* select, int32_t S, float* inv_norm, float* out, int32_t device,
                         trase_stream_t stream);
int trase_smooth_backward(const float* features, const float* inv_norm, int32_t P, int32_t F, int32_t K,
                          uint32_t select_mask, int32_t S, const int32_t* rev_ptr, const int32_t* rev_src,
                          const float* dL_dout, float* dL_dfeatures, int32_t device, trase_stream_t stream);

/* ---- photometric loss heads -----------------------------------------------------------
 * l1_loss (utils/loss_utils.py:30-31) and ssim (utils/loss_utils.py:56-86: 11x11 Gaussian window, sigma 1.5, zero
 * padding, C1 = 0.01^2, C2 = 0.03^2, mean over the map), combined at train.py:235-238.
 *   forward : img, gt (C,H,W) -> out2 = {mean |img - gt|, mean SSIM map} (device floats); keeps the SSIM partial
 *             derivatives in `ws` (trase_loss_sizes bytes; the caller holds it until the backward).
 *   backward: g2 = {dL/dl1, dL/dssim} (device floats, so no host sync) -> dL/dimg (C,H,W), overwritten. */
int trase_loss_sizes(int32_t C, int32_t H, int32_t W, size_t* ws_bytes);
int trase_loss_l1_ssim_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, float* out2, void* ws,
                               size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_loss_l1_ssim_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, const float* g2,
                                const void* ws, size_t ws_bytes, float* dL_dimg, int32_t device, trase_stream_t stream);
/* The combination train.py:235-238 forms with scalar tensor arithmetic, `(1 - lambda_dssim) * Ll1 + lambda_dssim * (1 - ssim(image,
 * gt))`, inside the same two launches: out3 = {l1, ssim, loss} (device floats); the backward takes ONE cotangent g = dL/dloss (a
 * device float) and scales it by (1 - lambda) and -lambda itself -- the ~10 scalar kernels autograd launches for the composition
 * (and their host time) disappear.  lambda_dssim in [0, 1], a DOUBLE as in the reference's Python: the two weights are (float)(1.0 -
 * lambda) and (float)lambda, what torch's tensor-times-Python-scalar arithmetic multiplies with (1.0f - (float)lambda is one ulp off
 * for lambda = 0.35). */
int trase_loss_photometric_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, double lambda_dssim,
                                   float* out3, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_loss_photometric_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W, double lambda_dssim,
                                    const float* g, const void* ws, size_t ws_bytes, float* dL_dimg, int32_t device,
                                    trase_stream_t stream);

/* ---- contrastive pixel-pair losses (second half) -------------------------------------------
 * positive_pixel_pair_loss / negative_pixel_pair_loss (utils/loss_utils.py:396-406), selected by
 * opt.contrastive_mode (arguments/__init__.py:131; default 'soft') and called at train.py:290-291 on the S x S matrices
 * C (0/1 pixel-mask correspondence), C_F (feature similarity) and weights (may be NULL).
 * kind = TRASE_PAIR_POSITIVE|NEGATIVE + TRASE_PAIR_SOFT|ALL|HARD:
 *   soft: utils/loss_utils.py:304-349   all: :275-302 (threshold unused)   hard: :351-394.
 * out2 = {loss, N} (device floats; N = number of candidate pairs, for 'hard' the selection size); `ws`
 * (trase_contrastive_sizes bytes) keeps the column flags for the backward, which writes the dense dL/dC_F (S x S) for
 * the upstream gradient g (device scalar).  Nothing synchronises (the reference calls torch.nonzero per loss).  An
 * empty candidate set gives loss 0 with a zero gradient (the reference: python 0.0 / tensor(0.) for soft / hard, and
 * 0/0 = nan for 'all'). */
#define TRASE_PAIR_POSITIVE 0
#define TRASE_PAIR_NEGATIVE 1
#define TRASE_PAIR_SOFT 0
#define TRASE_PAIR_ALL 2
#define TRASE_PAIR_HARD 4
int trase_contrastive_sizes(int32_t S, size_t* ws_bytes);
int trase_contrastive_forward(const float* C, const float* C_F, const float* weights, int32_t S, float threshold,
                              int32_t kind, float* out2, void* ws, size_t ws_bytes, int32_t device,
                              trase_stream_t stream);
int trase_contrastive_backward(const float* C, const float* C_F, const float* weights, int32_t S, float threshold,
                               int32_t kind, const float* out2, const float* g, const void* ws, size_t ws_bytes,
                               float* dL_dC_F, int32_t device, trase_stream_t stream);

/* ---- multi-tensor Adam (first half) --------------------------------------------------------
 * One launch steps up to 16 parameter tensors with per-tensor learning rate and step count, in place
 * (param, exp_avg, exp_avg_sq), with torch.optim.Adam's arithmetic (no weight decay, no amsgrad):
 * scene/gaussian_model.py:253-300 builds Adam(l, lr=0.0, eps=1e-15) over per-parameter groups; train.py:376-389
 * steps it.  The tables (pointers to device tensors, sizes, learning rates, steps) are HOST arrays. */
/* ---- nearest-neighbour feature matching style loss (utils/loss_utils.py:223-228; train_style_transfer_nnfm.py:201-203) ----
 * loss = mean_i min_j (1 - cos(feat1[:, i], feats2[:, j])) for feat1 (C, N1), feats2 (C, N2) fp32, C a multiple of 64 up
 * to 512 (VGG conv4_1), N2 <= 2 097 152.  The N1 x N2 cosine matrix is never formed: a bf16 MFMA GEMM with a running
 * two-entry row maximum short-lists two neighbours per row, fp32 cosines of the original data decide between them (ties ->
 * lowest column).  The backward differentiates through the arg-min w.r.t. feat1 only (the style reference carries no
 * graph).  `ws` from the forward is handed back to the backward unchanged. */
int trase_nnfm_sizes(int32_t C, int32_t N1, int32_t N2, size_t* ws_bytes);
int trase_nnfm_forward(const float* feat1, const float* feats2, int32_t C, int32_t N1, int32_t N2, float* loss, void* ws,
                       size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_nnfm_backward(const float* feat1, const float* feats2, int32_t C, int32_t N1, int32_t N2, const float* g_loss,
                        const void* ws, size_t ws_bytes, float* dL_dfeat1, int32_t device, trase_stream_t stream);

int trase_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, const float* lr, const int64_t* step, double beta1,
                    double beta2, float eps, int32_t device, trase_stream_t stream);
/* The same step behind a DEVICE-side guard: `guard` is the geom workspace of the forward whose gradients the step consumes
 * (its 64-word header holds the overflow flag and the binning guards).  When that forward overflowed its pair buffer the
 * kernel returns without touching parameters or moments -- the reference skips optimizer.step() on a bad iteration
 * (train.py:298-301, :378) after a host-side check; the sync-free policy cannot afford that check, so the decision is
 * taken where the flag lives.  guard == NULL: no guard. */
int trase_adam_step_guarded(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, const float* lr, const int64_t* step, double beta1,
                            double beta2, float eps, const void* guard, int32_t device, trase_stream_t stream);

/* Per-kernel timing with HIP events on the caller's stream (used by bench.py's
 * roofline leg).  enable=1 starts recording, the report call synchronises the
 * events and returns averaged milliseconds per kernel name. */
int trase_prof_enable(int enable);
int trase_prof_report(char* buf, size_t buf_bytes); /* JSON object {"kernel": {"ms":..,"n":..}, ...} */

/* On-device self test of the wave64 primitives the kernels rely on
 * (DPP reductions, ballots, MFMA fragment layouts).  Returns 0 when all pass. */
int trase_selftest(int32_t device, trase_stream_t stream, char* msg, size_t msg_bytes);

const char* trase_last_error(void);
const char* trase_version(void);

/* ---- densification bookkeeping + densify / prune compaction (second half) -------------------
 * trase_densify_stats: the per-iteration statistics of train.py:362-365 and scene/gaussian_model.py:637-639 in one
 * launch, with visibility_filter = radii > 0 (gaussian_renderer/__init__.py:137):
 *   max_radii2D = max(max_radii2D, radii);  xyz_gradient_accum += |viewspace_grad[:, :2]|;  denom += 1   on visible rows.
 * viewspace_grad is the [P][3] gradient of the screen-space points (means2D.grad).
 *
 * trase_densify_plan + trase_densify_apply replace GaussianModel.densify_and_prune (scene/gaussian_model.py:617-635:
 * densify_and_clone :594-615, densify_and_split :563-592, prune_points :491-509, optimizer surgery :472-489, :511-534).
 * plan: decides clone / split / prune per row from the accumulated statistics and the RAW scaling [P][3] and opacity
 *   [P] parameters, and builds the row map of the final model in `ws` (trase_densify_sizes bytes).
 *   dense_extent = percent_dense * scene_extent; big_extent = 0.1 * scene_extent, applied only when use_screen_size != 0
 *   (the reference's max_screen_size argument; its max_radii2D test is dead code there, see densify.hip).
 *   counts (5 device ints) = {kept originals, kept clones, kept split children per sample, split-selected rows M,
 *   clone-selected rows}; the new row count is counts[0] + counts[1] + 2 * counts[2].  The caller reads them (the one
 *   synchronisation: it has to allocate the new tensors), draws 2M x 3 standard normals and calls
 * apply: gathers `count` (<= 32 per call) row-major tensors of 4-byte elements through the map, dst[r] = src[map[r]],
 *   writing zeros into rows that are not kept originals where zero_new[i] != 0 (the Adam moments), then -- when
 *   normal_samples != NULL -- rewrites the children rows of new_xyz / new_scaling:
 *   xyz' = R(rotation) (z * exp(scaling)) + xyz,  scaling' = log(exp(scaling) / 1.6).  With more than 32 tensors pass
 *   normal_samples on the last call only (after xyz and scaling were gathered); it may be NULL only when M == 0. */
int trase_densify_stats(const float* viewspace_grad, const int32_t* radii, float* xyz_gradient_accum, float* denom,
                        float* max_radii2D, int32_t P, int32_t device, trase_stream_t stream);
/* guard: as for trase_adam_step_guarded -- the statistics of an overflowed view are not taken */
int trase_densify_stats_guarded(const float* viewspace_grad, const int32_t* radii, float* xyz_gradient_accum, float* denom,
                                float* max_radii2D, int32_t P, const void* guard, int32_t device, trase_stream_t stream);
int trase_densify_sizes(int32_t P, size_t* ws_bytes);
int trase_densify_plan(const float* xyz_gradient_accum, const float* denom, const float* scaling, const float* opacity,
                       int32_t P, float grad_threshold, float dense_extent, float min_opacity, int32_t use_screen_size,
                       float big_extent, int32_t* counts, void* ws, size_t ws_bytes, int32_t device,
                       trase_stream_t stream);
int trase_densify_apply(int32_t count, const void* const* src, void* const* dst, const int32_t* row_elems,
                        const int32_t* zero_new, int32_t P, int32_t new_P, const float* xyz, const float* scaling,
                        const float* rotation, const float* normal_samples, float* new_xyz, float* new_scaling,
                        const void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);

/* ---- FEATURE-state loss head without S x S matrices (train.py:251-296) ------------------------
 * trase_mask_stats: one pass over the N bool masks [N][HW] (1 byte each): cover_count[p] = number of masks covering pixel
 *   p (the sampler's non_mask_region is cover_count == 0, utils/feature_utils.py:23) and mask_size[n] = sam_masks[n].sum()
 *   (utils/feature_utils.py:30).
 * trase_pairhead_forward: for the S sampled pixels pix[] (flat indices into HW, ascending = boolean-index order) evaluates
 *   what train.py:272-296 computes through get_pixel_mask_correspondence_matrix (utils/feature_utils.py:40-49),
 *   get_features_correspondence_matrix (:51-57), get_pixel_weights (:28-38), positive_/negative_pixel_pair_loss[mode]
 *   (utils/loss_utils.py:275-406; mode 0 soft, 1 all, 2 hard) and the two mean similarities (train.py:295-296), from
 *   per-pixel factors only: feats is the [F = 32][HW] feature image at mask resolution, sampled_mask the N 0/1 bytes of the
 *   sampled masks (n_sampled_masks = an upper bound of their number, at most 256).
 *   out8 = {loss_pos, N_pos, loss_neg, N_neg, pos_similarity, neg_similarity, S, sampled masks} (device floats).
 *   use_weights = 0 evaluates the losses with weights = None.
 * trase_pairhead_backward: dL/dfeats [F][HW] (zero-filled here, then the S sampled columns) for the upstream gradients
 *   g2 = {dL/dloss_pos, dL/dloss_neg} (device floats), from the forward's workspace.  accumulate != 0 adds the S columns
 *   into an image that already holds a gradient (the regulariser's, trase_featnorm_backward) instead of zero-filling.
 * trase_featnorm_*: the regulariser (1 - mean_p |feats[:, p]|_2)^2 of train.py:281-282; out2 = {value, mean norm}. */
int trase_mask_stats(const uint8_t* sam_masks, int32_t N, int64_t HW, int32_t* cover_count, uint32_t* mask_size, int32_t device,
                     trase_stream_t stream);
int trase_pairhead_sizes(int32_t S, size_t* ws_bytes);
int trase_pairhead_forward(const float* feats, int32_t F, int64_t HW, const uint8_t* sam_masks, int32_t N,
                           const uint8_t* sampled_mask, int32_t n_sampled_masks, const uint32_t* mask_size, const int32_t* pix,
                           int32_t S, int32_t mode, float positive_th, float negative_th, int32_t use_weights, float* out8,
                           void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_pairhead_backward(int32_t F, int64_t HW, const int32_t* pix, int32_t S, int32_t mode, float positive_th,
                            float negative_th, int32_t use_weights, const float* out8, const float* g2, const void* ws,
                            size_t ws_bytes, int32_t accumulate, float* dL_dfeats, int32_t device, trase_stream_t stream);
/* The same head WITHOUT the host knowing the number of sampled pixels (round 5: the reference synchronises at every boolean index of
 * train.py:262-296; the round-4 head still did once, for S).  trase_compact_pixels turns the (H W) byte mask `sampled_pixel` into the
 * ascending pixel indices `pix[0 .. count[0])` -- the order of a boolean index -- entirely on the device: count[0] = min(number of set
 * bytes, cap), count[1] = the number itself (ws: trase_compact_pixels_sizes(HW) bytes).  The _n entry points take S = the CAPACITY the
 * workspace and the launch grids are sized for and S_dev = &count[0] (NULL: S is the count, as in the entry points above); rows at or
 * behind the device count do not exist for any kernel; S_dev == 0 gives zero losses, NaN similarities and a zero gradient. */
int trase_compact_pixels_sizes(int64_t HW, size_t* ws_bytes);
int trase_compact_pixels(const uint8_t* flags, int64_t HW, int32_t* pix, int32_t cap, int32_t* count2, void* ws, size_t ws_bytes,
                         int32_t device, trase_stream_t stream);
int trase_pairhead_forward_n(const float* feats, int32_t F, int64_t HW, const uint8_t* sam_masks, int32_t N,
                             const uint8_t* sampled_mask, int32_t n_sampled_masks, const uint32_t* mask_size, const int32_t* pix,
                             int32_t S, const int32_t* S_dev, int32_t mode, float positive_th, float negative_th, int32_t use_weights,
                             float* out8, void* ws, size_t ws_bytes, int32_t device, trase_stream_t stream);
int trase_pairhead_backward_n(int32_t F, int64_t HW, const int32_t* pix, int32_t S, const int32_t* S_dev, int32_t mode, float positive_th,
                              float negative_th, int32_t use_weights, const float* out8, const float* g2, const void* ws,
                              size_t ws_bytes, int32_t accumulate, float* dL_dfeats, int32_t device, trase_stream_t stream);
int trase_featnorm_sizes(int64_t HW, size_t* ws_bytes);
int trase_featnorm_forward(const float* feats, int32_t F, int64_t HW, float* out2, void* ws, size_t ws_bytes, int32_t device,
                           trase_stream_t stream);
int trase_featnorm_backward(const float* feats, int32_t F, int64_t HW, const float* out2, const float* g, float* dL_dfeats,
                            int32_t device, trase_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRASE_RAST_H */
