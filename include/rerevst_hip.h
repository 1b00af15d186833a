/*
 * rerevst_hip.h — C ABI of librerevst_hip.so, the MI355X (gfx950) per-frame stylization
 * path of ReReVST.
 *
 * Every entry point replaces one method of the reference's Python drop-in boundary,
 * class `Stylization` in test/framework.py:56-118 (multi-style twin:
 * "Multi-style Interpolation/stylization.py":42-100).  Plain pointers and sizes only; all
 * image buffers are caller-owned.  A handle owns its device weights, workspace, saved
 * state and one HIP stream; it is NOT thread-safe (the reference model is stateful and
 * non-reentrant as well).  Multi-GPU = one handle per process per GPU.
 *
 * Return value: 0 on success, negative RRV_E_* otherwise; rrv_last_error() gives the text.
 */
#ifndef REREVST_HIP_H
#define REREVST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrv_ctx* rrv_handle;

enum {
    RRV_OK = 0,
    RRV_E_ARG = -1,        /* bad argument / shape */
    RRV_E_HIP = -2,        /* HIP runtime error */
    RRV_E_WEIGHTS = -3,    /* unknown key, wrong shape, or weights incomplete */
    RRV_E_STATE = -4,      /* transfer before compute()/set_state, compute with no frames, ... */
    RRV_E_NOMEM = -5,      /* out of device (or page-locked host) memory */
    RRV_E_DEBUG = -6,      /* bounds-checked debug mode found a store outside a tensor's valid region */
    RRV_E_COMM = -7        /* RCCL: library not found, or a communicator / collective call failed */
};

/* Floats in the per-style shared-state blob: 11 norm layers x {mean,rstd,lo,hi}[C]
 * (2368 channels), 6 dynamic filters [32][32], style (mean,std) for relu1_1..relu4_1
 * (test/style_network_global.py:37-40,148,171,330).  Layout documented in DESIGN.md. */
#define RRV_STATE_FLOATS 17536
#define RRV_MAX_STYLES 8
#define RRV_MAX_SLOTS 4

/* Stylization.__init__ (test/framework.py:57-78): picks the device and builds the model.
 * `device` is the HIP device ordinal. */
int rrv_create(int device, rrv_handle* out);
int rrv_destroy(rrv_handle h);
const char* rrv_last_error(rrv_handle h);

/* model.load_state_dict(torch.load(checkpoint)) (test/framework.py:75): one call per
 * state_dict key used by the inference path (Encoder.*, EncoderStyle.*, Decoder.*; the
 * reference deletes Vgg19.* itself, style_network_global.py:467-469).  `data` is host
 * fp32, OIHW for convolutions / [out,in] for FC, exactly as stored in the checkpoint;
 * repacking to the kernel-native layout happens inside.  rrv_finalize_weights() checks
 * the full key set is present (strict load) and uploads. */
int rrv_load_weight(rrv_handle h, const char* key, const float* data, const int64_t* shape, int ndim);
int rrv_finalize_weights(rrv_handle h);

/* Stylization.prepare_style (test/framework.py:99-104; multi-style stylization.py:71-79).
 * style: uint8 BGR HWC [Hs][Ws][3].  style_id in [0, RRV_MAX_STYLES). */
int rrv_prepare_style(rrv_handle h, const uint8_t* style_bgr, int Hs, int Ws, int style_id);

/* Stylization.clean (test/framework.py:93-95). */
int rrv_clean(rrv_handle h);

/* Stylization.add (test/framework.py:82-86): encode one sampled frame (uint8 BGR HWC,
 * UNPADDED as the driver passes it, generate_real_video.py:139-143) and keep its
 * relu4_1 feature.  All frames added between clean() and compute() must share H, W. */
int rrv_add(rrv_handle h, const uint8_t* frame_bgr, int H, int W);

/* Stylization.compute (test/framework.py:88-91): batched decoder pass over the added
 * frames that records the saved statistics / dynamic filters for every prepared style. */
int rrv_compute(rrv_handle h);

/* Memory policy of rrv_compute.  Decoder.compute as written keeps every sampled frame's decoder activations resident
 * (O(B * 64 * H * W) floats at the last level; the reference authors' own long-sequence sketch streams them through a
 * disk cache, test/style_network.py:597-624).  When that workspace would exceed `bytes` (default 64 GiB) rrv_compute
 * STREAMS instead: one sync point (normalisation layer / filter prediction) at a time, groups of G frames re-run the
 * decoder prefix from their relu4_1 features and contribute partial statistics (sum, centred square sum, min, max;
 * pairwise merge in fp64) — workspace = one group, independent of B; the state blob equals the resident one to
 * rounding.  rrv_last_compute_info reports what the last rrv_compute did. */
int rrv_set_workspace_cap(rrv_handle h, size_t bytes);
int rrv_last_compute_info(rrv_handle h, int* groups, int* group_size, size_t* workspace_bytes);

/* The saved state of one style as a flat blob (what an RCCL broadcast ships to the other
 * ranks, and what the golden fixtures compare).  n must be RRV_STATE_FLOATS. */
int rrv_get_state(rrv_handle h, float* out, int n, int style_id);
int rrv_set_state(rrv_handle h, const float* in, int n, int style_id);

/* Multi-GPU (one process and one handle per GPU; the reference has no distributed code — SURVEY 8(e)): frames are
 * independent once the saved state exists, so the only collective is ONE ncclBroadcast of the blob per style and video,
 * from the rank that ran prepare_style / add / compute.  rrv_broadcast_state issues it on the handle's stream over an
 * ordinary ncclComm_t (pass the application's own, e.g. MPI- or torch-built, as void*); on the other ranks the style
 * then is exactly as after rrv_set_state.  librccl.so is opened on first use (RRV_RCCL_PATH overrides the search).
 * The three rrv_comm_* helpers build a communicator for callers without RCCL bindings (ctypes, cgo, JNI): rank 0 calls
 * rrv_comm_unique_id, ships the 128 bytes to the others by any means, every rank calls rrv_comm_init_rank. */
int rrv_comm_unique_id(char id[128]);
int rrv_comm_init_rank(rrv_handle h, const char id[128], int nranks, int rank, void** comm);
int rrv_comm_destroy(void* comm);
int rrv_broadcast_state(rrv_handle h, void* comm, int root, int my_rank, int style_id);

/* Stylization.transfer (test/framework.py:106-118): uint8 BGR HWC [H][W][3] in host
 * memory -> float32 BGR HWC [H][W][3] in 0..255 in host memory (includes the H2D / D2H crossings of
 * framework.py:109 `.to(device)` and :40 `.cpu()`).  Any H, W >= 8: as in the reference the three max pools floor the
 * size, so out_bgr is [Ho][Wo][3] with Ho = 8*(H/8), Wo = 8*(W/8) (= [H][W][3] for the multiples of 64 the reference
 * driver produces); (H+2)*(W+2)*64 < 2^31 (about 33 Mpixel per frame; RRV_E_ARG beyond).  The same holds for every
 * other transfer entry (batch, device, blend, features, frame mode).  Page-locked caller buffers (rrv_host_alloc / rrv_host_register)
 * are DMA'd directly; pageable ones are staged through the library's own pinned buffers. */
int rrv_transfer(rrv_handle h, const uint8_t* frame_bgr, int H, int W, float* out_bgr);

/* Look-ahead form of rrv_transfer for a one-frame-per-call driver loop (test/generate_real_video.py:152-171 calls
 * framework.transfer once per frame): rrv_transfer_async queues the frame and returns a ticket at once;
 * rrv_transfer_wait(ticket) blocks until `out_bgr` of that call is filled.  Up to FOUR tickets may be open (a fifth
 * submission first completes the oldest): each runs on its own stream and workspace with a quarter of the CUs per
 * launch, so four frames run side by side instead of queueing behind each other's partially filled last round of
 * workgroups.  The frame is copied to the device on the ticket's own stream (one H2D copy; no separate copy streams)
 * and the last kernel writes the result to page-locked host memory directly (the caller's buffer if it is page-locked,
 * the library's staging otherwise).  Keeping three frames submitted ahead of the one collected gives 551 frames/s at
 * 512x512 and 1418 at 256x256 against 431 / 868 for rrv_transfer.
 * `frame_bgr` may be reused as soon as the call returns (a pageable frame has been copied to staging by then; for a
 * page-locked one the call waits for its H2D copy); `out_bgr` must stay valid until its ticket is waited for.
 * Bit-identical to rrv_transfer. */
int rrv_transfer_async(rrv_handle h, const uint8_t* frame_bgr, int H, int W, float* out_bgr, long* ticket);
int rrv_transfer_wait(rrv_handle h, long ticket);

/* Same computation on device-resident buffers (HBM in, HBM out), asynchronous on the
 * handle's own (non-blocking) streams; rrv_sync() waits.
 * ORDERING: the library's streams are NOT ordered against any stream of the caller.  Either (a) make sure the input
 * is complete before the call (e.g. torch.cuda.synchronize()) and call rrv_sync() before reading the output, or
 * (b) register the producing / consuming stream once with rrv_set_caller_stream(): every *_device entry then waits
 * for the work queued on that stream so far and makes that stream wait for its own output (event based, no host
 * sync), i.e. the call behaves as if it had been enqueued on the caller's stream. */
int rrv_transfer_device(rrv_handle h, const void* d_frame_bgr_u8, int H, int W, void* d_out_bgr_f32);

/* B frames per launch ([B][H][W][3] in, [B][H][W][3] out, both in HBM): frames are independent
 * once the state exists (test/style_network_global.py:499-501), so batching only widens every
 * kernel's grid -- it removes the workgroup-quantisation loss of the small layers.
 * rrv_transfer_device(h, in, H, W, out) == rrv_transfer_batch_device(h, in, 1, H, W, out). */
int rrv_transfer_batch_device(rrv_handle h, const void* d_frames_bgr_u8, int B, int H, int W, void* d_out_bgr_f32);

/* Multi-style transfer (stylization.py:94-100 + style_network.py:432-460): every saved
 * quantity is replaced by sum_s weight[s]*q_s before the same forward.  Device buffers. */
int rrv_transfer_blend_device(rrv_handle h, const void* d_frame_bgr_u8, int H, int W,
                              const float* style_weight, int n_styles, void* d_out_bgr_f32);

/* Host-buffer forms of the two entries above (H2D copy, same device path, D2H copy). */
int rrv_transfer_batch(rrv_handle h, const uint8_t* frames_bgr, int B, int H, int W, float* out_bgr);
int rrv_transfer_blend(rrv_handle h, const uint8_t* frame_bgr, int H, int W, const float* style_weight, int n_styles,
                       float* out_bgr);

/* The reference driver's ReshapeTool + crop on the device (test/generate_real_video.py:61-83 process: reflect-pad
 * 64 px on every side and up to a multiple of 64, cv2.BORDER_REFLECT; :167 crop [64:64+H, 64:64+W]): UNPADDED
 * [B][H][W][3] uint8 frames in, [B][H][W][3] float32 stylized frames out.  The padded frame never exists: the first
 * kernel reads the source through the reflection, the last one writes only the crop window.  Bit-identical to
 * pad -> rrv_transfer_batch -> crop for a fixed kernel choice (rrv_set_f43 mode 0 / 2; in the default mode 1 the choice
 * follows the launch geometry, and the crop window is one).  _device: HBM buffers, asynchronous; the host form pipelines sub-batches. */
int rrv_transfer_frames_device(rrv_handle h, const void* d_frames_bgr_u8, int B, int H, int W, void* d_out_bgr_f32);
int rrv_transfer_frames(rrv_handle h, const uint8_t* frames_bgr, int B, int H, int W, float* out_bgr);

/* Multi-style feature API ("Multi-style Interpolation/stylization.py"): generate_content_features :87-92
 * (encode a frame once; the reference caches the tensor on disk, test.py:87-101 — here it stays in HBM and
 * an integer id is returned), add_patch :66-67 (sample a cached feature for the statistics pass; then
 * rrv_compute == compute_norm :81-83), transfer(feature, style_weight) :94-100 (decoder only, blended state).
 * rrv_release_features frees the cache. */
int rrv_generate_content_features(rrv_handle h, const uint8_t* frame_bgr, int H, int W, int* feature_id);
/* The caching pass of a run of frames (test.py:87-101 encodes every frame of the video once) in ONE call: frames_bgr[B][H][W][3],
 * feature_ids[B] out.  Sub-batches are pipelined inside (copy-in stream + two compute streams, several frames per encoder
 * launch, the per-frame path's kernel choice — rrv_set_f43), and the encoder's last layer stores straight into the cache
 * (one arena per call): no allocation, device copy or host wait per frame.  Frames beyond the cache cap are kept as pixels. */
int rrv_generate_content_features_batch(rrv_handle h, const uint8_t* frames_bgr, int B, int H, int W, int* feature_ids);
int rrv_add_patch(rrv_handle h, int feature_id);
int rrv_transfer_features(rrv_handle h, int feature_id, const float* style_weight, int n_styles, float* out_bgr);
/* n cached features with one weight vector each (style_weight[n][n_styles], out_bgr[n][H][W][3]) in one call — what the
 * reference driver's loop does frame by frame (test.py:127-131) — pipelined inside: consecutive frames alternate over
 * two (stream, workspace, blended-state) sets and the D2H copy of one frame overlaps the next frame's kernels. */
int rrv_transfer_features_batch(rrv_handle h, const int* feature_ids, const float* style_weight, int n, int n_styles, float* out_bgr);
int rrv_release_features(rrv_handle h);
/* Frames per launch sequence of rrv_transfer_features_batch: 1..16, or 0 (default) = by the frame size, ~6.6 Mpixel per
 * launch as in the single-style host entries (4 at 1152 x 1152, 16 at 640 x 640 and below).  With more than one, every
 * image of a launch carries its own blended state set (per-image normalisation parameters and folded KernelFilter
 * weights), which widens every layer's grid and lets the kernel choice (rrv_set_f43) count the group's frames; with a
 * fixed kernel mode the results are bit-identical for every setting.  Measured with four styles in the default mode
 * (round 5, conv_f43_k with per-image parameters), frames/s for 1 / 2 / 4 per launch: 1152 x 1152 371 / 382 / 381,
 * 640 x 640 941 / 1070 / 1167, 384 x 384 1750 / 2163 / 2590. */
int rrv_set_multistyle_group(rrv_handle h, int frames);

/* Kernel choice for the same-resolution 3x3 layers of the per-frame path that have a Winograd F(4x4,3x3) pack (conv_f43_k):
 * encoder conv1_2 .. conv3_4 (test/style_network_global.py:271-281) and the three ResidualBlock.conv2 (:104,119-122).
 * mode 0 = always F(2x2,3x3); 1 (default, also RRV_F43=) = F(4x4,3x3) where the launch geometry lets it win: a rule on the
 * layer, the frames per launch, the frame size and the CUs the launch may use (the device's CU count under HSA_CU_MASK /
 * RRV_CUS, divided by the grid share of the look-ahead tickets) — from four 640 x 640 frames per launch on every packed layer,
 * from two 1152 x 1152 frames, from one where the items are long; 1.25-1.38x per layer, +16 % frames/s at 512 x 512;
 * 2 = always F(4x4,3x3).  The preparation pass (prepare_style / add / compute) and the frame mode always run F(2x2,3x3).
 * RRV_F43_LAYERS (bit 0..6 = encoder conv1_2 .. conv3_4, bit 7..9 = slice4 / slice3 / slice2 .conv2; default all ten) narrows
 * the set.  In mode 1 a state whose dynamic filters are far from the O(1) scale (Frobenius norm above 4 sqrt 32: ill-conditioned
 * in float32 whoever evaluates it) keeps the seven encoder layers on F(2x2,3x3).  F(4x4,3x3) makes 1.4x the rounding error of
 * F(2x2,3x3): worst pre-clamp error over 32 inputs x 4 weight sets <= 0.73 of the stated bound in the default mode
 * (profiles/r05_parity_margin.txt).  In mode 1 a frame's low-order bits therefore depend on how it was submitted; with a fixed
 * mode every single-style entry delivers the same bits for the same frame, and every mode is run-to-run deterministic.  The
 * one-frame feature cache entry (rrv_generate_content_features) always runs F(2x2,3x3); rrv_generate_content_features_batch
 * and the grouped per-image-state launches (rrv_set_multistyle_group > 1) follow the mode. */
int rrv_set_f43(rrv_handle h, int mode);
/* Capacity policy of the feature cache (default 64 GiB).  The reference spills every frame's feature to disk
 * (test.py:87-101, cache/%d.pt), so its video length is unbounded; here a cached feature costs 42 MB of HBM per
 * 1152x1152 frame.  Once the cache would exceed `bytes`, rrv_generate_content_features keeps the frame's uint8 pixels
 * instead (a tenth of the size) and every use of that feature re-runs the encoder first (rrv_transfer_features[_batch]
 * = encoder + blended decoder, rrv_add_patch encodes on the spot) — slower, never an out-of-memory error.
 * rrv_feature_cache_info: cached features, spilled ones, bytes held by the cached ones. */
int rrv_set_feature_cache_cap(rrv_handle h, size_t bytes);
int rrv_feature_cache_info(rrv_handle h, int* resident, int* spilled, size_t* bytes);

/* Stylization(checkpoint, cuda, use_Global=False).transfer (test/framework.py:69-72,106-118 with
 * test/style_network_frame.py): per-frame InstanceNorm statistics (:39-43) and per-frame filter
 * prediction (:53-62,97-105); needs only rrv_prepare_style (style 0).  Host buffers. */
int rrv_transfer_frame_mode(rrv_handle h, const uint8_t* frame_bgr, int H, int W, float* out_bgr);

/* Debug/parity taps: pre-clamp network output (normalised RGB, NHWC [H][W][3]) of the last
 * transfer, copied to host. */
int rrv_get_preclamp(rrv_handle h, float* out, int H, int W);
/* The same tap for image `b` of the last launch (a batched entry runs up to 64 frames per launch; for the host-buffer
 * entries the last launch is the last sub-batch of the call). */
int rrv_get_preclamp_image(rrv_handle h, float* out, int H, int W, int b);

int rrv_sync(rrv_handle h);

/* Consecutive rrv_transfer[_batch]_device calls alternate over n_slots (1..RRV_MAX_SLOTS, default 2) internal
 * (HIP stream, workspace) pairs, so n_slots independent batches are in flight and one batch's kernel
 * tails overlap the other's kernels.  Callers must give consecutive calls distinct output buffers
 * and call rrv_sync() before reading them.  Host-buffer entries and blend transfers are serialised. */
int rrv_set_pipeline(rrv_handle h, int n_slots);

/* The persistent grids of the transform-domain kernels normally take every CU (one workgroup each).  share = 2..4 gives
 * each launch 1/share of them, so that the launches of `share` streams run side by side instead of queueing behind each
 * other's last partial round — for callers that keep several SMALL batches in flight (one frame per call with
 * look-ahead).  Results do not depend on it. */
int rrv_set_grid_share(rrv_handle h, int share);

/* How the host-buffer entries (rrv_transfer, _batch, _frames, _async) cross PCIe.  0 (default): staged — H2D copy into
 * HBM, kernels, D2H copy, on dedicated copy streams.  1: zero copy — the first kernel reads the uint8 frames straight
 * from page-locked host memory and the last one stores the stylized frame there (caller buffers if they are
 * page-locked, the library's pinned staging otherwise): no copy kernels, no copy-stream events.  2 / 3: zero copy for the
 * input / the output only, the other direction staged.  Same results bit for bit. */
int rrv_set_host_io(rrv_handle h, int mode);

/* Debugging aid: activation tensor `index` (0..8 encoder c11 p1 c21 p2 c31 c32 c33 p3 c41, 9..22 decoder d f1 f2 f3 xs4 a4
 * o4 xs3 a3 o3 xs2 a2 o2 dpart) of workspace slot `slot` for frames of H x W, first image, ring layout [H'+2][W'+2][C],
 * copied to the host after a full synchronisation.  *floats receives its size (nothing is copied when cap is smaller). */
int rrv_debug_copy_tensor(rrv_handle h, int slot, int index, int H, int W, float* host, size_t cap, size_t* floats);

/* Stream-ordered use of the *_device entries from a caller that produces / consumes the buffers on its own HIP
 * stream (e.g. torch.cuda.current_stream().cuda_stream): see ORDERING above.  enable = 0 switches it off. */
int rrv_set_caller_stream(rrv_handle h, void* hip_stream, int enable);

/* Page-locked host memory for the host-buffer entries (no staging copy, true async DMA): allocate, or pin an
 * existing range in place.  Plain wrappers of hipHostMalloc / hipHostFree / hipHostRegister / hipHostUnregister so
 * that a caller without HIP bindings (ctypes, cgo, JNI) can use them. */
int rrv_host_alloc(size_t bytes, void** out);
int rrv_host_free(void* p);
int rrv_host_register(void* p, size_t bytes);
int rrv_host_unregister(void* p);

/* Bounds-checked debug mode (also RRV_DEBUG=<level> in the environment at rrv_create).  Every activation tensor is
 * allocated between two 64 KB guard bands filled with a canary; level 1 synchronises and checks after every API call,
 * level 2 after every kernel launch (an asynchronous fault or a violation is reported with the kernel's name):
 * guard bands intact, the one-pixel zero ring and the slack rows of every tensor still zero (the kernels rely on both
 * and must never store outside the valid pixels).  A violation returns RRV_E_DEBUG.  Changing the level drops the
 * workspaces (saved state and prepared styles are kept).  rrv_debug_selftest plants one ring store and one guard-band
 * store and returns RRV_OK only if the checker reports both. */
int rrv_set_debug(rrv_handle h, int level);
int rrv_debug_selftest(rrv_handle h);
/* Failure injection: the nth next device allocation of this handle (1 = the very next one) reports out-of-memory
 * (RRV_E_NOMEM, as a real hipErrorOutOfMemory does); 0 disarms.  Workspaces are built transactionally — complete or
 * released — so the call after a failed one starts over instead of finding a half-built plan. */
int rrv_debug_fail_alloc(rrv_handle h, int nth);

/* Per-launch timing with HIP events recorded on the handle's own stream.
 * rrv_profile_begin() clears the log and starts bracketing every kernel launch with
 * events; rrv_profile_end() syncs and freezes the log.  Entry i: kernel name (with "@CinxCout@HxW"
 * for convolutions), elapsed ms, ALGORITHMIC FLOPs (the reference's direct convolution:
 * 2*B*H*W*Cin*Cout*taps) and algorithmic HBM bytes of that launch, and the FLOPs the kernel actually
 * executes (fewer for the Winograd and upsample-folded kernels). */
int rrv_profile_begin(rrv_handle h);
int rrv_profile_end(rrv_handle h);
int rrv_profile_count(rrv_handle h);
int rrv_profile_entry(rrv_handle h, int i, const char** name, float* ms, double* flops, double* bytes,
                      double* flops_executed);

#ifdef __cplusplus
}
#endif
#endif /* REREVST_HIP_H */
