/*
 * libowlhip -- C ABI of the MI355X-native OWL-ViT train path (gfx950 / CDNA4).
 *
 * The reference (stevebottos/owl-vit-object-detection) has NO native layer: its hot path is Python
 * over stock aten kernels.  The drop-in boundary is therefore the Python
 * call surface (`model(image)`, `PushPullLoss(...)`, `HungarianMatcher(...)`); this header is the
 * thin C ABI underneath it.  Each entry point names the reference arithmetic it replaces
 * (`ref:` = path under the reference repo, `HF5:` = transformers/models/owlvit/modeling_owlvit.py).
 *
 * Conventions:
 *   - every function enqueues on the hipStream_t passed as `stream` and returns immediately;
 *     0 = OK, <0 = error (message via owl_last_error(), thread-local);
 *   - re-entrant: the library keeps NO process-global mutable state (kernel choices are per-call arguments; the tuning
 *     switches of tools/ exist only in an OWL_TUNING build, include/owl_hip_tuning.h);
 *   - ops that need device scratch take it from the caller (`*_workspace_bytes` / `*_workspace` queries size it);
 *   - the library never allocates or frees device memory; all pointers are device pointers to
 *     contiguous row-major buffers owned by the caller; outputs are pre-allocated;
 *   - bf16 buffers are passed as `void*` (raw 16-bit words), f32 as `float*`;
 *   - activations are laid out [B * Tp, width] with Tp = tokens padded to a multiple of 8 per
 *     image (class token first); allocations are padded to a multiple of 128 rows.
 *
 * The prototypes below are parsed by owl_vit_object_detection_amd/_lib.py to build the ctypes
 * signatures -- keep one prototype per statement, plain C types only.
 */
#ifndef OWL_HIP_H
#define OWL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---------------------------------------------------------------------------------- */
/* Bumped whenever a prototype, an argument's meaning or a caller-provided scratch layout changes (1 = round 1; 2 = round 2: per-call `tile` /
 * `variant` arguments, partial-sum scratch of the row reductions, 5D+4 box_final_bwd partials; 3 = round 3; 4 = round 4: `slow_tiles` statistic of the attention forward; 5 = round 5: owl_patch_embed_bf16's weight layout for patch sizes that are not 2^n (gathered, no im2row); the V^T attention form, attention variants 3-5, GEMM epilogues 5 / 6 and tiles 8 / 9 / 5 / 4 moved to OWL_TUNING builds); 6 = round 6: + owl_patch_embed_scratch_bytes, owl_normalize_u8, owl_allreduce_sum_f32, `phases` of owl_attention_bwd_bf16; GEMM epilogue 1 saves quick_gelu'(u) and epilogue 8 multiplies by it; patch sizes must be even.  owl_abi_version() returns the value
 * the library was BUILT with: a binding compares it with the header it was generated from and refuses a mismatch (_lib.load() does). */
#define OWL_ABI_VERSION 6
const char* owl_last_error(void);
int owl_abi_version(void);

/* ---- GEMM  C[M,N] = A[M,K] . W[N,K]^T with fused epilogue ---------------------------------------
 * replaces aten::addmm/mm under HF5:437-439,457 (q/k/v/out proj), HF5:472,474 (fc1/fc2),
 * HF5:994-997 (box head dense0/1), ref src/models.py:25 (class dense0) and their autograd forms.
 * epi: 0 bias->bf16 | 1 bias+quick_gelu->bf16 (aux, optional = bf16 quick_gelu'(pre-activation): ABI 6; ABI <= 5 saved the pre-activation) |
 *      2 bias+erf-gelu->bf16 (aux, optional = pre-activation) |
 *      3 resid+acc+bias->f32 | 4 alpha*acc(+bias)->f32 | 8 acc*aux->bf16 (aux = what epi 1 saved: the dX GEMM through quick-GELU is one multiply, ABI 6) |
 *      9 acc*gelu'(aux)->bf16 | 10 out f32 += acc | 11 split-K slab
 *      (5 atomicAdd f32 and 6 per-head transposed bf16: OWL_TUNING builds only -- the train path uses neither).
 * a_rows / w_rows clamp the tile loads; M, N guard the stores; K % 64 == 0.
 * tile: kernel choice, per call (no global state): 0 = automatic (large shapes: 256x256x64 tiles on the two-phase ping-pong schedule, 8 waves,
 *       gemm_pp2.hip -- for narrow outputs with the remainder round on half-height 128x256 tiles, gemm_pph.hip --; 128x128x64 otherwise); tests pin one
 *       kernel with 128 | 256 (the single-phase REFERENCE kernel every other one is held to, bit for bit) | 7 (two-phase ping-pong on the whole
 *       problem) | 6 (= 0, plus: a problem of at most 128 tiles of 256 x 256 -- half a round of the 256 CUs -- runs every tile as two half-height tiles;
 *       for callers that know nothing else is in flight: batch 1 / 2, one stream).  All give identical bits.  (8, 9, 5, 4: the round-1 four-phase ping-pong, the round-4 free-running and the four-wave experiments,
 *       OWL_TUNING builds only.)                                                                                                                   */
int owl_gemm_nt_bf16(void* stream, int epi, const void* A, int64_t lda, int64_t a_rows, const void* W, int64_t ldw, int64_t w_rows, const float* bias, void* out, int64_t ldo, const float* resid, void* aux, int64_t ld_aux, int64_t M, int64_t N, int64_t K, float alpha, int splits, int64_t Tp, int tile);
/* epi 11 = split-K partial slabs out[split][M][ldo] (f32, no atomics); reduce them with owl_slab_reduce */
int owl_gemm_effective_splits(int64_t K, int splits);
/* f32 slab scratch of an epi-11 call (M, N, K, splits): bytes = owl_gemm_effective_splits(K, splits) * M * ldo * 4 (`bytes`: HOST pointer) */
int owl_gemm_slab_workspace_bytes(int64_t M, int64_t ldo, int64_t K, int splits, int64_t* bytes);
int owl_slab_reduce(void* stream, const float* slabs, float* out, int64_t n, int64_t slab_stride, int nsplit, int accumulate);

/* ---- patch embedding (HF5:282-288 Conv2d k=s=patch, no bias; HF5:336-343 flatten + positions) ----
 * im2row-free for EVERY even patch size 8 <= ps <= 64: the A-operand loader gathers 16-byte runs of each patch row straight from the
 * bf16 image [B,3,S,S] into LDS.  x_out[b*Tp + 1 + p, :] = W_pe . vec(patch) + pos[1+p, :].
 *   ps = 2^n      : w_pe = the conv weight [D, 3*ps*ps] as it lies.
 *   other ps (14) : the K index pads a patch row to psp = 2^n >= ps positions; position `pos` of a row holds pixel
 *                   min(8*(pos/8), ps - 8) + pos%8 -- the last 16-byte chunk of a row OVERLAPS its predecessor instead of leaving the
 *                   patch -- and w_pe is [D, Kg], Kg = 3*ps*psp rounded up to 64, with the conv weight at each pixel's FIRST position
 *                   and zeros elsewhere (Python: weights.patch_weight_gather_layout).  (ABI 5; ABI <= 4 took an explicit im2row there.)
 * scratch: bf16 [rows128(B*P), Kg], needed only when a single-phase reference kernel (tile 256 / 128, or a problem too small for the
 *          ping-pong kernel) meets a patch size that is not 2^n (explicit im2row in the same K order: identical bits); else NULL.
 *          owl_patch_embed_scratch_bytes answers "how many bytes, for this problem and tile" (0 = pass NULL): the dispatcher's rule lives there only (ABI 6).
 * tile: 0 = automatic (two-phase ping-pong kernel for big problems), 7 / 256 / 128 pin a kernel (tests); same bits.       */
int owl_patch_embed_scratch_bytes(int64_t B, int64_t S, int64_t ps, int64_t D, int tile, int64_t* bytes);
int owl_patch_embed_bf16(void* stream, const void* image_bf16, const void* w_pe, const float* pos, float* x_out, void* scratch, int64_t B, int64_t S, int64_t ps, int64_t D, int64_t Tp, int tile);
/* class-token rows x[b*Tp, :] = class_embedding + pos[0, :]  (HF5:338-343)                        */
int owl_cls_rows(void* stream, float* x, const float* cls, const float* pos, int64_t B, int64_t Tp, int64_t D);

/* ---- LayerNorm (HF5:484-486, 721-723; eps 1e-5).  out bf16 or f32 (may alias x); stats = (mean,rstd) */
int owl_layernorm_fwd(void* stream, const float* x, const float* gamma, const float* beta, void* out, int out_bf16, float* stats, int64_t rows, int64_t D, float eps);
/* fused residual add: x_out = x + delta (bf16 output of the previous branch's GEMM, HF5:500,507 `residual + hidden_states`),
 * out = LN(x_out).  x_out may alias x.                                                                              */
int owl_add_layernorm_fwd(void* stream, const float* x, const void* delta_bf16, float* x_out, const float* gamma, const float* beta, void* out, int out_bf16, float* stats, int64_t rows, int64_t D, float eps, const void* delta2_bf16);

/* ---- fused self-attention forward (HF5:377-402): softmax(Q K^T * scale) V, dh = 64 -----------------
 * q, k, v row-major [B*Tp, ld_qkv] (head h at column h*64 of each; normally three column slices of the one [B*Tp, 3D] QKV GEMM output,
 * HF5:437-439); out row-major [B*Tp, ld_out]; lse optional [B,H,Tp] (log2 domain).  V is read where the QKV GEMM leaves it: the
 * [64 key][64 d] tile is transposed by the LDS hardware (ds_read_b64_tr_b16) -- no V^T copy of anything exists.
 * variant (per call, no global state): 0 = the library's choice; 1 = plain tiling (tokens 0..T-1 in 64-key tiles / 128-query blocks);
 * 2 = class token peeled: token 0 enters every other query's online softmax as its initial state and is itself one VALU-only workgroup
 * per (image, head), the tiles cover tokens 1..T-1 -- needs T - 1 a positive multiple of 64 (else rc != 0); same values as 1 to bf16
 * round-off (other summation order), 5.8 % faster at T = 2305.  0 picks 2 wherever it is allowed.
 * slow_tiles (optional, may be NULL): device int, += 1 for every (wave = 32 queries, 64-key tile) pair that leaves the fast path -- the tile's row sums
 * against the offset the wave already holds exceed 2^40 (or are inf / NaN) and the tile is recomputed with an explicit maximum.  A wave's first tile
 * is not counted.  Zero on scores of ordinary size (HF-init weights); trained-like weights with attention sinks trip it about once per wave and layer.
 * (The round-1 V^T form, the one-wave-per-SIMD and the 12-wave experiments: OWL_TUNING builds, include/owl_hip_tuning.h.) */
int owl_attention_fwd_vrow_bf16(void* stream, const void* q, const void* k, const void* v, int64_t ld_qkv, void* out, int64_t ld_out, float* lse, int64_t B, int64_t H, int64_t T, int64_t Tp, float scale, int variant, int* slow_tiles);

/* backward of the fused attention (layers whose attention runs backward): qkv row-major [B*Tp,3D] (q|k|v), dO / O row-major
 * [B*Tp,D], lse from the forward; writes dqkv [B*Tp,3D] (dq|dk|dv, bf16).  Every transposed operand of the dK/dV/dQ MFMAs is read
 * out of the row-major tiles by the LDS hardware (ds_read_b64_tr_b16): no Q^T / K^T / dO^T copies exist.  dvec_ws: f32 [B,H,Tp]. */
int owl_attention_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Tp, int64_t* bytes);   /* dvec_ws; `bytes` is a HOST pointer */
int owl_attention_bwd_bf16(void* stream, const void* qkv, const void* dO, const void* O, const float* lse, float* dvec_ws, void* dqkv, int64_t B, int64_t H, int64_t T, int64_t Tp, float scale, int phases);   /* phases (ABI 6): 0 = all three kernels; else a mask 1 dvec | 2 dK, dV | 4 dQ -- the last two only share inputs and may go to two streams behind the first */

/* ---- post_layernorm on all tokens + class-token merge + post_post_layernorm (ref src/models.py:80-86);
 * optional fused final residual add (delta_bf16 -> x_out = x + delta, may alias x)                      */
int owl_merge_ln_fwd(void* stream, const float* x, const void* delta_bf16, float* x_out, const float* g1, const float* b1, const float* g2, const float* b2, float* cls_ln, void* feats_bf16, float* stats1, float* stats2, int64_t B, int64_t P, int64_t Tp, int64_t D, float eps);

/* ---- heads -------------------------------------------------------------------------------------- */
/* qhat = Q/|Q| + 1e-6 (ref src/models.py:31-33, eps placement literal); padded to 32 rows        */
int owl_query_normalize(void* stream, const float* queries, float* qhat32, float* qnorm, int64_t nq, int64_t Dt);
/* sims = max over 3 prompts of (e/(|e|+1e-6)) . qhat (ref src/models.py:25-36); f32 MFMA            */
int owl_class_sims_fwd(void* stream, const float* e, const float* qhat32, float* sims, unsigned char* argmax, float* inv_norm, int64_t rows, int64_t Dt, int64_t C);
/* dense2 + box bias + sigmoid + center_to_corners (HF5:998, 1071-1104; ref src/models.py:70-73); D % 8 == 0, D <= 1024 */
int owl_box_final_fwd(void* stream, const void* h_bf16, const float* w2, const float* b2, const float* box_bias, float* boxes, float* sig_out, int64_t rows, int64_t P, int64_t D);

/* ---- Hungarian-matched push-pull loss, fully on device (no host sync) --------------------------------
 * Targets are padded: labels [B,Nmax] i64, tgt_boxes [B,Nmax,4] f32, counts [B] i32.
 * costT[b][j][p] = w_bbox*|box_p - tgt_j|_1 - w_class*softmax(sims_p)[label_j] - w_giou*GIoU   (ref src/matcher.py:106-131).
 * A label outside [0, C) never indexes sims: its class term is 0 and owl_push_pull_loss returns loss_ce = NaN for that batch
 * (the reference raises IndexError / one_hot error; validate on the host where the labels are host tensors). */
/* ragged per-image target lists (ref main.py:77-79; src/matcher.py:94-104 `targets`) -> the padded form in ONE launch: labels_cat i64 [N],
 * boxes_cat f32 [N,4] = the lists concatenated in image order, offsets i32 [B+1] (device) = image boundaries; pads are zero-filled. */
int owl_pack_targets(void* stream, const int64_t* labels_cat, const float* boxes_cat, const int* offsets, int64_t* labels, float* boxes, int* counts, int64_t B, int64_t Nmax);
int owl_match_cost(void* stream, const float* sims, const float* boxes, const int64_t* labels, const float* tgt_boxes, const int* counts, float* costT, int64_t B, int64_t P, int64_t C, int64_t Nmax, float w_class, float w_bbox, float w_giou);
/* rectangular LSAP per image (replaces scipy.optimize.linear_sum_assignment at ref src/matcher.py:134-137, f64 duals,
 * scipy's tie rule) + target scatter (ref src/matcher.py:146-157): pairs ordered by prediction index             */
int owl_hungarian(void* stream, const float* costT, const int64_t* labels, const int* counts, int64_t* pred_idx, int64_t* tgt_idx, int64_t* target_classes, int64_t B, int64_t P, int64_t Nmax, int64_t bg);
/* sequential IoU > thr label spreading (ref src/losses.py:100-106), in place on target_classes [B,P]             */
int owl_spread_labels(void* stream, const float* boxes, int64_t* target_classes, int64_t B, int64_t P, int64_t bg, float thr);
/* class_loss (ref src/losses.py:16-40) + loss_boxes (ref src/losses.py:42-69); losses[4] = mean over images of
 * {loss_ce, loss_bg, loss_bbox, loss_giou}; optional per-term gradients wrt sims / boxes                          */
int owl_push_pull_loss(void* stream, const float* sims, const float* boxes, const int64_t* target_classes, const float* scales, const float* tgt_boxes, const int64_t* pred_idx, const int64_t* tgt_idx, const int* counts, float* per_image, float* losses, float* dsims, float* dl1, float* dgiou, int64_t B, int64_t P, int64_t C, int64_t Nmax, int64_t bg);
int owl_push_pull_loss_bwd(void* stream, const float* g4, const int64_t* target_classes, const float* dsims, const float* dl1, const float* dgiou, float* out_sims, float* out_boxes, int64_t B, int64_t P, int64_t C, int64_t bg);

/* ---- inference post-process, fully on device (ref src/models.py:122-146 PostProcess.__call__; top-k prefix = ref
 * main.py:114-117).  Per image: score = max_c sims, class = first arg-max, keep score > conf_thr, class-aware NMS
 * (torchvision.ops.batched_nms semantics: same class, IoU > iou_thr, lower score suppressed), results ordered by
 * descending score (ties: ascending patch index), at most max_out per image.
 * route: which of torchvision's two batched_nms routes -- 0 per class on the raw coordinates (_batched_nms_vanilla), 1 coordinate
 * offset boxes + class * (max + 1) then one class-agnostic NMS (_batched_nms_coordinate_trick), 2 / 3 torchvision's own choice for a
 * GPU / CPU tensor (coordinate trick up to 20 000 / 4 000 coordinates past the threshold).  They differ only where the f32 rounding of
 * the shifted coordinates moves an IoU across the threshold.
 * boxes [B,P,4] f32 xyxy, sims [B,P,C] f32 -> out_boxes [B,max_out,4], out_scores [B,max_out], out_classes [B,max_out] i64,
 * out_patch [B,max_out] i64 (source patch index), out_count [B] i32; entries beyond the count are left untouched.
 * workspace: device scratch of owl_postprocess_workspace() bytes (16-byte aligned); `bytes` is a HOST pointer.     */
int owl_postprocess_workspace(int64_t B, int64_t P, int64_t* bytes);
int owl_postprocess(void* stream, const float* boxes, const float* sims, void* workspace, int64_t ws_bytes, float* out_boxes, float* out_scores, int64_t* out_classes, int64_t* out_patch, int* out_count, int64_t B, int64_t P, int64_t C, int64_t max_out, float conf_thr, float iou_thr, int route);

/* ---- the one collective of the path: in-place SUM of the flat f32 gradient bucket over the data-parallel ranks on `stream`, through the
 * CALLER's RCCL communicator (`rccl_comm` = an `ncclComm_t`).  For hosts without PyTorch; this repo's Python host issues the same collective through
 * torch.distributed (torch owns its communicator).  librccl.so is bound lazily at the first call: no load-time dependency.  Scale by 1 / world in
 * owl_adamw_step (`grad_scale`).  ABI 6.                                                                                                                          */
int owl_allreduce_sum_f32(void* stream, void* rccl_comm, float* buf, int64_t n);

/* ---- device input pipeline (ref src/dataset.py:69-71: HF OwlViTImageProcessor = PIL bicubic resize -> x(1/255) ->
 * (x-mean)/std).  owl_bicubic_coeffs is HOST-side (all pointers host): Pillow's Resample.c tap tables for one axis,
 * bounds[2*out] = {first tap, tap count}, kk[out*ksize] 22-bit fixed point; kk_capacity in ints; ksize returned.
 * owl_preprocess_u8_batch (device pointers), per image: src RGB u8 [H,W,3] -> horizontal pass into tmp u8 [H,out_w,3] -> vertical pass
 * -> lut[3][256] (the reference's rescale+normalize value of each u8 level) -> out [3,out_h,out_w] f32 or bf16.      */
int owl_bicubic_coeffs(int64_t in_size, int64_t out_size, int* bounds, int* kk, int64_t kk_capacity, int* ksize_out);
/* one launch pair for a ragged batch: desc = n_images x 10 int64 in DEVICE memory, per image
 * {src ptr, H, W, bounds_x ptr, kk_x ptr, ksize_x, bounds_y ptr, kk_y ptr, ksize_y, byte offset of its intermediate in tmp};
 * out [n_images,3,out_h,out_w]                                                                                        */
int owl_preprocess_u8_batch(void* stream, const void* desc, int64_t n_images, int64_t max_h, unsigned char* tmp, const float* lut, void* out, int out_bf16, int64_t out_h, int64_t out_w);
/* images that already have the model's size (ABI 6): src u8 [n,H,W,3] (src_chw = 0) or [n,3,H,W] (src_chw = 1) -> lut -> out [n,3,H,W] f32 | bf16.  Pillow's resize
 * to the size an image already has is a copy, so this IS the reference pipeline for such images, bit for bit.  H*W % 4 == 0; src 4-byte, out 16-byte aligned.       */
int owl_normalize_u8(void* stream, const unsigned char* src, int src_chw, const float* lut, void* out, int out_bf16, int64_t n_images, int64_t H, int64_t W);

/* ---- query-bank initialisation: the CLIP-style text tower run once by ref src/models.py:155-169 (HF5:603-663, 945-970).
 * Linear / LayerNorm layers reuse owl_gemm_nt_bf16 / owl_layernorm_fwd; these are the text-only pieces:
 * x[n*S+t] = tok_emb[ids[n,t]] + pos_emb[t] (f32 [N*S,W]); causal softmax attention for S <= 64, dh = 64 on
 * qkv bf16 [N*S,3W] -> out bf16 [N*S,W]; final LN of the EOS row (arg-max id) -> text_projection (f32 [Pdim,W], no bias)
 * -> L2 normalise -> out f32 [N,Pdim].                                                                               */
int owl_text_embed(void* stream, const int64_t* ids, const float* tok_emb, const float* pos_emb, float* x, int64_t N, int64_t S, int64_t W, int64_t vocab);
int owl_causal_attention_small(void* stream, const void* qkv_bf16, void* out_bf16, int64_t N, int64_t S, int64_t heads, float scale);
int owl_text_pool_project(void* stream, const float* x, const int64_t* ids, const float* gamma, const float* beta, const float* wproj, float* out, int64_t N, int64_t S, int64_t W, int64_t Pdim, float eps);

/* weight gradients without transposed copies:  slab[s][n][k] = sum_{m in split s} dY[m][n] * X[m][k]  (dW = dY^T X; both
 * operands token-major bf16 as the other kernels leave them; the transposition happens in LDS via ds_read_b64_tr_b16).
 * zero_row: >= 512 B of device zeros (source of rows m >= M); slabs [splits_used][N][K] f32 are reduced by owl_slab_reduce;
 * splits_used is a HOST pointer.  variant: 0 = the library's choice (the ping-pong schedule), 1 = single-phase kernel, 2 = ping-pong; same bits.
 * owl_colsum_bf16: colsum[c] += sum_r in[r][c] (bias gradients).                                                                   */
int owl_gemm_tn_slab_bf16(void* stream, const void* dY, int64_t ldy, const void* X, int64_t ldx, const void* zero_row, float* slab, int64_t M, int64_t N, int64_t K, int splits, int* splits_used, int variant, float* bias_slab);   /* bias_slab (ABI 6, optional): f32 [splits_used][N] receives the column sums of dY over each split's token range from the same pass (variant 0 / 2) -- the bias gradient's partial sums; add them with owl_slab_reduce(bias_slab, db, N, N, splits_used, 1) */
int owl_colsum_bf16(void* stream, const void* in_bf16, int64_t ld, float* colsum, int64_t R, int64_t C, float* partials, int64_t partials_floats);
/* f32 slab scratch of the call above for (M, N, K, splits): bytes = splits_used * N * K * 4 (`bytes` is a HOST pointer) */
int owl_gemm_tn_slab_workspace_bytes(int64_t M, int64_t N, int64_t K, int splits, int64_t* bytes);

/* pairwise out3 = {iou, union, giou} each [N,M] (free functions box_iou / generalized_box_iou, ref src/matcher.py:8-44) */
int owl_box_pairwise(void* stream, const float* boxes1, const float* boxes2, float* out3, int64_t N, int64_t M);

/* ---- backward of the row-wise pieces (autograd forms of the entries above) --------------------------
 * parameter-gradient outputs (dgamma, dbeta, dw2, db2, dqueries, colsum) ACCUMULATE into buffers the caller zeroes once per
 * step -- they are views of the flat gradient bucket.  No atomics: every workgroup writes its partial sums into `partials`
 * (caller-provided f32 scratch, `partials_floats` elements; size it once with owl_rowreduce_workspace_bytes) and a second
 * kernel adds them in a fixed order, so the bucket is bitwise reproducible run to run.                                  */
/* bytes of partial-sum scratch that serve every entry below for activations of `groups` images x `rows_per_group` rows x up
 * to C columns (C = the widest matrix reduced over rows, e.g. the MLP width for the fc1 bias gradient); HOST pointer.   */
int owl_rowreduce_workspace_bytes(int64_t groups, int64_t rows_per_group, int64_t C, int64_t* bytes);
/* dx_bf16 (optional): bf16 copy of dx, the operand of the next dX GEMM.  dx_colsum (optional, with dgamma / dbeta): += column sums of dx
 * = the bias gradient of the linear layer whose output sits at this residual position (no separate pass over the f32 dx).             */
int owl_layernorm_bwd(void* stream, const void* dy, int dy_bf16, const float* x, const float* stats, const float* gamma, const float* dres, float* dx, float* dgamma, float* dbeta, int64_t rows, int64_t D, void* dx_bf16, float* partials, int64_t partials_floats, float* dx_colsum);
int owl_merge_ln_bwd(void* stream, const float* dfeats, const float* x, const float* cls_ln, const float* stats1, const float* stats2, const float* g1, const float* b1, const float* g2, float* dx, float* dcls_ws, float* dg1, float* db1, float* dg2, float* db2, int64_t B, int64_t P, int64_t Tp, int64_t D, float* partials, int64_t partials_floats, void* dx_bf16, float* dx_colsum);
/* (owl_merge_ln_bwd: partials >= 6 D floats per 64-row block and image; dx_colsum (optional): += column sums of dx over every token of the batch,
 * class-token rows included = the bias gradient of the last encoder layer's fc2)                                                          */
/* class head backward, row-parallel part: de (bf16 [rows,Dt]), routed upstream G (bf16 [rows,32]) and a bf16 copy of e;
 * dqhat[32,Dt] = G^T e is then a split-K owl_gemm_nt_bf16, and owl_query_normalize_bwd maps it onto dqueries            */
int owl_class_sims_bwd(void* stream, const float* dsims, const float* sims, const unsigned char* argmax, const float* inv_norm, const float* e, const float* qhat32, void* de_bf16, void* g_bf16, void* e_bf16, int64_t rows, int64_t Dt, int64_t C);
int owl_query_normalize_bwd(void* stream, const float* dqhat, const float* queries, float* dqueries, int64_t nq, int64_t Dt);
/* dw2 [4,D] and db2 [4] must be contiguous (dw2 then db2); partials = f32 [owl_box_final_bwd_blocks(rows)][5*D+4];
 * du1_colsum (optional, [D]): += column sums of du1 = dense1's bias gradient, from the same pass */
int owl_box_final_bwd_blocks(int64_t rows);
int owl_box_final_bwd(void* stream, const float* dboxes, const float* sig, const void* h1_bf16, const void* u1_bf16, const float* w2, void* du1_bf16, float* partials, float* dw2_db2, int64_t rows, int64_t D, float* du1_colsum);
int owl_transpose_colsum_bf16(void* stream, const void* in, int64_t ld_in, void* out_t, int64_t ld_out, float* colsum, int64_t R, int64_t C, float* partials, int64_t partials_floats);
int owl_colsum_f32(void* stream, const float* in, float* colsum, int64_t R, int64_t C, float* partials, int64_t partials_floats);

/* ---- fused AdamW on the flat trainable bucket (replaces torch.optim.AdamW.step, ref main.py:56-60,91) -----
 * decoupled weight decay, bias-corrected; g is pre-scaled by grad_scale (1/world after the sum all-reduce);
 * optionally refreshes the bf16 compute copy in the same pass                                              */
int owl_adamw_step(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale);

/* ---- utilities ------------------------------------------------------------------------------------ */
int owl_cast_f32_bf16(void* stream, const float* in, void* out, int64_t n);
int owl_transpose_bf16(void* stream, const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C);

#ifdef __cplusplus
}
#endif
#endif /* OWL_HIP_H */
