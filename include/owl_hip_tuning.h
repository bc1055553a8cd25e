/*
 * Tuning / race-hunting switches of libowlhip -- NOT part of the product ABI.  They are process-global mutable state and
 * only exist when the library is built with OWL_TUNING=1 (owl-vit-object-detection_amd/csrc/build.sh); the default build
 * does not export them and `_lib.py` only binds them when OWL_TUNING=1 is set in the environment.  Used by tools/*.py.
 */
#ifndef OWL_HIP_TUNING_H
#define OWL_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif
/* persistent scheduling (one workgroup per CU walks tiles with cross-tile prefetch): 1 on (default), 0 off */
int owl_gemm_set_persistent(int on);
/* 1: run the GEMM main loop but skip every epilogue store; 8: ping-pong trace run (tools/pp_trace.py) */
int owl_gemm_debug_nostore(int on);
/* override the persistent grid size */
int owl_gemm_debug_slots(int n);
/* attention forward: bit 0 always rescale, bit 1 plain block mapping, bit 2 the first tile always sets the softmax offset */
int owl_attention_debug(int flags);
/* two-phase ping-pong GEMM: device buffer of 128 x u64 -- while set, bias-epilogue launches run the s_memtime-stamped kernel (tools/pp2_trace.py); NULL = off */
/* two-phase ping-pong GEMM: persistent grid size (default 256 = one workgroup per CU; multiples of 8) */
int owl_gemm_pp2_slots(int n);
/* ... 1: skip every epilogue store of that kernel (tools/gemm_nostore_ab.py) */
int owl_gemm_pp2_nostore(int on);
int owl_gemm_pp2_block_width(int epi, int bw);   /* tile order of the two-phase GEMM: epi 0 bias / 1 quick-GELU; bw 0 = the launcher's rule, else column blocks of the largest divisor of tiles_n up to bw */
int owl_gemm_pp2_lines(int on);              /* quad-contiguous epilogue stores: 0 off, 1 bias epilogue (default, = the product), 2 quick-GELU epilogue too, 3 + dX through quick-GELU' (its saved tile is loaded that way too) */
int owl_gemm_pp2_stagger(int n);             /* round 6 experiment: every second workgroup of an XCD starts n x ~8 k cycles late (epilogue bursts of the two halves interleave) */
int owl_gemm_pp2_ablate(int a);              /* timing-only ablations of the shipped two-phase GEMM: bit 0 no LDS-DMA requests after the prologue, bit 1 fragments read once per tile, bit 2 no epilogue */
int owl_gemm_pp2_trace(void* buf);
/* ... which of workgroup 0's tiles is stamped (0 = its first; later tiles see the sustained clock and warm queues) */
int owl_gemm_pp2_trace_tile(int n);
/* ... and which K-tile of it (default 4; 0-2 show the refill behind the previous tile's epilogue) */
int owl_gemm_pp2_trace_ktile(int n);
/* round-1 attention forward with V^T per head [B][heads*64][Tp] as written by GEMM epilogue 6 (bit-identical to variant 1 of owl_attention_fwd_vrow_bf16) */
int owl_attention_fwd_bf16(void* stream, const void* q, const void* k, int64_t ld_qk, const void* vt, int64_t vt_img_stride, void* out, int64_t ld_out, float* lse, int64_t B, int64_t H, int64_t T, int64_t Tp, float scale);
/* attention forward experiments: variant 3 = one wave per SIMD, 64 queries per wave (csrc/attention_fwd_w64.hip; blocks whose scores leave the range of its
 * offset-free softmax are flagged in redo_ws and redone by the classic kernel in the same call), 4 = its s_memtime-stamped form, 5 = one 12-wave workgroup
 * per CU sharing the stage buffers; 0-2 as owl_attention_fwd_vrow_bf16.  redo_ws: owl_attention_fwd_workspace_bytes(B, H, T) bytes (`bytes`: HOST pointer) */
int owl_attention_fwd_workspace_bytes(int64_t B, int64_t H, int64_t T, int64_t* bytes);
int owl_attention_fwd_w64_bf16(void* stream, const void* q, const void* k, const void* v, int64_t ld_qkv, void* out, int64_t ld_out, float* lse, int64_t B, int64_t H, int64_t T, int64_t Tp, float scale, int variant, int* redo_ws);
/* free-running GEMM (csrc/gemm_fr.hip, tile 5): timing-only ablations (1 no LDS-DMA after the prologue, 2 fragments read once per tile, 3 both), persistent grid
 * size (default 512 = two workgroups per CU), column-block width of the tile order */
int owl_gemm_fr_ablate(int a);
int owl_gemm_fr_slots(int n);
int owl_gemm_fr_block_width(int epi, int bw);
#ifdef __cplusplus
}
#endif
#endif
