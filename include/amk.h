/*
 * amk.h -- C ABI of libamk.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * attention / MoE / VQ hot path of pranoyr/attention-models.
 *
 * The reference has no FFI: its hot path is five Python nn.Modules (SURVEY.md
 * section 8b).  This header is the boundary a maintainer binds instead of the
 * eager-PyTorch op sequences inside those modules; every entry point cites the
 * reference lines it replaces.  The Python binding that ships with this repo is
 * attention-models_amd/amk/lib.py (ctypes); INTEGRATION.md shows the stub.
 *
 * Contract (all entry points)
 *   - extern "C", plain pointers and sizes; no C++ / torch types cross the ABI.
 *   - every pointer is a DEVICE pointer (hipMalloc'ed, fp32 unless stated),
 *     allocated and owned by the caller; the library never allocates, frees or
 *     synchronises.  Workspaces are caller-allocated, sizes stated per function.
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream).
 *   - return value: AMK_OK (0) or a negative AMK_E* code; amk_last_error()
 *     returns a thread-local message for the last failure on this thread.
 *   - tensors are row-major; "stride" arguments are in ELEMENTS, the innermost
 *     (head-dim / feature) axis is always contiguous.
 *   - indices are int64 where the reference returns torch.int64.
 */
#ifndef AMK_H_
#define AMK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMK_VERSION 140 /* 0.4.0: amk_attn_bf16_fwd / _bwd take key_mask / causal_mask (signatures changed), the optimizer step takes a per-parameter table (decoupled weight decay, device-resident learning rate), one-pass backward for head dims 32 / 128 behind amk_attn_bwd; 0.3.5: amk_agent_conv_grad_reduce; 0.3.4: amk_moe_route_distinct, _rows forms of combine / gate_grad; 0.3.3: amk_moe_expert_sums; 0.3.2: bf16 forward / input-gradient GEMM; 0.3.1: bf16 weight-gradient GEMM; 0.3.0: dense f32 GEMMs with LayerNorm / SwiGLU / residual / bias-gradient fusions (amk_gemm_f32) */

enum {
  AMK_OK = 0,
  AMK_EINVAL = -1,       /* bad argument (null pointer, non-positive size, ...) */
  AMK_EUNSUPPORTED = -2, /* shape outside what the kernels are built for        */
  AMK_ELAUNCH = -3       /* hipLaunchKernel / hipMemsetAsync reported an error  */
};

int amk_version(void);
/* Offload architecture the code objects were built for ("gfx950"). */
const char* amk_arch(void);
const char* amk_last_error(void);

/* --------------------------------------------------------------------------
 * Fused softmax attention core.
 * Replaces models/softmax_attention.py:62-76 (and the identical core of
 * models/switchhead_attention.py:98-111): scaled QK^T, key-padding fill,
 * causal fill, row softmax, PV -- without materialising the (B,h,I,J) scores.
 *
 *   S[b,h,i,j] = sum_d (q[b,h,i,d]*scale) * k[b,h,j,d]
 *   S = -1e9 where key_mask[b,j] == 0          (reference: masked_fill(~context_mask))
 *   S = -1e9 where causal_mask[i,j] != 0       (reference: masked_fill(causal_mask))
 *   P = softmax_j(S);  o[b,h,i,:] = sum_j P[b,h,i,j] v[b,h,j,:]
 *   stats[b,h,i] = { m, l }: a row reference m in the log2 domain and l = sum_j exp2(S*log2(e) - m)
 *                 (saved for the backward, which forms P = exp2(S*log2(e) - m) / l; two floats per row so
 *                  that a fully masked row, S = -1e9 everywhere, keeps its exact 1/J weights).
 *                 With a mask m = max_j S*log2(e); without masks (D = 64) m is a reference within 8 of
 *                 that maximum -- it moves only when a tile's maximum passes it by more than 8 -- and l
 *                 is the sum relative to it: the same P.
 *
 * q/o are addressed as  base + b*sb + t*st + h*sh + d   (d < D contiguous), so
 * the (B,T,h*D) projection outputs are consumed in place (st = h*D, sh = D) as
 * well as (B,h,T,D) tensors (sh = T*D, st = D).  k and v likewise with J rows.
 * key_mask: uint8 (B,J) contiguous or NULL; causal_mask: uint8 (I,J) contiguous
 * or NULL.  stats: (B,H,I,2) contiguous.  D is 32, 64 or 128 (64: the tuned kernels and the split-bf16
 * forward; 32 / 128: csrc/attn_generic.hip -- forward, with kept scores when there is no mask -- and
 * csrc/attn_bwd_fused_gen.hip, the one-pass backward with dq by atomics; the reproducible backward of
 * those head dims is the two recompute kernels of csrc/attn_generic.hip).
 * -------------------------------------------------------------------------- */
int amk_attn_fwd(const float* q, const float* k, const float* v, float* o, float* stats,
                 const uint8_t* key_mask, const uint8_t* causal_mask,
                 int B, int H, int I, int J, int D,
                 int64_t q_sb, int64_t q_st, int64_t q_sh,
                 int64_t k_sb, int64_t k_st, int64_t k_sh,
                 int64_t v_sb, int64_t v_st, int64_t v_sh,
                 int64_t o_sb, int64_t o_st, int64_t o_sh,
                 float scale, void* stream);

/* The same forward, which also leaves the raw scores for the backward (training): scores must hold
 * amk_attn_scores_bytes(B, H, I, J) bytes (4 bytes per (b, h, i, j), both sequence lengths padded to the
 * kernels' tiles; 32x32 tiles, [b][h][key block][query block][key][query]).  The reference keeps the same
 * tensor alive for autograd (models/softmax_attention.py:62, the einsum output); here it is written
 * once and read once, by amk_attn_bwd_kept, which then runs four matrix products instead of five. */
int64_t amk_attn_scores_bytes(int B, int H, int I, int J);
int amk_attn_fwd_keep(const float* q, const float* k, const float* v, float* o, float* stats, float* scores,
                      const uint8_t* key_mask, const uint8_t* causal_mask,
                      int B, int H, int I, int J, int D,
                      int64_t q_sb, int64_t q_st, int64_t q_sh,
                      int64_t k_sb, int64_t k_st, int64_t k_sh,
                      int64_t v_sb, int64_t v_st, int64_t v_sh,
                      int64_t o_sb, int64_t o_st, int64_t o_sh,
                      float scale, void* stream);

/* The same forward with split-bf16 products: every f32 operand is split into three bf16 parts
 * (24 mantissa bits) and each product is the sum of six exact partial products accumulated in f32
 * (v_mfma_f32_32x32x16_bf16).  Same arguments, outputs, statistics and masks; the error against a
 * double-precision result is that of the f32 MFMA path (1.5e-6 vs 2.6e-6 on K = 64 dot products,
 * tools/ubench_bf16x6.hip), not that of a bf16 computation.  The backward is amk_attn_bwd either way.
 * ws: amk_attn_fwd_x6_ws_bytes(B, H, J) bytes (48 KiB per (batch, head, 64-key tile): K and V split into
 * bf16 planes once per call by a pre-pass); contents undefined on return. */
int64_t amk_attn_fwd_x6_ws_bytes(int B, int H, int J);
int amk_attn_fwd_x6(const float* q, const float* k, const float* v, float* o, float* stats, void* ws,
                    const uint8_t* key_mask, const uint8_t* causal_mask,
                    int B, int H, int I, int J, int D,
                    int64_t q_sb, int64_t q_st, int64_t q_sh,
                    int64_t k_sb, int64_t k_st, int64_t k_sh,
                    int64_t v_sb, int64_t v_st, int64_t v_sh,
                    int64_t o_sb, int64_t o_st, int64_t o_sh,
                    float scale, void* stream);

/* Backward of amk_attn_fwd (autograd of the same reference lines).
 * Inputs: q,k,v,o,stats as in the forward, d_o (gradient of o, same addressing
 * as o with its own strides).  Outputs: dq (q-like), dk (k-like), dv (v-like),
 * each fully overwritten.  delta_ws: workspace of amk_attn_bwd_ws_floats(B, H, I, J, stages) floats
 * (B*H*I rounded up to 4, unless AMK_ATTN_BWD_DQ_REPRO asks for more).
 * Gradients do not flow through filled (-1e9) positions, as in masked_fill.
 * `stages` selects the launches.  bit 0: delta = rowsum(dO*O) into delta_ws (must have run
 * before any other bit).  Then EITHER bit 3 (AMK_ATTN_BWD_FUSED): one pass, each of the five
 * products computed once, dq accumulated with f32 atomics (dq must be the dense (B,I,H,D) layout;
 * it is zeroed here; dq differs in the last bits from run to run; falls back to the two-kernel
 * path when a causal mask is given or dq is strided) -- OR bits 1 and 2: two recompute kernels,
 * dK/dV and dQ, no atomics, bitwise reproducible.  Profilers time one stage by passing its bit. */
#define AMK_ATTN_BWD_DELTA 1
#define AMK_ATTN_BWD_DKDV 2
#define AMK_ATTN_BWD_DQ 4
#define AMK_ATTN_BWD_FUSED 8
#define AMK_ATTN_BWD_ALL 7          /* delta + dK/dV + dQ: reproducible */
#define AMK_ATTN_BWD_FAST 9         /* delta + fused */
/* fused pass only: keys per workgroup (neither bit: 256 when J >= 256, else 128).  256 halves the
 * number of atomic adds per dq element (J / 256 instead of J / 128). */
#define AMK_ATTN_BWD_KEYS128 16
#define AMK_ATTN_BWD_KEYS256 32
/* fused pass only: bitwise reproducible dq.  No atomics: a workgroup that holds every key of its (batch, head)
 * (J <= keys per workgroup) stores dq directly; otherwise every key block stores its partial and a second
 * launch sums the partials in key-block order.  delta_ws must then hold amk_attn_bwd_ws_floats(...) floats
 * (the deltas followed by the partials). */
#define AMK_ATTN_BWD_DQ_REPRO 64
#define AMK_ATTN_BWD_FAST_REPRO 73  /* delta + fused + reproducible dq */
int64_t amk_attn_bwd_ws_floats(int B, int H, int I, int J, int stages);
int amk_attn_bwd(const float* q, const float* k, const float* v, const float* o,
                 const float* stats, const float* d_o,
                 float* dq, float* dk, float* dv, float* delta_ws,
                 const uint8_t* key_mask, const uint8_t* causal_mask,
                 int B, int H, int I, int J, int D,
                 int64_t q_sb, int64_t q_st, int64_t q_sh,
                 int64_t k_sb, int64_t k_st, int64_t k_sh,
                 int64_t v_sb, int64_t v_st, int64_t v_sh,
                 int64_t o_sb, int64_t o_st, int64_t o_sh,
                 int64_t do_sb, int64_t do_st, int64_t do_sh,
                 int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                 int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                 int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                 float scale, int stages, void* stream);
/* amk_attn_bwd reading the scores amk_attn_fwd_keep left (same q, k, masks, scale): the fused pass skips
 * its S = QK^T product; the results are those of amk_attn_bwd bit for bit (dq up to the atomics' order).
 * `stages` must contain AMK_ATTN_BWD_FUSED; when the fused pass cannot run (causal mask, strided dq) the
 * two recompute kernels run and the scores are not read. */
int amk_attn_bwd_kept(const float* scores, const float* q, const float* k, const float* v, const float* o,
                      const float* stats, const float* d_o,
                      float* dq, float* dk, float* dv, float* delta_ws,
                      const uint8_t* key_mask, const uint8_t* causal_mask,
                      int B, int H, int I, int J, int D,
                      int64_t q_sb, int64_t q_st, int64_t q_sh,
                      int64_t k_sb, int64_t k_st, int64_t k_sh,
                      int64_t v_sb, int64_t v_st, int64_t v_sh,
                      int64_t o_sb, int64_t o_st, int64_t o_sh,
                      int64_t do_sb, int64_t do_st, int64_t do_sh,
                      int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                      int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                      int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                      float scale, int stages, void* stream);

/* --------------------------------------------------------------------------
 * VQ codebook nearest-neighbour lookup.
 * Replaces models/vitvqgan.py:151-171 (Codebook.forward) and :16-17 (l2_norm).
 *
 *   zn = z / max(||z||, 1e-12)                   (N,C)
 *   en = E / max(||E||, 1e-12)                   (K,C)
 *   dist[n,k] = (sum zn[n]^2 + sum en[k]^2) - 2 * <zn[n], en[k]>
 *   idx[n] = argmin_k dist[n,k]                  (first minimum on ties)
 *   zq[n]  = E[idx[n]] / max(||E[idx[n]]||, 1e-12)
 *   out[n] = zn[n] + (zq[n] - zn[n])             (straight-through value)
 *   sqerr_partial[w] = partial sums of (zq - zn)^2 ; the caller forms
 *        loss = (1 + beta) * sum(sqerr_partial) / (N*C)
 *
 * C must be 32, 64, 128 or 256; K is any positive size (the reference takes any codebook_size,
 * models/vitvqgan.py:141).  nsplit >= 1 splits the codebook over workgroups; internally the
 * normalised codebook is padded to Kp = amk_vq_padded_codes(K, nsplit) rows (whole 32-code tiles
 * per slice; the padding scores -inf and is never chosen).  Workspaces: en_ws Kp*C floats,
 * ee_ws Kp floats, pmin_ws N*nsplit floats, pidx_ws N*nsplit int32.  sqerr_partial holds
 * amk_vq_num_partials(N) floats.  The distance matrix is never written to HBM.
 * -------------------------------------------------------------------------- */
int64_t amk_vq_num_partials(int64_t N);
int amk_vq_padded_codes(int K, int nsplit);
int amk_vq_lookup_fwd(const float* z, const float* codebook, int64_t N, int K, int C, int nsplit,
                      float* en_ws, float* ee_ws, float* pmin_ws, int32_t* pidx_ws,
                      int64_t* idx, float* out, float* zq, float* zn, float* sqerr_partial,
                      void* stream);

/* Backward of Codebook.forward (autograd of models/vitvqgan.py:151-171).
 *   g_out (N,C): gradient of `out`; g_loss: DEVICE pointer to the scalar
 *   gradient of the loss.  dz (N,C) is overwritten; dcodebook (K,C) is zeroed
 *   here and then scatter-added by idx.
 *   dzn = g_out + g_loss*2*beta*(zn - zq)/(N*C);   dz = J_norm(z)^T dzn
 *   dzq = g_loss*2*(zq - zn)/(N*C);  dE[idx[n]] += J_norm(E[idx[n]])^T dzq */
int amk_vq_lookup_bwd(const float* z, const float* codebook, const float* zn, const float* zq,
                      const int64_t* idx, const float* g_out, const float* g_loss, float beta,
                      int64_t N, int K, int C, float* dz, float* dcodebook, void* stream);

/* The same backward without atomics: instead of scatter-adding into dcodebook, every row's contribution
 * J_norm(E[idx[n]])^T dzq[n] is written to ge_rows (N,C); the caller adds the rows into dcodebook[idx[n]] in an
 * order of its choosing (the Python binding uses a deterministic index_add_ when reproducible gradients are asked
 * for: AMK_DETERMINISTIC=1 / torch.use_deterministic_algorithms). */
int amk_vq_lookup_bwd_rows(const float* z, const float* codebook, const float* zn, const float* zq,
                           const int64_t* idx, const float* g_out, const float* g_loss, float beta,
                           int64_t N, int K, int C, float* dz, float* ge_rows, void* stream);

/* Codebook.indices_to_embeddings (models/vitvqgan.py:173-176): out[n] = l2norm(E[idx[n]]).
 * The indices are the caller's: one outside [0, K) is never dereferenced (it reads row 0 or K-1) and
 * is counted in *bad_count (device int32, zeroed by the caller; NULL = do not count) -- the reference
 * raises IndexError there (nn.Embedding), and so does the Python binding from the count.
 * amk_vq_lookup_bwd clamps its (forward-produced) indices the same way. */
int amk_vq_gather(const int64_t* idx, const float* codebook, int64_t N, int K, int C,
                  float* out, int32_t* bad_count, void* stream);

/* --------------------------------------------------------------------------
 * Top-k expert routing and grouped expert GEMMs.
 * Replace the per-expert Python loops (torch.where + gather + Linear + index_put) of
 * MoELayer.forward (models/moe.py:23-38) and SwitchHeadAttention.moe_v / moe_out
 * (models/switchhead_attention.py:58-88).  A routed "unit" u is a token (MoELayer) or a
 * (token, head) (SwitchHead); a "pair" is p = u*k + slot.
 * -------------------------------------------------------------------------- */

/* torch.topk(logits, k) + sigmoid of the selected logits, and the expert-major ordering.
 *   logits (U,E) -> ids int64 (U,k) descending by logit (lowest index on equal logits),
 *   gate (U,k) = sigmoid(selected logit); offsets int32 (E+1), perm int32 (U*k): pairs grouped
 *   by expert, ascending pair index inside an expert (deterministic).
 *   Workspaces: counts int32 (E), rank int32 (U*k), blockhist int32 (amk_moe_route_ws_ints(U,E,k)).
 *   k <= 8, E <= 1024. */
int64_t amk_moe_route_ws_ints(int64_t U, int E, int k);
int amk_moe_route(const float* logits, int64_t U, int E, int k,
                  int64_t* ids, float* gate, int32_t* counts, int32_t* rank, int32_t* blockhist,
                  int32_t* offsets, int32_t* perm, void* stream);

/* Y[p,:] = A[p / a_div, :] * W[e(p)]^T (+ bias[e(p)]) for every pair: the expert Linear layers
 * (models/moe.py:34-36, switchhead_attention.py:69-71,86).  A rows have Kd floats at stride lda;
 * W is (E,N,Kd) contiguous, bias (E,N) or NULL, Y is (P,N).  N, Kd, lda multiples of 4. */
int amk_grouped_gemm_nt(const float* A, int64_t lda, int a_div, const float* W, const float* bias,
                        const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                        float* Y, void* stream);

/* Y[p,:] = scale[p] * (A[p / a_div, :] * W[e(p)]): input gradient of the expert Linear.
 * A rows have N floats, Y is (P,Kd); scale (P) or NULL. */
int amk_grouped_gemm_nn(const float* A, int64_t lda, int a_div, const float* W, const float* scale,
                        const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                        float* Y, void* stream);

/* The same two products with the sum over the pairs of an output row folded in: pair p ADDS its row into
 * Y[p / y_div, :] (f32 atomics; Y is ((P-1)/y_div + 1, N) resp. (.., Kd) and must be ZEROED by the caller).
 * Replaces amk_grouped_gemm_* followed by amk_moe_combine where the combine is un-weighted -- the head / slot sum
 * of SwitchHead's output experts (switchhead_attention.py:86-87,115: 16 pairs per token, a 272 MB per-pair
 * intermediate at the ViTMoE layer size) and the sum of the input gradients of its V experts -- at the price of
 * the reference's fixed accumulation order (ascending expert id): sums differ in the last bits from run to run.
 * Needs the wide kernels: N >= 128 and Kd % 32 == 0 (nt), Kd >= 128 and N % 32 == 0 (nn), buffers below 2 GB;
 * AMK_EUNSUPPORTED otherwise. */
int amk_grouped_gemm_nt_acc(const float* A, int64_t lda, int a_div, const float* W, const float* bias,
                            const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                            float* Y, int y_div, void* stream);
int amk_grouped_gemm_nn_acc(const float* A, int64_t lda, int a_div, const float* W, const float* scale,
                            const int32_t* offsets, const int32_t* perm, int64_t P, int E, int N, int Kd,
                            float* Y, int y_div, void* stream);

/* dW[e] = sum_{p in e} scale[p] * G[p / g_div, :]^T (x) X[p / x_div, :]   (E,N,Kd), fully
 * overwritten; dbias[e] = sum_{p in e} scale[p] * G[p / g_div, :] (E,N) or NULL.  (With few output tiles and long
 * experts the entry zeroes dW and two workgroups per tile add their halves of the pairs: 0 + a + b, the same bits in
 * either order.) */
int amk_grouped_gemm_wgrad(const float* G, int64_t ldg, int g_div, const float* X, int64_t ldx, int x_div,
                           const float* scale, const int32_t* offsets, const int32_t* perm,
                           int64_t P, int E, int N, int Kd, float* dW, float* dbias, void* stream);

/* out[g,:] = sum_{o<outer} ( sum over the k slots of unit g*outer+o, in ASCENDING EXPERT ID,
 * of scale[p]*Y[p,:] ): the accumulation order of the reference loops (moe.py:32-36) and the
 * head sum of switchhead_attention.py:115.  scale (G*outer*k) or NULL (moe_out is un-weighted). */
int amk_moe_combine(const float* Y, const int64_t* ids, const float* scale, int64_t G, int outer, int k,
                    int N, float* out, void* stream);

/* The selection alone: ids (U,k) = top-k expert indices of every logits row (descending value, lowest index on ties, as
 * torch.topk), gate (U,k) = sigmoid of the selected logits -- the first stage of amk_moe_route, for callers that build
 * their own lists (amk_moe_route_distinct). */
int amk_moe_topk(const float* logits, int64_t U, int E, int k, int64_t* ids, float* gate, void* stream);

/* Lists of the DISTINCT (row group, expert) combinations of a routing: group g = pairs g*fan .. g*fan+fan-1 (in
 * SwitchHead one token's heads x slots).  Where all pairs of a group read the same input row (moe_v,
 * switchhead_attention.py:58-73: the token's row for every head) or their products are summed over the group anyway
 * (moe_out, :75-88,115), an expert's product is needed once per distinct (group, expert) -- 12.9 instead of 16 per
 * token at E 32, h 8, top-2.  offsets (E+1) / perm (up to G*min(fan,E) entries) list "virtual pairs" g*E + e by expert,
 * groups ascending; the amk_grouped_gemm_* entry points run on them unchanged with P = G*E, a_div = E (input row g,
 * output row g*E + e).  mask (G) is a workspace.  E <= 64, fan <= 4096. */
int amk_moe_route_distinct(const int64_t* ids, int64_t G, int fan, int E, uint64_t* mask,
                           int32_t* offsets, int32_t* perm, void* stream);

/* amk_moe_combine / amk_moe_gate_grad reading such per-(group, expert) rows: pair p reads Y row
 * (p / v_div) * E + ids[p]  (v_div = pairs per group; v_div == 0: row p, the plain entry points). */
int amk_moe_combine_rows(const float* Y, const int64_t* ids, const float* scale, int64_t G, int outer, int k,
                         int N, int v_div, int E, float* out, void* stream);
int amk_moe_gate_grad_rows(const float* d_out, const float* Y, const int64_t* ids, const float* gate,
                           int64_t P, int k, int E, int N, int g_div, int v_div, float* dlogits, void* stream);

/* Z[g, e, :] = sum over the fan pairs of row g (pairs g*fan .. g*fan+fan-1) that chose expert e = ids[p] of
 * scale[p] * A[p / a_div, :]  (d floats at stride lda; scale (G*fan) or NULL).  Z (G, E*d) is fully overwritten, in a
 * fixed order.  With it the head / slot sum of SwitchHead's output experts (switchhead_attention.py:86-87,115) is one
 * dense product Z x (E*d, N) instead of amk_grouped_gemm_nt + amk_moe_combine and their (pairs, N) intermediate, and
 * likewise the input gradient of its V experts (switchhead_attention.py:69-71) -- worth it where a row has at least
 * E/2 pairs (the dense product does E/fan times the routed FLOPs).  d, lda multiples of 4; fan <= 4096. */
int amk_moe_expert_sums(const float* A, int64_t lda, int a_div, const int64_t* ids, const float* scale,
                        int64_t G, int fan, int E, int d, float* Z, void* stream);

/* Gradient of the gate logits through sigmoid(topk): dlogits (P/k, E) fully overwritten,
 * dlogits[p/k, ids[p]] = gate[p]*(1-gate[p]) * <d_out[p / g_div, :], Y[p, :]>, zero elsewhere. */
int amk_moe_gate_grad(const float* d_out, const float* Y, const int64_t* ids, const float* gate,
                      int64_t P, int k, int E, int N, int g_div, float* dlogits, void* stream);

/* --------------------------------------------------------------------------
 * AgentAttention core.
 * Replaces models/agent_attention.py:55-73: agent tokens A = adaptive average pool of q over
 * the sequence into P bins per head (AdaptiveAvgPool2d over the (t, h) plane with
 * h == P, the only configuration in which the reference runs -- SURVEY.md section 0.5),
 *   V_a = softmax((A*scale) K^T) V,   O = softmax((q*scale) A^T) V_a + dwc(v)
 * where dwc is the depthwise 3x3 convolution (weights (D,1,3,3), bias (D)) over the
 * (head, token) plane.  The reference's scalar bias1 / bias2 are added to whole softmax rows,
 * so they change nothing and receive zero gradient; they do not cross the ABI.
 * q,k,v,o addressed like amk_attn_fwd (T rows).  Saved for the backward: agents (B,H,P,D),
 * vagent (B,H,P,D), stats1 (B,H,P,2) = {row max, row sum} of the aggregation softmax.
 * D must be 64, P <= 16, P <= T.
 * The sequence is processed in chunks of 128 tokens, one workgroup per (batch, head, chunk);
 * sums over tokens go through per-chunk partials in the caller's workspace `ws`
 * (amk_agent_ws_floats(B,H,T,P,backward) floats, contents undefined on return) and are folded
 * in chunk order, so results are bitwise reproducible.
 * -------------------------------------------------------------------------- */
int amk_agent_num_chunks(int T);                                  /* ceil(T / 128); 0 for T <= 0 */
int64_t amk_agent_ws_floats(int B, int H, int T, int P, int backward); /* workspace size in floats */

int amk_agent_attn_fwd(const float* q, const float* k, const float* v, const float* conv_w, const float* conv_b,
                       float* o, float* agents, float* vagent, float* stats1, float* ws,
                       int B, int H, int T, int D, int P,
                       int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                       int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                       float scale, void* stream);

/* Backward of amk_agent_attn_fwd: dq, dk, dv fully overwritten (q/k/v-like addressing);
 * dconvw_part (B*H*NC, 9, D) and dconvb_part (B*H*NC, D), NC = amk_agent_num_chunks(T), are
 * per-(batch, head, chunk) partial sums of the convolution weight / bias gradients (the caller
 * sums over the first axis; weight element [c][0][a][b] is partial [.., a*3+b, c]);
 * ws: amk_agent_ws_floats(B,H,T,P,1) floats. */
int amk_agent_attn_bwd(const float* q, const float* k, const float* v, const float* conv_w, const float* d_o,
                       const float* agents, const float* vagent, const float* stats1,
                       float* dq, float* dk, float* dv, float* ws, float* dconvw_part, float* dconvb_part,
                       int B, int H, int T, int D, int P,
                       int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                       int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t do_sb, int64_t do_st, int64_t do_sh,
                       int64_t dq_sb, int64_t dq_st, int64_t dq_sh, int64_t dk_sb, int64_t dk_st, int64_t dk_sh,
                       int64_t dv_sb, int64_t dv_st, int64_t dv_sh, float scale, void* stream);

/* The sums over the first axis of amk_agent_attn_bwd's partials, in a fixed order, in one launch: dconvw (D, 1, 3, 3)
 * = the depthwise convolution's weight gradient in the parameter's layout (agent_attention.py:41-45), dconvb (D). */
int amk_agent_conv_grad_reduce(const float* dconvw_part, const float* dconvb_part, int64_t rows, int D,
                               float* dconvw, float* dconvb, void* stream);

/* --------------------------------------------------------------------------
 * Residual-add + LayerNorm and the bias-gradient column sum (the
 * pre-LN / residual epilogues around the attention and FFN of models/vitvqgan.py:44-61 and
 * models/transformer.py:11-19,58-76).
 *   forward : h = x + res (written only when res != NULL; then h != NULL too);  y = LN(h)*gamma + beta
 *             mean, rstd (M) are saved for the backward.  eps as torch.nn.LayerNorm (1e-5 default).
 *   backward: dh = LN'(dy) + dh_in (dh_in may be NULL); dgb_part (amk_rowsum_num_partials(M), 2, D)
 *             holds per-workgroup partial sums of dgamma (row 0) and dbeta (row 1): the caller sums axis 0.
 *             `h` is the forward's h (or x when there was no residual).
 *   colsum  : part (amk_rowsum_num_partials(M), N) partial column sums of x (M, N); caller sums axis 0.
 * Widths: multiples of 4 up to 4096; all matrices row-major contiguous; HBM-bound, one pass each.
 * -------------------------------------------------------------------------- */
int amk_rowsum_num_partials(int64_t M);
int amk_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                          int64_t M, int D, float eps, float* h, float* y, float* mean, float* rstd, void* stream);
int amk_add_layernorm_bwd(const float* dy, const float* h, const float* dh_in, const float* gamma,
                          const float* mean, const float* rstd, int64_t M, int D, float* dh, float* dgb_part,
                          void* stream);
int amk_colsum(const float* x, int64_t M, int N, float* part, void* stream);

/* --------------------------------------------------------------------------
 * Fused SwiGLU gate (epilogue of the ViT-VQGAN FFN):
 *   out[m, j] = silu(ab[m, j]) * ab[m, H + j]      ab: (M, 2H) = w12(x), out: (M, H)
 * replaces chunk -> silu -> mul of the SwiGLU the reference's FeedForward keywords describe
 * (models/vitvqgan.py:20-34).  Backward writes d_ab (M, 2H) = (d_a | d_b) in one pass
 * (no chunk-backward concatenation).  H must be a multiple of 4.
 * -------------------------------------------------------------------------- */
int amk_swiglu_fwd(const float* ab, int64_t M, int H, float* out, void* stream);
int amk_swiglu_bwd(const float* ab, const float* d_out, int64_t M, int H, float* d_ab, void* stream);

/* GEGLU gate of the transformer FFN (models/transformer.py:22-27): out[m, j] = gelu(ab[m, j]) * ab[m, H + j]
 * with the exact (erf) GELU of F.gelu; same layout and rules as the SwiGLU pair above. */
int amk_geglu_fwd(const float* ab, int64_t M, int H, float* out, void* stream);
int amk_geglu_bwd(const float* ab, const float* d_out, int64_t M, int H, float* d_ab, void* stream);

/* --------------------------------------------------------------------------
 * Optimizer step over flat gradient buckets.
 * Replaces, per phase of the train step, accelerator.clip_grad_norm_ + Adam.step + zero_grad
 * (trainers/vitgqgan.py:67-68,159-163,185-189; trainers/vit.py:29-31,74-76 for AdamW):
 *
 *   amk_sumsq_partials   partials[w] (w < amk_opt_num_partials()) = sum of x^2 over the w-th fixed range
 *   amk_adam_flat_step   norm = sqrt(sum of all `partials`), coef = min(1, max_norm / (norm + 1e-6))
 *                        (max_norm <= 0: no clipping), then for every element of an ACTIVE parameter
 *                          g *= coef; [Adam: g += wd * p | AdamW: p -= lr * wd * p]
 *                          m += (1 - beta1) (g - m); v = beta2 v + (1 - beta2) g^2
 *                          p -= tab.step_size * m / (sqrt(v) / tab.bc2_sqrt + eps)
 *                        and grad is left zeroed (all parameters).  torch.optim.Adam / AdamW arithmetic.
 * Buffers are flat fp32 arrays of n elements, n a multiple of 256; the i-th segment of 256 elements
 * belongs to parameter seg_param[i]; param_tab is (P, 4) floats per step: {active (0 / 1),
 * lr / (1 - beta1^t), sqrt(1 - beta2^t), weight-decay factor} with t the parameter's own step count.
 * `decoupled`: bit 0 selects AdamW (1) or Adam-with-L2 (0); bit 1 set = the decay factor is read per parameter from
 * param_tab[.][3] (wd for Adam, lr * wd for AdamW: per-group weight decay as trainers/muse.py:48-58 sets it up, and a
 * learning rate that lives on the device for a graph-captured step) instead of the `lr` / `weight_decay` arguments.
 * Inactive parameters (no gradient this step) are left untouched, as optimizers skip .grad None.
 * -------------------------------------------------------------------------- */
int amk_opt_num_partials(void);
int amk_sumsq_partials(const float* x, int64_t n, float* partials, void* stream);
int amk_adam_flat_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const int32_t* seg_param, const float* param_tab,
                       const float* partials, int n_partials,
                       float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int decoupled, float* norm_out, void* stream);
/* The same step, also refreshing param_bf16 (n bf16 values, 8-byte aligned; NULL: none): the copy of the parameters the
 * mixed-precision GEMMs read, so that torch.autocast's per-call weight casts (accelerator.autocast, trainers/
 * vitgqgan.py:139-190) have nothing left to do. */
int amk_adam_flat_step_shadow(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const int32_t* seg_param, const float* param_tab,
                              const float* partials, int n_partials,
                              float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int decoupled, float* norm_out, void* param_bf16, void* stream);

/* --------------------------------------------------------------------------
 * One step of the masked-token parallel decode.
 * Replaces, per step of MUSE.generate / MaskGitTransformer.generate (models/muse.py:211-236,
 * models/maskgit.py:255-272), the chain CFG combine -> softmax -> filter_logits (top-k, scatter) ->
 * gumbel_softmax(tau).argmax -> probs.gather -> masked writes over the (R, V) logits:
 *   s      = null + cfg_scale * (logits - null)       (null_logits NULL: s = logits)
 *   pred   = argmax over the `keep` largest s of (s + g)    (tau > 0; tau == 0 gives 0, as the reference's
 *            division by zero does -- every Gumbel-softmax entry is NaN and argmax returns 0)
 *   ids[r] = pred where mask[r] != 0 (mask NULL: every row)
 *   scores[r] = softmax(s)[pred]; rows with mask[r] == 0 get `unmasked_score` when it is >= 0
 * g: `gumbel` (R, V) when given, else -log(-log u) with u from Philox4x32-10 keyed by (seed, offset, row, j/4)
 * -- torch's distribution, not its stream.  R rows of V logits (V % 4 == 0, V <= 36864), contiguous.
 * -------------------------------------------------------------------------- */
int amk_sample_step(const float* logits, const float* null_logits, float cfg_scale,
                    const float* gumbel, uint64_t seed, uint64_t offset, float tau,
                    int64_t R, int V, int keep, const uint8_t* mask, float unmasked_score,
                    int64_t* ids, float* scores, void* stream);

/* --------------------------------------------------------------------------
 * Dense GEMM with split-bf16 products (the projections around the attention
 * core and the FFN -- nn.Linear in models/softmax_attention.py:30-42,80 and models/vitvqgan.py:20-61):
 *   C[m, n] = sum_k A[m, k] * W[n, k] (+ bias[n])        A (M, K), W (N, K), C (M, N), all f32, row-major
 * i.e. F.linear(A, W, bias); the input gradient dX = dY W is the same call with W^T in W's place.
 * Every operand is split into three bf16 parts (24 mantissa bits) and every product is the sum of six
 * exact partial products accumulated in f32 (v_mfma_f32_32x32x16_bf16): f32-level error (tests compare
 * with float64), 6/16 of the matrix-pipe time of the exact-f32 MFMA.  Its bound is the bf16 MFMA peak / 6.
 * The weight is split once per call by amk_gemm_x6_split into `planes` (amk_gemm_x6_planes_bytes(N, K)
 * bytes: three bf16 planes, K padded to 32); the activations are split inside the GEMM.
 * lda / ldw / ldc: row strides in elements (multiples of 4); K a multiple of 4; pointers 16-byte aligned.
 * -------------------------------------------------------------------------- */
int64_t amk_gemm_x6_planes_bytes(int N, int K);
int amk_gemm_x6_split(const float* W, int64_t ldw, int N, int K, void* planes, void* stream);
int amk_gemm_x6_nt(const float* A, int64_t lda, const void* w_planes, const float* bias,
                   float* C, int64_t ldc, int M, int N, int K, void* stream);

/* --------------------------------------------------------------------------
 * Dense exact-f32 GEMMs with the surrounding element-wise passes folded in.
 * Replaces, around the attention core and in the FFN of the ViT-VQGAN blocks, nn.Linear forward / backward
 * (models/softmax_attention.py:30-42,80; models/vitvqgan.py:20-34) together with nn.LayerNorm applied to its
 * input (models/vitvqgan.py:44-48), the residual add behind it (:50-61), the SwiGLU gate between w12 and w3 and
 * the bias gradients.  All operands f32 row-major; products on v_mfma_f32_32x32x2_f32 (exact f32).
 *
 *   op AMK_GEMM_NT   C[m, n] = sum_k A'[m, k] W[n, k] (+ bias[n]) (+ resid[m, n])        F.linear(A', W, bias)
 *        A (m x k, lda), W (n x k, ldw), C (m x n, ldc).  A' = ((A - ln_mean[m]) * ln_rstd[m]) * ln_gamma[k] +
 *        ln_beta[k] when ln_mean != NULL (LayerNorm applied while the tile is staged), else A.
 *        split > 0: output columns [split, n) use (w2, ldw2, bias2, c2, ldc2) instead -- two projections of the
 *        same input in one launch; split a multiple of 128.
 *        epilogue AMK_EPI_BIAS, AMK_EPI_RESID (adds resid (m x n, ldr); resid may be c itself), or
 *        AMK_EPI_SWIGLU: W is w12 (2n x k), bias (2n) or NULL; gate[m, j] = silu(a) * b with a = column j,
 *        b = column n + j of A' W^T + bias; gate (m x n, ldg) is written, and c (m x 2n, ldc) = (a | b) too
 *        unless c == NULL.
 *   op AMK_GEMM_NN   C[m, n] = sum_k A[m, k] W[k, n]                                     dX = dY W
 *        A (m x k, lda), W (k x n, ldw).  split > 0: contraction indices [split, k) read (a2, lda2) column
 *        k - split and (w2, ldw2) row k - split.
 *        epilogue AMK_EPI_BIAS (bias must be NULL: plain store), or AMK_EPI_SWIGLU_BWD: the product is dGate
 *        (m x n); ab (m x 2n, ldab) is the forward's (a | b); c (m x 2n, ldc) = (dA | dB) =
 *        (dGate * b * silu'(a) | dGate * silu(a)).
 *   op AMK_GEMM_TN   C[n, k] = sum_m Y[m, n] X'[m, k]                                    dW = dY^T X', db = colsum(dY)
 *        a = Y (m x n, lda), w = X (m x k, ldw), c (n x k, ldc); X' = LayerNorm(X) as above when ln_mean != NULL
 *        (ln_gamma / ln_beta indexed by k).  split > 0: output rows [split, n) read Y = (a2, lda2) column
 *        n - split and are written to (c2, ldc2); split a multiple of 128.  dbias (n) != NULL: column sums of Y.
 *        The m rows are cut into chunks; partial tiles go through `workspace` (amk_gemm_f32_ws_bytes(d) bytes,
 *        16-byte aligned) and are summed in chunk order: bitwise reproducible.
 * k and every leading dimension multiples of 4, pointers 16-byte aligned, operand panels below 1 GiB.
 * amk_row_stats: mean and rstd = 1 / sqrt(var + eps) (biased variance, two-pass) of every row of x (M x D),
 * the statistics nn.LayerNorm uses; D a multiple of 4, at most 4096.
 * -------------------------------------------------------------------------- */
enum { AMK_GEMM_NT = 0, AMK_GEMM_NN = 1, AMK_GEMM_TN = 2 };
enum { AMK_EPI_BIAS = 0, AMK_EPI_RESID = 1, AMK_EPI_SWIGLU = 2, AMK_EPI_SWIGLU_BWD = 3 };
typedef struct amk_gemm_desc {
  int32_t op, epilogue;
  int64_t m;
  int32_t n, k, split, reserved;
  const float *a, *a2, *w, *w2;
  float *c, *c2;
  int64_t lda, lda2, ldw, ldw2, ldc, ldc2;
  const float *bias, *bias2, *resid;
  int64_t ldr;
  const float *ln_mean, *ln_rstd, *ln_gamma, *ln_beta;
  const float* ab;
  int64_t ldab;
  float* gate;
  int64_t ldg;
  float* dbias;
} amk_gemm_desc;
int64_t amk_gemm_f32_ws_bytes(const amk_gemm_desc* d);
int amk_gemm_f32(const amk_gemm_desc* d, void* workspace, int64_t ws_bytes, void* stream);
int amk_row_stats(const float* x, int64_t M, int D, float eps, float* mean, float* rstd, void* stream);

/* --------------------------------------------------------------------------
 * Softmax attention on bf16 tensors (the reference's shipped training precision: cfg/vitvqgan.yaml:73 runs
 * models/softmax_attention.py:62-76 under bf16 autocast -- projections and both einsums in bf16, softmax in f32).
 * q, k, v, o, d_o, dq, dk, dv are bf16 (2-byte elements, strides in ELEMENTS, multiples of 8, 16-byte aligned);
 * contractions on v_mfma_f32_32x32x16_bf16 with f32 accumulation; scores, softmax and stats (B,H,I,2) are f32 as in
 * amk_attn_fwd.  Head dim 64.  key_mask (B, J) 1 = keep and causal_mask (I, J) 1 = masked, bytes, either may be NULL:
 * the masked_fill(-1e9) semantics of amk_attn_fwd (models/softmax_attention.py:65-71; a fully masked row keeps uniform
 * weights), as template variants of the same kernels -- what Muse's padded text prompts (models/muse.py:88-96) and every
 * masked call under the reference's autocast need.
 * Backward: ws = amk_attn_bf16_bwd_ws_floats(B,H,I,J) floats (four row constants per query and per-key-block dq
 * partials, summed in key-block order by a second launch: no atomics, bitwise reproducible).
 * -------------------------------------------------------------------------- */
int amk_attn_bf16_fwd(const void* q, const void* k, const void* v, void* o, float* stats,
                      const uint8_t* key_mask, const uint8_t* causal_mask,
                      int B, int H, int I, int J, int D,
                      int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                      int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                      float scale, void* stream);
int64_t amk_attn_bf16_bwd_ws_floats(int B, int H, int I, int J);
int amk_attn_bf16_bwd(const void* q, const void* k, const void* v, const void* o, const float* stats, const void* d_o,
                      void* dq, void* dk, void* dv, float* ws,
                      const uint8_t* key_mask, const uint8_t* causal_mask,
                      int B, int H, int I, int J, int D,
                      int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st, int64_t k_sh,
                      int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st, int64_t o_sh,
                      int64_t do_sb, int64_t do_st, int64_t do_sh, int64_t dq_sb, int64_t dq_st, int64_t dq_sh,
                      int64_t dk_sb, int64_t dk_st, int64_t dk_sh, int64_t dv_sb, int64_t dv_st, int64_t dv_sh,
                      float scale, void* stream);

/* --------------------------------------------------------------------------
 * Element-wise passes of the mixed-precision (bf16 autocast) mode: the ops of amk_swiglu_* / amk_add_layernorm_*
 * (models/vitvqgan.py:20-61) reading and writing bf16 where the neighbouring GEMM is a bf16 GEMM, so that no separate
 * cast pass runs (cfg/vitvqgan.yaml:73: nn.Linear in bf16, LayerNorm in f32).  Arithmetic in f32.
 *   swiglu_bf16   : ab (M, 2H) bf16 -> out (M, H) bf16;  backward: ab, d_out bf16 -> d_ab (M, 2H) bf16
 *   mixed LN fwd  : h = x (+ res): x f32 or bf16 (x_is_bf16), res f32 or NULL; h f32 (may be NULL without res and with
 *                   f32 x), y = LN(h) * gamma + beta written as bf16, mean / rstd f32
 *   mixed LN bwd  : dy bf16 or f32 (dy_is_bf16), h f32 -> dh f32 (+ dh_in), dh_bf16 = the same rounded to bf16 (or NULL);
 *                   dgb_part as amk_add_layernorm_bwd
 * -------------------------------------------------------------------------- */
int amk_swiglu_bf16_fwd(const void* ab, int64_t M, int H, void* out, void* stream);
int amk_swiglu_bf16_bwd(const void* ab, const void* d_out, int64_t M, int H, void* d_ab, void* stream);
int amk_add_layernorm_mixed_fwd(const void* x, int x_is_bf16, const float* res, const float* gamma, const float* beta,
                                int64_t M, int D, float eps, float* h, void* y_bf16, float* mean, float* rstd, void* stream);
int amk_add_layernorm_mixed_bwd(const void* dy, int dy_is_bf16, const float* h, const float* dh_in, const float* gamma,
                                const float* mean, const float* rstd, int64_t M, int D, float* dh, void* dh_bf16,
                                float* dgb_part, void* stream);

/* Weight and bias gradient of nn.Linear in the mixed-precision mode (csrc/gemm_bf16.hip): c[n, k] = sum_m y[m, n] x[m, k]
 * with y (M, N) and x (M, K) in bf16 as the autocast GEMMs leave them, c (N, K) and dbias (N, optional: column sums of y)
 * in f32 as the parameters' gradients are kept.  Replaces the autograd backward of nn.Linear under torch.autocast
 * (reference: models/softmax_attention.py:30-42,80, models/vitvqgan.py:20-34 under cfg/vitvqgan.yaml:73).  N, K, ldy, ldx
 * multiples of 8, ldc of 4; sums in a fixed order (bitwise reproducible).  workspace: amk_gemm_tn_bf16_ws_bytes(). */
int64_t amk_gemm_tn_bf16_ws_bytes(int64_t M, int N, int K);
int amk_gemm_tn_bf16(const void* y, int64_t ldy, const void* x, int64_t ldx, float* c, int64_t ldc, float* dbias,
                     int64_t M, int N, int K, void* workspace, int64_t ws_bytes, void* stream);

/* Forward and input gradient of nn.Linear in the mixed-precision mode (csrc/gemm_bf16.hip), bf16 in and out, f32
 * accumulation, the f32 bias added before the one rounding:
 *   op 0: c (M, N) = a (M, K) w^T + bias, w (N, K) as nn.Linear stores it;   op 1: c (M, N) = a (M, K) w, w (K, N) (dX = dY W)
 *   epi 1 (op 0): the SwiGLU gate of models/vitvqgan.py:20-34 folded in: w = w12 (2 H, K), g (M, H) = silu(a-half) * b-half,
 *   c (M, 2 H) optional (NULL: the pre-activations are not written).
 * N, K and the leading dimensions multiples of 8 (N of 16 with epi 1); pointers 16-byte aligned; bias may be NULL. */
int amk_gemm_bf16(int op, int epi, const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                  void* c, int64_t ldc, void* g, int64_t ldg, int64_t M, int N, int K, void* stream);

/* dab (M, 2 H) = (dA | dB): the backward of the SwiGLU gate applied to dG = dy (M, K) w3 (K, H), with the forward's
 * ab (M, 2 H) = (a | b) -- the input gradient of the FFN's second projection and the gate's backward in one launch
 * (models/vitvqgan.py:20-34, backward).  dG is rounded to bf16 before the gate's arithmetic (f32), exactly as
 * amk_gemm_bf16(op 1) followed by amk_swiglu_bf16_bwd would; H, K and the leading dimensions multiples of 8. */
int amk_gemm_bf16_swiglu_bwd(const void* dy, int64_t lddy, const void* w3, int64_t ldw, const void* ab, int64_t ldab,
                             void* dab, int64_t lddab, int64_t M, int H, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMK_H_ */
