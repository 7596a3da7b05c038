/*
 * bevrender_hip.h -- C ABI of libbevrender_hip.so: the MI355X (gfx950) kernels behind the
 * BEV-lift + correlation hot path of rpl-cmu/bevrender.
 *
 * The reference has no FFI: its boundary is Python nn.Module.forward.  The
 * Python modules in bevrender_amd/model/ mirror that interface and bind these entry points with
 * ctypes (bevrender_amd/_lib.py); INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch CUDA storage) unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls are asynchronous
 *     on that stream, allocate nothing and never synchronise (graph-capture safe);
 *   - return value: 0 on success, a negative BEVR_E_* code when the arguments violate the stated
 *     contract (nothing is launched), a positive hipError_t if the launch itself failed;
 *   - tensors are dense row-major with the layouts written next to each argument.
 *
 * Reference functions replaced (paths relative to the reference repo):
 *   bevr_project_bev_grid[_masked]  model/bev_cmr_proj.py:61-124  (bev_grid_to_camera + get_in_bound_mask)
 *   bevr_sample_fwd/bwd     F.grid_sample at model/SCA_deform_attn.py:290-301, model/TSA_deform_attn.py:210-217
 *   bevr_attn_fwd/bwd_*     model/SCA_deform_attn.py:331-413, model/TSA_deform_attn.py:245-333
 *                           (QK^T*scale + bilinear RPE bias + softmax + PV, never materialised)
 *   bevr_attn_cell_*        the same lines, for key segments sorted by rpe-table cell (bias as a matrix product)
 *   bevr_pack_kv/unpack_dkv model/SCA_deform_attn.py:312-321 (projection outputs -> per-head operand layouts)
 *   bevr_dwconv_fwd/bwd_w   model/encoder.py:363-411, model/model_utils.py:6-35 (depthwise 3x3 of the layer glue)
 *   bevr_offset_head_fwd/bwd  model/SCA_deform_attn.py:56-77, model/TSA_deform_attn.py:54-68 (offset heads, fused)
 *   bevr_layernorm_fwd/bwd  model/model_utils.py:37-49, model/encoder.py:275 (LayerNormProxy)
 *   bevr_merge_views_fwd/bwd  model/SCA_deform_attn.py:415-420, model/TSA_deform_attn.py:325-333 (view concat before proj_out)
 *   bevr_kv_project         model/SCA_deform_attn.py:290-321, model/TSA_deform_attn.py:210-236 (sample + proj_k | proj_v + pack)
 *   bevr_key_positions_fwd/bwd  model/SCA_deform_attn.py:248-277, model/TSA_deform_attn.py:170-196 (row split, tanh range, + ref, key order)
 *   bevr_affine_warp_fwd/bwd  model/encoder.py:413-466 (project_history_bev_feat: torchvision F.affine, twice)
 *   bevr_corr_fwd/bwd       train.py:554 (2 - 2 cam map^T) and the pairwise distance inside
 *                           loss/contrastive_loss.py:10-19 / loss/lift_loss.py:13-22
 */
#ifndef BEVRENDER_HIP_H
#define BEVRENDER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: bevr_attn_fwd writes TWO LSE planes (round 2 changed that under version 2: a version-2 caller's [n_prob][heads][Mp]
 *    buffer is too small), the key workspace carries group boxes, bevr_attn_bwd_q takes grad_scale; new: bevr_attn_cell_*,
 *    problem strides of bevr_pack_kv / bevr_unpack_dkv, BEVR_PREC_F16, grad_scale[8] for every backward entry point. */
/* 4: new entry points bevr_attn_tap_* (the projector-pinned keys without K / V), bevr_attn_gather_fwd (the forward over
 *    scattered keys with the bias on the matrix cores) and bevr_attn_*_dropout; nothing else
 *    changed. */
/* 5: new entry points bevr_attn_slab_ws_bytes / _slab_prep / _slab_bwd_q (the query-side backward cut along the rpe
 *    table: a workgroup owns a slab of table columns); bevr_attn_tap_bwd_k's table operand is declared as what it always
 *    was (the plain packed table, not the pair table); bevr_kv_project takes the channel-group count (before `stream`);
 *    nothing else changed. */
#define BEVR_ABI_VERSION 6

enum {
  BEVR_OK = 0,
  BEVR_E_NULL = -1,      /* a required pointer is NULL */
  BEVR_E_SHAPE = -2,     /* a dimension violates the contract below */
  BEVR_E_PRECISION = -3, /* unknown precision code */
  BEVR_E_ALIGN = -4      /* a pointer is not 16-byte aligned */
};

/* BEVR_PREC_F16: fp16 operands (v_mfma_f32_32x32x16_f16), f32 accumulate -- BASELINE config 5.  fp16 has 5 exponent bits:
 * the kernels keep softmax weights and logit gradients inside its normal range with power-of-two scales (forward: the
 * softmax reference sits 10 binades below the running maximum; backward: grad_scale[2..5]). */
/* BEVR_PREC_BF16X3: the fp32-tolerance mode that is not bound by the f32 matrix rate.  Every matrix operand is split
 *   x = hi + lo, hi = bf16(x), lo = bf16(x - hi) (round to nearest), and a product runs as three bf16 MFMAs
 *   (lo*hi + hi*lo + hi*hi; the lo*lo term, <= 2^-16 of the product, is dropped); everything per pair -- bias taps, f32
 *   table window, softmax -- is BEVR_PREC_F32's arithmetic.  Results agree with BEVR_PREC_F32 to ~1e-5 relative.
 *   Packed operands (Q, K, V, dO and their transposes) keep BEVR_PREC_F32's shapes, strides and byte sizes (E = float
 *   as a container) but hold the two bf16 planes:
 *     row layout, per 16-element half [e0..e15] of a 32-element row (64 bytes):
 *         hi(e0..e7) | hi(e8..e15) | lo(e0..e7) | lo(e8..e15)
 *     transposed (bits 2 <-> 3 swapped) layout, per block of 32 (128 bytes = 8 chunks of 16 bytes), y[q] = the block in
 *     the f32 mode's order:   chunk 2 h + s = hi(y[16 s + 8 h .. + 7]),   chunk 4 + 2 h + s = lo(same),   h, s in {0, 1}
 *   bevr_pack_kv writes K, V in this format; bevrender_amd/ops.py (_split_rows, _split_perm_t) is the definition the
 *   tests compare it with and what packs Q and dO. */
enum { BEVR_PREC_F32 = 0, BEVR_PREC_BF16 = 1, BEVR_PREC_F16 = 2, BEVR_PREC_BF16X3 = 3 };

int bevr_abi_version(void);
/* Human-readable text for a BEVR_E_* code (static storage). */
const char* bevr_strerror(int code);

/* ------------------------------------------------------------------------------------------------
 * Attention core geometry shared by the three attention kernels.
 *
 * Queries are the S x S BEV grid.  The kernels index them column-major with the rows of a column
 * padded to Sp = 32*ceil(S/32): packed query index  mq = j*Sp + i  (i = BEV row, j = BEV column),
 * Mp = S*Sp.  Rows i >= S are padding (zero on input, ignored on output).
 * Keys: N real keys padded to Np (multiple of 64); keys >= N are masked inside the kernels.
 * head_dim is fixed at 32 (the reference's dims/heads always give 32; smaller heads are zero-padded
 * by the caller).  heads % groups == 0; the heads of one group share key positions.
 *
 * Relative-position bias: the reference samples rpe_table[h, Ht=2S-1, Wt] bilinearly at
 *   ty = i + a_n,         a_n = (1 - py_n) * (S-1)/2
 *   tx = j * rx + b_n,    b_n = (1 - px_n) * (Wt-1)/4,   rx = (Wt-1) / (2(S-1))
 * (algebraically identical to grid_sample(align_corners=True) of (q_grid - pos)/2).
 * The table is handed over transposed, zero-padded and pair-packed:
 *   table_pair[h][Wp][Hp][2] = ( T2[y - y_off][x - x_off], T2[y + 1 - y_off][x - x_off] ),  T2 = T * log2(e),
 * zero outside the real table, with  y_off = Sp + 2, Hp = Ht + 2*Sp + 4,
 * x_off = (Wt-1)/2 + 4, Wp = 2*Wt + 8  (bevr_attn_table_dims computes them).
 * Keys carry key_a[n] = a_n, key_b[n] = b_n (float, any value; the kernels clamp to the padded range,
 * where every tap is zero exactly as grid_sample's zero padding).
 * Q is pre-multiplied by head_dim^-0.5 * log2(e); logits and LSE are in log2 units.
 * ---------------------------------------------------------------------------------------------- */
typedef struct bevr_attn_desc {
  int32_t n_prob;    /* B' = batch * views: independent softmax problems per head            */
  int32_t q_div;     /* query batch index = prob / q_div (views of one sample share the query) */
  int32_t heads;     /* h                                                                    */
  int32_t groups;    /* g                                                                    */
  int32_t S;         /* BEV side                                                             */
  int32_t Sp;        /* padded rows per column, 32*ceil(S/32)                                */
  int32_t N;         /* real keys                                                            */
  int32_t Np;        /* padded keys, multiple of 64                                          */
  int32_t Ht, Wt;    /* rpe table height (must equal 2S-1) and width                         */
  int32_t Hp, Wp;    /* padded table dims                                                    */
  int32_t y_off, x_off;
  int32_t precision; /* BEVR_PREC_F32: exact-f32 MFMA; BEVR_PREC_BF16 / BEVR_PREC_F16: 16-bit operands, f32 accumulate */
  int32_t reserved;
} bevr_attn_desc;

/* Fill Sp, Hp, Wp, y_off, x_off from S, Wt.  Returns 0 or BEVR_E_SHAPE. */
int bevr_attn_table_dims(bevr_attn_desc* d);

/* Key preparation, once per attention call (shared by bevr_attn_fwd and bevr_attn_bwd_q of the same keys);
 * replaces the query-grid / key-position displacement of model/SCA_deform_attn.py:352-378 and
 * model/TSA_deform_attn.py:264-291 (reference), reduced to per-key table coordinates:
 * clamps key_a/key_b, splits them into table row / fraction / column parts and reduces, per 64-key step, the
 * bounding box of the table taps the step needs.  key_ws: caller-allocated device buffer of
 * bevr_attn_key_ws_bytes(d) bytes, 16-byte aligned; its content is opaque to the caller.
 *   key_a, key_b [n_prob*groups][Np] float */
size_t bevr_attn_key_ws_bytes(const bevr_attn_desc* d);
int bevr_attn_key_prep(const bevr_attn_desc* d, const float* key_a, const float* key_b, void* key_ws, void* stream);

/* Forward.  Element type E = float (F32), bf16 (BF16) or fp16 (F16).
 *   Q   [n_prob/q_div][heads][Mp][32]  E     K  [n_prob][heads][Np][32] E
 *   Vt  [n_prob][heads][32][Np] E, keys permuted inside each aligned block of 32: the key with
 *       in-block index r is stored at position (r & 19) | ((r & 4) << 1) | ((r & 8) >> 1)
 *       (bits 2 and 3 swapped) -- the order the MFMA consumes the P accumulator in.
 *   key_ws: output of bevr_attn_key_prep for these keys      table_pair as above, float
 *   O   [n_prob][heads][Mp][32] float (normalised)
 *   LSE [2][n_prob][heads][Mp] float: plane 0 the log2-sum-exp of the row (what the backward entry points read:
 *       they take a pointer to plane 0), plane 1 an upper bound of log2 of the row's largest softmax weight (may exceed 0: clamp;
 *       rows past the grid: unspecified), from which the caller may tighten grad_scale of bevr_attn_bwd_q */
int bevr_attn_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                  const void* key_ws, const float* table_pair,
                  float* O, float* LSE, void* stream);

/* Backward, query side (same tiling as the forward):  dQ [like Q, float] and the table gradient
 *   dtable [heads][Wp][Hp+1] float, transposed/padded like table_pair but one value per entry;
 *   it is ACCUMULATED into (caller zeroes it).
 *   V  [n_prob][heads][Np][32] E (row layout), Kt [like Vt] E, dO [n_prob][heads][Mp][32] E,
 *   delta [n_prob][heads][Mp] float = rowsum(dO * O).
 *   grad_scale [8] float (device): { s, 1/s, kp, c2, 1/(2^kp c2), 2^-kp, 0, 0 }.  s: a power of two such that s * max|P (dP - delta)| <= 2^30; a
 *   valid bound is Pmax * (max_q |dO_q| * max_n |V_n| + max |delta|) (Euclidean norms over the 32 channels), Pmax = 1
 *   or the largest softmax weight of the launch (2^max of LSE plane 1, with a margin for the recomputation).  The
 *   kernel multiplies dO and delta by s as it loads them (exact) and accumulates the table gradient in 64-bit fixed
 *   point with unit ln2 / s (each contribution rounded to nearest; sums are exact and order-independent within a
 *   workgroup's window, and a cell cannot wrap: 2^33 contributions of the largest size fit); ln2 / s is applied
 *   when a cell is flushed and when dQ is stored.
 *   Entries 2..5 are read in BEVR_PREC_F16 mode only, by every backward entry point (the others take the same array
 *   and ignore it in the other modes): kp, an integer-valued exponent with Pmax 2^kp <= 2^14 -- the kernels form
 *   P' = P 2^kp so that fp16 holds the softmax weights in its normal range; c2, a power of two with
 *   max|P' (dP - delta) c2| <= 2^14 -- the logit gradient as fp16 operand; then their inverses.  bevr_attn_bwd_q
 *   additionally needs s = 2^16 2^kp c2 there (its fixed-point cells then count units of 2^-16 of an fp16 operand,
 *   and s max|P (dP - delta)| <= 2^30 as in the other modes); dO and delta are NOT pre-scaled by s in this mode.
 *   All gradients are with respect to the log2-domain logits' inputs as handed in (Q pre-scaled,
 *   table pre-multiplied): the caller's autograd undoes the scaling. */
int bevr_attn_bwd_q(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                    const void* key_ws, const float* table_pair,
                    const void* dO, const float* LSE, const float* delta, const float* grad_scale,
                    float* dQ, float* dtable, void* stream);

/* Backward, key side:  dK, dV [n_prob][heads][Np][32] float, dkey_a, dkey_b [n_prob*groups][Np] float
 *   (dkey_* are ACCUMULATED over the heads of a group; caller zeroes them).
 *   Qt, dOt: [..][heads][32][Mp] E with the same in-32 permutation as Vt (over the packed query index). */
int bevr_attn_bwd_k(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                    const float* key_a, const float* key_b, const float* table_pair,
                    const void* dO, const void* dOt, const float* LSE, const float* delta, const float* grad_scale,
                    float* dK, float* dV, float* dkey_a, float* dkey_b, void* stream);

/* Attention dropout (nn.Dropout on the softmax weights: model/SCA_deform_attn.py:155,402-409,
 * model/TSA_deform_attn.py:90,313-323): the three entry points above with a keep mask.  The mask is not stored: pair
 * (problem-head ph = prob * heads + head, packed query mq, key n) is kept iff
 *     (mix(seed ^ ph * 0x9E3779B1 ^ mq * 0x85EBCA77 ^ n * 0xC2B2AE3D) >> 16) >= drop_thr,
 *     mix(x): x ^= x >> 16; x *= 0x7FEB352D; x ^= x >> 15; x *= 0x846CA68B; x ^= x >> 16      (32-bit arithmetic)
 * with drop_thr = round(p * 65536) < 65536, and a kept weight is scaled by 65536 / (65536 - drop_thr).  The forward
 * masks the weights AFTER the normalisation (LSE is of the unmasked logits, as the reference's softmax-then-dropout);
 * the backward entry points evaluate the same function (same seed): dS = P (D dP - delta), dV from D P.  delta is
 * still rowsum(dO * O).  bevrender_amd/ops.py:dropout_keep_mask is the host twin.  Region kernels only: a caller with
 * dropout keeps every key on these entry points (the cell / tap entry points have no mask). */
int bevr_attn_fwd_dropout(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                          const void* key_ws, const float* table_pair, float* O, float* LSE,
                          unsigned drop_thr, unsigned drop_seed, void* stream);
int bevr_attn_bwd_q_dropout(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                            const void* key_ws, const float* table_pair, const void* dO, const float* LSE,
                            const float* delta, const float* grad_scale, float* dQ, float* dtable,
                            unsigned drop_thr, unsigned drop_seed, void* stream);
int bevr_attn_bwd_k_dropout(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                            const float* key_a, const float* key_b, const float* table_pair, const void* dO,
                            const void* dOt, const float* LSE, const float* delta, const float* grad_scale,
                            float* dK, float* dV, float* dkey_a, float* dkey_b, unsigned drop_thr, unsigned drop_seed,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Cell-sorted key segments: the same attention with the relative-position bias (and its table gradient) as a small
 * matrix product on the matrix cores (csrc/attn_cell.h).  Same descriptor, operand layouts, table, key workspace
 * (bevr_attn_key_prep) and gradient semantics as the entry points above.  Fast when the keys of every aligned run
 * of 32 span fewer than 4 table columns and 4 table rows including their second taps -- what sorting keys that
 * crowd a few table cells by cell gives (the pillar points a camera does not see are all pinned to pixel (0, 0),
 * model/bev_cmr_proj.py:76 of the reference: two thirds of an SCA view's keys) -- correct for any keys (runs that do
 * not fit are gathered per pair from the table in global memory).  A workgroup is one BEV column: a wave per 32-row
 * block + a producer wave, 16 at most, and a Q (+ dO) slot per wave in LDS: BEVR_E_SHAPE for Sp > 480, and from
 * bevr_attn_cell_bwd_q with float-sized operands (BEVR_PREC_F32, BEVR_PREC_BF16X3) for Sp > 256 (160 KB of LDS); the
 * caller then keeps every key on the region entry points (bevrender_amd/ops.py:attention_core does).
 *
 * One softmax over two key segments: run bevr_attn_fwd on the scattered keys, then bevr_attn_cell_fwd on the sorted
 * ones with (O_in, LSE_in) = the first call's (O, LSE plane 0): the result is the softmax over both.  O_in == NULL:
 * a single segment.  The backward entry points of both segments take the FINAL LSE and delta;
 * bevr_attn_cell_bwd_q ADDS its share to dQ and dtable (call it after bevr_attn_bwd_q, or on zeroed buffers).
 *   O_in [n_prob][heads][Mp][32] float, LSE_in [n_prob][heads][Mp] float; O, LSE as bevr_attn_fwd (LSE: 2 planes). */
int bevr_attn_cell_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* Vt,
                       const void* key_ws, const float* table_pair, const float* O_in, const float* LSE_in,
                       float* O, float* LSE, void* stream);
int bevr_attn_cell_bwd_q(const bevr_attn_desc* d, const void* Q, const void* K, const void* Kt, const void* V,
                         const void* key_ws, const float* table_pair, const void* dO, const float* LSE,
                         const float* delta, const float* grad_scale, float* dQ, float* dtable, void* stream);
/* dK, dV written; dkey_a, dkey_b ACCUMULATED (as bevr_attn_bwd_k). */
int bevr_attn_cell_bwd_k(const bevr_attn_desc* d, const void* Q, const void* Qt, const void* K, const void* V,
                         const void* key_ws, const float* table_pair, const void* dO, const void* dOt,
                         const float* LSE, const float* delta, const float* grad_scale, float* dK, float* dV,
                         float* dkey_a, float* dkey_b, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TAP entry points (csrc/attn_tap.h): the same attention for a key segment whose keys all SAMPLE INSIDE THE TOP-LEFT
 * 4 x 3 PIXELS of their feature map -- the pillar points a camera does not see: the projector pins them to pixel (0, 0)
 * (model/bev_cmr_proj.py:76) and the learned offset moves them by less than +-2.5 (Hi-1)/(Hk-1) x +-2.5 (Wi-1)/(Wk-1)
 * pixels (model/SCA_deform_attn.py:261-277).  For those keys K_n = sum_t w_t(n) Kpix_t + bk and V_n alike (bilinear
 * weights w_t over the 12 pixels t = 3 r + c; proj_k / proj_v are linear), so K and V are never formed:
 *   logits  S[n][q] = sum_t w_t(n) G[t][q] + Gb[q] + bias[n][q],     G[t][q] = Q_q . Kpix_t (Q pre-scaled), Gb = Q_q . bk
 *   output  O_q     = sum_t Rn[t][q] Vpix_t + bv,                    Rn = R / R[15],  R[t][q] = sum_n w_t(n) P[n][q]
 * The caller forms G, Gb (thin GEMMs) before and O after the launch, and merges the segment with the other keys of the
 * same softmax through (mref, R[15]): the segment's log2-sum-exp is mref[q] + log2 R[15][q].
 * Geometry, table, descriptor (groups == 1, precision BEVR_PREC_BF16 or BEVR_PREC_F16) as above; keys cell-sorted for
 * speed (any key set is handled: a 32-key run that does not fit one table chunk is processed in several masked passes).
 *   key_a, key_b, key_y, key_x [n_prob][Np] float: table coordinates as above; sampling position in FEATURE PIXELS
 *       ys = (py + 1)/2 (Hi - 1), xs likewise (ys < 3 and xs < 2 or outside the image: the caller's contract)
 *   tap_ws: bevr_attn_tap_ws_bytes(d) bytes, written by bevr_attn_tap_prep, opaque
 *   G   [n_prob][heads][Mp][16] E: slots 0..11 the taps; slots 12, 13 the hi and lo 16-bit parts of the row's logit
 *       offset c = Gb[q] - mref[q] (hi = E(c), lo = E(c - hi): the matrix product adds them against ones on the key side);
 *       slot 14 = -1e30 (E = bf16) / -60000 (fp16): the logit of a masked key; slot 15 zero
 *   mref [n_prob][heads][Mp] float IN/OUT: the softmax reference of each row, AS THE KERNEL SEES IT (Gb - (hi + lo)).  In:
 *       an upper bound of the row's logits minus a HEADROOM (bevrender_amd/ops.py uses max(0, max_t G) + Gb +
 *       max(0, max table) - 64).  The weights 2^(S - mref) <= 2^headroom are rounded to the operand type E before the R
 *       product, so the headroom is bounded per precision: BEVR_PREC_BF16 <= 100 (8 exponent bits), BEVR_PREC_F16 <= 8
 *       (5 exponent bits: 2^15.9 is fp16's largest value; the row sum over up to 2^17 keys is taken in f32.  A larger fp16
 *       headroom gives inf / NaN in R -- the kernel cannot see the headroom, the limit is the caller's to keep;
 *       tools/tap_check.py runs fp16 at 8, bevrender_amd/ops.py keeps fp16 calls off these entry points).  Out: the reference R is relative to -- unchanged unless
 *       every weight of the row underflowed against the bound (looser than ~190 binades), in which case the column is
 *       recomputed with an online maximum and its rows' mref are replaced.
 *   R   [n_prob][heads][Mp][16] float (written; rows 12, 13 equal row 15)   flags [n_prob*heads][S] int32, ZEROED by the
 *       caller (scratch)
 * ---------------------------------------------------------------------------------------------- */
size_t bevr_attn_tap_ws_bytes(const bevr_attn_desc* d);
int bevr_attn_tap_prep(const bevr_attn_desc* d, const float* key_a, const float* key_b, const float* key_y,
                       const float* key_x, void* tap_ws, void* stream);
int bevr_attn_tap_fwd(const bevr_attn_desc* d, const void* G, const void* tap_ws, const float* table_pair, float* mref,
                      float* R, int32_t* flags, void* stream);
/* Query-side backward.  With P[n][q] = exp2(S[n][q] - LSE[q]) (LSE: the log2-sum-exp over ALL key segments of the softmax):
 *     dS[n][q] = P[n][q] (sum_t w_t(n) H[t][q] + Hc[q]),     dG[k][q] = sum_n A[n][k] dS[n][q]   (A = the 16 slot weights)
 *   G as above with the offset c = Gb - LSE in slots 12, 13 (padding rows i >= S: c <= -1e30, their weights vanish)
 *   H  [n_prob][heads][Mp][16] E = ln2 * (dO_q . Vpix_t) in slots 0..11, the hi and lo parts of Hc = ln2 * (dO_q . bv - delta_q)
 *      in slots 12, 13, zero in 14, 15
 *   dG [n_prob][heads][Mp][16] float WRITTEN: slots 0..11 the gradient of G, slot 15 the gradient of Gb
 *   dtable as bevr_attn_bwd_q: ACCUMULATED (float atomics). */
int bevr_attn_tap_bwd_q(const bevr_attn_desc* d, const void* G, const void* H, const void* tap_ws,
                        const float* table_pair, float* dG, float* dtable, void* stream);
/* Key-side backward: the gradients of every key's table coordinates and sampling position (G, H as bevr_attn_tap_bwd_q),
 *   dkey_a, dkey_b, dkey_y, dkey_x [n_prob][Np] float, ACCUMULATED over the heads (caller zeroes them):
 *   d/d key_a, d/d key_b through the bias (bilinear derivative of the table), d/d key_y, d/d key_x through the tap
 *   weights -- both paths of it: the logits (G) and the values (sum_q P H / ln2: H carries ln2 dO . Vpix).  Kinks as F.grid_sample's backward
 *   (floor-based: the derivative of the tap pair the position sits between). */
/* NOTE the table operand: NOT the pair table of the other entry points but the PLAIN packed table
 *   table [heads][Wp][Hp + 1] float, table[h][x][y] = log2(e) * rpe_table[h][y - y_off][x - x_off], zero in the padding --
 *   the transposed, padded table the pair table is built from and the buffer `dtable` mirrors (row pitch Hp + 1).  The
 *   kernel differences it along both axes in hi + lo 16-bit parts; handing it the (h, Wp, Hp, 2) pair table gives silently
 *   wrong dkey_a / dkey_b (tests/test_gpu_tap.py::test_tap_bwd_k_through_the_c_abi_as_the_header_describes). */
int bevr_attn_tap_bwd_k(const bevr_attn_desc* d, const void* G, const void* H, const void* tap_ws,
                        const float* table, float* dkey_a, float* dkey_b, float* dkey_y, float* dkey_x, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gather kernel: the FORWARD of bevr_attn_fwd for SCATTERED keys with the relative-position bias on the matrix cores
 * (csrc/attn_gather_fwd.hip: a sparse contraction whose table side the lanes gather from an LDS window of the table;
 * any key set is handled; there is no gather backward: bevr_attn_bwd_q / _bwd_k or the slab entry point below take the
 * same operands).  BEVR_PREC_BF16 only, S <= 224.  Replaces model/SCA_deform_attn.py:331-413 and
 * model/TSA_deform_attn.py:245-333 (reference) like the entry points above.
 *   Q, K, key_ws, O, LSE: as bevr_attn_fwd        V [n_prob][heads][Np][32] bf16 rows (not transposed)
 *   table_pk [heads][Wp][Hp] dwords: the pair table in bf16, (T2[y - y_off][x - x_off], T2[y + 1 - y_off][x - x_off]),
 *            round-to-nearest of table_pair
 *   mref [n_prob][heads][Mp] float: the softmax reference of every row, as bevr_attn_tap_fwd's -- an upper bound of the
 *            row's logits minus a headroom (64 binades); rows whose weights all underflow against it (bound looser than
 *            ~160 binades) are recomputed with an online maximum: flags [n_prob * heads][S] int, zeroed by the caller,
 *            marks their columns (scratch)
 * LSE plane 1 is log2 of an upper bound (within 2 binades) of the row's largest softmax weight. */
int bevr_attn_gather_fwd(const bevr_attn_desc* d, const void* Q, const void* K, const void* V,
                         const void* key_ws, const void* table_pk, const float* mref, float* O, float* LSE,
                         int* flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SLAB entry points (csrc/attn_slab_bwd_q.hip): bevr_attn_bwd_q -- dQ and the rpe-table gradient of a key segment, same
 * arithmetic per (query, key) pair, same operands and gradient semantics -- with the work cut along the TABLE: a work item
 * owns a slab of <= 24 consecutive table columns of one (problem, head), all rows of them in LDS (values and 64-bit
 * fixed-point gradient cells), and processes the pairs whose left tap column floor(j rx + b_n) falls into the slab; the
 * keys are handed over SORTED BY b per problem, so those are one contiguous run per BEV column.  Every table cell leaves
 * the workgroup once per item (no moving window, no flush traffic to speak of); dQ leaves as float atomics.
 * 16-bit operand modes (BEVR_PREC_BF16, BEVR_PREC_F16), S <= 211; BEVR_E_SHAPE / BEVR_E_PRECISION otherwise (the caller
 * keeps bevr_attn_bwd_q: bevrender_amd/ops.py:slab_supported).  Any key set is handled.
 *   order  [n_prob*groups][N] int32: per problem-group, a permutation of 0..N-1 that sorts key_b ascending
 *   Ks, Vs [n_prob][heads][N][32] E: the K / V rows IN THAT ORDER (row t of (prob, head) = key order[pg][t]), unpadded
 *   slab_ws: bevr_attn_slab_ws_bytes(d) bytes (0: unsupported shape), written by bevr_attn_slab_prep, opaque; one
 *           bevr_attn_slab_bwd_q launch per prep (the launch consumes the work list's counter)
 *   Q, table_pair, dO, LSE, delta, grad_scale: as bevr_attn_bwd_q
 *   dQ [n_prob][heads][Mp][32] float ACCUMULATED with float atomics (caller zeroes; bevr_attn_bwd_q WRITES its dQ)
 *   dtable ACCUMULATED (as bevr_attn_bwd_q)
 * ---------------------------------------------------------------------------------------------- */
size_t bevr_attn_slab_ws_bytes(const bevr_attn_desc* d);
int bevr_attn_slab_prep(const bevr_attn_desc* d, const float* key_a, const float* key_b, const int32_t* order,
                        void* slab_ws, void* stream);
int bevr_attn_slab_bwd_q(const bevr_attn_desc* d, const void* Q, const void* Ks, const void* Vs, const void* slab_ws,
                         const float* table_pair, const void* dO, const float* LSE, const float* delta,
                         const float* grad_scale, float* dQ, float* dtable, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bilinear feature sampling, align_corners=True, zero padding (grid_sample semantics).
 *   feat [nb][Hi][Wi][C] float (channels-last)     pos [nb][N][2] float, (y, x) in [-1, 1] units
 *   out  [nb][N][C] float
 * backward: dfeat [nb][Hi][Wi][C] ACCUMULATED (caller zeroes), dpos [nb][N][2] written.
 * C must be a multiple of 4 and <= 1024.
 * ---------------------------------------------------------------------------------------------- */
int bevr_sample_fwd(const float* feat, const float* pos, float* out,
                    int nb, int Hi, int Wi, int C, int N, void* stream);
int bevr_sample_bwd(const float* feat, const float* pos, const float* dout, float* dfeat, float* dpos,
                    int nb, int Hi, int Wi, int C, int N, void* stream);
/* The same on a bf16 feature map (the backbone's output dtype in the bf16 configurations; `feat` = raw bf16 bits, same
 * shape): half the tap bytes, features stay bf16 in HBM.  out, dout, dfeat (the master gradient), dpos stay float. */
int bevr_sample_fwd_bf16(const void* feat, const float* pos, float* out,
                         int nb, int Hi, int Wi, int C, int N, void* stream);
int bevr_sample_bwd_bf16(const void* feat, const float* pos, const float* dout, float* dfeat, float* dpos,
                         int nb, int Hi, int Wi, int C, int N, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BEV pillar grid -> camera pixels (model/bev_cmr_proj.py:61-113).
 *   points_3d [4][P] float homogeneous IMU-frame points
 *   cam_inv   [ncam][4][4] float = inverse(imu_to_cam)       Kmat [ncam][3][3] float (already rescaled)
 *   out       [ncam][2][P] float, (x, y) normalised to [-1, 1]; points whose int-truncated pixel is
 *             outside [0, W-1) x [0, H-1) are pinned to pixel (0, 0) -> (-1, -1).
 * ---------------------------------------------------------------------------------------------- */
int bevr_project_bev_grid(const float* points_3d, const float* cam_inv, const float* Kmat, float* out,
                          int ncam, int P, int img_w, int img_h, void* stream);
/* The same with the optional grey-pixel mask of model/bev_cmr_proj.py:114-122 (remove_ref_in_gray):
 *   ref_img [ncam][ref_c][ref_h][ref_w] uint8 (device), one reference image per camera, ref_h >= img_h - 1,
 *   ref_w >= img_w - 1; a point whose truncated pixel has exactly three channels equal to 128 is pinned like an
 *   out-of-bound point.  ref_img == NULL: identical to bevr_project_bev_grid. */
int bevr_project_bev_grid_masked(const float* points_3d, const float* cam_inv, const float* Kmat, float* out,
                                 int ncam, int P, int img_w, int img_h, const uint8_t* ref_img, int ref_c,
                                 int ref_h, int ref_w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ground <-> aerial correlation: D[i][j] = 2 - 2 * <cam_i, map_j>  (train.py:554) on raw or
 * L2-normalised rows.   cam [n][E], map [m][E] float; D [n][m] float.
 *   normalize != 0: rows are L2-normalised first (the LpDistance(normalize_embeddings=True) of the
 *   retrieval losses); inv_norm_cam [n], inv_norm_map [m] receive 1/||row|| for the backward.
 * backward: dcam [n][E], dmap [m][E] written from dD [n][m].
 * ---------------------------------------------------------------------------------------------- */
int bevr_corr_fwd(const float* cam, const float* map, float* D, float* inv_norm_cam, float* inv_norm_map,
                  int n, int m, int E, int normalize, void* stream);
int bevr_corr_bwd(const float* cam, const float* map, const float* D, const float* dD,
                  const float* inv_norm_cam, const float* inv_norm_map, float* dcam, float* dmap,
                  int n, int m, int E, int normalize, void* stream);
/* Rank of the diagonal within its column (train.py:559-563): rank[k] = #{ i : D[i][k] < D[k][k] }. */
int bevr_recall_rank(const float* D, int32_t* rank, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depthwise k x k convolution, one filter per channel, stride 1, zero "same" padding, k odd and <= 5: the local
 * perception units and the MLP's 3x3 of the reference's EncoderLayer (model/encoder.py:363-411,
 * model/model_utils.py:6-35) and TSA's offset head (model/TSA_deform_attn.py:54-68).
 *   x, y  [B][H][W][C] (nhwc = 1) or [B][C][H][W] (nhwc = 0) float      w [C][k][k] float      bias [C] or NULL
 *   flip = 1 correlates with the flipped filter: the input gradient is bevr_dwconv_fwd(dy, w, NULL, dx, ..., flip = 1).
 * bevr_dwconv_bwd_w ACCUMULATES dw [C][k][k] and dbias [C] (dbias may be NULL); the caller zeroes them.
 * ---------------------------------------------------------------------------------------------- */
int bevr_dwconv_fwd(const float* x, const float* w, const float* bias, float* y,
                    int B, int H, int W, int C, int k, int nhwc, int flip, void* stream);
int bevr_dwconv_bwd_w(const float* x, const float* dy, float* dw, float* dbias,
                      int B, int H, int W, int C, int k, int nhwc, void* stream);
/* The MLP's  act(y + dwc(y))  of model/model_utils.py:51-59 (nn.GELU, erf form) with the depthwise 3 x 3, one kernel
 * (ABI 6; channels-last [B][H][W][C], C % 4 == 0, k = 3):
 *   mode 1  y = gelu(x + conv(x) + bias)                               the forward
 *   mode 2  y = aux * gelu'(x + conv(x) + bias)                        backward: gradient at the pre-activation (aux = d out)
 *   mode 3  y = x + conv^T(x)    (flipped filter, bias ignored)        backward: the input gradient from mode 2's result
 * The weight / bias gradients are bevr_dwconv_bwd_w(x, <mode 2's result>). */
int bevr_dwconv_res_gelu(const float* x, const float* w, const float* bias, const float* aux, float* y, int B, int H, int W,
                         int C, int k, int mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Operand packing for bevr_attn_* (the reshapes of model/SCA_deform_attn.py:312-321 / TSA_deform_attn.py:226-236:
 * proj_k / proj_v outputs -> per-head operands), one pass instead of a permute / pad / cast / transpose chain.
 *   k, v  rows (n_prob, N, heads*c) float with row stride ld floats (ld >= heads*c; K | V of one GEMM: ld = 2 heads c)
 *         and problem stride pstride ROWS (pstride >= N; a key segment of a longer row array: pstride = its row count)
 *   Kr, Vr [n_prob][heads][Np][32] E   row layout: head_dim c <= 32 zero padded, keys N..Np-1 zero (Np % 64 == 0)
 *   Kt, Vt [n_prob][heads][32][Np] E   transposed, bits 2 <-> 3 of the in-32 key index swapped (either may be NULL)
 *   E = bf16 (BEVR_PREC_BF16) or fp16 (BEVR_PREC_F16), round to nearest even, or float (BEVR_PREC_F32).
 * bevr_unpack_dkv is the adjoint on the gradients of the row layout: dK, dV [n_prob][heads][Np][32] float ->
 * dk, dv rows (n_prob, N, heads*c) with row stride ld (every element of the rows written).
 * ---------------------------------------------------------------------------------------------- */
int bevr_pack_kv(const float* k, const float* v, long long ld, long long pstride, int n_prob, int N, int Np, int heads,
                 int c, int precision, void* Kr, void* Vr, void* Kt, void* Vt, void* stream);
int bevr_unpack_dkv(const float* dK, const float* dV, float* dk, float* dv, long long ld, long long pstride, int n_prob,
                    int N, int Np, int heads, int c, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the channel axis of channels-last rows, forward and backward (LayerNormProxy,
 * model/model_utils.py:37-49; the norm an EncoderLayer shares between its four uses, model/encoder.py:275).
 *   x, y, dy, dx [rows][C] float;  gamma, beta [C];  mean, rstd [rows] (written by the forward, read by the backward)
 *   C % 4 == 0, C / 4 a power of two <= 64.   backward: dx written, dgamma / dbeta [C] ACCUMULATED (caller zeroes).
 * ---------------------------------------------------------------------------------------------- */
int bevr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       long long rows, int C, float eps, void* stream);
int bevr_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* mean, const float* rstd,
                       float* dx, float* dgamma, float* dbeta, long long rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The attention output on its way to proj_out (ABI 6): the softmax merge of two key segments and the unpacking of the
 * packed per-(problem, head) rows into the layout the projection contracts, one pass each way.  Replaces the
 * concatenation of the per-view outputs along channels before proj_out (model/SCA_deform_attn.py:415-420) and the
 * (B, S S, C) flattening of TSA (model/TSA_deform_attn.py:325-333); with two segments also the merge
 *     L = log2(2^L_r + 2^L_c),  O = 2^(L_r - L) O_r + 2^(L_c - L) O_c
 * of the halves of ONE softmax that two kernel families computed (segment r: bevr_attn_fwd / _gather_fwd (+ _cell_fwd),
 * segment c: bevr_attn_tap_fwd + the caller's R Vpix product).
 *   O_r, O_c, dO_r, dO_c  [n_prob][heads][S * Sp][32] float, row j * Sp + i (the attention kernels' O layout);
 *                         n_prob = B * views, problem b * views + v;  channels c .. 31 and rows i >= S: padding
 *   L_r, L_c, dL_r, dL_c  [n_prob][heads][S * Sp]  log2-sum-exp of the segment (LSE plane 0) and its gradient
 *   out, dout             [B][S * S][views * heads * c], row i * S + j, channel (v * heads + hh) * c + cc
 *   O_c == NULL: one segment (L_r, L_c, dL_r, dO_c, dL_c unused): unpack only.
 *   backward: every element of dO_r (dO_c, dL_r, dL_c) is WRITTEN, the padding with zeros.
 *   Sp a multiple of 32, Sp >= S; c a multiple of 4, <= 32; n_prob a multiple of views.
 * ---------------------------------------------------------------------------------------------- */
int bevr_merge_views_fwd(const float* O_r, const float* L_r, const float* O_c, const float* L_c, float* out, int n_prob,
                         int views, int heads, int S, int Sp, int c, void* stream);
int bevr_merge_views_bwd(const float* dout, const float* O_r, const float* L_r, const float* O_c, const float* L_c,
                         float* dO_r, float* dL_r, float* dO_c, float* dL_c, int n_prob, int views, int heads, int S,
                         int Sp, int c, void* stream);
/* The same with the tap segment's half formed on the way instead of passed in:  O_c = Rn Vp + bv  (csrc/attn_tap.h:
 * V_n = sum_t w_t(n) Vpix_t + bv, so the tap kernels' output is the per-pixel weight sum Rn, not O).
 *   Rn, dRn  [n_prob][heads][S * Sp][12] float (bevr_attn_tap_fwd's R, normalised by the caller)
 *   Vp, dVp  [n_prob][heads][12][32] float: the 12 pixels' value rows per head (channels c .. 31 zero);  bv, dbv [heads][32]
 *   dVp, dbv are ACCUMULATED (the caller zeroes them); everything else is written.  Other arguments as above. */
int bevr_merge_tap_fwd(const float* O_r, const float* L_r, const float* Rn, const float* L_c, const float* Vp, const float* bv,
                       float* out, int n_prob, int views, int heads, int S, int Sp, int c, void* stream);
int bevr_merge_tap_bwd(const float* dout, const float* O_r, const float* L_r, const float* Rn, const float* L_c, const float* Vp,
                       const float* bv, float* dO_r, float* dL_r, float* dRn, float* dL_c, float* dVp, float* dbv, int n_prob,
                       int views, int heads, int S, int Sp, int c, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The cotangent's way into the 16-bit backward kernels (ABI 6), one pass over dO and O:
 *   dO, O   [n_ph][Mp][32] float (n_ph = n_prob * heads): the output's cotangent and the forward's output
 *   scale   one float on the device or NULL: a power of two applied to dO (and to dLSE) before the rounding (fp16 mode)
 *   dLSE, LSE0  [n_ph][Mp] float or NULL: the cotangent of the log2-sum-exp output (a caller that merged this softmax with
 *           another segment sends one) and LSE plane 0 (rows with a non-finite LSE take no dLSE)
 *   dOe     [n_ph][Mp][32] E: the rounded rows -- the dO operand of bevr_attn_bwd_q / _slab_bwd_q / _cell_bwd_q / _bwd_k
 *   dOt     [n_ph][32][Mp] E: the same transposed, position p of a 32-block holding row perm32(p) (bits 2 and 3 of p
 *           swapped): the dOt operand of bevr_attn_bwd_k / _cell_bwd_k
 *   delta   [n_ph][Mp] float: rowsum(dOe o O) - log2(e) scale dLSE: the `delta` operand of every backward entry point
 *   stats   2 floats the caller zeroed: raised (atomic max) to max ||dOe row||^2 and max |delta| -- the bound behind
 *           grad_scale.   Mp % 32 == 0; precision BEVR_PREC_BF16 or BEVR_PREC_F16.
 * Replaces the autograd bookkeeping between the output's cotangent and the kernels (model/SCA_deform_attn.py:331-413).
 * ---------------------------------------------------------------------------------------------- */
int bevr_attn_bwd_prep(const float* dO, const float* O, const float* scale, const float* dLSE, const float* LSE0, void* dOe,
                       void* dOt, float* delta, float* stats, int n_ph, int Mp, int precision, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K | V operands straight from the feature map (16-bit operand modes): bilinear sampling at `pos` -> proj_k | proj_v
 * as one 1x1 GEMM on the matrix cores -> the packed layouts below, in one pass; the sampled features and the projected
 * rows never reach HBM.  Replaces bevr_sample_fwd -> GEMM -> bevr_pack_kv, i.e. F.grid_sample + proj_k / proj_v + the
 * per-head reshapes of model/SCA_deform_attn.py:290-321, model/TSA_deform_attn.py:210-236.
 *   feat [nb][Hi][Wi][C]  float, or bf16 bits if feat_bf16 (channels-last)
 *   groups: channel groups (ABI 5; the reference's n_groups): the C / groups channels of group gi are sampled at group gi's
 *        positions -- x.reshape(B g, C / g, Hi, Wi) against pos (B g, N, 2), model/SCA_deform_attn.py:290-301 -- and the
 *        projection then mixes all C channels as before.  C % groups == 0, (C / groups) % 4 == 0.
 *   pos  (y, x) of key n of problem b, group gi at pos[((b * groups + gi) * pos_pstride + n) * 2]   (a key segment:
 *        pass pos + 2 n0)
 *   Wkv  [2C][C] E (E = bf16 / fp16 per `precision`: the caller rounds the float weights once), rows 0..C-1 = proj_k
 *   bkv  [2C] float or NULL
 *   outputs as bevr_pack_kv: Kr, Vr [nb][heads][Np][32] E, Kt (or NULL), Vt [nb][heads][32][Np] E; keys N..Np-1 zero.
 *   vnorm2_max: one float the caller zeroed, or NULL: raised (atomic max) to the largest squared row norm of V -- the
 *            bound the backward's fixed-point scale needs (grad_scale), for free instead of a pass over V.
 *   knorm2_max (ABI 6): [nb][heads] floats the caller zeroed, or NULL: raised (atomic max) to the largest squared row
 *            norm of K per (problem, head) -- the gather / tap forward's static softmax reference needs max ||K_n||
 *            (bevr_attn_gather_fwd's mref), for free instead of a pass over K.  Both norms are of the UNROUNDED rows.
 *   C = heads * c, c <= 32, C % 16 == 0, C <= 256; Np % 64 == 0.
 * The adjoint stays unfused: bevr_unpack_dkv -> GEMMs -> bevr_sample_bwd on samples recomputed with bevr_sample_fwd.
 * ---------------------------------------------------------------------------------------------- */
int bevr_kv_project(const void* feat, int feat_bf16, const float* pos, long long pos_pstride, const void* Wkv,
                    const float* bkv, int nb, int Hi, int Wi, int C, int N, int Np, int heads, int c, int precision,
                    void* Kr, void* Vr, void* Kt, void* Vt, float* vnorm2_max, float* knorm2_max, int groups,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Key positions from the offset heads' outputs, in the attention's key order, and the adjoint (one launch each).
 * Replaces model/SCA_deform_attn.py:248-277 (row split of the (b g) d (h n) w head output, tanh * range, + reference
 * point -- or clamp --, per view) and model/TSA_deform_attn.py:170-196, plus the gather into the static key order.
 *   off   [V][P][block] float   one view's head output per problem p = b * G + gi (P = B * G):
 *           sca = 1: block = [S][S][D] (the fused head's channels-last output); component c (0 = y, 1 = x) of key
 *                    n = hk * (S D) + w * D + d is element [(2 hk + c)][w][d]           (N = (S / 2) * S * D)
 *           sca = 0: block = [N][2]   (y, x) per key-grid pixel
 *   ref   [V][N][2] float (y, x)   reference points (SCA) / the regular grid (TSA), in the UNordered key index
 *   order [V][N] int32 or NULL     static key order: output key n' is key order[v][n']
 *   pos   [B][V][G][N][2] float    use_tanh = 1: tanh(off) * (sy, sx) + ref;   0: clamp(off + ref, -1, 1)
 * backward: doff (off's shape) is WRITTEN (the order is a permutation: every element belongs to one key).
 * ---------------------------------------------------------------------------------------------- */
int bevr_key_positions_fwd(const float* off, const float* ref, const int* order, float* pos, int V, int P, int G, int N,
                           int sca, int S, int D, int use_tanh, float sy, float sx, void* stream);
int bevr_key_positions_bwd(const float* off, const float* ref, const int* order, const float* dpos, float* doff, int V,
                           int P, int G, int N, int sca, int S, int D, int use_tanh, float sy, float sx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Offset heads of the deformable attention blocks, fused per BEV pixel (model/SCA_deform_attn.py:56-77 conv_offset_m{v};
 * the LayerNorm -> GELU -> 1x1 tail of model/TSA_deform_attn.py:54-68):
 *   z[c*Mx + m] = x[c] * w0[c*Mx + m] + b0[c*Mx + m]   (depthwise 1x1, channel multiplier Mx; w0 == NULL: z = x, Mx = 1)
 *   out[d] = sum_k W3[d][k] * GELU_erf( LayerNorm_{Cg*Mx}(z)[k] * gamma[k] + beta[k] )
 *   x   [P pixels][xstride] float, the head's Cg <= 64 input channels first (a channel group of a channels-last tensor:
 *       pass the pointer to the group's first channel and xstride = the tensor's channel count)
 *   w0, b0, gamma, beta [Cg*Mx], W3 [Dout][Cg*Mx], out [P][Dout] float.  (Mx, Dout) = (1, 2) or Mx == Dout <= 8.
 * backward: dx [P][xstride] (may be NULL) and every parameter gradient are ACCUMULATED (caller zeroes them);
 *   dw0 / db0 may be NULL (TSA form).
 * ---------------------------------------------------------------------------------------------- */
int bevr_offset_head_fwd(const float* x, const float* w0, const float* b0, const float* gamma, const float* beta,
                         const float* W3, float* out, long long P, int Cg, int xstride, int Mx, int Dout, float eps,
                         void* stream);
int bevr_offset_head_bwd(const float* x, const float* w0, const float* b0, const float* gamma, const float* beta,
                         const float* W3, const float* dout, float* dx, float* dw0, float* db0, float* dgamma,
                         float* dbeta, float* dW3, long long P, int Cg, int xstride, int Mx, int Dout, float eps,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ego-motion warp of the history BEV (model/encoder.py:413-466): one torchvision-style affine resampling,
 * bilinear, zero padding, fill = 0 -- out = bilinear(img) * bilinear(ones) -- for a whole batch.
 *   img, out [B][C][H][W] float      theta [B][6] float (device): torchvision's INVERSE affine matrix in pixel units
 *   about the image centre, rows (m0 m1 m2; m3 m4 m5): the source pixel of output pixel (x, y) is
 *   (m0 X + m1 Y + m2 + (W-1)/2, m3 X + m4 Y + m5 + (H-1)/2), X = x - W/2 + 1/2, Y = y - H/2 + 1/2.
 * backward: dimg ACCUMULATED from dout with the same theta (caller zeroes dimg).
 * ---------------------------------------------------------------------------------------------- */
int bevr_affine_warp_fwd(const float* img, const float* theta, float* out, int B, int C, int H, int W, void* stream);
int bevr_affine_warp_bwd(const float* dout, const float* theta, float* dimg, int B, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVRENDER_HIP_H */
