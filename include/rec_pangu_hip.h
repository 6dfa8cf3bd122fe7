/*
 * rec_pangu_hip.h — C ABI of librecpangu_hip.so, the MI355X (gfx950) kernels behind the
 * rec_pangu ranking hot path.
 *
 * The reference (HaSai666/rec_pangu v0.4.1) is pure Python over ATen and has NO native
 * interface of its own.  Each entry point below therefore replaces one
 * ATen op *site* of the reference; the site is cited as `file:line` relative to the reference
 * root.  A maintainer binds these with ctypes (rec_pangu_amd/hip.py; INTEGRATION.md shows the
 * stub that goes into the reference's own modules).
 *
 * Conventions
 *   - plain C: raw device pointers + sizes, no torch / C++ types; `stream` is a hipStream_t
 *     passed as void* (0 = the null stream).  All calls are asynchronous on `stream`, never
 *     synchronise, never allocate: every buffer (outputs and workspaces) is owned by the caller.
 *   - pointers named *_ptrs are HOST arrays of DEVICE pointers (copied into the launch packet).
 *   - return 0 on success, <0 on failure (RP_ERR_*); rp_last_error() gives the message for the
 *     calling thread.  Nothing throws or aborts.
 *   - fp32 storage, fp32 accumulation.  Matrix products run on the bf16 matrix core (v_mfma_f32_32x32x16_bf16)
 *     over split-bf16 pieces of the fp32 operands; the number of partial products per flop is the library's
 *     matmul precision (rp_set_matmul_precision, default RP_MATMUL_AUTO: six products = fp32-faithful for
 *     HBM-bound launches, three for matrix-core-bound ones; RP_MATMUL_FP32 = the exact-fp32
 *     v_mfma_f32_32x32x2_f32 reference kernels).
 *   - callable from any host thread (autograd's backward thread included).
 */
#ifndef REC_PANGU_HIP_H
#define REC_PANGU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_OK 0
#define RP_ERR_ARG (-1)         /* bad size / alignment / null pointer            */
#define RP_ERR_LAUNCH (-2)      /* hipGetLastError() after a launch                */
#define RP_ERR_UNSUPPORTED (-3) /* shape outside what the kernels were built for   */

#define RP_MAX_FIELDS 64 /* sparse fields (and dense columns) per launch            */

/* epilogue selectors for rp_linear_fwd */
#define RP_ACT_NONE 0
#define RP_ACT_RELU 1
#define RP_ACT_MASK 2 /* out = acc * (aux > 0): ReLU backward fused into a dgrad GEMM */
/* round 5: the other activations rec_pangu/models/layers/activation.py:37-59 hands to an MLP by name, as epilogues of
 * rp_linear_fwd (one extra elementwise launch behind the short-K kernel, fused in the others) */
#define RP_ACT_TANH 3
#define RP_ACT_SIGMOID 4
#define RP_ACT_LEAKY 5 /* nn.LeakyReLU() with its default negative_slope = 0.01 */

/* Matrix-core precision of the GEMM entry points (process-wide; fp32 operands and fp32 accumulation in every mode).
 * The reference computes torch.nn.Linear in fp32 (ATen); BF16X6 reproduces fp32 products to ~2^-23 on the bf16
 * matrix core by splitting each operand into three bf16 pieces (6 MFMA products), BF16X3 to ~2^-16 with two pieces,
 * BF16 rounds the operands to bf16 (outside the 1e-4 parity gate: opt-in only), FP32 uses v_mfma_f32_32x32x2_f32.
 * AUTO (the default) chooses per launch: BF16X3 when the launch is matrix-core bound even at three products
 * (3 x flops / algorithmic bytes above the part's MFMA/HBM ridge of 312), BF16X6 otherwise — HBM-bound launches (every
 * GEMM of the reference's default MLP [64,64,64]) keep the fp32-faithful products at no cost; the model-level parity
 * gates (logits / loss within 1e-4) are tested in AUTO, BF16X6 and forced BF16X3 (tests/test_hip_models.py). */
#define RP_MATMUL_FP32 0
#define RP_MATMUL_BF16 1
#define RP_MATMUL_AUTO 2
#define RP_MATMUL_BF16X3 3
#define RP_MATMUL_BF16X6 6
int rp_set_matmul_precision(int mode); /* default RP_MATMUL_AUTO */
int rp_get_matmul_precision(void);

typedef void *rp_stream_t;

/* ---- library ------------------------------------------------------------------------------ */
#define RP_ABI_VERSION 106 /* rp_version(): bumped with every change of an entry point's prototype (106 = round 6) */
int rp_version(void);
const char *rp_last_error(void);
/* number of kernel launches issued through this library since load (tests use it to prove the
 * HIP path, not a fallback, produced a result) */
uint64_t rp_launch_count(void);

/* ---- K1/K2/K3: fused multi-table gather + dense concat + FM second order -------------------
 * replaces  layers/embedding.py:59-63 (F x nn.Embedding + stack),  models/utils.py:122-137
 * (dense stack), ranking/deepfm.py:57-58 (flatten + cat) and layers/interaction.py:38-44 (FM).
 *   arena      [R_total, D] all tables back to back (table f starts at row row_base[f])
 *   row_base   device int64[F];  row_count device int64[F] (= vocab_size+1, bounds check)
 *   idx_ptrs   F device pointers to int64[B] ids (one index per field per sample, bag size 1)
 *   dense_ptrs ND device pointers to float[B]
 *   x          [B, ldx] out: cols [f*D,(f+1)*D) = arena[row_base[f]+idx_f[b]], then ND dense
 *              columns, then zeros up to ldx.  ldx >= F*D+ND.
 *   fm_out     [B]    0.5*sum_d((sum_f v)^2 - sum_f v^2)          (NULL to skip)
 *   sum_out    [B, D] sum_f v, kept for the FM backward           (NULL to skip)
 *   keys_out   int32[F*B], keys_out[f*B+b] = global arena row     (NULL to skip)
 *   err_flag   device int32, set to 1 if any id is outside [0,row_count[f]) (the id is then
 *              clamped to 0; the Python side turns the flag into the reference's IndexError)
 */
int rp_embed_gather_fwd(const float *arena, const int64_t *row_base, const int64_t *row_count,
                        const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND,
                        int64_t B, int D, float *x, int64_t ldx, float *fm_out, float *sum_out,
                        int32_t *keys_out, int32_t *err_flag, rp_stream_t stream);

/* The gather FUSED with the Linear (+ ReLU) that consumes its output (DeepFM: embedding.py:59-63 + utils.py:122-137 +
 * deepfm.py:57-58 + interaction.py:38-44 + deep.py:62-72 for dnn.net.0 in ONE launch): the gathered rows go from global
 * memory into MFMA fragments once and are stored to x (NULL: x is not materialised), summed for the FM term and
 * multiplied with the field's 64 x 64 slice of W on the matrix core (split-bf16, six products), so x is never re-read.
 *   W [64, K] (ldw floats per row), K = F*64 + ND;  h1 [B, 64] = relu(x[:, :K] . W^T + bias)
 * rp_embed_gather_linear_fits: D == 64, a 64-wide layer, ND <= 16 (and F <= 32); otherwise RP_ERR_UNSUPPORTED
 * (compose rp_embed_gather_fwd + rp_linear_fwd).
 *   xd [B, 64] (instead of x; NULL otherwise): ONLY the dense columns are stored (zero padded to 64) — the layer's weight
 *   gradient then gathers the embedding rows again itself (rp_linear_wgrad_gather below) and the 4*F*D bytes per sample of
 *   the x store (436 MB at Criteo shape) are never written. */
int rp_embed_gather_linear_fits(int D, int ND, int hidden, int64_t ldx, int64_t ldw);
int rp_embed_gather_linear_fwd(const float *arena, const int64_t *row_base, const int64_t *row_count,
                               const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND, int64_t B,
                               int D, float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias, float *h1,
                               float *fm_out, float *sum_out, int32_t *keys_out, int32_t *err_flag, float *xd, rp_stream_t stream);
/* bf16-STORAGE inference (SURVEY D6's secondary mode; the fp32 tables stay the parity path and the training path): the same
 * launch over a bf16 copy of the arena (rows of D bf16 = 128 bytes at D = 64), fp32 accumulation everywhere, nothing stored
 * but h1 and the FM term.  The looked-up values differ from the fp32 tables' by bf16 rounding (2^-9 relative per element):
 * logits within 6e-2 (measured 3.7e-2 at the Criteo shape: tests/test_hip_models.py), not within the 1e-4 parity gate.
 * bf16-STORAGE TRAINING (round 4, secondary mode with a stated tolerance: SURVEY D6): the same launch also stores what the
 * backward needs — x_bf16 [B, ldx] as BF16 (the gathered values are bf16 already: exact; the dense columns are rounded to
 * bf16 for the weight gradient only, the forward uses them in fp32), sum_out [B, 64] fp32, keys_out — all NULL for
 * inference.  The bf16 arena then is the lookup copy the deferred optimizer kernels keep current (rp_lazy_adam_catchup's
 * shadow_bf16); rp_linear_wgrad_xbf16 is the weight gradient over the bf16 activation.  Round 5: with x_bf16 = NULL and
 * xd [B, 64] given (the dense columns in fp32, zero padded — as rp_embed_gather_linear_fwd's xd) NO activation is stored:
 * the weight gradient's embedding columns then come from rp_embed_grad_seg over the fp32 master rows. */
int rp_embed_gather_linear_fwd_bf16(const void *arena_bf16, const int64_t *row_base, const int64_t *row_count,
                                    const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND, int64_t B,
                                    int D, const float *W, int64_t ldw, const float *bias, float *h1, float *fm_out,
                                    void *x_bf16, int64_t ldx, float *sum_out, int32_t *keys_out, int32_t *err_flag,
                                    float *xd, rp_stream_t stream);

/* ---- gather backward: sort by arena row, then segmented reduce into the dense grad arena ----
 * replaces aten::embedding_dense_backward under layers/embedding.py:62 and the autograd of
 * interaction.py:38-44.  rp_sort_pairs_i32 sorts (key, position) pairs by the low end_bit bits of the key
 * (stable, so equal rows keep ascending sample order; end_bit = 32 orders the keys as SIGNED ints);
 * position = index into keys_in.  keys_in / keys_out / pos_out must not alias.  Kernel launches only (no
 * memset, nothing carried between calls): capturable into a hipGraph at any n.  Workspace size from
 * rp_sort_workspace_bytes.  RP_SORT=rocprim in the environment selects rocPRIM's radix sort instead. */
int rp_sort_workspace_bytes(int64_t n, size_t *bytes);
int rp_sort_pairs_i32(void *workspace, size_t workspace_bytes, const int32_t *keys_in, int32_t *keys_out,
                      int32_t *pos_out, int64_t n, int end_bit, rp_stream_t stream);
/* The same sort for the pair list of ONE lookup over F tables that sit in the arena in field order (round 6; csrc/sort.hip):
 * keys_in[f * B + b] = arena row of pair (field f, sample b) (rp_embed_keys); field_base / field_rows: host arrays [F], first
 * arena row and row count of each field's table.  The list is sorted segment by segment — a field's B pairs by the row inside
 * its table, ceil(log2 rows) key bits instead of the arena's — and comes out as rp_sort_pairs_i32(end_bit = bits of the arena)
 * gives it, bit for bit (the stable sort by arena row; pos_out = field * B + sample).  9-bit passes over the fields that still
 * have bits left: 49 field-passes instead of 78 at Criteo shape.  Kernel launches only; workspace from
 * rp_sort_pairs_fields_workspace_bytes(B, F). */
int rp_sort_pairs_fields_workspace_bytes(int64_t B, int F, size_t *bytes);
int rp_sort_pairs_fields_i32(void *workspace, size_t workspace_bytes, const int32_t *keys_in, int32_t *keys_out,
                             int32_t *pos_out, int64_t B, int F, const int64_t *field_base, const int64_t *field_rows,
                             rp_stream_t stream);
/*   grad_arena[key] (+)= sum over pairs p=(f,b) with that key of
 *        dx[b, f*D:(f+1)*D]  +  (gfm ? gfm[b] * (sum_in[b,:] - arena[key,:]) : 0)
 *        (sum_in may be NULL with gfm given: the gfm[b]*sum_in[b,:] part was already added to dx by
 *         rp_linear_fwd_rowadd, only the -gfm[b]*arena[key,:] part is applied here)
 *   dx may be NULL (FM-only models), gfm may be NULL (no FM term), not both.
 *   accumulate=0: every row that has pairs is OVERWRITTEN with its sum (rows without pairs are not touched: the caller
 *                 keeps them zero, rp_zero_rows);   accumulate=1: the sum is added to what is there.
 *   Deterministic: every row has one writer and every sum a fixed order (ascending sorted position — the sort is
 *   stable, so ascending sample index); runs that cross workgroups are chained through `workspace`
 *   (rp_embed_grad_reduce_workspace_bytes) by a second launch.  No floating-point atomics.                        */
int rp_embed_grad_reduce_workspace_bytes(int64_t n, int D, size_t *bytes);
int rp_embed_grad_reduce(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D,
                         const float *dx, int64_t ldx, const float *gfm, const float *sum_in,
                         const float *arena, float *grad_arena, int accumulate, void *workspace,
                         size_t workspace_bytes, rp_stream_t stream);
/* The same reduce FUSED with the dgrad of the Linear that consumes the gathered rows (DeepFM's dnn.net.0, deep.py:62-72 +
 * embedding.py:62): the dX row of pair (f, b) is dh[b, 0:64] . W1[:, f*64:(f+1)*64], formed on the matrix core
 * (split-bf16, six products) inside the reduce, so the [B, F*D] gradient of the MLP input is never written or read.
 *   dh  [B, 64]            gradient w.r.t. the layer's pre-activation (lddh floats per row)
 *   wt  [>= F*64, 64]      the layer's weight transposed (rp_transpose): row f*64+d = column f*64+d of W1 [64, K]
 *   dx  optional           sum of the gradients of x's other consumers, added per pair (NULL: none)
 *   skip_fields            bit f set: the pairs of field f are left out (their tables' gradient rows come from
 *                          rp_embed_grad_tiny); only with FIELD-MAJOR positions (position = field * B + sample), 0 otherwise
 * rp_embed_grad_gemm_fits: D == 64, a 64-wide layer, row strides multiples of 4 floats; otherwise RP_ERR_UNSUPPORTED
 * (compose rp_linear_fwd + rp_embed_grad_reduce).  Workspace: rp_embed_grad_reduce_workspace_bytes(n, D). */
int rp_embed_grad_gemm_fits(int D, int hidden, int64_t lddh, int64_t ldwt);
int rp_embed_grad_gemm(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D, const float *dh,
                       int64_t lddh, const float *wt, int64_t ldwt, const float *dx, int64_t ldx, const float *gfm,
                       const float *sum_in, const float *arena, float *grad_arena, int accumulate, uint64_t skip_fields,
                       void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* The same gradient for TINY tables (each <= 254 rows, <= 224 rows together, <= 16 tables), SAMPLE-major (csrc/embed_tiny.hip):
 * G[r] = (sum_b dH[b]) . W1_f^T + sum_b g_fm[b] S[b] - (sum_b g_fm[b]) v_r over the samples b with id_f[b] = r; the 129-wide row
 * sums are one-hot GEMMs on the matrix core (one-hot exact in bf16, the data split into three bf16 pieces, fp32 accumulation
 * in a fixed order: deterministic), every sample's dH / S row is read once for all tiny tables.  Writes the rows somebody
 * looked up in this batch (accumulate != 0 adds), like rp_embed_grad_gemm: other rows are not touched.  keys [F * B]: arena row of pair (field, sample) at
 * field * B + sample (rp_embed_keys / the gather's keys_out).  Host arrays: tiny_field (field index), tiny_base (first arena
 * row), tiny_rows.  rp_embed_grad_gemm(skip_fields = bits of those fields) then covers the other fields.  dw != NULL
 * (round 5): also  dw[:, f*64:(f+1)*64] = sum over the table's rows of  (sum_b dH[b])^T (x) v_r  — the tiny tables' columns
 * of the first layer's weight gradient (see rp_embed_grad_seg). */
int rp_embed_grad_tiny_workspace_bytes(int64_t B, size_t *bytes);
int rp_embed_grad_tiny(const int32_t *keys, int64_t B, const int32_t *tiny_field, const int64_t *tiny_base,
                       const int32_t *tiny_rows, int n_tiny, const float *dh, int64_t lddh, const float *wt, int64_t ldwt,
                       const float *gfm, const float *sum_in, const float *arena, float *grad_arena, int accumulate,
                       float *dw, int64_t lddw, void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* The first layer's WHOLE backward on the embedding columns, segment-sum first (csrc/embed.hip, round 5; replaces
 * rp_embed_grad_gemm + the rp_linear_wgrad over a stored activation on the single-device path).  Reference ops:
 * aten::embedding_dense_backward of rec_pangu/models/layers/embedding.py:61-63, the `** 2` backward of
 * layers/interaction.py:38-44, the first Linear's dgrad and the embedding columns of its weight gradient
 * (layers/deep.py:62-72 on the input built at ranking/deepfm.py:57-59).  Per run of equal keys in the row-sorted pair list
 * (pairs p = (f, b) of one table row r):  Hs = sum dh[b, :],  u = sum gfm[b] sum_in[b, :],  s = sum gfm[b];
 *     grad_arena[r, :] (+)= Hs . w[:, f*64:(f+1)*64] + u - s arena[r, :]
 *     dw[:, f*64:(f+1)*64]  = sum over the field's runs of  Hs^T (x) arena[r, :]      (dw != NULL; written, not added)
 * w = the layer's weight [64, ldw] (row-major, NOT transposed), dh [B, 64] the gradient of its pre-activation.  FIELD-MAJOR
 * positions only: sorted_pos[i] = field * B + sample, n = F * B, F <= 64.  skip_fields: bit f set = field f is left out
 * (its table is rp_embed_grad_tiny's, which fills the same dw columns when given dw).  field_rows: host array [F] of table
 * sizes (scheduling only: long chunks first) or NULL.  gfm and sum_in both NULL = no FM term.  Deterministic (one writer
 * per row, fixed summation orders).  Workspace: rp_embed_grad_seg_workspace_bytes(n, B, D).
 * rp_embed_grad_seg_fits: D == 64, a 64-wide layer, lddh % 4 == 0; otherwise RP_ERR_UNSUPPORTED. */
int rp_embed_grad_seg_fits(int D, int hidden, int64_t lddh);
int rp_embed_grad_seg_workspace_bytes(int64_t n, int64_t B, int D, size_t *bytes);
int rp_embed_grad_seg(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D, const float *dh,
                      int64_t lddh, const float *w, int64_t ldw, const float *gfm, const float *sum_in, const float *arena,
                      float *grad_arena, int accumulate, uint64_t skip_fields, const int64_t *field_rows, float *dw,
                      int64_t lddw, void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* rp_embed_grad_seg's work in two launches (csrc/embed_ss.hip, round 6; the same reference ops, contract and results up to
 * fp32 summation order): a STREAMING segment-sum pass with no LDS and no barrier (a 16-lane group per 32 sorted positions
 * writes one record [sum dh | sum gfm sum_in | sum gfm] per run piece) and a matrix pass over tiles of 128 UNIQUE rows of a
 * field (a row's pieces summed in position order first; one writer per row, no run is cut at a tile border).  This form is
 * for the tables whose runs are long (the mid-size ones: rp_embed_grad_smp takes the big tables, rp_embed_grad_tiny the
 * tiny ones).
 * The matrix pass reads the fields' unique-row lists (ustart, ukey, offs): rp_embed_grad_ss_mark makes them from the sorted
 * keys alone (three short launches — with the sort, a step ahead); NULL x 3 = made inside this call (workspace).  Sizes in
 * int32 elements: rp_embed_grad_ss_mark_sizes(B, kept fields) -> n_rows (ustart and ukey each), n_offs.
 * phases: 1 = the segment-sum launch (reads dh / sum_in / gfm and the sorted pairs only), 2 = the launches behind it (same
 * workspace, ordered behind phase 1), 3 = both.  Workspace: rp_embed_grad_ss_workspace_bytes(n, B, D, skip_fields). */
int rp_embed_grad_ss_workspace_bytes(int64_t n, int64_t B, int D, uint64_t skip_fields, size_t *bytes);
int rp_embed_grad_ss_mark_sizes(int64_t B, int n_kept, size_t *n_rows, size_t *n_offs);
int rp_embed_grad_ss_mark(const int32_t *sorted_keys, int64_t n, int64_t B, uint64_t skip_fields, int32_t *ustart, int32_t *ukey,
                          int32_t *offs, rp_stream_t stream);
int rp_embed_grad_ss(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D, const float *dh,
                     int64_t lddh, const float *w, int64_t ldw, const float *gfm, const float *sum_in, const float *arena,
                     float *grad_arena, int accumulate, uint64_t skip_fields, const int64_t *field_rows, float *dw,
                     int64_t lddw, const int32_t *ustart, const int32_t *ukey, const int32_t *offs, int phases,
                     void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* The same backward for the BIG tables, SAMPLE-major (csrc/embed_smp.hip, round 6; reference ops as rp_embed_grad_seg:
 * rec_pangu/models/layers/embedding.py:61-63 backward, layers/interaction.py:38-44 backward, layers/deep.py:62-72 dgrad and
 * the embedding columns of the weight gradient).  For a table whose runs in the sorted pair list are mostly singletons the
 * sort buys nothing and the (field, row) order re-gathers every sample's dh / sum_in row once per field; this launch walks
 * the batch in sample order instead (a unit = 64 samples x one field: dh staged once in LDS, the table rows gathered into
 * the matrix core's accumulator layout, dgrad + FM term + the field's weight-gradient columns from the same registers):
 *     pair (f, b), r = keys[f * B + b]:   row = dh[b, :] . w[:, f*64:(f+1)*64] + gfm[b] (sum_in[b, :] - arena[r, :])
 *     grad_arena[r, :] (+)= row                       when the pair is alone in its run of the sorted list (one writer), else
 *     the row goes to a side buffer at its number among such pairs and rp_embed_grad_reduce_rows sums the runs in list order;
 *     dw[:, f*64:(f+1)*64] = sum_b dh[b, :]^T (x) arena[r, :]                      (dw != NULL; written, not added)
 * rp_embed_grad_smp_mark (from the sorted list; may run ahead of the backward: it depends on the batch's ids only): the
 *     pairs whose run has >= 2 pairs are numbered c = 0, 1, ... in sorted order (field by field);
 *     dupq[fi * B + b] = c of pair (fields[fi], b), or -1 for a pair alone in its run;  dupkeys[c] = the key of pair c, -1
 *     from the number of such pairs on.  Both [n_fields * B] int32, caller-owned; scratch: rp_embed_grad_smp_mark_scratch
 *     int32 words.
 * fields: host array, ascending, <= 16; field_base / field_rows: host arrays [n_fields], first arena row and row count of each
 * field's table (< 2^24 rows: 32-bit row offsets; n_fields * B < 2^24); FIELD-MAJOR positions (n = F * B, F <= 64).
 * rp_embed_grad_seg(skip_fields = their bits) covers the other fields.  phases: 1 = the main launch, 2 = the launches behind it
 * (the partial sums of dw, the duplicate runs: they touch no row of another launch and may run beside rp_embed_grad_seg on a
 * second stream, ordered behind phase 1 on the same workspace), 3 = both.  Deterministic.  Workspace: rp_embed_grad_smp_workspace_bytes(B, n_fields).
 * rp_embed_grad_reduce_rows: grad_arena[key, :] (+)= sum of rows[i, :] over the entries i of a key-sorted list with
 * keys[i] == key; key -1 = no entry (its row is not read).  Workspace: rp_embed_grad_reduce_workspace_bytes(n, D). */
int rp_embed_grad_smp_fits(int D, int hidden, int64_t lddh);
int rp_embed_grad_smp_workspace_bytes(int64_t B, int n_fields, size_t *bytes);
int rp_embed_grad_smp_mark_scratch(int64_t B, int n_fields, size_t *n_int32);
int rp_embed_grad_smp_mark(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B,
                           const int32_t *fields, int n_fields, int32_t *dupq, int32_t *dupkeys, int32_t *scratch,
                           rp_stream_t stream);
int rp_embed_grad_smp(const int32_t *keys, const int32_t *dupq, const int32_t *dupkeys, int64_t B, int F,
                      const int32_t *fields, const int64_t *field_base, const int64_t *field_rows, int n_fields,
                      const float *dh, int64_t lddh, const float *w, int64_t ldw,
                      const float *gfm, const float *sum_in, const float *arena, float *grad_arena, int accumulate,
                      float *dw, int64_t lddw, int phases, void *workspace, size_t workspace_bytes, rp_stream_t stream);
int rp_embed_grad_reduce_rows(const int32_t *keys, const float *rows, int64_t n, int D, float *grad_arena, int accumulate,
                              void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* grad_arena[keys[i], :] = 0 for i < n (duplicates allowed) */
int rp_zero_rows(const int32_t *keys, int64_t n, int D, float *grad_arena, rp_stream_t stream);
/* dst[0 .. n_words) = value, 32-bit words, 16-byte aligned buffer: the library's own fill (a recorded step must hold no
 * ATen fill kernel / memset node: csrc/plan.hip) */
int rp_fill_words(void *dst, int64_t n_words, uint32_t value, rp_stream_t stream);

/* ---- K4: Linear (+bias +activation) on the fp32 MFMA ----------------------------------------
 * replaces layers/deep.py:62-72 (nn.Linear + ReLU chain) and its autograd.
 *   out[M,N] = act(a[M,K] . w[N,K]^T + bias[N]);  bias may be NULL;
 *   act = RP_ACT_MASK multiplies by (aux[m,n] > 0) (aux: [M, ldaux]); RP_ACT_TANH / SIGMOID / LEAKY: see above. */
int rp_linear_fwd(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias, float *out,
                  int64_t ldo, int64_t M, int N, int K, int act, const float *aux, int64_t ldaux,
                  rp_stream_t stream);
/* weight/bias gradient: dw[N,K] (+)= dy[M,N]^T . x[M,K];  db[N] (+)= column sums of dy.
 * Deterministic two-stage reduction through `workspace` (rp_linear_wgrad_workspace_bytes).  */
int rp_linear_wgrad_workspace_bytes(int64_t M, int N, int K, size_t *bytes);
int rp_linear_wgrad(const float *dy, int64_t lddy, const float *x, int64_t ldx, float *dw, int64_t lddw,
                    float *db, int64_t M, int N, int K, int accumulate, void *workspace,
                    size_t workspace_bytes, rp_stream_t stream);
/* the same with the activation stored as bf16 (x_bf16 [M, ldx] bf16, ldx even, 4-byte aligned; the bf16-storage training
 * mode): half the bytes of the dominant operand; bf16 matrix-core modes and N <= 128 only (RP_ERR_UNSUPPORTED otherwise) */
int rp_linear_wgrad_xbf16(const float *dy, int64_t lddy, const void *x_bf16, int64_t ldx, float *dw, int64_t lddw,
                          float *db, int64_t M, int N, int K, int accumulate, void *workspace, size_t workspace_bytes,
                          rp_stream_t stream);
/* The weight gradient of a layer whose input is the embedding lookup, WITHOUT the stored activation: the first Kg = F*64
 * columns of X are gathered — X[m, f*64 + j] = arena[keys[f*M + m]*64 + j], keys = the arena rows the forward saved
 * (rp_embed_gather_linear_fwd keys_out / rp_embed_keys) — the remaining K - Kg <= 64 columns (dense features) come from
 * xd [M, ldxd].  N == 64, Kg a multiple of 128, bf16 matrix-core modes (rp_linear_wgrad_gather_fits); workspace as
 * rp_linear_wgrad_workspace_bytes(M, N, K). */
int rp_linear_wgrad_gather_fits(int64_t M, int N, int K, int Kg);
int rp_linear_wgrad_gather(const float *dy, int64_t lddy, const float *arena, const int32_t *keys, int Kg, const float *xd,
                           int64_t ldxd, float *dw, int64_t lddw, float *db, int64_t M, int N, int K, int accumulate,
                           void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* out[C,R] = in[R,C]^T (weights for the dgrad GEMM); rows C .. C_out-1 of out (C_out >= C) are written as zeros */
int rp_transpose(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, int C_out, rp_stream_t stream);
/* rp_transpose and rp_copy_rows of the same matrix in ONE launch: out = in^T as rp_transpose, copy[r, 0:C] = in[r, 0:C] with
 * row stride ldcopy (the first layer's weight: its transposed copy for the gather backward and its aligned copy for the
 * gather forward are both made in front of the forward) */
int rp_transpose_copy(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, int C_out, float *copy,
                      int64_t ldcopy, rp_stream_t stream);
/* out[r, 0:C] = in[r, 0:C], r < R, with another row stride (staging copy of a weight with unaligned rows) */
int rp_copy_rows(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, rp_stream_t stream);
/* out[r, 0:C] += in[r, 0:C] (independent row strides): the gradient of a column block of the gathered activation that a
 * second consumer produced (AutoInt's attention over x[:, :F*D], autoint.py:44-46) added into the first consumer's dX */
int rp_add_rows(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, rp_stream_t stream);
/* ---- nn.Linear forward on PRE-SPLIT bf16 operands (csrc/gemm_pieces.hip; round 4) ---------------------------------
 * Same contract as rp_linear_fwd (deep.py:62-72: out = act(a . w^T + bias)), but both operands arrive as bf16 PIECES in
 * the "interleaved" layout: a row is a sequence of 128-byte k-tiles,
 *   np = 2 (the bf16x3 products hi.lo + lo.hi + hi.hi):  [32 x bf16 hi | 32 x bf16 lo]  per 32 values of K
 *   np = 1 (plain bf16, a stated-tolerance mode):        [64 x bf16]                    per 64 values of K
 * K zero-padded to whole tiles: a row holds rp_pieces_ld(K, np) bf16 elements (lda / ldw >= that, multiples of 8).  The
 * k-tiles go HBM -> LDS by LDS-DMA and the inner loop is fragment reads and MFMAs only.  np = 2 is bit-identical to
 * rp_linear_fwd under RP_MATMUL_BF16X3.  rp_pieces_pack makes the layout from an fp32 matrix [M, K] (hi = RN(x),
 * lo = RN(x - hi)); ldo in bf16 elements, a multiple of 64.  act: RP_ACT_NONE / RELU / MASK (aux as in rp_linear_fwd). */
int64_t rp_pieces_ld(int K, int np);
int rp_pieces_pack(const float *in, int64_t ld, int64_t M, int K, int np, void *out, int64_t ldo, rp_stream_t stream);
int rp_linear_fwd_pieces(const void *a, int64_t lda, const void *w, int64_t ldw, const float *bias, float *out,
                         int64_t ldo, int64_t M, int N, int K, int np, int act, const float *aux, int64_t ldaux,
                         rp_stream_t stream);
/* y = dy * (act_out > 0), elementwise over [M,N] */
int rp_relu_bwd(const float *dy, int64_t lddy, const float *act_out, int64_t ldact, float *out, int64_t ldo,
                int64_t M, int N, rp_stream_t stream);
/* the activations as launches of their own (elementwise over [M,N]; act = RP_ACT_RELU / TANH / SIGMOID / LEAKY):
 * rp_act_fwd: y = act(x) (y == x allowed);  rp_act_bwd: out = dy * act'(.) expressed through the activation's OUTPUT
 * (relu / leaky: sign of y; tanh: 1 - y^2; sigmoid: y (1 - y)) — the backward of rp_linear_fwd(act) needs no pre-activation */
int rp_act_fwd(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t M, int N, int act, rp_stream_t stream);
int rp_act_bwd(const float *dy, int64_t lddy, const float *act_out, int64_t ldact, float *out, int64_t ldo, int64_t M,
               int N, int act, rp_stream_t stream);

/* ---- K5: DCN-v1 CrossNet, all L layers in one pass ------------------------------------------
 * replaces layers/interaction.py:119-141 (CrossInteractionLayer/CrossNet) and, when wfc is given,
 * the `fc` that follows it in ranking/dcn.py:64.
 *   X_{l+1} = X_l + (X_l . W[l]) * X_0 + Bv[l];   W, Bv: [L, d] (the L Linear(d,1) weights / biases stacked)
 *   xout [B, ldo] = X_L (NULL to skip);  logit[B] = X_L . wfc + bfc[0] (NULL to skip)
 *   s_out [B, L] = the per-layer scalars X_l . W[l], all the backward needs besides X_0.
 * backward: rp_crossnet_bwd_rows below.  d <= 2048, L <= 6.        */
int rp_crossnet_fwd(const float *x0, int64_t ldx, int d, int L, const float *W, const float *Bv, const float *wfc,
                    const float *bfc, float *xout, int64_t ldo, float *logit, float *s_out, int64_t B,
                    rp_stream_t stream);

/* ---- K6: xDeepFM CIN layer on the fp32 MFMA ------------------------------------------------------
 * replaces layers/interaction.py:164-168 (einsum "bhd,bmd->bhmd" + view + Conv1d(k=1) + sum over d).
 *   x0 [B, ld0]: X_0[b] as H rows of D floats;  xp [B, ldp]: X_{k-1}[b] as M rows of D floats (xp == x0 for layer 1)
 *   W  [O, H*M] (Conv1d weight, channel c = h*M + m), bias [O] or NULL
 *   out [B, O, D] = X_k (NULL to skip: the collapsed last layer only needs the pooling)
 *   pooled [B, ldpool]: pooled[b, o] = sum_d X_k[b,o,d]  (NULL to skip)
 * The outer product [B, H*M, D] is never materialised.  H <= 32 fields; must fit the 160 KB LDS.
 * backward (g_out [B,O,D] = dL/dX_k and/or g_pool [B, ldgp] = dL/dpooled, broadcast over d):
 *   rp_cin_layer_bwd_x: dx0 [B, lddx0] (+)= , dxp [B, lddxp] = gradients of both factors
 *   rp_cin_layer_bwd_w: dW [O, H*M], dbias [O] (batch-reduced through `workspace`)                           */
int rp_cin_layer_fwd(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *W, const float *bias,
                     float *out, float *pooled, int64_t ldpool, int H, int M, int O, int D, int64_t B,
                     rp_stream_t stream);
/* add_dx0 != 0: dx0 += (several layers share X_0), else dx0 = ; for the first layer (xp == x0) both factor
 * gradients land in dx0 and dxp is ignored.  Middle layers with O >= 4 need M <= 32 (register-tiled). */
int rp_cin_layer_bwd_x(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *W, const float *g_out,
                       const float *g_pool, int64_t ldgp, float *dx0, int64_t lddx0, int add_dx0, float *dxp,
                       int64_t lddxp, int H, int M, int O, int D, int64_t B, rp_stream_t stream);
int rp_cin_layer_bwd_w_workspace_bytes(int64_t B, int H, int M, int O, size_t *bytes);
int rp_cin_layer_bwd_w(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *g_out,
                       const float *g_pool, int64_t ldgp, float *dW, float *dbias, int H, int M, int O, int D,
                       int64_t B, void *workspace, size_t workspace_bytes, rp_stream_t stream);

/* ---- K7: AutoInt field self-attention layer ----------------------------------------------------
 * replaces layers/attention.py:63-101 (MultiHeadSelfAttention, align_to="output", no dropout/LayerNorm):
 * QKV(+residual) projections, RAW-view head split (:73-75), scores [/scale], softmax, PV, +residual, ReLU.
 *   x   [B, ldx]: T tokens of Din floats, contiguous per sample (x[b, t*Din + k])
 *   W   [(3|4)*H*a, Din]: Wq, Wk, Wv (and Wres when has_res) stacked; without Wres Din must equal H*a
 *   scale: attention_dim**0.5 when use_scale, 0 for none
 *   out [B, T*H*a];   backward: gout [B, T*H*a] -> dx [B, lddx] (NULL to skip), dW like W.          */
int rp_field_attention_fits(int T, int Din, int H, int a, int has_res); /* 1 if fwd+bwd fit the 160 KB LDS */
int rp_field_attention_fwd(const float *x, int64_t ldx, const float *W, int T, int Din, int H, int a, int has_res,
                           float scale, float *out, int64_t B, rp_stream_t stream);
int rp_field_attention_bwd_workspace_bytes(int64_t B, int T, int Din, int H, int a, int has_res, size_t *bytes);
int rp_field_attention_bwd(const float *x, int64_t ldx, const float *W, int T, int Din, int H, int a, int has_res,
                           float scale, const float *gout, float *dx, int64_t lddx, float *dW, int64_t B,
                           void *workspace, size_t workspace_bytes, rp_stream_t stream);

/* ---- K8: MMOE gate softmax + gate-weighted expert combine -----------------------------------
 * replaces multi_task/mmoe.py:92-104.  The expert einsum (mmoe.py:86) and the T gate products (:94)
 * are ONE rp_linear_fwd over the concatenated [experts | gates] matrix; its output z [B, ldz] has the
 * expert outputs at columns k*E+e (k < K) and the gate logits at K*E + t*E + e.
 *   gate [B, T*E] = per-task softmax over experts (kept for the backward)
 *   out  [T, B, K] : out[t,b,k] = sum_e z[b,k*E+e] * gate[b,t,e]
 * backward: dout [T,B,K] -> dz [B, lddz] (expert columns and gate-logit columns).  T, E <= 8, T*E <= 32. */
int rp_mmoe_combine_fwd(const float *z, int64_t ldz, int K, int E, int T, float *out, float *gate, int64_t B,
                        rp_stream_t stream);
int rp_mmoe_combine_bwd(const float *z, int64_t ldz, int K, int E, int T, const float *gate, const float *dout,
                        float *dz, int64_t lddz, int64_t B, rp_stream_t stream);

/* Streaming CrossNet backward.  X_l = A_l X_0 + C_l (A_l = 1 + sum_{k<l} s_k per sample,
 * C_l = sum_{k<l} b_k per feature), so one wave per row produces dx0 and V[B, 2L+2] =
 * [t_l A_l (l<L) | g_logit A_L | t_l (l<L) | g_logit]; then dW_l = (V^T X_0)[l] + C_l colsum(V)[L+1+l] etc. are one
 * rp_linear_wgrad(V, X_0) plus [L,d]-sized weight-space arithmetic.  d <= 2048.                            */
int rp_crossnet_bwd_rows(const float *x0, int64_t ldx, int d, int L, const float *W, const float *wfc,
                         const float *s_in, const float *g_x, int64_t ldg, const float *g_logit, float *dx0,
                         int64_t lddx, float *V, int64_t B, rp_stream_t stream);
/* the CrossNet's parameter gradients from P [2L+2, ldp] = V^T X_0 (rp_linear_wgrad over V) and cs [2L+2] = column sums of V:
 * dW [L, d], dB [L, d] and — with the fused fc (wfc [d]) — dwfc [d]; colg [d] = column sums of the incoming gradient when
 * there is no fused fc (else NULL).  One launch for what autograd of interaction.py:119-141 spreads over the layers. */
int rp_crossnet_param_grads(const float *P, int64_t ldp, const float *cs, const float *W, const float *Bv, const float *wfc,
                            const float *colg, int L, int d, float *dW, float *dB, float *dwfc, rp_stream_t stream);

/* ---- K6 (bf16 matrix core): a CIN layer with at most 32 x 32 (field, map) pairs per channel ----------------
 * replaces interaction.py:157-171 for such a layer (a FIRST layer: X_{k-1} = X_0, H = M <= 32) on
 * v_mfma_f32_32x32x16_bf16 with split-bf16 operands (fp32-faithful, 6 products).  The X_{k-1} fragments stay in
 * registers, the channels' weights stream past as bf16 pieces  wp [O][3][32][32]  = (hi, mid, lo) of W[o, row, col]
 * zero-padded to 32 x 32 (hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)).
 *   fwd   : out [B, O, D] and/or pooled [B, O] (= sum_d out); bias [O] optional
 *   bwd_x : dx[b,r,:] = sum_o (gout[b,o,:] + gpool[b,o]) sum_c wp[o][r][c] xk[b,c,:]   — X_0-role gradient with
 *           wp from W[o,h,m], X_{k-1}-role gradient with W[o,m,h]; for a first layer W[o,h,m] + W[o,m,h] gives the
 *           whole gradient in one pass.
 * Limits: rows, contraction <= 32, D in {32, 64} (rp_cin_bs_fits). */
int rp_cin_bs_fits(int H, int M, int D);
/* A layer fed by MORE than 32 maps (a middle layer of e.g. cin_layer_units = [128, 128, 128]) is the sum over chunks of
 * <= 32 maps of such layers: rp_cin_bs_fwd per chunk (xp = the chunk's columns of X_{k-1}, leading dimension M*D),
 * partial outputs summed with rp_accumulate; in the backward g_out is shared, rp_cin_bs_bwd_x gives the X_0-role
 * gradient per chunk (accumulated) and the chunk's own X_{k-1}-role gradient (written into its columns), and
 * rp_cin_bs_bwd_w the chunk's slice of dW (functional._CINChunked).  dst[0:n] += src[0:n]: */
int rp_accumulate(float *dst, const float *src, int64_t n, rp_stream_t stream);
int rp_cin_bs_fwd(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const void *wp, const float *bias, int H,
                  int M, int O, int D, float *out, float *pooled, int64_t B, rp_stream_t stream);
int rp_cin_bs_bwd_x(const float *xk, int64_t ldk, const void *wp, const float *gout, const float *gpool, int R, int Cn,
                    int O, int D, float *dx, int64_t lddx, int64_t B, rp_stream_t stream);
/* dW [O, H*M], db [O] (optional): dW[o,h,m] = sum_{b,d} (gout + gpool)[b,o,d] X_0[b,h,d] X_{k-1}[b,m,d] */
int rp_cin_bs_bwd_w_workspace_bytes(int64_t B, int O, size_t *bytes);
int rp_cin_bs_bwd_w(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *gout, const float *gpool,
                    int H, int M, int O, int D, float *dW, float *db, int64_t B, void *workspace, size_t workspace_bytes,
                    rp_stream_t stream);
/* first layer only (X_{k-1} = X_0, O <= 128): the same dW [O, H*H] / db through the symmetric pair form — the products
 * X_0[h] X_0[m] are formed once per (pair, contraction row) instead of once per channel, and dWs[o, pair] is ONE TN GEMM */
int rp_cin_pair_fits(int H, int O, int D); /* H <= 32, O <= 128, D in {32, 64} */
/* forward of the same layer as ONE GEMM over the pair products: X_1[b,o,d] = bias[o] + sum_p Ws[o,p] X_0[b,h_p,d] X_0[b,m_p,d],
 * Ws[o,(h,m)] = W[o,h,m] + W[o,m,h] (h < m; W[o,h,h] on the diagonal), pairs row-major over the upper triangle.
 * wsp: Ws split into bf16 pieces (hi, mid, lo), [3][128][KP] with KP = 32*ceil(H(H+1)/2 / 32), zero padded.
 * out [B, O*D] or NULL, pooled [B, O] (sum over d) or NULL. */
int rp_cin_pair_fwd(const float *x0, int64_t ld0, const void *wsp, const float *bias, int H, int O, int D, float *out,
                    float *pooled, int64_t B, rp_stream_t stream);
/* dX_0 of the same layer (X_0 in both roles): T[p] = sum_o Ws[o,p] G[o], dX_0[h] = sum_{p=(h,m)|(m,h)} T[p] X_0[m].
 * wst: Ws^T as bf16 pieces [3][KPT][128], KPT = 128*ceil(H(H+1)/2 / 128), zero padded.  gout [B,O*D] / gpool [B,O] (packed)
 * as in rp_cin_bs_bwd_x.  lstart / lent (device int32): for pair tile t, half c (64 pairs) and field h the entries
 * lent[lstart[(2t+c)*H + h] .. lstart[(2t+c)*H + h + 1]) = (local pair row) | (other field) << 8 of that half's pairs
 * containing h, the diagonal pair listed twice.  dx rows [B, lddx]: the first H*D floats of each row are written —
 * accumulate != 0 (round 6): ADDED to what is there (the collapsed last layer's gradient of X_0, rp_cin_last_bwd_x: the sum of
 * the two without a pass of its own). */
int rp_cin_pair_bwd_x(const float *x0, int64_t ld0, const void *wst, const float *gout, const float *gpool,
                      const int32_t *lstart, const int32_t *lent, int H, int O, int D, float *dx, int64_t lddx, int64_t B,
                      int accumulate, rp_stream_t stream);
int rp_cin_pair_bwd_w_workspace_bytes(int64_t B, int H, int O, size_t *bytes);
int rp_cin_pair_bwd_w(const float *x0, int64_t ld0, const float *gout, const float *gpool, int H, int O, int D, float *dW,
                      float *db, int64_t B, void *workspace, size_t workspace_bytes, rp_stream_t stream);

/* ---- K6 (last layer): the collapsed final CIN layer ------------------------------------------------------
 * replaces interaction.py:157-171 for the LAST layer: with no activation and linear pooling + fc behind it, it
 * enters the logit only as  p[b] = sum_d sum_{h,m} V[h,m] X_0[b,h,d] X_{L-1}[b,m,d],  V = sum_o c[o] W_L[o].
 * x0 [B, ld0] holds H rows of D floats per sample, xp [B, ldp] M rows of D floats; vt = V^T zero-padded to [M, 32].
 * bwd_x: dx0[b,h,:] = g[b] sum_m V[h,m] Xp[b,m,:], dxp[b,m,:] = g[b] sum_h V[h,m] X0[b,h,:];  bwd_v: dV [H, M].
 * Limits: H <= 32, D <= 64 (rp_cin_last_fits). */
int rp_cin_last_fits(int H, int M, int D);
int rp_cin_last_fwd(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *vt, int H, int M, int D,
                    float *pooled, int64_t B, rp_stream_t stream);
int rp_cin_last_bwd_x(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *vt, const float *g, int H,
                      int M, int D, float *dx0, int64_t ldd0, float *dxp, int64_t lddp, int64_t B, rp_stream_t stream);
int rp_cin_last_bwd_v_workspace_bytes(int64_t B, int H, int M, size_t *bytes);
int rp_cin_last_bwd_v(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *g, int H, int M, int D,
                      float *dV, int64_t B, void *workspace, size_t workspace_bytes, rp_stream_t stream);
/* The weight-space arithmetic around the pair kernels and the collapsed last layer (round 6, csrc/cin_head.hip) — what the host
 * side did with ~50 elementwise / index / matmul launches per step (interaction.py:157-171 has no counterpart: the reference
 * runs every layer at full width):
 *   rp_cin_pair_pieces      W [O, H, H] -> the symmetric pair weights Ws[o, (h <= m)] = W[o,h,m] + W[o,m,h] (W[o,h,h] on the
 *                           diagonal) as three bf16 pieces (round to nearest even, piece k of what the pieces before it left):
 *                           wsp [3][128][KP] (rp_cin_pair_fwd / _bwd_w; KP = pairs rounded up to 32) and / or wst [3][KPT][128]
 *                           (rp_cin_pair_bwd_x; KPT = pairs rounded up to 128); either may be NULL; zero padded.
 *   rp_cin_head_params_fwd  vt [M, 32] = V^T zero padded, V[h, m] = sum_o c[o] WL[o, h M + m];  vb[0] = sum_o c[o] bL[o]
 *                           (bL may be NULL: 0) — c = the slice of fc.weight that multiplies the last layer's pooling.
 *   rp_cin_head_params_bwd  dWL[o, j] = c[o] dV[j];  dbL[o] = Dscale sg[0] c[o];  dc[o] = sum_j WL[o, j] dV[j] + Dscale sg[0] bL[o]
 *                           (dV [H, M] from rp_cin_last_bwd_v, sg[0] = sum_b g[b], Dscale = the embedding width D).
 *   rp_add_scalars          out[i] += scale a[0] + (b0 ? b0[0] : 0) for i < n (the head's D vb + fc.bias onto the [B] logit).
 *   rp_sum_all              out[0] = sum_i x[i], fixed order (one workgroup). */
int rp_cin_pair_pieces(const float *W, int O, int H, void *wsp, void *wst, rp_stream_t stream);
int rp_cin_head_params_fwd(const float *WL, const float *bL, const float *c, int O, int H, int M, float *vt, float *vb,
                           rp_stream_t stream);
int rp_cin_head_params_bwd(const float *WL, const float *bL, const float *c, const float *dV, const float *sg, float Dscale,
                           int O, int H, int M, float *dWL, float *dbL, float *dc, rp_stream_t stream);
int rp_add_scalars(float *out, int64_t n, const float *a, float scale, const float *b0, rp_stream_t stream);
int rp_sum_all(const float *x, int64_t n, float *out, rp_stream_t stream);

/* ---- K7 (split form): the T x T core of the field self-attention; projections are rp_linear_fwd GEMMs ------
 * replaces attention.py:20-33,73-94 for one AutoInt layer once QKVR = X . [Wq|Wk|Wv|Wres]^T has been computed for
 * all B*T token rows (row-major [B*T, ldq], columns Q | K | V | R, each H*a wide; nproj = 3 without W_res, the
 * residual then being xres [B*T, ldr] = X itself).  Heads are the reference's RAW view of a sample's flat [T*H*a]
 * buffer.  out [B*T, H*a] = relu(softmax(Q K^T [/scale]) V + R); stats [B, H*T, 2] = per-row (max, sum) of the
 * softmax, consumed by the backward.  Backward writes dqkvr (same layout; the R block = ReLU-masked dout) and, when
 * nproj == 3, dxres.  Limits: a <= 16 and a sample's Q,K,V,dO within 64 KB of LDS (rp_attention_core_fits). */
int rp_attention_core_fits(int T, int H, int a);
int rp_attention_core_fwd(const float *qkvr, int64_t ldq, int nproj, const float *xres, int64_t ldr, int T, int H,
                          int a, float scale, float *out, float *stats, int64_t B, rp_stream_t stream);
int rp_attention_core_bwd(const float *qkvr, int64_t ldq, int nproj, const float *out, const float *dout,
                          const float *stats, int T, int H, int a, float scale, float *dqkvr, int64_t lddq,
                          float *dxres, int64_t lddr, int64_t B, rp_stream_t stream);

/* dgrad with a fused row-scaled periodic addend (the DeepFM gather backward's FM term folded into dX):
 *   out[M,N] = a[M,K] . w[N,K]^T + row_scale[m] * row_add[m, n % 64]  for n < add_cols   (no bias / activation)
 * Only K <= 64 (multiple of 4), M % 128 == 0, N % 64 == 0, add_cols % 64 == 0, 16-byte aligned rows and a split-bf16
 * matmul mode; otherwise RP_ERR_UNSUPPORTED and the caller uses rp_linear_fwd + the sum_in path of the reduce. */
int rp_linear_fwd_rowadd(const float *a, int64_t lda, const float *w, int64_t ldw, float *out, int64_t ldo, int64_t M,
                         int N, int K, const float *row_scale, const float *row_add, int64_t ld_add, int add_cols,
                         rp_stream_t stream);

/* ---- stand-alone FM pooling on a [B,F,D] tensor -------------------------------------------------------
 * replaces layers/interaction.py:36-44 when the caller already holds the stacked embeddings (in DeepFM/FM the
 * gather kernel produces the term itself).  x[b] = F rows of D floats at x + b*ldb.
 *   out_sum [B] = product_sum_pooling, out_bi [B,D] = Bi_interaction_pooling (either may be NULL)
 *   backward: dx[b,f,:] = (g_sum[b] + g_bi[b,:]) * (sum_f x[b,f,:] - x[b,f,:])                             */
int rp_fm_pool_fwd(const float *x, int64_t ldb, int F, int D, float *out_sum, float *out_bi, int64_t B,
                   rp_stream_t stream);
int rp_fm_pool_bwd(const float *x, int64_t ldb, int F, int D, const float *g_sum, const float *g_bi, float *dx,
                   int64_t lddx, int64_t B, rp_stream_t stream);

/* ---- K9: BatchNorm1d of the MMOE towers ---------------------------------------------------------------
 * replaces nn.BatchNorm1d at multi_task/mmoe.py:54.  Training: batch mean / biased variance per column
 * (deterministic two-stage reductions, variance taken around the mean), y = (x-mean)*rstd*gamma+beta; the caller
 * updates the running statistics from mean/var.  Backward: dgamma = sum dy*xhat, dbeta = sum dy,
 * dx = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)).  rp_batchnorm_apply(_bwd): given statistics (eval mode). */
int rp_batchnorm_workspace_bytes(int64_t M, int N, size_t *bytes);
int rp_batchnorm_train_fwd(const float *x, int64_t ldx, const float *gamma, const float *beta, float eps, float *y,
                           int64_t ldy, float *mean, float *var, float *rstd, int64_t M, int N, void *workspace,
                           size_t workspace_bytes, rp_stream_t stream);
int rp_batchnorm_train_bwd(const float *x, int64_t ldx, const float *dy, int64_t lddy, const float *mean,
                           const float *rstd, const float *gamma, float *dx, int64_t lddx, float *dgamma,
                           float *dbeta, int64_t M, int N, void *workspace, size_t workspace_bytes,
                           rp_stream_t stream);
int rp_batchnorm_apply(const float *x, int64_t ldx, const float *mean, const float *rstd, const float *gamma,
                       const float *beta, float *y, int64_t ldy, int64_t M, int N, rp_stream_t stream);
int rp_batchnorm_apply_bwd(const float *dy, int64_t lddy, const float *rstd, const float *gamma, float *dx,
                           int64_t lddx, int64_t M, int N, rp_stream_t stream);

/* ---- Dice activation (layers/activation.py:10-34: p = sigmoid(BatchNorm1d(x, affine=False, eps=1e-9, momentum=0.01)),
 * y = p x + (1 - p) alpha x).  The normalisation is rp_batchnorm_train_fwd / rp_batchnorm_apply with gamma = beta = NULL;
 * these are the gate around it on xhat = the normalised tensor (row-major, leading dimensions in floats):
 *   rp_dice_gate_fwd  y = x (alpha + s (1 - alpha)), s = sigmoid(xhat)
 *   rp_dice_gate_bwd  dx_direct = dy (alpha + s (1 - alpha)),  dxhat = dy x (1 - alpha) s (1 - s),  dal = dy x (1 - s)
 *                     (three packed [M, N] outputs; dalpha = column sums of dal: rp_batchnorm_colsum; dxhat goes on
 *                     through rp_batchnorm_train_bwd / rp_batchnorm_apply_bwd) */
int rp_dice_gate_fwd(const float *x, int64_t ldx, const float *xhat, int64_t ldh, const float *alpha, float *y, int64_t ldy,
                     int64_t M, int N, rp_stream_t stream);
int rp_dice_gate_bwd(const float *x, int64_t ldx, const float *xhat, int64_t ldh, const float *alpha, const float *dy,
                     int64_t lddy, float *dx_direct, float *dxhat, float *dal, int64_t M, int N, rp_stream_t stream);

/* building blocks of a BatchNorm1d whose batch statistics span several ranks (SyncBatchNorm1d of rec_pangu_amd/sharded.py;
 * mmoe.py:54 on the global batch).  The all-reduce between the stages is the caller's.
 *   rp_batchnorm_colsum     out[n] = sum_m x[m,n]  (center NULL)  |  sum_m (x[m,n] - center[n])^2  (center given)
 *   rp_batchnorm_bwd_sums   dbeta[n] = sum_m dy, dgamma[n] = sum_m dy * xhat  (xhat from the given mean / rstd)
 *   rp_batchnorm_bwd_apply  dx = gamma * rstd * (dy - mean_dy - xhat * mean_dyx) with the given (global) means
 * workspace: rp_batchnorm_workspace_bytes(M, N). */
/* nn.BatchNorm1d's running statistics of one training forward in ONE launch (mmoe.py:54 towers; torch/nn/modules/batchnorm.py):
 * num_batches_tracked (int64[1], may be NULL) += 1; running_x = (1 - m) running_x + m batch_x, the variance unbiased by
 * M / (M - 1); momentum < 0 = None: the cumulative average m = 1 / num_batches_tracked. */
int rp_batchnorm_update_running(const float *mean, const float *var, float *running_mean, float *running_var,
                                int64_t *num_batches_tracked, float momentum, int64_t M, int N, rp_stream_t stream);
int rp_batchnorm_colsum(const float *x, int64_t ldx, const float *center, float *out, int64_t M, int N, void *workspace,
                        size_t workspace_bytes, rp_stream_t stream);
int rp_batchnorm_bwd_sums(const float *x, int64_t ldx, const float *dy, int64_t lddy, const float *mean,
                          const float *rstd, float *dgamma, float *dbeta, int64_t M, int N, void *workspace,
                          size_t workspace_bytes, rp_stream_t stream);
int rp_batchnorm_bwd_apply(const float *x, int64_t ldx, const float *dy, int64_t lddy, const float *mean,
                           const float *rstd, const float *gamma, const float *mean_dy, const float *mean_dyx,
                           float *dx, int64_t lddx, int64_t M, int N, rp_stream_t stream);

/* ---- the narrow tail of the MLP as one launch each way (layers/deep.py:62-72 with hidden_units [.., 64, 64], output_dim 1:
 * DeepFM's dnn.net.{2,4,6}) --------------------------------------------------------------------------------------------
 *   hin [M, 64] (a ReLU output) -> [Linear 64x64 + ReLU] x n_hidden (1..3) -> Linear 64 -> 1 = logit [M]
 * rp_mlp_tail_fwd   W_hidden[l] [64 out, 64 in] (ldw[l] floats per row), b_hidden[l] [64] or NULL, h_out[l] [M, 64]: the
 *                   hidden outputs, saved for the backward; w_out [64], b_out [1] or NULL
 * rp_mlp_tail_bwd   dz [M] -> dhin [M, lddh] (already masked by hin > 0) and `grads` = dW_0 | .. | db_0 | .. | dw_out | db_out
 *                   packed (n_hidden*4096 + n_hidden*64 + 64 + 1 floats); acts[0] = hin (ldact0), acts[l] = h_out[l-1];
 *                   per-workgroup partials through `workspace`, summed in a fixed order (deterministic)
 * Split-bf16 six-product MFMAs (fp32-faithful) in every precision mode except RP_MATMUL_FP32 (callers compose the plain
 * entry points there).  rp_mlp_tail_fits: width 64, 1..3 hidden layers, row strides multiples of 4 floats. */
int rp_mlp_tail_fits(int n_hidden, int width, int64_t ldin);
int rp_mlp_tail_fwd(const float *hin, int64_t ldin, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                    const float *const *b_hidden, float *const *h_out, const float *w_out, const float *b_out, float *logit,
                    int64_t M, rp_stream_t stream);
int rp_mlp_tail_bwd_workspace_bytes(int64_t M, int n_hidden, size_t *bytes);
int rp_mlp_tail_bwd(const float *dz, int n_hidden, const float *const *W_hidden, const int64_t *ldw, const float *const *acts,
                    int64_t ldact0, const float *w_out, float *dhin, int64_t lddh, float *grads, int64_t M, void *workspace,
                    size_t workspace_bytes, rp_stream_t stream);
/* the same in two separately issued parts: parts = 1 the per-workgroup launch (dhin + partial sums into the workspace),
 * 2 = the second stage (workspace -> grads), 3 = both (= rp_mlp_tail_bwd).  The second stage reads only the workspace, so
 * a captured step issues it beside the launches that follow the first one. */
int rp_mlp_tail_bwd_parts(const float *dz, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                          const float *const *acts, int64_t ldact0, const float *w_out, float *dhin, int64_t lddh, float *grads,
                          int64_t M, void *workspace, size_t workspace_bytes, int parts, rp_stream_t stream);

/* The tail with the model's loss head inside (ranking/deepfm.py:61-66 on top of layers/deep.py:61-84; three launches of a
 * DeepFM step in one each way).  Forward: pred [M] = sigmoid(addends[0][m] + .. + tail logit) (0..3 other addends, summed in
 * that order, the tail's logit last), partial [rp_mlp_tail_loss_partials(M)] = per-workgroup sums of the BCE terms of
 * (pred + p_eps, label), logs clamped at -100: rp_loss_finish(partial, n, weight / M, loss) gives the mean loss — term for
 * term rp_sigmoid_bce_fwd's arithmetic (pred is bit-identical to the separate launches; the loss differs by the order of
 * its sum).  Backward: the logit's gradient is formed per row from (pred, label, gloss[0] * weight / M) — rp_sigmoid_bce_bwd
 * bit for bit — and written to dz_out [M] (NULL: not) for the other addends; everything else as rp_mlp_tail_bwd_parts. */
int rp_mlp_tail_loss_partials(int64_t M);
int rp_mlp_tail_fwd_bce(const float *hin, int64_t ldin, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                        const float *const *b_hidden, float *const *h_out, const float *w_out, const float *b_out,
                        const float *const *addends, int n_addends, const float *label, float p_eps, float *pred,
                        float *partial, int64_t M, rp_stream_t stream);
int rp_mlp_tail_bwd_bce(const float *pred, const float *label, const float *gloss, float p_eps, float weight, float *dz_out,
                        int n_hidden, const float *const *W_hidden, const int64_t *ldw, const float *const *acts,
                        int64_t ldact0, const float *w_out, float *dhin, int64_t lddh, float *grads, int64_t M,
                        void *workspace, size_t workspace_bytes, int parts, rp_stream_t stream);

/* ---- Dropout (layers/deep.py:66-68 inside the MLP chain; the multi-task towers, mmoe.py:55) -------------------------
 * y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from Philox4x32-10(counter = (element group, offset), key = seed): a
 * pure function of (seed, offset, element index).  mask: uint8 [M*N] (1 = kept), saved for the backward
 * dx = dy * keep / (1 - p).  One call consumes ONE offset value (the host advances its generator by one per call). */
int rp_dropout_fwd(const float *x, int64_t ldx, float *y, int64_t ldy, uint8_t *mask, int64_t M, int N, float p,
                   uint64_t seed, uint64_t offset, rp_stream_t stream);
/* the same with offset = *offset_dev + offset_delta read on the device: what a captured step records (frozen launch
 * arguments): offset_dev holds the generator offset of the step's start and is advanced at the step's end with
 * rp_counter_add_u64, so a replay draws the masks the eager loop would draw at that point of the generator's stream */
int rp_dropout_fwd_dev(const float *x, int64_t ldx, float *y, int64_t ldy, uint8_t *mask, int64_t M, int N, float p,
                       uint64_t seed, uint64_t offset_delta, const uint64_t *offset_dev, rp_stream_t stream);
int rp_counter_add_u64(uint64_t *counter, uint64_t delta, rp_stream_t stream);
int rp_dropout_bwd(const float *dy, int64_t lddy, const uint8_t *mask, float *dx, int64_t lddx, int64_t M, int N, float p,
                   rp_stream_t stream);

/* ---- K10: logit sum + sigmoid + BCE(mean) ---------------------------------------------------
 * replaces ranking/deepfm.py:61-63 (sigmoid + torch.nn.BCELoss) and multi_task/mmoe.py:127.
 *   z = sum_i z_ptrs[i][b] (n_addends <= 4; pass apply_sigmoid=0 when z is already a probability)
 *   pred[b] = sigmoid(z);  loss = weight * mean_b BCE(pred + p_eps, label), logs clamped at -100
 *   partial: float[rp_loss_partials(B)] workspace;  loss: device float[1] (written unless NULL) */
int rp_loss_partials(int64_t B);
int rp_sigmoid_bce_fwd(const float *const *z_ptrs, int n_addends, int apply_sigmoid, const float *label,
                       int64_t B, float p_eps, float weight, float *pred, float *partial, float *loss,
                       rp_stream_t stream);
/* rp_sigmoid_bce_fwd with loss[0] += instead of = : the second .. last task of a multi-task loss (mmoe.py:127: the sum of the
 * tasks' weighted means, added in task order like the reference's python sum) */
int rp_sigmoid_bce_fwd_accum(const float *const *z_ptrs, int n_addends, int apply_sigmoid, const float *label,
                             int64_t B, float p_eps, float weight, float *pred, float *partial, float *loss,
                             rp_stream_t stream);
/* loss[0] = scale * sum(partial[0..n)), fixed order (the second stage of rp_sigmoid_bce_fwd; also ends rp_mlp_tail_fwd_bce) */
int rp_loss_finish(const float *partial, int n, float scale, float *loss, rp_stream_t stream);
/* dz[b] = gloss[0] * weight/B * dBCE/dp * (apply_sigmoid ? p(1-p) : 1) */
int rp_sigmoid_bce_bwd(const float *pred, const float *label, const float *gloss, int64_t B, float p_eps,
                       float weight, int apply_sigmoid, float *dz, rp_stream_t stream);

/* ---- multi-id ("bag") lookups with pooling: the CSR / segmented embedding gather + sum-pool of the north star ----------
 * replaces EmbeddingLayer.forward(X, name="<col>_seq") (rec_pangu/models/layers/embedding.py:64-71: [B, L] ids ->
 * [B, L, D]) FOLLOWED BY MaskedSumPooling (layers/sequence.py:38-59: sum over dim 1) or MaskedAveragePooling
 * (layers/sequence.py:13-36: sum / (count of non-zero ELEMENTS per (b, d) + 1e-16)) — the [B, L, D] tensor never exists.
 *   ids: int64; dense bags: offsets = NULL, bag b = ids[b*L .. b*L+L); CSR bags: offsets int64 [B+1] into ids.
 *   The table is rows [row_base, row_base + row_count) of the arena; an id outside [0, row_count) sets *err_flag and
 *   reads row 0 (the reference raises IndexError).  mode 0 = sum, 1 = masked average.
 *   out [B, ldo >= D]; inv_out [B, D] (mode 1, for the backward: 1 / (count + 1e-16)) or NULL;
 *   bag_out int32 [nnz] (CSR, for the backward: the bag of every id) or NULL.
 * rp_embed_pool_bwd: grad_arena[row] (+)= sum over the ids p of that row of g[bag(p), :] (* scale[bag(p), :]), from the
 *   (arena row, flat id position) pairs sorted by row (rp_embed_keys with F = 1 over the flat ids, rp_sort_pairs_i32);
 *   bag(p) = bag_of[p], or p / L when bag_of is NULL.  Deterministic (the segmented reduction of rp_embed_grad_reduce,
 *   same workspace): a padding id present in every bag is one long run like any hot row.
 * rp_seq_pool_fwd / _bwd: the same two poolings on an explicit, contiguous [B, L, D] tensor (the drop-in modules). */
int rp_embed_gather_pool_fwd(const float *arena, int64_t row_base, int64_t row_count, const int64_t *ids,
                             const int64_t *offsets, int64_t L, int64_t B, int D, int mode, float *out, int64_t ldo,
                             float *inv_out, int32_t *bag_out, int32_t *err_flag, rp_stream_t stream);
int rp_embed_pool_bwd(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int D, const float *g, int64_t ldg,
                      const float *scale, const int32_t *bag_of, int64_t L, float *grad_arena, int accumulate,
                      void *workspace, size_t workspace_bytes, rp_stream_t stream);
int rp_seq_pool_fwd(const float *e, int64_t B, int64_t L, int D, int mode, float *out, float *inv_out, rp_stream_t stream);
int rp_seq_pool_bwd(const float *g, const float *inv, int64_t B, int64_t L, int D, float *de, rp_stream_t stream);

/* ---- K11: fused Adam (torch.optim.Adam single-tensor operation order) -----------------------
 * replaces trainer.py:75 + model_pipeline.py:57-58 (optimizer.step(); model.zero_grad()).
 * n_tensors <= RP_MAX_FIELDS per call; zero_grad=1 also clears g (the fused zero_grad).
 * lr, betas and eps are DOUBLES, like the python floats torch.optim.Adam computes its bias corrections from (a float
 * beta2 = 0.999f shifts 1 - beta2^t by 5e-6 relative at t ~ 600).
 * The second-moment arrays (`v_ptrs`, and `v` of the lazy entry points below) hold s = sqrt(v), not v: Adam only uses
 * sqrt(v), and a zero-gradient step is then s *= sqrt(b2) (one multiply, no transcendental) — chosen per element on
 * g == 0 by the one update function all these kernels share, so dense and lazy execution stay bit-identical.  The
 * host side squares s when it exports torch.optim.Adam-style state (rec_pangu_amd/optim.py).  */
int rp_adam_step(float *const *p_ptrs, float *const *g_ptrs, float *const *m_ptrs, float *const *v_ptrs,
                 const int64_t *sizes, int n_tensors, double lr, double beta1, double beta2, double eps,
                 int64_t step, int zero_grad, const float *step_scalars, const int32_t *t_dev, rp_stream_t stream);
/* DEVICE-RESIDENT STEP COUNTERS (hipGraph replays freeze every launch argument at capture, so the step number cannot be
 * one).  Wherever an entry point of this section takes `t_dev` (device int32[1] = number of COMPLETED optimizer steps,
 * NULL = use the host arguments), the kernel reads the step from it: rp_adam_step applies step *t_dev + 1 with the
 * scalars step_scalars[*t_dev + 1] (the float2 table of rp_adam_step_scalars rows indexed by step; lr / step arguments
 * ignored), rp_lazy_adam_rows replays to *t_dev (real_step: applies *t_dev + 1), rp_lazy_adam_cf_table builds the table
 * for t_end = *t_dev (a fixed-size window of the youngest stamps: consecutive replays keep it complete).  rp_counter_add advances a
 * counter on the stream.  rec_pangu_amd/graph_step.py captures fwd + bwd + optimizer on top of these. */
int rp_counter_add(int32_t *counter, int32_t delta, rp_stream_t stream);
/* the same for 1..8 distinct counters in one launch (counters: host array of device pointers) */
int rp_counters_add(int32_t *const *counters, int n, int32_t delta, rp_stream_t stream);

/* ---- exact LAZY dense Adam for arena rows -----------------------------------------------------
 * Same semantics as rp_adam_step over the whole arena (trainer.py:75: DENSE Adam, every row every step), but a
 * row's zero-gradient steps are replayed in registers when the row is next needed instead of being streamed
 * through HBM every step.  last[row] (int32, 0 = never updated) is the step the stored (p,m,v) are current at;
 * step_scalars is a device float2 table indexed by step: {A_t, B_t} from rp_adam_step_scalars (the eps given
 * there is the one that counts for the serial replay).  The serial replay runs the dense kernel's update function with
 * g = 0: bit-identical results.
 *   rp_embed_keys       arena-row keys of a batch (same check/flag/clamp as the gather) — needed before the gather
 *   rp_lazy_adam_rows   for every UNIQUE row of `sorted_keys` (sorted; duplicates skipped): replay steps
 *                       last+1 .. t_target(-1); if real_step also apply step t_target with g = grad row
 *                       (and clear it if zero_grad); last[row] = t_target
 *   rp_lazy_adam_flush  replay every row up to t_target (before a checkpoint / state_dict / eval of raw tables)
 * Any D >= 1.  Pure replays (real_step = 0) and the flush of rows wider than 16 floats run ONE ROW PER WAVE (the chain
 * length differs per row: side by side in a wave every row would wait for the longest), candidates strided over the
 * waves so that every wave sees the same mix of tables; the real step and narrow rows (D = 1: the LR_Layer's tables)
 * use 16-byte vector lanes when D % 4 == 0 and the arenas are 16-byte aligned, scalar lanes otherwise. */
int rp_embed_keys(const int64_t *row_base, const int64_t *row_count, const int64_t *const *idx_ptrs, int F, int64_t B,
                  int32_t *keys_out, int32_t *err_flag, rp_stream_t stream);
int rp_adam_step_scalars(double lr, double beta1, double beta2, double eps, int64_t step, float *sa,
                         float *sb);
/* the same for n consecutive steps step0 .. step0 + n - 1: out[2 i], out[2 i + 1] = {A, B} of step step0 + i (HOST array) */
int rp_adam_step_scalars_range(double lr, double beta1, double beta2, double eps, int64_t step0, int64_t n, float *out); /* {A_t, B_t} = {1/sqrt(1-b2^t), eps} / (-lr/(1-b1^t)), computed in double: the update
                                      * is p += m * rcp(s * A_t + B_t) (s = sqrt(v)); lr = 0 gives (0, -inf) */
int rp_lazy_adam_rows(const int32_t *sorted_keys, int64_t n, int D, float *p, float *g, float *m, float *v,
                      int32_t *last, const float *step_scalars, int64_t t_target, int real_step, int zero_grad,
                      double beta1, double beta2, double eps, const float *cf_table, int64_t cf_from, const int32_t *t_dev,
                      rp_stream_t stream);
int rp_lazy_adam_flush(int64_t rows, int D, float *p, float *m, float *v, int32_t *last, const float *step_scalars,
                       int64_t t_target, double beta1, double beta2, double eps, const float *cf_table, int64_t cf_from,
                       rp_stream_t stream);
/* CLOSED-FORM replay (cf_table != NULL in the two entry points above; tolerance mode, NULL = the bit-exact serial
 * replay).  The k zero-gradient steps l+1 .. l+k of an element are  m b1^k,  s r^k  and
 *     p + m * sum_i w_i / (s a_i + eps),  w_i = -lr_{l+i}/(1-b1^{l+i}) b1^i,  a_i = r^i / sqrt(1-b2^{l+i}),  r = sqrt(b2);
 * every term is expanded around the w-weighted mean abar of the a_i:  q [N0 + y^2 N2 - y^3 N3 + y^4 N4],
 * q = 1/(s abar + eps), y = s q, N_n = sum_i w_i (a_i - abar)^n — uniformly convergent in s (|y (a_i - abar)| <=
 * |a_i/abar - 1|), relative truncation error of the summed update <= 9e-8 once l >= 256 at b2 = 0.999
 * (profiles/microbench/probes/closed_form_replay.py).  Steps up to `cf_from` are still replayed serially; steps cf_from+1 .. t cost one
 * reciprocal and ~10 fp32 operations per element whatever their number.  Here `eps` counts (it is the eps of the
 * expansion) and must be the one the step scalars were built with.
 *   rp_lazy_adam_cf_table  cf_table: `capacity` entries of 8 floats (abar, N0, N2, -N3, N4, b1^k, r^k, 0) with TWO index
 *                          meanings: the first five of entry l describe a replay that starts at stamp l (steps l+1 ..
 *                          t_end), cf_from <= l < t_end; the two powers of entry k belong to a replay of k steps.  Valid
 *                          for replays that END at t_end (rp_lazy_adam_rows(real_step=0, t_target=t_end), (real_step=1,
 *                          t_target=t_end+1), rp_lazy_adam_flush(t_target=t_end)).  Built on the device, in double:
 *                          built_to < 0 = a fresh buffer (the power columns and every stamp: O(t_end) once); otherwise
 *                          built_to = the t_end of the previous call on this buffer and only the stamps that were not
 *                          final then ( >= built_to - J, J = rp_lazy_adam_cf_terms) are rebuilt: O(1) per training step
 *                          whatever the step count.  With t_dev (consecutive graph replays) the window is the J + 4
 *                          youngest stamps below *t_dev.  ns_d: device double2 table indexed by step:
 *                          {-lr_j/(1-b1^j), 1/sqrt(1-b2^j)} (rows of steps already taken never change).
 *   rp_lazy_adam_cf_terms  number of leading terms that carry weight (b1^J < 1e-17) */
int rp_lazy_adam_cf_terms(double beta1, int *terms);
/* DEFERRED real step (opt-in execution mode of the same optimizer; no reference counterpart beyond trainer.py:75).
 * A row's real step needs only its own gradient row, which stays in the (dense) gradient arena: it can wait, like the
 * zero-gradient steps, until the row is next needed.  One launch per training step then does what the rows of the
 * incoming batch are owed — 8 rows of traffic per unique row instead of 6 (replay) + 8 (step).
 *   last[row] >= 0: as above.  last[row] < 0: (p,m,v) current through step l = -last-1 and g[row] = the gradient of
 *   step l+1, not applied yet.
 *   rp_lazy_adam_catchup  for every unique row of sorted_keys: a pending gradient whose step has been taken
 *                         (l+1 <= t_done) is applied with the scalars of step l+1 and its row cleared, then the
 *                         zero-gradient steps up to t_done (serial up to cf_from, closed form beyond if cf_table);
 *                         mark != 0: the row is stamped pending for step t_done+1 (the backward of the forward this
 *                         launch precedes writes its gradient; a row that receives none holds zeros, and the real
 *                         step with g = 0 IS the zero-gradient step).  mark == 2: the same stamp, and the applied
 *                         rows are NOT cleared — the caller guarantees that the coming backward overwrites the gradient
 *                         row of every stamped row (and clears them itself, rp_zero_rows, if that backward never comes).
 *                         t_dev: device counter of completed steps.
 *   rp_lazy_adam_flush_deferred  the same for every row of the arena through t_target (g may be NULL when no
 *                         gradient arena exists yet); stamps last[row] = t_target.
 * Per row the same operations on the same values in the same order as rp_lazy_adam_rows: identical bits after a flush.
 * shadow_bf16 (both; NULL = none): the bf16 LOOKUP COPY of the tables ([rows, D] bf16, 16-byte aligned) of the bf16-storage
 * training mode — every parameter row the launch writes is also written there, rounded to nearest even; rp_rows_to_bf16
 * refreshes the run heads of a sorted key list after an update that went another way. */
int rp_lazy_adam_catchup(const int32_t *sorted_keys, int64_t n, int D, float *p, float *g, float *m, float *v,
                         int32_t *last, const float *step_scalars, int64_t t_done, int mark, double beta1, double beta2,
                         double eps, const float *cf_table, int64_t cf_from, const int32_t *t_dev, void *shadow_bf16,
                         rp_stream_t stream);
int rp_lazy_adam_flush_deferred(int64_t rows, int D, float *p, float *g, float *m, float *v, int32_t *last,
                                const float *step_scalars, int64_t t_target, double beta1, double beta2, double eps,
                                const float *cf_table, int64_t cf_from, void *shadow_bf16, rp_stream_t stream);
int rp_rows_to_bf16(const int32_t *sorted_keys, int64_t n, int D, const float *p, void *shadow_bf16, rp_stream_t stream);
int rp_lazy_adam_cf_table(const double *ns_d, int64_t t_end, int64_t cf_from, double beta1, double beta2, float *cf_table,
                          int64_t capacity, int64_t built_to, const int32_t *t_dev, rp_stream_t stream);

/* ---- request routing for row-sharded tables (rec_pangu_amd/sharded.py; no reference counterpart: the reference is
 * single-device).  Arena row r lives on rank r % world at local row r / world.
 *   rp_shard_keys   keys_out[f*B + b] = (owner << lbits) | local_row of id_f[b] (int32; range check / flag / row 0 as
 *                   the gather does); owner bits + lbits <= 31
 *   rp_route_build  from the SORTED keys (rp_sort_pairs_i32): slot_sorted [n] int32 = unique-request slot of the j-th
 *                   sorted request; slot_of_pair [n] int64 = slot of request p; uniq_rows [n] int64 whose first
 *                   counts[world] entries are the local rows to ask for, grouped by ascending owner (the all-to-all
 *                   send order); counts [world+1] int64 = unique requests per owner, then their total */
int rp_shard_keys(const int64_t *row_base, const int64_t *row_count, const int64_t *const *idx_ptrs, int F, int64_t B,
                  int world, int lbits, int32_t *keys_out, int32_t *err_flag, rp_stream_t stream);
int rp_route_workspace_bytes(int64_t n, int world, size_t *bytes);
int rp_route_build(void *workspace, size_t workspace_bytes, const int32_t *sorted_keys, const int32_t *sorted_pos,
                   int64_t n, int world, int lbits, int32_t *slot_sorted, int64_t *slot_of_pair, int64_t *uniq_rows,
                   int64_t *counts, rp_stream_t stream);
/* fixed-capacity form of the exchange (no host-side split sizes, hence no host sync per step): rewrites the compact
 * slots of rp_route_build as owner * capacity + index within the owner (slot_sorted in place, slot_of_pair) and fills
 * rows_padded[world * capacity] (zero-filled by the caller: unused slots ask for local row 0 and receive a zero
 * gradient).  More than `capacity` unique requests for one owner: bit 1 of *err_flag is set, the step is invalid. */
int rp_route_pad(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int world, int lbits,
                 int64_t capacity, const int64_t *counts, int32_t *slot_sorted, int64_t *slot_of_pair,
                 int64_t *rows_padded, int32_t *err_flag, rp_stream_t stream);
/* FIELD-MAJOR copy of a route's sorted request list (round 6).  Sorted by composite key the list is ordered (owner, field,
 * row); the first layer's backward launches (rp_embed_grad_seg / _ss / _smp) want field f's B requests at [f B, (f + 1) B)
 * with equal keys adjacent.  slot_fm / pos_fm [n] int32: the (slot, position) pairs of slot_sorted / sorted_pos moved
 * segment by segment — inside a field the order is (owner, row) = ascending slot, positions of a run keep their order.
 * n = F * B requests, world * F <= 1024 segments; delta: int64 scratch [world * F].  world == 1: the list is field-major
 * already (callers skip the call).  No reference counterpart (the reference is single-device: trainer.py:75). */
int rp_route_field_major(const int32_t *sorted_keys, const int32_t *sorted_pos, const int32_t *slot_sorted, int64_t n,
                         int64_t B, int world, int lbits, int32_t *slot_fm, int32_t *pos_fm, int64_t *delta,
                         rp_stream_t stream);

/* ---- LAUNCH PLANS: record a sequence of this library's kernel launches once, re-issue it with one call (csrc/plan.hip).
 * No reference counterpart (the reference's step loop is Python: model_pipeline.py:47-61); this is the host-side
 * mechanism behind rec_pangu_amd/graph_step.py's captured training step.  Between rp_plan_begin and rp_plan_end every
 * kernel launch any entry point of this library issues — from any host thread — is appended to the plan with its packed
 * arguments (and still issued: under a stream capture it becomes a graph node, otherwise it runs).  rp_plan_replay
 * re-issues the recorded launches with the recorded arguments on `stream`; launches recorded under rp_plan_section(1)
 * are independent of the others and go to a side stream owned by the plan, forked at rp_plan_fork_here (default: the start
 * of the replay) and joined at its end.  The caller guarantees that every address baked into the plan stays valid and that step numbers are read
 * on the device (t_dev arguments).  One plan may be recorded at a time, process-wide.
 *   rp_plan_info          launches recorded, those of section 1, distinct streams they were issued on while recording
 *   rp_graph_node_counts  kernel nodes / other nodes (memset, memcpy, ...) of a captured hipGraph_t: the check that a
 *                         captured step holds no launch the plan has not seen */
int rp_plan_begin(void **plan_out);
int rp_plan_section(int section);
int rp_plan_fork_here(void); /* the side section is forked in front of the NEXT main launch (default: start of the replay) */
/* rp_plan_section(2): an INLINE fork — the launches recorded under it run on a second side stream beside the main launches
 * recorded after them, from the point where they were recorded until rp_plan_join() (the first layer's weight gradient
 * beside the fused gather backward: both depend only on the masked dH).  The caller keeps every buffer those launches use
 * alive until the join: the capture's allocator assumes ONE stream and would hand a freed workspace to the next launch. */
int rp_plan_join(void);
/* explicit fork point of the inline section (2): its launches recorded after this mark depend on what the main stream held
 * HERE (main launches recorded between the mark and them run beside them), until the next rp_plan_join */
int rp_plan_fork2_mark(void);
/* Input rebinding: after rp_plan_end, name the addresses of the static input buffers the step was recorded on (n of them);
 * every 8-byte word of the recorded launch arguments that holds one is remembered (n_sites).  rp_plan_set_inputs(addrs[n])
 * then makes the next replays read buffer i from addrs[i] instead — the current batch's own tensors (same shape, dtype,
 * contiguity; alive until the replay has run), no staging copy. */
int rp_plan_bind_inputs(void *plan, const uint64_t *addrs, int n, int *n_sites);
int rp_plan_set_inputs(void *plan, const uint64_t *addrs, int n);
/* what rp_plan_bind_inputs would find, per input: sites[i] = argument words holding exactly addrs[i]; *n_interior = words that
 * point INSIDE buffer i (addrs[i] < word < addrs[i] + nbytes[i]): arguments derived from an input, which rp_plan_set_inputs
 * cannot re-point — the caller keeps its staging copy then (graph_step.py) */
int rp_plan_bind_report(void *plan, const uint64_t *addrs, const uint64_t *nbytes, int n, int32_t *sites, int *n_interior);
/* HOST MARKS: a step interleaved with work the library does not issue (the collectives of the row-sharded path) is recorded
 * as segments — rp_plan_host_mark() at each such point while recording (*index_out = 0, 1, ...; the foreign work is NOT
 * issued under the capture), rp_plan_replay_segment(plan, k, stream) for the launches between mark k - 1 and mark k, the
 * foreign work issued by the caller in between on the same stream (every launch of the segment on that one stream, in
 * recorded order, whatever section it was recorded under). */
int rp_plan_host_mark(int *index_out);
int rp_plan_host_marks(void *plan, int *n_marks);
int rp_plan_replay_segment(void *plan, int seg, rp_stream_t stream);
/* the main stream waits HERE for the side section (1) of the replay (default: at the end of the replay) — for a step that
 * itself consumes what the side section produces (the next batch's sorted keys: graph_step.py, catch-up ahead) */
int rp_plan_join_side(void);
/* the inline section (2) waits here for what the main stream holds at this point (a second dependency edge for a section
 * forked earlier; the fork mark while the section is not open yet) */
int rp_plan_side2_sync(void);
/* a non-blocking stream of the lowest priority the device offers (side streams that should yield to the main stream's
 * launches); the caller owns it */
int rp_stream_create_low(void **stream_out);
int rp_plan_is_recording(void);
int rp_plan_end(void *plan);
int rp_plan_info(void *plan, int *n_nodes, int *n_side, int *n_streams);
/* the streams sections 1 / 2 are re-issued on (NULL: the plan creates one); before the first replay */
int rp_plan_set_streams(void *plan, rp_stream_t side, rp_stream_t side2);
int rp_plan_inline_count(void *plan, int *n_inline);
int rp_plan_replay(void *plan, rp_stream_t stream);
/* Live timing of ONE launch inside replayed steps (bench.py's roofline: a replay runs no host code between its launches, so
 * nothing outside the library can bracket one of them).  rp_plan_set_probe(plan, k): the k-th recorded launch (0 ..
 * n_nodes - 1 of rp_plan_info, recorded order) of the following replays is bracketed by a HIP timing-event pair on the stream it
 * is issued on; -1 = off.  rp_plan_probe_ms: elapsed milliseconds of the last replay's pair (waits for it).
 * rp_plan_launch_name: the launch's (demangled) kernel name and section (0 main stream, 1 side, 2 inline side). */
int rp_plan_set_probe(void *plan, int launch);
int rp_plan_probe_ms(void *plan, float *ms);
/* Completion markers for the host's run-ahead bound (graph_step.py keeps at most 6 replayed steps in flight): an event
 * without timing and without the system-scope fence of a default event record (the host waits for it; it does not read
 * device memory on its strength).  (No counterpart in the reference.) */
int rp_marker_create(void **marker);
int rp_marker_record(void *marker, rp_stream_t stream);
int rp_marker_wait(void *marker);
int rp_marker_destroy(void *marker);
/* Host-side stall finder: the slowest single HIP call (kind 0 = kernel launch, 1 = event record, 2 = stream wait) issued by
 * the replays since the last reset, the plan node it belongs to (nodes, not launches: markers count) and its host time in
 * milliseconds.  reset != 0 starts a new window.  (No counterpart in the reference: its step is eager torch.) */
int rp_plan_slowest_call(void *plan, int *kind, int *node, double *ms, int reset);
int rp_plan_launch_name(void *plan, int launch, char *buf, int buf_len, int *section);
int rp_plan_destroy(void *plan);
int rp_graph_node_counts(void *graph, int *n_kernel, int *n_other);
/* dst_ptrs[i][0 : bytes[i]] = src_ptrs[i][..] for i < n in ONE launch (per 96 buffers): a batch of ~40 columns into the static
 * input buffers of a captured step.  Host arrays of device addresses; buffers must not overlap. */
int rp_multi_copy(void *const *dst_ptrs, const void *const *src_ptrs, const uint64_t *bytes, int n, rp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REC_PANGU_HIP_H */
