/*
 * xdfm.h -- C ABI of libxdfm.so: the MI355X (gfx950) implementation of the
 * xDeepFM CTR training path of recommenders.models.deeprec.
 *
 * The reference has no FFI for this path: its compute boundary is
 * `sess.run(fetches, feed_dict)` (reference base_model.py:379-388, :402, :416)
 * into TensorFlow 2.7.  Each entry point below replaces the TensorFlow ops one
 * reference call site builds; the comment on it names that call site
 * (paths are relative to /root/reference/recommenders/models/deeprec/).
 *
 * Conventions
 *  - every function returns an int status: 0 = ok, <0 = error (see XDFM_E_*);
 *    xdfm_last_error() returns the message of the calling thread's last error.
 *  - device pointers are plain `void*`-compatible pointers into HBM owned by the
 *    caller (PyTorch-ROCm allocations in the Python host); nothing is retained
 *    after a call returns, nothing is allocated or freed by the library on the
 *    device (workspaces are sized by *_workspace_bytes and passed in).
 *  - `stream` is a hipStream_t passed as void*; all device entry points only
 *    enqueue work on it and return without synchronising.
 *  - all floating point tensors are IEEE fp32, row-major, densely packed.
 *  - `err_flag` is a device int32 the kernels OR error bits into
 *    (XDFM_DEVERR_*); the host reads it at its next natural sync point.
 */
#ifndef XDFM_H_
#define XDFM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XDFM_ABI_VERSION 4

#define XDFM_OK 0
#define XDFM_E_INVALID (-1)   /* bad argument (null pointer, size, enum) */
#define XDFM_E_HIP (-2)       /* a HIP runtime call failed */
#define XDFM_E_WORKSPACE (-3) /* workspace too small */
#define XDFM_E_PARSE (-4)     /* malformed libffm text (host parser) */
#define XDFM_E_IO (-5)        /* file could not be opened / mapped */
#define XDFM_E_EOF (-6)       /* reader exhausted */

/* bits OR-ed into the device error flag */
#define XDFM_DEVERR_ID_RANGE 1  /* feature id outside [0, FEATURE_COUNT) */
#define XDFM_DEVERR_ROW_RANGE 2 /* row (sample*FIELD_COUNT+field) outside [0, B*F) */
#define XDFM_DEVERR_ROW_ORDER 4 /* dnn_feat_indices rows not ascending */

/* optimizer enum: base_model.py:283-302 (TF1 optimizers, defaults of TF 2.7) */
#define XDFM_OPT_SGD 0
#define XDFM_OPT_ADAM 1
#define XDFM_OPT_ADAGRAD 2
#define XDFM_OPT_RMSPROP 3  /* beta2 = decay, eps; slot_m = rms (starts at 1) */
#define XDFM_OPT_ADADELTA 4 /* beta2 = rho, eps; slot_m = accum, slot_v = accum_update */
#define XDFM_OPT_FTRL 5     /* lr_power -0.5, l1 = l2 = 0; slot_m = accum (starts at 0.1), slot_v = linear */

/* loss enum: base_model.py:215-262 */
#define XDFM_LOSS_LOG 0           /* tf.compat.v1.losses.log_loss, eps 1e-7 */
#define XDFM_LOSS_CROSS_ENTROPY 1 /* sigmoid_cross_entropy_with_logits, mean */
#define XDFM_LOSS_SQUARE 2        /* sqrt(mean((pred-y)^2)) */

/* activation enum: base_model.py:339-353 */
#define XDFM_ACT_IDENTITY 0
#define XDFM_ACT_RELU 1
#define XDFM_ACT_SIGMOID 2
#define XDFM_ACT_TANH 3
#define XDFM_ACT_ELU 4

int xdfm_abi_version(void);
const char* xdfm_last_error(void);

/* Measurement only (no reference counterpart): when enabled, every kernel launch of the
 * library is bracketed by two HIP events on its launch stream.  xdfm_prof_collect waits for
 * the recorded events, writes {"<kernel name>": {"ms": total, "n": launches}, ...} as JSON
 * text into `json` and forgets them.  bench.py's roofline line is computed from this. */
int xdfm_prof_enable(int32_t on);
int xdfm_prof_collect(char* json, size_t cap);
/* Measurement only (no reference counterpart): device-side time stamps, for the timeline of a replayed HIP
 * graph, which neither events nor a tracer show unchanged.  With a device buffer of XDFM_STAMP_SLOTS pairs
 * installed (the caller fills it with {UINT64_MAX, 0} pairs first), every kernel of the training step adds
 * the constant 100 MHz clock to its pair: [2 id] = minimum over its blocks and launches (start), [2 id + 1]
 * = maximum (end); ids in csrc/stamp.h, names in tools/stamp_timeline.py.  NULL removes the buffer (the
 * cost is then one scalar load per block).  Applies to the calling thread's current device. */
#define XDFM_STAMP_SLOTS 32
int xdfm_debug_stamps(uint64_t* device_buf);

/* ------------------------------------------------------------------------
 * libffm text ingest (host).  Replaces FFMTextIterator.parser_one_line /
 * load_data_from_file / _convert_data  (io/iterator.py:90-219).
 * ---------------------------------------------------------------------- */
typedef struct xdfm_ffm_reader xdfm_ffm_reader;

typedef struct xdfm_ffm_batch_info {
  int64_t n_rows;      /* samples in this batch (== batch_size except the last) */
  int64_t nnz;         /* field:feature:value tokens in this batch */
  int64_t dnn_width;   /* dnn_feat_shape[1] = max per-(sample,field) count (0 if nnz==0) */
  int64_t imp_bytes;   /* total bytes of the impression-id strings */
  int64_t first_line;  /* 1-based line number of the first sample (for messages) */
} xdfm_ffm_batch_info;

/* Opens (mmaps) `path`.  col_sep / id_sep are the reference's col_spliter / ID_spliter
 * (single characters here).  iterator.py:50-63, :132. */
int xdfm_ffm_open(const char* path, int64_t field_count, int64_t feature_count,
                  char col_sep, char id_sep, xdfm_ffm_reader** out);
/* Parses the next `batch_size` lines (fewer at end of file) into the reader's
 * staging area and reports the sizes.  Returns XDFM_E_EOF when no line is left,
 * XDFM_E_PARSE on a malformed line.  iterator.py:118-150, :90-116. */
int xdfm_ffm_next(xdfm_ffm_reader* r, int64_t batch_size, xdfm_ffm_batch_info* info);
/* Copies the staged batch into caller buffers, in the reference's layout
 * (iterator.py:152-219).  Any pointer may be NULL to skip that array.
 *   labels[n_rows]                      float32   (labels[:,0])
 *   fm_feat_indices[nnz*2]              int64     [sample, feature] in file order
 *   fm_feat_values[nnz]                 float32
 *   dnn_feat_indices[nnz*2]             int64     [sample*F+field, k-th of that field], sorted
 *   dnn_feat_values[nnz]                int64     feature id, same order
 *   dnn_feat_weights[nnz]               float32   value, same order
 *   row_ptr[n_rows*F+1]                 int32     CSR offsets of the dnn view (build extra;
 *                                                 only valid when every row is in range)
 *   imp_offsets[n_rows+1], imp_chars[imp_bytes]   impression ids; offset pair equal and
 *   imp_present[n_rows]                           present==0 means "no id" (reference: int 0)
 */
int xdfm_ffm_fill(xdfm_ffm_reader* r, float* labels, int64_t* fm_feat_indices,
                  float* fm_feat_values, int64_t* dnn_feat_indices, int64_t* dnn_feat_values,
                  float* dnn_feat_weights, int32_t* row_ptr, int64_t* imp_offsets,
                  char* imp_chars, uint8_t* imp_present);
int xdfm_ffm_close(xdfm_ffm_reader* r);

/* ------------------------------------------------------------------------
 * Device batch preparation.
 * ---------------------------------------------------------------------- */
/* CSR offsets + per-token row from the (sorted) dnn_feat_indices[:,0] column that
 * `tf.SparseTensor(dnn_feat_indices, ...)` carries (xDeepFM.py:82-91).
 *   dnn_feat_indices : int64 [nnz,2] (device)      row_ptr : int32 [n_rows+1]
 *   tok_row          : int32 [nnz]  */
int xdfm_build_row_ptr(const int64_t* dnn_feat_indices, int64_t nnz, int64_t n_rows,
                       int32_t* row_ptr, int32_t* tok_row, int32_t* err_flag, void* stream);

typedef struct xdfm_opt_args {
  int32_t kind;       /* XDFM_OPT_* */
  float lr;           /* learning_rate */
  float beta1, beta2; /* Adam */
  float eps;          /* Adam epsilon (outside the sqrt, TF1 placement) */
  float beta1_power;  /* beta1^t, t = 1-based step being applied */
  float beta2_power;  /* beta2^t */
  float l2;           /* coefficient of sum(p^2)/2, gradient l2*p */
  float l1;           /* coefficient of sum|p|, gradient l1*sign(p) */
  float clip_norm;    /* >0: per-tensor tf.clip_by_norm(g, clip_norm); 0: off */
  int32_t reserved;   /* keeps the pointer below 8-byte aligned; set to 0 */
  /* Optional device scalars (NULL: use beta1_power / beta2_power above).  opt_state[2] is the
   * Adam step size lr*sqrt(1-beta2^t)/(1-beta1^t) kept up to date by xdfm_opt_advance; with it
   * no host value changes from step to step, so a step can be captured in a HIP graph. */
  const float* opt_state;
} xdfm_opt_args;

/* opt_state[4] = {beta1^t, beta2^t, step size, t}: advance t by one (one small kernel).
 * Initialise the buffer to {1, 1, 0, 0} before the first step.  Build extra.
 * drift_tab (NULL ok): the table row ageing reads under Adam (below); the call rebuilds its entries for
 * the new step count t in slot t & 1, so that kernels still ageing rows to t - 1 keep reading theirs. */
int xdfm_opt_advance(float* opt_state, float lr, float beta1, float beta2, float eps, float* drift_tab,
                     void* stream);

/* ------------------------------------------------------------------------
 * Row ageing: sparse row updates (xdfm_embed_bwd mode 0) that give the reference's numbers.
 * In the reference the table's gradient is dense (the regulariser ops of base_model.py:121-149 sit on
 * the table even at coefficient 0), so apply_gradients (base_model.py:322) moves
 * every row every step: a row without a token in the batch still takes Adam's m <- b1 m, v <- b2 v,
 * p <- p - lr_t m / (sqrt(v) + eps).  row_step[id] (int32) records the number of optimizer steps a
 * row's stored (p, m, v) are current for; a call that is handed an xdfm_row_age applies the
 * G = opt_state[3] + step_offset - row_step[id] steps the row missed as it reads it:
 *   xdfm_embed_fwd / xdfm_embed_gather   aged row formed in registers, nothing written
 *   xdfm_embed_bwd (mode 0)              (p, m, v) aged in the registers of the row update, then the
 *                                        gradient applied, row_step <- opt_state[3] + step_offset + 1
 *   xdfm_embed_catchup                   aged rows written back (checkpoints, table exports)
 * Adam's G steps are summed in closed form (error below 1e-6 of the movement at any learning rate,
 * tools/drift_sim.py) with coefficients from the drift table: xdfm_drift_tab_floats() floats, filled by
 * xdfm_drift_tab_init (once, and after opt_state was set from a checkpoint) and kept current by
 * xdfm_opt_advance.  RMSProp / Adadelta decay their accumulators; FTRL re-derives p = -linear /
 * quadratic as TF's dense ApplyFtrl does every step (a row that never had a gradient becomes 0);
 * SGD and Adagrad have no drift.  opt_state[3] is a float: exact up to 2^24 steps.
 * No counterpart in the reference (its optimizer streams the whole table every step).
 * ---------------------------------------------------------------------- */
typedef struct xdfm_row_age {
  int32_t* row_step;       /* device, one entry per table row */
  int64_t step_stride;     /* 4-byte words between consecutive entries (0 = dense) */
  const float* slot_m;     /* the table's optimizer slots, strides as the table (read by the gathers; */
  const float* slot_v;     /*   xdfm_embed_bwd / xdfm_embed_catchup take them as arguments of their own) */
  const float* lin_m;      /* lin_w's slots (NULL without a linear part) */
  const float* lin_v;
  const float* drift_tab;  /* Adam: the drift table; NULL for the other optimizers */
  const xdfm_opt_args* opt; /* kind, lr, betas, eps and opt_state (required: the step count is read there) */
  int32_t step_offset;     /* 0, or -1 once xdfm_opt_advance has counted the step being applied */
  int32_t reserved;
} xdfm_row_age;

int xdfm_drift_tab_floats(size_t* floats);
int xdfm_drift_tab_init(float* drift_tab, const float* opt_state, float lr, float beta1, float beta2, float eps,
                        void* stream);

/* ------------------------------------------------------------------------
 * Field-embedding lookup + linear part + FM part (forward).
 *   tf.nn.embedding_lookup_sparse(..., combiner="sum")   xDeepFM.py:92-100
 *   sparse_dense_matmul(x, w)                            xDeepFM.py:128   (bias added by the caller)
 *   0.5*sum((xE)^2 - (x^2)(E^2))                         xDeepFM.py:153-158
 *   table[N,D], lin_w[N] (NULL: no linear part)
 *   table_stride / lin_stride: floats between consecutive rows of table / entries of lin_w;
 *     0 = dense (D and 1).  A caller may interleave a row's parameter, its linear weight and their
 *     optimizer slots in one record (the sparse-row update then touches two cache lines per row
 *     instead of six); every xdfm_embed_* call takes the same two strides, which also apply to the
 *     slot arrays.
 *   row_ptr[B*F+1] int32, ids[nnz] int64, weights[nnz] float32 : the dnn view
 *   emb_out[B,F,D]           out[r,:] = sum_j w_j*table[id_j,:] in column order; rows
 *                            without a token are zero (also trailing ones)
 *   lin_out[B]  (NULL ok)    sum_j w_j*lin_w[id_j]
 *   fm_out[B]   (NULL ok)    FM second-order term
 *   fm_sum[B,D] (NULL ok)    s[b,d] = sum_j w_j*table[id_j,d]   (kept for the backward)
 *   age (NULL ok)            rows are aged as they are read (xdfm_row_age above)
 * ---------------------------------------------------------------------- */
int xdfm_embed_fwd(const float* table, const float* lin_w, int64_t table_stride, int64_t lin_stride,
                   int64_t N, int32_t D, int32_t F, int64_t B, const int32_t* row_ptr, const int64_t* ids, const float* weights,
                   int64_t nnz, float* emb_out, float* lin_out, float* fm_out, float* fm_sum,
                   int32_t* err_flag, const xdfm_row_age* age, void* stream);

/* Plain row gather: out[i,:] = table[rows[i],:], lin_out[i] = lin_w[rows[i]] (NULL ok).
 * No counterpart in the reference (one device, one table): it is the owner-side lookup of the
 * row-sharded table ("owners run K1 locally"), the same copy xdfm_embed_fwd
 * does with one token of weight 1 per row, without spending a wave per row.
 * out_stride: floats between consecutive rows of `out` and between consecutive entries of
 * `lin_out` (0 = dense D / 1): with out and lin_out pointing into one buffer of
 * [row | linear weight | pad] records, rows and weights travel in one all-to-all.
 * age (NULL ok): rows are aged as they are read (xdfm_row_age above). */
int xdfm_embed_gather(const float* table, const float* lin_w, int64_t table_stride,
                      int64_t lin_stride, int64_t N, int32_t D, const int64_t* rows, int64_t n, float* out,
                      float* lin_out, int64_t out_stride, int32_t* err_flag, const xdfm_row_age* age,
                      void* stream);

/* ------------------------------------------------------------------------
 * Routing for the row-sharded table (owner = id mod G, local row = id div G).
 * No counterpart in the reference (one device, one table).  Index work, bit-exact.
 *   xdfm_route_keys: keys[i] = (id mod G) * rpr + id div G, rpr = ceil(N / G) (ids clamped into
 *     [0, N)).  A stable sort of these keys (xdfm_embed_plan with N = G * rpr) groups the tokens by
 *     owner, orders them by row inside an owner and puts duplicates side by side.
 *   xdfm_route_uniq: from that sort (sorted_keys, sorted_tok) the de-duplicated request:
 *     slot_sorted[n]  slot (0 .. n_uniq-1) of every sorted position        (int32)
 *     inverse[n]      slot of every token                                   (int64)
 *     uniq_rows[..]   local row of every slot, grouped by owner, ascending  (int64, n_uniq used)
 *     first_slot[G+1] first slot of every owner (-1: owner without a row), [G] = n_uniq
 *     block_scratch   xdfm_route_scratch_ints(n) int32 of scratch
 * ---------------------------------------------------------------------- */
int xdfm_route_keys(const int64_t* ids, int64_t n, int64_t N, int32_t G, int64_t* keys, void* stream);
/* The request in fixed-capacity form (what lets every exchange of a step move equally sized messages without
 * sizes from the host).  S = cap + 1 slots per owner: rows_padded[G*S] = owner g's distinct rows in
 * [g*S, g*S + count_g), pad_row (a scratch row the owner keeps past its real ones) in its unused slots, and
 * in the block's last slot a header: pad_row in the low 32 bits, this rank's largest per-owner count in the
 * high ones -- the message that carries the rows also tells every owner whether this source's batch fits.
 * slot_sorted_p / inverse_p = xdfm_route_uniq's slot numbers renumbered to owner*S + position; counts[G+1] =
 * distinct rows per owner and, in [G], their maximum (a batch whose maximum exceeds cap does not fit: its
 * padded arrays must not be used).  At most 64 owners.
 * xdfm_route_decode (owner side): the G blocks that arrived -> rows[G*S] plain row indices (headers -> the
 * scratch row) and src_max[G] every source's largest count. */
int xdfm_route_pad(const int32_t* first_slot, const int64_t* uniq_rows, const int32_t* slot_sorted,
                   const int64_t* inverse, int64_t n, int32_t G, int64_t cap, int64_t pad_row,
                   int64_t* rows_padded, int32_t* slot_sorted_p, int64_t* inverse_p, int32_t* counts,
                   void* stream);
int xdfm_route_decode(const int64_t* raw, int32_t G, int64_t cap, int64_t* rows, int32_t* src_max, void* stream);
int xdfm_route_scratch_ints(int64_t n, size_t* ints);
int xdfm_route_uniq(const int32_t* sorted_keys, const int32_t* sorted_tok, int64_t n, int64_t N, int32_t G,
                    int32_t* slot_sorted, int64_t* inverse, int64_t* uniq_rows, int32_t* first_slot,
                    int32_t* block_scratch, void* stream);

/* ------------------------------------------------------------------------
 * Embedding backward: deterministic sorted-key segment sum (+ row update).
 * Replaces TF's gradient of embedding_lookup_sparse / sparse_dense_matmul and
 * AdamOptimizer/AdagradOptimizer.apply_gradients on the table
 * (base_model.py:290, :286, :314-322).
 * ---------------------------------------------------------------------- */
/* bytes of `plan_ws` needed by xdfm_embed_plan for nnz tokens and ids < N */
int xdfm_embed_plan_workspace_bytes(int64_t nnz, int64_t N, size_t* bytes);
/* Sorts the tokens of one batch by feature id (stable: equal ids keep token order).
 *   sorted_ids[nnz] int32, sorted_tok[nnz] int32 (token index into ids/weights) */
int xdfm_embed_plan(const int64_t* ids, int64_t nnz, int64_t N, int32_t* sorted_ids,
                    int32_t* sorted_tok, void* plan_ws, size_t plan_ws_bytes, void* stream);
/* xdfm_embed_plan for callers that cannot allocate or size a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same kernels, same output bit for bit; the sort's scratch is a file-scope __device__ array of
 * 9 MiB that every GPU the library runs on holds until the process ends.  At most 524 288 tokens
 * (XDFM_E_WORKSPACE beyond).  The first call on a device resolves the array's device address (make
 * it outside a capture); calls that overlap on one GPU share the scratch and race. */
int xdfm_embed_plan_static(const int64_t* ids, int64_t nnz, int64_t N, int32_t* sorted_ids,
                           int32_t* sorted_tok, void* stream);

/* tok_row and weights in sorted order (row_sorted[p] = tok_row[sorted_tok[p]], w_sorted[p] =
 * weights[sorted_tok[p]]): xdfm_embed_bwd called with sorted_tok = NULL reads them in place of its
 * two dependent 4-byte gathers per token.  Depends on the batch only: a step runs it beside its
 * forward pass. */
int xdfm_embed_plan_gather(const int32_t* sorted_tok, const int32_t* tok_row, const float* weights,
                           int64_t nnz, int32_t* row_sorted, float* w_sorted, void* stream);

/* Segment sum of the per-token gradient contributions
 *     c_j = w_j*d_emb[row_j,:] + g[b_j]*w_j*(fm_sum[b_j,:] - w_j*table[id_j,:])   (FM term optional)
 *     l_j = g[b_j]*w_j                                                            (linear, optional)
 * over equal ids in sorted order (the FM term's common factor table[id,:] is taken out of the sum: the
 * row is read once per segment, where it is updated), then
 *   mode 0 (sparse_rows): apply `opt` to the touched rows of table (+slots) and lin_w in place;
 *   (table_stride / lin_stride as for xdfm_embed_fwd: they address table, slot_m, slot_v and
 *   lin_w, lin_m, lin_v; grad_table / grad_lin are always dense)
 *   mode 1 (scatter)    : store the sums into grad_table[N,D] / grad_lin[N] (rows of untouched
 *                         ids are left as they are: the caller keeps them zero).
 *   mode 2 (scatter from records): as mode 1 into DENSE grad_table[N,D] / grad_lin[N], but d_emb /
 *                         g_logit are read at grad_stride like mode 0 does (the owner side of the
 *                         row-sharded table under the reference's dense-exact update: the records that
 *                         arrived are summed per row into the dense gradient image xdfm_dense_update reads).
 *   pieces: float scratch of 2*(nnz/64 + 1)*(D + 2) elements (partial sums of the segments that
 *   cross a 64-token chunk boundary).
 *   sorted_tok = NULL: tok_row and weights are already in sorted order (xdfm_embed_plan_gather).
 *   d_emb[B,F,D]; d_emb2: NULL, or a second [B,F,D] array added to d_emb as it is read (two
 *   producers, e.g. the CIN and the DNN tower, then need no common accumulator);
 *   g_logit[B] and fm_sum[B,D] only when the FM / linear parts are on.
 *   grad_stride (0 = dense): floats between consecutive gradient records.  Mode 1 then writes
 *   row i at grad_table + i*grad_stride and its linear gradient at grad_lin + i*grad_stride;
 *   mode 0 reads d_emb / d_emb2 rows and g_logit entries at that stride (the owner side of the
 *   row-sharded table gets [gradient row | linear gradient | pad] records from one all-to-all).
 *   age (NULL ok, mode 0): the rows are aged in the registers of their update (xdfm_row_age above;
 *   step_offset = -1 when xdfm_opt_advance has already counted this step). */
int xdfm_embed_bwd(int32_t mode, float* table, float* slot_m, float* slot_v, float* lin_w,
                   float* lin_m, float* lin_v, int64_t table_stride, int64_t lin_stride,
                   float* grad_table, float* grad_lin, int64_t N,
                   int32_t D, int32_t F, int64_t B, const int32_t* sorted_ids,
                   const int32_t* sorted_tok, const int32_t* tok_row, const float* weights,
                   int64_t nnz, const float* d_emb, const float* g_logit, const float* fm_sum,
                   int32_t use_fm, const xdfm_opt_args* opt, float* pieces, const float* d_emb2,
                   int64_t grad_stride, const xdfm_row_age* age, void* stream);
/* Writes aged rows back: the rows named by sorted_ids (the first position of each run of equal ids
 * works; NULL: all N rows, n ignored) are brought up to opt_state[3] + age->step_offset steps and
 * row_step is set to that count (xdfm_row_age above; age->slot_* are ignored, the slots are arguments).
 * Needed only where the table is read by something that does not age rows itself: checkpoints, exports. */
int xdfm_embed_catchup(float* table, float* slot_m, float* slot_v, float* lin_w, float* lin_m,
                       float* lin_v, int64_t table_stride, int64_t lin_stride, int64_t N, int32_t D,
                       const int32_t* sorted_ids, int64_t n, const xdfm_row_age* age, void* stream);
/* zero the rows of grad_table / grad_lin named by sorted_ids (restores the all-zero invariant) */
int xdfm_embed_clear_rows(float* grad_table, float* grad_lin, int32_t D,
                          const int32_t* sorted_ids, int64_t nnz, void* stream);

/* ------------------------------------------------------------------------
 * Dense optimizer step over a flat fp32 buffer made of `n_seg` tensors
 * (apply_gradients, base_model.py:305-322; regulariser gradients of
 * base_model.py:121-167 folded in).
 *   seg_off[n_seg+1] int64 (host), per segment: l2, l1 coefficients, norm_l2 coefficient
 *   (gradient norm_l2 * p/||p||_F, the `tf.norm(ord=2)` term of cross_params) -- host arrays.
 *   grad may be NULL (regulariser-only step, used for untouched state in tests).
 *   stats: device float scratch of at least n_seg*260 floats (per-segment sums of p^2, |p|,
 *   g^2 and their per-block partials); n_seg <= 32 per call
 *   reg_loss (device float, may be NULL): receives the regularisation loss of these
 *   segments (value BEFORE the update), base_model.py:264-272.
 *   accumulate_reg_loss: 0 write, 1 add, XDFM_REG_PRECOMPUTED: `stats` already holds what a
 *   preceding xdfm_dense_reg_loss over the same, unchanged parameters left there and reg_loss is
 *   not touched: only the update runs, one launch (clip_norm must be 0).
 * ---------------------------------------------------------------------- */
#define XDFM_REG_PRECOMPUTED 2
int xdfm_dense_update(float* param, const float* grad, float* slot_m, float* slot_v,
                      int32_t n_seg, const int64_t* seg_off, const float* seg_l2,
                      const float* seg_l1, const float* seg_norm_l2, const xdfm_opt_args* opt,
                      float* stats, float* reg_loss, int32_t accumulate_reg_loss, void* stream);
/* The regularisation loss of xdfm_dense_update alone (same kernels, same value).  It depends on
 * the parameters only, so a step may compute it beside its forward pass and then call
 * xdfm_dense_update with reg_loss = NULL: without clipping and norm_l2 terms the update is then
 * one launch.  stats as above (a buffer of its own when the two calls can overlap). */
int xdfm_dense_reg_loss(const float* param, int32_t n_seg, const int64_t* seg_off, const float* seg_l2,
                        const float* seg_l1, const float* seg_norm_l2, float* stats, float* reg_loss,
                        int32_t accumulate_reg_loss, void* stream);

/* ------------------------------------------------------------------------
 * Compressed Interaction Network, one layer.   xDeepFM.py:191-267
 *   x0[B,F,D], xk[B,H,D] (layer 0: xk == x0), w[F*H, L] (filter f_k, index f*H+h)
 *   masked != 0: layer-0 strictly-upper mask times 2 (xDeepFM.py:208-217)
 *   curr[b,n,d] = sum_{f,h} s*M[f,h]*x0[b,f,d]*xk[b,h,d]*w[f*H+h,n]
 *   n <  n_hidden : xnext[b,n,d]            = curr   ([B,n_hidden,D], the next layer's input)
 *   n >= n_hidden : pooled[b,pool_off+n-n_hidden] = sum_d curr   (pooled is [B,pool_stride])
 *   workspace: device float scratch of xdfm_cin_fwd_workspace_floats(...) elements (the
 *   filter re-packed in matrix-core operand order; may be NULL when that size is 0)
 * ---------------------------------------------------------------------- */
int xdfm_cin_fwd_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                  size_t* floats);
int xdfm_cin_fwd(const float* x0, const float* xk, const float* w, int64_t B, int32_t F,
                 int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                 float* xnext, float* pooled, int32_t pool_stride, int32_t pool_off,
                 float* workspace, void* stream);
/* Backward of one layer.  Upstream gradient of curr[b,n,d]:
 *   n <  n_hidden : d_xnext[b,n,d]
 *   n >= n_hidden : d_pooled[b,pool_off+n-n_hidden]        (same for every d)
 * Outputs: d_xk[B,H,D] (written; NULL for layer 0 where both roles are x0),
 *          d_x0[B,F,D] (accumulated into: +=), d_w partial sums in dw_part, reduced
 *          into d_w[F*H,L] (written).  dw_part: float scratch of
 *          xdfm_cin_bwd_workspace_floats(...) elements. */
int xdfm_cin_bwd_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                  size_t* floats);
int xdfm_cin_bwd(const float* x0, const float* xk, const float* w, int64_t B, int32_t F,
                 int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                 const float* d_xnext, const float* d_pooled, int32_t pool_stride,
                 int32_t pool_off, float* d_xk, float* d_x0, float* d_w, float* dw_part,
                 void* stream);

/* Filters packed ahead of the layer calls.  xdfm_cin_fwd / xdfm_cin_bwd re-pack their filter into
 * matrix-core operand order on every call; a training step can instead pack the filters of all
 * its layers in ONE launch, before they are needed, and call the *_packed variants (same
 * results, bit for bit).  No counterpart in the reference (TF re-reads its variables).
 *   xdfm_cin_pack_floats: sizes of a layer's images; 0 = this shape has no packed path for that
 *     direction (use xdfm_cin_fwd / xdfm_cin_bwd).
 *   xdfm_cin_pack: up to 8 jobs per call; fwd_image / bwd_image may be NULL (skip).  Images are
 *     16-byte aligned device buffers, valid for batch size B and dim D until the filter changes.
 *   xdfm_cin_bwd_packed: dw_part sized by xdfm_cin_bwd_workspace_floats as for xdfm_cin_bwd.
 *     phases: XDFM_CIN_BWD_DX (d_xk, d_x0), XDFM_CIN_BWD_DW (d_w) or both.  The two read the same
 *     inputs and share no output, so a caller may run them on different streams (the filter
 *     gradient beside whatever consumes d_x0 next).
 *     XDFM_CIN_BWD_DW = XDFM_CIN_BWD_DW_PARTIAL (per-split partial sums into dw_part) followed by
 *     XDFM_CIN_BWD_DW_REDUCE (dw_part -> d_w); the reduction is a short kernel that nothing but
 *     the optimizer waits for, so a caller may give it to another stream as well (dw_part must
 *     stay untouched in between).
 *     d_x0 = this layer's contribution + d_x0_in + d_x0_in2 (each [B,F,D] or NULL; d_x0_in = d_x0 is
 *     xdfm_cin_bwd's "accumulated into").  The layer that runs first can write instead of adding to
 *     a cleared array, the one that runs last can fold a second producer's gradient (the DNN
 *     tower's) in: the table gradient then reads one array per token.
 *     ldw: floats between consecutive rows of d_w (0 = L), as in xdfm_cin_pack_job. */
#define XDFM_CIN_BWD_DX 1
#define XDFM_CIN_BWD_DW 2
#define XDFM_CIN_BWD_DW_PARTIAL 4
#define XDFM_CIN_BWD_DW_REDUCE 8
typedef struct xdfm_cin_pack_job {
  const float* w;    /* [F*H, L] */
  int32_t F, H, L, masked;
  float* fwd_image;  /* xdfm_cin_pack_floats fwd_floats elements, or NULL */
  float* bwd_image;  /* bwd_floats elements, or NULL */
  int32_t ldw;       /* floats between consecutive filter rows (0 = L): L may be a leading column
                      * range of a wider filter (a layer's hidden channels, see xdfm_cin_pair_sum) */
  int32_t reserved;  /* set to 0 */
} xdfm_cin_pack_job;
int xdfm_cin_pack_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, int32_t masked,
                         size_t* fwd_floats, size_t* bwd_floats);
int xdfm_cin_pack(const xdfm_cin_pack_job* jobs, int32_t n_jobs, int64_t B, int32_t D, void* stream);
int xdfm_cin_fwd_packed(const float* x0, const float* xk, const float* fwd_image, int64_t B, int32_t F,
                        int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked, float* xnext,
                        float* pooled, int32_t pool_stride, int32_t pool_off, void* stream);
int xdfm_cin_bwd_packed(const float* x0, const float* xk, const float* bwd_image, int64_t B, int32_t F,
                        int32_t H, int32_t D, int32_t L, int32_t n_hidden, int32_t masked,
                        const float* d_xnext, const float* d_pooled, int32_t pool_stride,
                        int32_t pool_off, float* d_xk, float* d_x0, float* d_w, float* dw_part,
                        int32_t phases, const float* d_x0_in, const float* d_x0_in2, int32_t ldw, void* stream);

/* The pooled ("direct") channels of a layer (xDeepFM.py:253-270: channels n >= n_hidden are summed
 * over the embedding dimension) without the factor D of work: the sum over d is taken first,
 *     S[b,p] = sum_d x0[b,f,d] xk[b,h,d]          p = f*H + h, or the index of (f,h) in the row-major
 *                                                  strict upper triangle on the masked layer
 *     pooled[b,j] = sum_p S[b,p] wp[p,j]           wp = the filter's columns n_hidden.. in pair order,
 *                                                  the masked layer's factor 2 folded in
 * -- the same sums in another association, 1/D of the multiply-adds.  A layer then runs its hidden
 * channels alone through xdfm_cin_fwd_packed / xdfm_cin_bwd_packed (L = n_hidden, ldw = the filter's
 * width) and its pooled channels as
 *   forward : xdfm_cin_pool_pack, xdfm_cin_pair_sum, xdfm_dense_fwd(S, wp) -> pooled columns
 *   backward: xdfm_dense_bwd(S, wp, d_pooled columns) -> dS, dwp; xdfm_cin_pool_unpack(dwp) -> d_w
 *             columns; xdfm_cin_pair_bwd(dS) -> d_x0 (+ d_xk), written or added to.
 * xdfm_cin_pair_count: number of pairs P. */
int xdfm_cin_pair_count(int32_t F, int32_t H, int32_t masked, int64_t* pairs);
int xdfm_cin_pair_sum(const float* x0, const float* xk, int64_t B, int32_t F, int32_t H, int32_t D,
                      int32_t masked, float* S, void* stream);
int xdfm_cin_pair_bwd(const float* x0, const float* xk, const float* dS, int64_t B, int32_t F, int32_t H,
                      int32_t D, int32_t masked, float* d_x0, int32_t accumulate_x0, float* d_xk,
                      int32_t accumulate_xk, void* stream);
/* The same two products with their operands rounded to bf16 on the bf16 matrix cores (fp32 sums), for the bf16
 * path's wide layers (north_star's bf16 bar, 2e-2): one wave per sample, no staging through LDS.  Served: F <= 64
 * and D a multiple of 16 (pair_sum) / 32 (pair_bwd) up to 64, the masked layer (xk == x0) and the unmasked ones
 * (xk != x0; pair_bwd then also wants H a multiple of 8); x0, xk, dS 16-byte aligned.  Every other shape is handed
 * to the exact fp32 forms above, so the calls can replace them anywhere.  (Same reference lines: xDeepFM.py:253-270.) */
int xdfm_cin_pair_sum_bf16(const float* x0, const float* xk, int64_t B, int32_t F, int32_t H, int32_t D,
                           int32_t masked, float* S, void* stream);
int xdfm_cin_pair_bwd_bf16(const float* x0, const float* xk, const float* dS, int64_t B, int32_t F, int32_t H,
                           int32_t D, int32_t masked, float* d_x0, int32_t accumulate_x0, float* d_xk,
                           int32_t accumulate_xk, void* stream);
int xdfm_cin_pool_pack(const float* w, int32_t F, int32_t H, int32_t L, int32_t n_hidden, int32_t masked,
                       float* wp, void* stream);
int xdfm_cin_pool_unpack(const float* dwp, int32_t F, int32_t H, int32_t L, int32_t n_hidden,
                         int32_t masked, float* d_w, void* stream);

/* A layer whose channels are ALL pooled (n_hidden = 0: the last layer of the network,
 * xDeepFM.py:253-270), unmasked, small shapes (F <= 64, H <= 16, D <= 16 and a multiple of 4, L <= 16),
 * as ONE forward and ONE backward kernel on the f32 matrix cores with the sum over the embedding
 * dimension taken first (the pair sums never leave the registers).  Same arguments as xdfm_cin_fwd /
 * xdfm_cin_bwd with n_hidden = 0 and masked = 0; the same sums in another association (1e-6).
 *   xdfm_cin_pooled_workspace_floats: 0 = this shape is not served (use xdfm_cin_fwd / xdfm_cin_bwd).
 *   xdfm_cin_pooled_bwd: d_xk[B,H,D] written, d_x0[B,F,D] written or added to (accumulate_x0),
 *     d_w[(f,h), 0..L) written with row stride ldw (0 = L).  The filter gradient is a short second
 *     kernel (a sum of per-block partial sums from the workspace) that nothing of the backward pass waits
 *     for: d_w == NULL leaves the partial sums in the workspace, and a later call with x0 == NULL (same B
 *     and shape, same workspace, any stream ordered after the first) writes d_w from them. */
int xdfm_cin_pooled_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, size_t* floats);
int xdfm_cin_pooled_fwd(const float* x0, const float* xk, const float* w, int64_t B, int32_t F, int32_t H,
                        int32_t D, int32_t L, int32_t ldw, float* pooled, int32_t pool_stride, int32_t pool_off,
                        void* stream);
/* xdfm_cin_pooled_fwd with the rest of a training step's forward pass in the same launch: the CIN head over
 * pooled[B, n] (n = pooled channels of ALL layers = the row length of pooled and d_pooled; the other layers'
 * channels must be there already), logit, pred, g_logit, cin_out, d_pooled as xdfm_head_loss computes them
 * (log_loss / cross_entropy_loss; the same expressions, the head's two sums as wave sums).  The mean of the
 * loss is left as *n_partial per-block sums in scratch[<= 1024]: xdfm_loss_finish_partials(scratch, n_partial,
 * mean_count or B, loss, stream) adds them in index order, on any stream ordered after this call.  At most
 * 16384 samples per call.  No counterpart in the reference (one sess.run there). */
int xdfm_cin_pooled_fwd_head(const float* x0, const float* xk, const float* w, int64_t B, int32_t F, int32_t H,
                             int32_t D, int32_t L, int32_t ldw, float* pooled, int32_t pool_off,
                             const float* part0, const float* part1, const float* part3, const float* bias,
                             const float* head_w, const float* head_b, int32_t n, const float* labels,
                             int64_t mean_count, int32_t classification, int32_t loss_kind, float* logit,
                             float* pred, float* cin_out, float* g_logit, float* d_pooled, float* scratch,
                             int32_t* n_partial, void* stream);
int xdfm_loss_finish_partials(const float* scratch, int32_t n_partial, int64_t count, float* loss, void* stream);
int xdfm_cin_pooled_bwd(const float* x0, const float* xk, const float* w, const float* d_pooled,
                        int32_t pool_stride, int32_t pool_off, int64_t B, int32_t F, int32_t H, int32_t D,
                        int32_t L, int32_t ldw, float* d_xk, float* d_x0, int32_t accumulate_x0, float* d_w,
                        float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * The same layer on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation), for the
 * wide shapes: embedding dim 64, up to 128 inputs per side and 128 channels per call
 * (xDeepFM.py:191-267; BASELINE configs[4]).  Operands (x, filter, upstream gradient) are rounded
 * to bf16, sums stay fp32: results within 2e-2 relative of the fp32 entry points.
 * These calls run a layer's HIDDEN channels (L = their number, ldw = the filter's full width);
 * the pooled channels go through xdfm_cin_pair_sum / xdfm_cin_pair_bwd and bf16 dense products.
 *   xdfm_cin_bf16_image_bytes / xdfm_cin_bf16_pack: the filter in matrix-core operand order, one
 *     image for the forward pass and one for the input gradient (256-byte aligned device buffers;
 *     either may be NULL), valid until the filter changes.  On the masked layer the images hold
 *     the symmetrised filter w[(min(f,h), max(f,h))] with a zero diagonal (the same sum as
 *     "pairs f < h, times 2").
 *   xdfm_cin_bf16_fwd:    xnext[B,L,D].
 *   xdfm_cin_bf16_bwd_dx: d_xk[B,H,D] written (NULL on the masked layer, where both roles are x0);
 *                         d_x0[B,F,D] written, or added to when accumulate_x0 != 0.
 *   xdfm_cin_bf16_bwd_dw: d_w[(f,h), 0..L) written with row stride ldw (masked layer: rows f < h,
 *                         times 2, zero elsewhere); workspace of
 *                         xdfm_cin_bf16_bwd_dw_workspace_floats elements.
 * ---------------------------------------------------------------------- */
int xdfm_cin_bf16_image_bytes(int32_t F, int32_t H, int32_t L, size_t* fwd_bytes, size_t* dx_bytes);
int xdfm_cin_bf16_pack(const float* w, int32_t F, int32_t H, int32_t L, int32_t ldw, int32_t masked,
                       void* fwd_image, void* dx_image, void* stream);
int xdfm_cin_bf16_fwd(const float* x0, const float* xk, const void* fwd_image, int64_t B, int32_t F,
                      int32_t H, int32_t D, int32_t L, int32_t masked, float* xnext, void* stream);
int xdfm_cin_bf16_bwd_dx(const float* x0, const float* xk, const void* dx_image, const float* d_xnext,
                         int64_t B, int32_t F, int32_t H, int32_t D, int32_t L, int32_t masked,
                         float* d_xk, float* d_x0, int32_t accumulate_x0, void* stream);
int xdfm_cin_bf16_bwd_dw_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                          size_t* floats);
int xdfm_cin_bf16_bwd_dw(const float* x0, const float* xk, const float* d_xnext, int64_t B, int32_t F,
                         int32_t H, int32_t D, int32_t L, int32_t masked, float* d_w, int32_t ldw,
                         float* workspace, void* stream);

/* cross_activation (xDeepFM.py:242, :402): the activation between a CIN layer's product and its
 * split.  With a non-identity activation run the layer with n_hidden = L (raw curr[B,L,D] into a
 * buffer of its own, no pooled output), then
 *   xdfm_cin_act_fwd: curr <- act(curr) in place (kept for the backward), xnext / pooled as in
 *                     xdfm_cin_fwd;
 *   xdfm_cin_act_bwd: d_curr[B,L,D] = upstream * act'(y), to be handed to the layer's backward as
 *                     its d_xnext with n_hidden = L.
 * act: XDFM_ACT_* (softmax is not supported). */
int xdfm_cin_act_fwd(float* curr, int64_t B, int32_t L, int32_t D, int32_t n_hidden, int32_t act,
                     float* xnext, float* pooled, int32_t pool_stride, int32_t pool_off, void* stream);
int xdfm_cin_act_bwd(const float* y, const float* d_xnext, const float* d_pooled, int32_t pool_stride,
                     int32_t pool_off, int64_t B, int32_t L, int32_t D, int32_t n_hidden, int32_t act,
                     float* d_curr, void* stream);

/* ------------------------------------------------------------------------
 * Fast (low-rank) CIN, one layer.   xDeepFM.py:295-451 (`_build_fast_CIN`, fast_CIN_d > 0)
 *   x0[B,F,D], xk[B,H,D]; w[F, d*L], v[H, d*L] (the [1,F,d*L] / [1,H,d*L] conv1d filters);
 *   c = l*d + dd.
 *   first != 0 (layer 0, xk and v unused): curr[b,l,dim] = sum_dd 0.5*(sum_f x0*w - sum_f x0^2*w^2)
 *   else:                                  curr[b,l,dim] = sum_dd (sum_f x0*w) * (sum_h xk*v)
 *   split of curr into xnext (l < n_hidden) and pooled (sum over dim) as in xdfm_cin_fwd.
 * Backward: d_x0 accumulated (+=), d_xk written (layer k), d_w[F,d*L], d_v[H,d*L] written.
 *   workspace: float scratch of xdfm_fast_cin_bwd_workspace_floats(...) elements.
 * ---------------------------------------------------------------------- */
int xdfm_fast_cin_fwd(const float* x0, const float* xk, const float* w, const float* v, int64_t B,
                      int32_t F, int32_t H, int32_t D, int32_t L, int32_t d, int32_t n_hidden,
                      int32_t first, float* xnext, float* pooled, int32_t pool_stride,
                      int32_t pool_off, void* stream);
int xdfm_fast_cin_bwd_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                       int32_t d, size_t* floats);
int xdfm_fast_cin_bwd(const float* x0, const float* xk, const float* w, const float* v, int64_t B,
                      int32_t F, int32_t H, int32_t D, int32_t L, int32_t d, int32_t n_hidden,
                      int32_t first, const float* d_xnext, const float* d_pooled,
                      int32_t pool_stride, int32_t pool_off, float* d_xk, float* d_x0, float* d_w,
                      float* d_v, float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * Dense layer  y = act(x*W + b)     xDeepFM.py:491-507 (xw_plus_b + activation),
 * also the two output heads (xDeepFM.py:290-292, :531-533) with act = identity.
 *   x[B,K], w[K,Nout], b[Nout], y[B,Nout]
 * Backward: dy is the gradient w.r.t. y (post-activation); y is the saved output.
 *   dx[B,K] written or accumulated (accumulate_dx != 0), dw[K,Nout], db[Nout] written.
 *   dx may be NULL (weight gradient only); dw, db and part may all be NULL (input gradient only).
 *   part: float scratch of xdfm_dense_bwd_workspace_floats(...) elements.
 * ---------------------------------------------------------------------- */
int xdfm_dense_fwd(const float* x, const float* w, const float* b, int64_t B, int32_t K,
                   int32_t Nout, int32_t act, float* y, void* stream);
int xdfm_dense_bwd_workspace_floats(int64_t B, int32_t K, int32_t Nout, size_t* floats);
int xdfm_dense_bwd(const float* x, const float* w, const float* y, const float* dy, int64_t B,
                   int32_t K, int32_t Nout, int32_t act, float* dx, int32_t accumulate_dx,
                   float* dw, float* db, float* part, void* stream);

/* Several weight gradients in one launch: the dw / db halves of xdfm_dense_bwd for up to 8 layers that share
 * the batch (the tower's layers, the linear bias as a [1] -> [1] layer, the CIN head as [n] -> [1]), over
 * common row splits, and one second launch that adds the partial sums (fixed order, as xdfm_dense_bwd).
 * Taken one by one these are a dozen ~5 us launches, each waiting for room beside the kernels that fill the
 * chip.  x[B,K], y[B,Nout] (the layer's saved output), dy[B,Nout], dw[K,Nout], db[Nout]; part: float scratch
 * of xdfm_dense_bwd_dw_multi_workspace_floats elements.  No counterpart in the reference (one sess.run). */
typedef struct xdfm_dw_job {
  const float* x;
  const float* y;
  const float* dy;
  float* dw;
  float* db;
  int32_t K;
  int32_t Nout;
  int32_t act;
  int32_t reserved;
} xdfm_dw_job;
int xdfm_dense_bwd_dw_multi_workspace_floats(const xdfm_dw_job* jobs, int32_t n_jobs, int64_t B, size_t* floats);
int xdfm_dense_bwd_dw_multi(const xdfm_dw_job* jobs, int32_t n_jobs, int64_t B, float* part, void* stream);

/* A plain DNN tower in_dim -> size[0] -> ... -> size[n_hidden-1] -> 1 (xDeepFM.py:473-533 without batch norm,
 * dropout or softmax; size[0] <= 32, the other widths <= 64, at most 4 hidden layers) as ONE launch per pass:
 * a row's activations depend on that row alone, so 16 rows go through all layers in one block.
 *   xdfm_tower_fwd:    h[l][B,size[l]] = act_l(...) for every hidden layer and out[B] (identity output).
 *   xdfm_tower_bwd_dx: from g[B] = d(loss)/d(out): d_h[l][B,size[l]] (the gradient w.r.t. each hidden output,
 *                      what the weight-gradient pass takes as dy) and dx[B,in_dim], written or added to.
 * The same sums as the per-layer calls in another association (fp32, 1e-6). */
typedef struct xdfm_tower_desc {
  int32_t n_hidden;
  int32_t in_dim;
  int32_t size[4];
  int32_t act[4];
  const float* w[4];   /* [in][out] per hidden layer */
  const float* b[4];
  const float* w_out;  /* [size[n_hidden-1]] */
  const float* b_out;  /* [1] */
  float* h[4];
  float* d_h[4];       /* xdfm_tower_bwd_dx only */
} xdfm_tower_desc;
int xdfm_tower_fwd(const xdfm_tower_desc* desc, const float* x, int64_t B, float* out, void* stream);
int xdfm_tower_bwd_dx(const xdfm_tower_desc* desc, const float* g, int64_t B, float* dx, int32_t accumulate_dx,
                      void* stream);

/* The same layer on the bf16 matrix cores (fp32 accumulation; x, w and the upstream gradient are
 * rounded to bf16 on their way into the product: 2e-2 relative bar), for wide layers
 * (BASELINE configs[4]: 2496 -> 1024 -> 512) and the pooled-channel products of the CIN.
 * Arguments as xdfm_dense_fwd / xdfm_dense_bwd; workspace: float scratch of
 * xdfm_dense_bf16_workspace_floats(...) elements (products with few output tiles are split along
 * their reduction and summed in split order), may be NULL when that size is 0. */
int xdfm_dense_bf16_workspace_floats(int64_t B, int32_t K, int32_t Nout, size_t* floats);
int xdfm_dense_bf16_fwd(const float* x, const float* w, const float* b, int64_t B, int32_t K,
                        int32_t Nout, int32_t act, float* y, float* workspace, void* stream);
int xdfm_dense_bf16_bwd(const float* x, const float* w, const float* y, const float* dy, int64_t B,
                        int32_t K, int32_t Nout, int32_t act, float* dx, int32_t accumulate_dx,
                        float* dw, float* db, float* workspace, void* stream);

/* Dropout of a DNN layer (user_dropout, base_model.py:324-337, :355-365): tf.nn.dropout on the
 * pre-activation, then the activation.  Run xdfm_dense_fwd with act = identity, then
 *   xdfm_dropout_act_fwd: z[B,n] <- act(z * m / keep) in place, mask_scale[B,n] = m / keep kept for
 *                         the backward; m is a counter-based hash of (seed, *step, element) -- step
 *                         is a device float (the optimizer's step count), may be NULL;
 *   xdfm_dropout_act_bwd: dz = dy * act'(y) * mask_scale, to be handed to xdfm_dense_bwd as dy with
 *                         act = identity.
 * TF's random stream is not reproducible: parity holds given the same masks. */
int xdfm_dropout_act_fwd(float* z, float* mask_scale, int64_t B, int32_t n, float keep, int32_t act,
                         uint64_t seed, const float* step, void* stream);
int xdfm_dropout_act_bwd(const float* y, const float* dy, const float* mask_scale, int64_t B, int32_t n,
                         int32_t act, float* dz, void* stream);

/* ------------------------------------------------------------------------
 * Batch normalisation (hparams.enable_BN): tf.compat.v1.layers.batch_normalization(momentum 0.95,
 * epsilon 1e-4, training = is_train_stage) after a CIN layer's product (xDeepFM.py:234-240,
 * :394-400) and after a DNN layer's xw_plus_b (xDeepFM.py:497-503).
 *   x, y: [B, C, I] with the channel in the middle: I = 1 for the DNN ([B, n]), I = dim for the CIN
 *   (curr lives as [B, L, dim] here).  gamma[C], beta[C]; moving_mean[C], moving_var[C].
 *   training != 0: batch moments (biased variance), the moving statistics are moved by
 *   v -= (v - batch) * (1 - momentum); else the moving statistics normalise.
 *   stats[2C] receives (mean, 1/sqrt(var + eps)) as used: the backward needs it.
 *   y = act((x - mean) * inv * gamma + beta), act = XDFM_ACT_* (identity when something else,
 *   e.g. dropout, sits between the normalisation and the activation).  y may alias x only when
 *   no backward pass follows (the backward needs x).
 * Backward (training statistics): g = dy * act'(y); dgamma[C] = sum g*xhat, dbeta[C] = sum g,
 *   dx = gamma*inv*(g - dbeta/n - xhat*dgamma/n), n = B*I.  dx may alias dy.
 * ---------------------------------------------------------------------- */
int xdfm_bn_fwd(const float* x, int64_t B, int32_t C, int32_t I, const float* gamma, const float* beta,
                float* moving_mean, float* moving_var, float momentum, float eps, int32_t training,
                int32_t act, float* stats, float* y, void* stream);
int xdfm_bn_bwd(const float* x, const float* y, const float* dy, int64_t B, int32_t C, int32_t I,
                const float* gamma, const float* stats, int32_t act, float* dx, float* dgamma,
                float* dbeta, void* stream);

/* The same normalisation when the batch is split between ranks (one process per GPU; the reference
 * is one process, its statistics are those of the whole batch: xDeepFM.py:234-240, :497-503).
 *   xdfm_bn_local_moments: moments[c] = mean of this rank's B*I values of channel c,
 *     moments[C + c] = their sum of squared deviations from that mean.  The ranks combine these
 *     (count-weighted mean; M2 = sum_r M2_r + n_r (mean_r - mean)^2) into
 *   mean_var[2C] = (mean, biased variance) of the whole batch, and
 *   xdfm_bn_fwd_moments normalises with it: stats, moving statistics and y as xdfm_bn_fwd(training).
 *   xdfm_bn_bwd_sums: dgamma[C], dbeta[C] over THIS rank's samples (its share of the parameter
 *     gradients); the ranks add them into sum_gamma / sum_beta, and
 *   xdfm_bn_bwd_apply: dx = gamma*inv*(g - sum_beta/n_total - xhat*sum_gamma/n_total), n_total =
 *     the whole batch's count per channel (sum of the ranks' B*I).  dx may alias dy. */
int xdfm_bn_local_moments(const float* x, int64_t B, int32_t C, int32_t I, float* moments, void* stream);
int xdfm_bn_fwd_moments(const float* x, int64_t B, int32_t C, int32_t I, const float* gamma, const float* beta,
                        float* moving_mean, float* moving_var, float momentum, float eps,
                        const float* mean_var, int32_t act, float* stats, float* y, void* stream);
int xdfm_bn_bwd_sums(const float* x, const float* y, const float* dy, int64_t B, int32_t C, int32_t I,
                     const float* stats, int32_t act, float* dgamma, float* dbeta, void* stream);
int xdfm_bn_bwd_apply(const float* x, const float* y, const float* dy, int64_t B, int32_t C, int32_t I,
                      int64_t n_total, const float* gamma, const float* stats, int32_t act,
                      const float* sum_gamma, const float* sum_beta, float* dx, void* stream);

/* softmax activation (`_activate`, base_model.py:343-344: tf.nn.softmax over the last axis), the
 * one activation the fused forms above do not carry: run the layer with the identity and then
 *   y[b,:,i] = softmax(x[b,:,i])            x, y: [B, C, I] as for xdfm_bn_fwd; y may alias x
 *   dx = y * (dy - sum_c y*dy)              dx may alias dy */
int xdfm_softmax_fwd(const float* x, int64_t B, int32_t C, int32_t I, float* y, void* stream);
int xdfm_softmax_bwd(const float* y, const float* dy, int64_t B, int32_t C, int32_t I, float* dx,
                     void* stream);

/* ------------------------------------------------------------------------
 * CIN output head with the residual connection (res=True):
 *   exFM_out[b] = sum_n r[b,n] + (sum_n r[b,n]*w[n] + bias)        xDeepFM.py:269-293
 *   pooled r[B,n], w[n], bias[1], out[B].
 * Backward (g = d loss / d exFM_out [B]): d_pooled[b,n] = g[b] + g[b]*w[n] (written),
 *   dw[n] = sum_b g[b]*r[b,n], db = sum_b g[b].  part: scratch of
 *   xdfm_dense_bwd_workspace_floats(B, n, 1) floats.  d_pooled may be NULL (weight gradients
 *   only), or dw, db and part together (d_pooled only).
 * ---------------------------------------------------------------------- */
int xdfm_cin_head_fwd(const float* pooled, const float* w, const float* bias, int64_t B,
                      int32_t n, float* out, void* stream);
int xdfm_cin_head_bwd(const float* pooled, const float* w, const float* g, int64_t B, int32_t n,
                      float* d_pooled, float* dw, float* db, float* part, void* stream);

/* ------------------------------------------------------------------------
 * logit sum + prediction + data loss + d(loss)/d(logit).
 *   _build_graph sum xDeepFM.py:46-71, _get_pred base_model.py:91-112,
 *   _compute_data_loss base_model.py:215-262.
 *   parts: up to 4 device arrays [B] (NULL entries skipped) summed in the given order,
 *   plus `bias` (device float scalar or NULL; the linear part's b).
 *   classification != 0: pred = sigmoid(logit) else pred = logit.
 *   labels[B]; outputs logit[B], pred[B], g_logit[B] (NULL: forward only),
 *   loss: device float (mean data loss; NULL: skip), scratch[1024] floats.
 *   mean_count: number of samples the mean runs over: 0 or B on one GPU; the global batch
 *   size when the batch is split over ranks (then `loss` is this rank's share of the mean
 *   and g_logit is scaled accordingly; build extra for the sharded path).
 * ---------------------------------------------------------------------- */
int xdfm_logit_loss(const float* part0, const float* part1, const float* part2,
                    const float* part3, const float* bias, const float* labels, int64_t B,
                    int64_t mean_count, int32_t classification, int32_t loss_kind, float* logit,
                    float* pred, float* g_logit, float* loss, float* scratch, void* stream);

/* The training step's chain xdfm_cin_head_fwd -> xdfm_logit_loss -> d_pooled of xdfm_cin_head_bwd
 * as ONE launch (same expressions, bit-identical results): part2 of the logit is the CIN head
 * computed here from pooled[B,n], head_w[n], head_b[1]; writes cin_out[B], logit, pred, g_logit,
 * d_pooled[B,n] and the mean loss.  part0 / part1 / part3 and bias as in xdfm_logit_loss.
 *   scratch: 1025 floats; element 1024 (a block counter) must be zero before the first call and
 *   is left zero by every call.  square_loss runs the separate kernels (it needs the mean first).
 *   loss = NULL defers the mean: the call leaves its per-block partial sums in scratch and
 *   xdfm_loss_finish(scratch, B, mean_count, loss_kind, loss, stream), with the same B, mean_count and
 *   loss_kind, adds them up in the same order -- on any stream ordered after this call and before
 *   the next one that uses scratch; nothing in a training step waits for the loss value. */
int xdfm_head_loss(const float* part0, const float* part1, const float* part3, const float* bias,
                   const float* pooled, const float* head_w, const float* head_b, int32_t n,
                   const float* labels, int64_t B, int64_t mean_count, int32_t classification,
                   int32_t loss_kind, float* logit, float* pred, float* cin_out, float* g_logit,
                   float* d_pooled, float* loss, float* scratch, void* stream);
int xdfm_loss_finish(const float* scratch, int64_t B, int64_t mean_count, int32_t loss_kind, float* loss,
                     void* stream);
/* out2 = {loss2[0], loss2[0] + loss2[1]}: data loss and total loss (data + regularisation) as a training
 * step reports them (base_model.py:80-89, :379-388), one tiny launch. */
int xdfm_loss_total(const float* loss2, float* out2, void* stream);

/* ------------------------------------------------------------------------
 * Fused evaluation: AUC (ties grouped, == sklearn.roc_auc_score) and logloss
 * (float64, preds clipped to [1e-11, 1-1e-11]) of a whole file's predictions.
 *   cal_metric deeprec_utils.py:535-545 as called from run_eval base_model.py:588.
 *   preds[n], labels[n] float32 (device; labels 0 / 1); out[4] double (device):
 *   out[0]=auc (NaN when only one class), out[1]=logloss, out[2]=#positives, out[3]=n
 *   Hand-written throughout: order-preserving 32-bit keys of the scores with the label as payload
 *   through the library's own stable radix sort, then ONE pass over the sorted pairs that forms the
 *   tie-aware pair count in integers (twice the average rank of every positive) and the logloss sum.
 * ---------------------------------------------------------------------- */
int xdfm_eval_workspace_bytes(int64_t n, size_t* bytes);
int xdfm_eval_auc_logloss(const float* preds, const float* labels, int64_t n, double* out,
                          void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XDFM_H_ */
