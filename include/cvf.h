/* cvf.h - C ABI of libcvf_hip.so, the MI355X (gfx950) hot path of colvarsfinder.
 *
 * The reference (zwpku/colvars-finder v0.1.14) exposes NO FFI for this path: it is
 * plain PyTorch behind Python objects.  This header is therefore
 * the boundary the reference's Python layer would bind with ctypes; every entry point
 * names the reference code it replaces (file:line under the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every function is stream-ordered on `stream` (a hipStream_t passed as void*),
 *     never synchronises the host, never allocates, never frees: the caller owns all
 *     buffers (sizes are returned by the cvf_*_size helpers);
 *   - return value: 0 on success, negative on error (cvf_last_error() has the text);
 *   - "tiled" tensors are laid out [tile][row][64]: frame b lives in tile b/64, lane
 *     b%64; rows are features / nets / auxiliary slots.  Tail tiles are padded with
 *     copies of the last frame whose weight is forced to 0.
 *   - theta / grad / adam moments are flat fp32 buffers in torch's parameters() order
 *     (per Linear: weight [out,in] row-major, then bias [out]); cvf_mlp_desc holds the
 *     offsets.
 */
#ifndef CVF_H
#define CVF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVF_TILE 64
#define CVF_MAX_NETS 8
#define CVF_MAX_LAYERS 12
#define CVF_AUX_ROWS 18 /* R (9, row-major), centroid (3), Kinv (6: 00 01 02 11 12 22) */

/* feature record types: molann's ids (examples/dipeptide/main.ipynb:306-307 prints
 * "position ... type_id 3") */
enum { CVF_FEAT_ANGLE = 0, CVF_FEAT_BOND = 1, CVF_FEAT_DIHEDRAL = 2, CVF_FEAT_POSITION = 3 };
enum { CVF_PP_IDENTITY = 0, CVF_PP_ALIGN = 1 };
/* cvf_mlp_desc.act[l]: what follows Linear layer l (nn.py:29-59 takes any torch activation module).  The chain kernels
 * (cvf_ae_*, cvf_regae_*, cvf_mlp_eval_rows) and the eigenfunction kernels on 64-frame tiles (cvf_ef_*: they use the
 * activation's first TWO derivatives, expressed through its output) take all of these, one code for every hidden layer of a
 * net; the 16-frames-per-wave kernels (cvf_ef16_*) take CVF_ACT_TANH only - cvf_ef16_supported() answers 0 otherwise. */
enum {
  CVF_ACT_NONE = 0,
  CVF_ACT_TANH = 1,
  CVF_ACT_SIGMOID = 2,
  CVF_ACT_RELU = 3,
  CVF_ACT_ELU = 4,        /* alpha = 1 */
  CVF_ACT_LEAKY_RELU = 5, /* negative_slope = 0.01 */
  CVF_ACT_SOFTPLUS = 6    /* beta = 1, threshold = 20 */
};
/* cvf_pp_desc.flags: structure of the tables, enabling kernels whose LDS addresses are affine in the atom index */
enum {
  CVF_PP_ALIGN_CONTIG = 1,   /* align_idx[b] == b for all b (the align atoms are the first n_align frame atoms) */
  CVF_PP_PURE_POSITION = 2,  /* record r is {POSITION, atom r, out 3r}: n_rec atoms emit their aligned positions in order */
  CVF_PP_SLOT_BATCHED = 4    /* rec_slot holds n_rec_slot entries in batches of 64 of ONE feature type (one entry per lane of a wave),
                              * entries of type -1 are padding */
};

/* The preprocessing layer r(x): torch.nn.Identity (examples/2d/2d.ipynb:485) or the
 * Kabsch alignment + feature map the reference gets from molann
 * (examples/dipeptide/main.ipynb:333-348; consumed at core.py:403,414,635). */
typedef struct cvf_pp_desc {
  int32_t mode;            /* CVF_PP_* */
  int32_t n_coord;         /* floats per frame: 3*N (align) or d (identity) */
  int32_t n_align;         /* number of align atoms */
  int32_t n_rec;           /* number of feature records */
  int32_t d_r;             /* output dimension */
  int32_t use_angle_value; /* 0: angle->cos, dihedral->(cos,sin); 1: radians */
  int32_t has_position;    /* any CVF_FEAT_POSITION record */
  int32_t flags;           /* CVF_PP_* hints the caller vouches for (0 is always valid) */
  const int32_t* align_idx; /* [n_align] atom indices into the frame */
  const float* ref_c;       /* [n_align*3] reference positions minus their centroid */
  const int32_t* rec;       /* [n_rec*6]: type, a0, a1, a2, a3, out_offset (one record per
                               position ATOM: a0 = atom, 3 outputs) */
  /* optional (large molecules): per-atom tables that let the streaming kernel copy the atoms the features
   * use ("slots") to LDS while the frame goes by, instead of gathering them from HBM afterwards */
  const int32_t* atom_align; /* [N]: index b of the atom in align_idx / ref_c, or -1 */
  const int32_t* atom_slot;  /* [N] (16-byte aligned): slot of the atom, or -1 when no feature uses it */
  const int32_t* rec_slot;   /* [n_rec_slot*6]: rec with the atom fields replaced by slots; any order (best grouped by type) */
  const int32_t* slot_atom;  /* [n_slot]: atom of each slot */
  int32_t n_slot;
  int32_t n_rec_slot;        /* entries of rec_slot (0: n_rec, unbatched) */
  /* optional per-atom alignment weights ("weighted Kabsch": the rotation and translation minimise
   * sum_b w_b |(x_b - c) R - ref_b|^2, c = sum_b w_b x_b / sum_b w_b).  NULL = uniform weights.  When set:
   *   align_w[b] = n_align * w_b / sum w   (mean 1), and ref_c[b] = align_w[b] * (ref_b - weighted centroid of ref),
   * flags must be 0 and 3 N must fit the lane-per-frame kernels (the layouts built for speed assume uniform weights). */
  const float* align_w;      /* [n_align] or NULL */
  /* large molecules, derivative kernel (cvf_metric_apply; csrc/metric_large.hip): the scatter J^T g -> feature atoms as a table
   * of rows.  Every (record, atom position) pair owns one row; the rows of one slot are contiguous:
   *   slot_row[t] .. slot_row[t+1]-1 (slot_row[n_slot] = n_ref = total number of pairs, < 65536).
   * mrec[r] = { (type + 1) | out_offset << 3, slot[0] | slot[1] << 16, slot[2] | slot[3] << 16, urow[0] | urow[1] << 16,
   *             urow[2] | urow[3] << 16, off[0] | off[1] << 8 | off[2] << 16 | off[3] << 24, 0, 0 }   (16-byte aligned)
   * with urow[p] = slot_row[slot[p]] and urow[p] + off[p] = the row of atom p of record r (off < 256; unused positions 0;
   * n_slot, n_ref < 65536); records of one type side by side keep a wave on one code path. */
  const int32_t* mrec;       /* [n_mrec*8] */
  const int32_t* slot_row;   /* [n_slot+1] */
  int32_t n_mrec;
  int32_t n_ref;
} cvf_pp_desc;

/* k identical feed-forward nets (colvarsfinder.nn.EigenFunctions, nn.py:242-293) or one
 * chain (AutoEncoder = encoder followed by decoder, nn.py:61-114). */
typedef struct cvf_mlp_desc {
  int32_t n_nets;
  int32_t n_layers;                      /* Linear layers per net */
  int32_t dims[CVF_MAX_LAYERS + 1];      /* widths d0..dL */
  int32_t act[CVF_MAX_LAYERS];           /* 1: tanh after this Linear (nn.py:55-57) */
  int32_t w_off[CVF_MAX_NETS][CVF_MAX_LAYERS];
  int32_t b_off[CVF_MAX_NETS][CVF_MAX_LAYERS];
  int32_t n_params;
} cvf_mlp_desc;

/* Scalars of EigenFunctionTask (core.py:293-354). */
typedef struct cvf_ef_cfg {
  int32_t k;
  int32_t lag_idx;       /* 0: generator (core.py:418-426,438); >0: transfer operator (428,440) */
  int32_t sort_eigvals;  /* core.py:430-434 */
  int32_t pad_;
  double alpha;          /* core.py:455 */
  double beta;           /* core.py:426,438 */
  double dt;             /* core.py:428,440 */
  double eig_w[CVF_MAX_NETS];
} cvf_ef_cfg;

/* layout of the fp64 statistics vector (collective #1) */
#define CVF_NPAIR(k) ((k) * ((k) + 1) / 2)
/* generator: [W, S1(k), S2(k(k+1)/2, i<=j row-major), E(k)]
 * transfer : [W, S1(k), S2(...), W', S1'(k), S2'_ii(k), T(k)] */
int cvf_ef_nstats(int k, int lag_idx);
/* loss vector written by cvf_ef_loss: [loss, npl, pen, eig_sorted(k), cvec(k) as doubles] */
#define CVF_LOSS_LEN(k) (3 + 2 * (k))
/* coefficient vector written by cvf_ef_loss and read by cvf_ef_backward:
 * [gS1(k), gS2(k*k symmetric full), gE_or_gT(k), gS1'(k), gS2'_ii(k)] */
#define CVF_COEF_LEN(k) (4 * (k) + (k) * (k))

/* Adam scalars + state for the fused "reduce the gradient and update" calls (single-process runs, where no
 * cross-rank all-reduce sits between the gradient and the update).  torch.optim.Adam as built at core.py:164. */
typedef struct cvf_adam_args {
  float* theta;
  float* m;
  float* v;
  double lr, beta1, beta2, eps;
  const int32_t* step_count;   /* device: number t >= 1 of the current step */
  const cvf_mlp_desc* mlp;     /* with `packed`: the nets whose fragment copy is refreshed, else NULL */
  float* packed;
  const float* lr_dev;         /* device scalar that overrides `lr` when non-NULL: a captured hipGraph of the step then follows
                                * a learning rate the host changes between replays (scheduler, manual decay) */
} cvf_adam_args;

int cvf_version(void);
const char* cvf_last_error(void);

/* --- K1: alignment + features, forward.  Replaces pp_layer(X) at core.py:403,414,635.
 * x [B, n_coord] row-major fp32.  feat_tiled [T][d_r][64] and/or feat_rows [B][d_r]
 * (either may be NULL); aux_tiled [T][18][64] (may be NULL; identity mode ignores it).
 * Frames of thousands of atoms take the streaming path (groups of 8 frames; large batches with a tiled output run the resident
 * role-split kernel of csrc/k1_large.hip - same numbers, bit for bit - which writes every tiled row of 32 consecutive frames as one
 * 128-byte line: feat_tiled / aux_tiled / scratch must cover the padded frame count, ceil(B / 64) * 64, as documented).
 * `scratch` (may be NULL; cvf_align_feature_scratch_bytes() bytes): on the streaming path it receives the compact
 * copy [padded frames / 8][n_slot*3][8] (coordinate rows of groups of eight frames) of the atoms the features use, which
 * cvf_metric_apply consumes as `slot_xyz`. */
int64_t cvf_align_feature_scratch_bytes(const cvf_pp_desc* pp, int64_t B); /* 0 for small molecules */
int cvf_align_feature_fwd(const cvf_pp_desc* pp, const float* x, int64_t B, float* feat_tiled, float* feat_rows,
                          float* aux_tiled, void* scratch, void* stream);

/* --- K2+K3: per frame and net, q = J A J^T g and E = g^T J A J^T g with J the Jacobian
 * of r at the frame and A = diag(a).  Replaces the k autograd.grad calls through
 * pp_layer at core.py:424 and the a-weighted square sums at core.py:426,438; the
 * [B,3N,k] gradient tensor is never materialised.
 * g_tiled, q_tiled [T][k][d_r][64]; e_tiled [T][k][64]; a [n_coord]. */
int cvf_metric_apply(const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled, const float* a,
                     int k, const float* g_tiled, float* q_tiled, float* e_tiled, const float* slot_xyz,
                     const double* dense, void* stream);
/* The same launch with the first stage of K5 (cvf_ef_stats in generator mode, below) folded in: every block reduces its
 * tile's share of the batch sums, one short launch adds the tiles in a fixed order and, when loss_vec != NULL,
 * evaluates cvf_ef_loss.  Shapes the fused stage does not cover run cvf_metric_apply and cvf_ef_stats back to back -
 * the results agree to fp64 summation order.  scratch: cvf_metric_stats_scratch_doubles() doubles. */
int64_t cvf_metric_stats_scratch_doubles(int64_t B, int k);
int cvf_metric_apply_stats(const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled, const float* a,
                           int k, const float* g_tiled, float* q_tiled, float* e_tiled, const float* slot_xyz,
                           const double* dense, const cvf_ef_cfg* cfg, const float* w, const float* y_tiled,
                           double* scratch, double* stats, double* loss_vec, double* coef, void* stream);
/* large molecules only (slot_xyz / dense may be NULL otherwise): dense[cvf_metric_dense_doubles(pp)], prepared once per (pp, a):
 * dense[0..41] = moments of (a, ref) over the align atoms, T0[c] = sum a_bc, T1[c][j] = sum a_bc ref_bj,
 * T2[c][j][k] = sum a_bc ref_bj ref_bk, R1[j] = sum ref_bj (fp64); behind them n_slot x 8 floats of per-slot constants
 * (a, ref, align flag, rows of the slot) for the derivative kernel. */
int64_t cvf_metric_dense_doubles(const cvf_pp_desc* pp);
int cvf_metric_dense_tensors(const cvf_pp_desc* pp, const float* a, double* dense, void* stream);

/* --- packed MFMA weight fragments of the nets (a second copy of the weights in the order the
 * matrix-core kernels consume them; see csrc/cvf_pack.hpp).  cvf_ef_pack rebuilds it from theta;
 * cvf_adam_step / cvf_sgd_step keep it in step when given the buffer. */
int64_t cvf_ef_pack_floats(const cvf_mlp_desc* mlp);
int cvf_ef_pack(const cvf_mlp_desc* mlp, const float* theta, float* packed, void* stream);

/* --- K4a: k nets forward (+ gradient of each output w.r.t. the features).
 * Replaces self.model(...) at core.py:403,414 (nn.py:293).
 * y_tiled [T][k][64]; g_tiled [T][k][d0][64] or NULL. */
int cvf_ef_mlp_fwd(const cvf_mlp_desc* mlp, const float* theta, const float* packed, const float* feat_tiled,
                   int64_t n_tiles, float* y_tiled, float* g_tiled, float* saved, void* stream);
/* `saved` (may be NULL): cvf_ef_saved_floats() floats in which the forward kernel leaves the hidden activations (and
 * the back-propagated output sensitivities) of every (tile, net) for cvf_ef_backward, which then skips recomputing
 * them; for first layers wider than 128 inputs the buffer also has room for the tangent chain's first product, which
 * cvf_ef_backward computes with a launch of its own ahead of its main kernel).  Opaque layout; 0 floats = this shape has no
 * hand-off, pass NULL to both calls. */
int64_t cvf_ef_saved_floats(const cvf_mlp_desc* mlp, int64_t n_tiles);

/* --- K4a + K2/K3 + K5 in one go for the fast layout (pure position features on a contiguous align set, d_r <= 72):
 * cvf_ef_mlp_fwd, cvf_metric_apply_stats fused per tile - g = dy/dfeat never leaves the chip.  Same outputs (y, saved,
 * q, e, stats[, loss_vec, coef]) up to summation order; g_tiled is not produced.  Check cvf_ef_fwd_metric_supported()
 * first; otherwise make the two calls.  scratch as for cvf_metric_apply_stats. */
int cvf_ef_fwd_metric_supported(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp);
int cvf_ef_fwd_metric_stats(const cvf_mlp_desc* mlp, const float* theta, const float* packed, const float* feat_tiled,
                            const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled, const float* a,
                            float* y_tiled, float* saved, float* q_tiled, float* e_tiled, const cvf_ef_cfg* cfg,
                            const float* w, double* scratch, double* stats, double* loss_vec, double* coef, void* stream);

/* With stats == NULL the fused launches stop after leaving cvf_ef_fused_stats_rows() (> 0 required) rows of per-tile
 * sums in `scratch`; cvf_ef_stats_finish_rows adds them in a fixed order (and evaluates cvf_ef_loss when loss_vec != NULL). */
/* Transfer-operator mode (lag_tau > 0; core.py:403,414): alignment + features + nets forward in one launch for the frames x
 * and their lagged partners x_lag (may be NULL: T tiles only) -> feat_tiled [2T][d_r][64] (tiles T.. = lagged), y_tiled
 * [2T][k][64], saved (cvf_ef_saved_floats(mlp, 2T); may be NULL).  Same shapes as cvf_ef_align_fwd_metric_supported().
 * Follow with cvf_ef_stats (lag_idx > 0) and cvf_ef_backward as after cvf_align_feature_fwd x 2 + cvf_ef_mlp_fwd. */
int cvf_ef_align_fwd(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                     const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled, float* saved,
                     void* stream);
int64_t cvf_ef_fused_stats_rows(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp, int64_t B, int with_align);
int cvf_ef_stats_finish_rows(const cvf_ef_cfg* cfg, int64_t n_rows, const double* partial, double* stats, double* loss_vec,
                             double* coef, void* stream);
/* The same with K1 folded in: from the coordinates to q, E and the batch sums in one launch (feat_tiled and aux_tiled
 * become outputs; aux_tiled may be NULL).  Replaces cvf_align_feature_fwd + cvf_ef_fwd_metric_stats. */
int cvf_ef_align_fwd_metric_supported(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp);
int cvf_ef_align_fwd_metric_stats(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                  const cvf_pp_desc* pp, const float* x, int64_t B, float* aux_tiled, const float* a,
                                  float* y_tiled, float* saved, float* q_tiled, float* e_tiled, const cvf_ef_cfg* cfg,
                                  const float* w, double* scratch, double* stats, double* loss_vec, double* coef,
                                  void* stream);

/* --- The generator-mode step with SIXTEEN frames per wave (csrc/ef16_front.hip, ef16_back.hip), for the shapes of cvf_ef16_supported(): pure position
 * features on a contiguous align set, d_r <= 72 (the dipeptide-sized configurations 3 and 4).  Same mathematics and the same
 * outputs as cvf_ef_align_fwd_metric_stats + cvf_ef_backward; what differs is the decomposition: a wave owns 16 frames (the
 * matrix instruction's N), so the launch has four times the waves of a quarter of the dependent chain each, 3-5 waves per SIMD.
 *  cvf_ef16_front   : x [B][n_coord] -> feat_tiled [T][d_r][64], y_tiled [T][k][64], saved (cvf_ef16_saved_floats(); opaque
 *                     hand-off of the hidden activations), q_tiled [T][k][d_r][64], e_tiled [T][k][64], stats (+ loss_vec, coef
 *                     when non-NULL: cvf_ef_loss in the same call).  Replaces pp_layer(X), model(...), the k autograd.grad calls
 *                     and the batch sums of core.py:403-452.  scratch: cvf_ef16_scratch_doubles(B, k) doubles.
 *  cvf_ef16_backward: flat parameter gradient as slab rows, as cvf_ef_backward (core.py:517); follow with cvf_slab_reduce. */
int cvf_ef16_supported(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp);
int64_t cvf_ef16_scratch_doubles(int64_t B, int k);
int64_t cvf_ef16_saved_floats(const cvf_mlp_desc* mlp, int64_t n_tiles);
/* With stats == NULL cvf_ef16_front stops after leaving cvf_ef16_rows(B) (> 0 required) rows of per-unit batch sums in `scratch`;
 * cvf_ef16_finish adds them in a fixed order (and evaluates cvf_ef_loss when loss_vec != NULL).  Two calls = two launches that
 * can be timed apart; one call with stats != NULL does both. */
int64_t cvf_ef16_rows(int64_t B);
int cvf_ef16_finish(const cvf_ef_cfg* cfg, int64_t B, const double* scratch, double* stats, double* loss_vec, double* coef,
                    void* stream);
int cvf_ef16_front(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled, const cvf_pp_desc* pp,
                   const float* x, int64_t B, const float* a, float* y_tiled, float* saved, float* q_tiled, float* e_tiled,
                   const cvf_ef_cfg* cfg, const float* w, double* scratch, double* stats, double* loss_vec, double* coef,
                   void* stream);
int cvf_ef16_backward(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed, int64_t B,
                      const float* w, const float* feat_tiled, const float* y_tiled, const float* q_tiled, const double* coef,
                      float* slab, int32_t* step_count, const float* saved, void* stream);
/* Transfer-operator mode (lag_tau > 0; core.py:403,414: y = model(pp_layer(X)) on the frames and on their lagged partners, then
 * core.py:420-431,440 and loss.backward()) on the same kernels:
 *  cvf_ef16_front_transfer   : x, x_lag [B][n_coord] -> feat_tiled [2T][d_r][64], y_tiled [2T][k][64] (the partners' tiles follow
 *                              the frames'), saved (cvf_ef16_saved_floats(mlp, 2T)).  Follow with cvf_ef_stats (transfer mode).
 *  cvf_ef16_backward_transfer: slab rows of the parameter gradient from both passes; follow with cvf_slab_reduce. */
int cvf_ef16_front_transfer(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                            const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled, float* saved,
                            void* stream);
/* ... and, for k <= 4 (cvf_ef16_transfer_rows(B, k) > 0), the units' rows of the TIME-LAGGED batch sums in the same launch: a block
 * takes a unit of the frames AND the same unit of their lagged partners, so that sum w (y' - y)^2 (core.py:428) is formed where both
 * values are; w, w_lag [B]; scratch as cvf_ef16_scratch_doubles.  Follow with cvf_ef16_finish / cvf_ef16_finish_dp (cfg.lag_idx > 0)
 * instead of cvf_ef_stats (two launches). */
int64_t cvf_ef16_transfer_rows(int64_t B, int k);
int cvf_ef16_front_transfer_rows(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                 const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled, float* saved,
                                 const float* w, const float* w_lag, double* scratch, void* stream);
int cvf_ef16_backward_transfer(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed, int64_t B,
                               const float* w, const float* w_lag, const float* feat_tiled, const float* y_tiled,
                               const double* coef, float* slab, int32_t* step_count, const float* saved, void* stream);

/* --- K5: batch statistics (core.py:406-416,426,428,446-452), fp64, fixed-order two
 * stage reduction.  w [B]; y_tiled [T][k][64]; generator: e_tiled [T][k][64];
 * transfer: y_lag_tiled, w_lag.  scratch: cvf_ef_stats_scratch_doubles() doubles. */
int64_t cvf_ef_stats_scratch_doubles(int k, int lag_idx);
int cvf_ef_stats(const cvf_ef_cfg* cfg, int64_t B, const float* w, const float* y_tiled, const float* e_tiled,
                 const float* w_lag, const float* y_lag_tiled, double* scratch, double* stats, double* loss_vec,
                 double* coef, void* stream); /* loss_vec/coef non-NULL: also run cvf_ef_loss in the same launch */

/* --- loss, eigenvalues, ordering and the partial derivatives d loss / d stat
 * (core.py:426-457 after the sums).  One wave; runs after the cross-rank all-reduce of
 * `stats` when there is one. */
int cvf_ef_loss(const cvf_ef_cfg* cfg, const double* stats, double* loss_vec, double* coef, void* stream);

/* --- K4b: parameter gradient of the loss given the coefficients (what loss.backward()
 * does at core.py:517): reverse mode over the nets and, in generator mode, over their
 * directional derivative along q.  Each block writes its partial sum to one row of
 * `slab` [cvf_ef_backward_slab_rows(n_tiles)][n_params]; cvf_slab_reduce then sums the rows
 * in fixed order into grad [n_params] (bitwise reproducible, no atomics).
 * transfer mode: feat/y hold 2T tiles (frames then their lagged partners); n_tiles = 2T. */
int64_t cvf_ef_backward_slab_rows(int64_t n_tiles);
int cvf_ef_backward(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed, int64_t B,
                    const float* w, const float* w_lag, const float* feat_tiled, const float* y_tiled,
                    const float* q_tiled, const double* coef, float* slab, int32_t* step_count, float* saved,
                    void* stream);
                    /* step_count (may be NULL): the optimiser's device step counter, advanced by one.  saved: the hand-off buffer of
                     * cvf_ef_mlp_fwd, cvf_ef_saved_floats(mlp, n_tiles) floats - NOT const: for first layers wider than 128 inputs
                     * the call writes t0 = W0 q behind the activations (the extra vector per (tile, net) that size includes). */
int cvf_slab_reduce(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const cvf_adam_args* adam,
                    void* stream); /* adam (may be NULL): apply the update in the same launch */

/* --- AutoEncoder: weighted reconstruction loss and its parameter gradient in one pass
 * (core.py:664-666,708).  feat_rows [n][d0] row-major (the precomputed feature
 * trajectory of core.py:635); idx NULL or [B] frame indices into it; w [B]; inv_wsum =
 * 1/sum(w) (host-known: batches are static).  out2 [3] doubles: {sum w*err, sum w, their ratio = the loss}.
 * grad may be NULL (test pass, core.py:725-735). */
int64_t cvf_ae_scratch_floats(const cvf_mlp_desc* mlp, int64_t B);
int cvf_ae_step(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx, int64_t B,
                const float* w, double inv_wsum, float* scratch, double* out2, float* grad, int32_t* step_count,
                const cvf_adam_args* adam, void* stream);
                /* step_count (may be NULL) is advanced by one when grad != NULL; adam (may be NULL): update in the same call */

/* --- RegAutoEncoderTask (core.py:746-1217): time-lagged reconstruction loss
 * (weighted_MSE_loss, core.py:883-885) + transfer-operator eigenfunction regulariser (reg_eigen_loss with
 * lag_tau_reg > 0, core.py:973-1036) + the variance / covariance penalties on the latent vector (reg_enc_norm_loss,
 * reg_enc_orthognal_loss, core.py:912-971) + backward + optimizer step (core.py:1117,1128).  `mlp` is ONE chain: the
 * encoder's n_enc_layers layers, then the decoder and the K regulariser nets side by side as block-structured layers (built
 * by the host, colvarsfinder/core.py:_RegFlatParams), so that its last layer is [reconstruction (d_0 rows) | y_1..y_K].
 * feat_rows [n][d_0]: the feature trajectory; idx [B] (or NULL: 0..B-1): the batch's rows; rows idx + lag_target are
 * the reconstruction targets, rows idx + lag_input the lagged arguments of the regularisers (the generator-mode
 * regulariser and the gradient-norm penalty eta_0 of the reference are not built).
 *  cvf_regae_forward : y_tiled [2 T][K][64] (tiles 0..T-1: y on the rows idx, T..2T-1: on the lagged rows), enc_tiled
 *                      [T][k][64] (latent vector on the rows idx; NULL: not wanted) and out2 [3] = {sum w |dec(enc(f)) -
 *                      f_target|^2, sum w, their ratio}.  Then cvf_ef_stats (lag_idx > 0, k = K) on y_tiled gives the eigenfunction
 *                      terms and `coef`; cvf_ef_stats (lag_idx = 0, zero e_tiled) on enc_tiled + cvf_regae_enc_loss the
 *                      latent penalties and `enc_coef`.
 *  cvf_regae_backward: flat gradient of  mse_scale * sum w |..|^2 + head_scale * (npl + cfg.alpha * pen) + eta_1 norm +
 *                      eta_2 orth  in the chain's own parameter order (mse_scale = alpha / sum w, head_scale = gamma_0,
 *                      cfg.alpha = gamma_1 / gamma_0; coef / enc_coef NULL: that part is off), times `mask` (may be NULL;
 *                      zeros at the structural zeros of the block layers and at frozen parameters), + the Adam update
 *                      when adam != NULL; step_count as cvf_ae_step. */
int64_t cvf_regae_scratch_floats(const cvf_mlp_desc* mlp, int64_t B);
int cvf_regae_forward(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx, int64_t B,
                      int64_t lag_target, int64_t lag_input, int K, const float* w, float* scratch, float* y_tiled,
                      int n_enc_layers, float* enc_tiled, double* out2, void* stream);
int cvf_regae_backward(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx, int64_t B,
                       int64_t lag_target, int64_t lag_input, int K, const float* w, const float* w_lag, double mse_scale,
                       double head_scale, const float* y_tiled, const double* coef, int n_enc_layers, const double* enc_coef,
                       float* scratch, float* grad, const float* mask, int32_t* step_count, const cvf_adam_args* adam,
                       void* stream);
/* The two passes of ONE step on the same (theta, rows, lags): _keep leaves every tile's activations in `scratch`, _reuse reads
 * them instead of running the chain forward again (same arguments as the plain calls). */
int cvf_regae_forward_keep(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx, int64_t B,
                      int64_t lag_target, int64_t lag_input, int K, const float* w, float* scratch, float* y_tiled,
                      int n_enc_layers, float* enc_tiled, double* out2, void* stream);
int cvf_regae_backward_reuse(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx, int64_t B,
                       int64_t lag_target, int64_t lag_input, int K, const float* w, const float* w_lag, double mse_scale,
                       double head_scale, const float* y_tiled, const double* coef, int n_enc_layers, const double* enc_coef,
                       float* scratch, float* grad, const float* mask, int32_t* step_count, const cvf_adam_args* adam,
                       void* stream);
/* latent penalties from the latent vector's batch sums (cvf_ef_stats layout [W, S1(k), S2(i<=j), ..]):
 * terms = {sum_j (var_j - 1)^2, sum_{i<j} cov_ij^2} (core.py:934, 966); enc_coef [k + k*k] for cvf_regae_backward */
int cvf_regae_enc_loss(const double* stats, int k, double eta1, double eta2, double* terms, double* enc_coef, void* stream);
/* row [7 + K] of the task's loss list (core.py:1112-1124): [loss, ae, npl, pen, eig_1..K, 0 (gradient-norm term), norm, orth]
 * with loss = alpha ae + gamma_0 npl + gamma_1 pen + eta_1 norm + eta_2 orth; from out2 (cvf_regae_forward), loss_vec
 * (cvf_ef_stats; NULL: regulariser off), enc_terms (cvf_regae_enc_loss; NULL: off); alpha = 0 leaves ae = 0 (core.py:1090). */
int cvf_regae_loss_row(const double* out2, const double* loss_vec, double alpha, double gamma0, double gamma1, int K,
                       const double* enc_terms, double eta1, double eta2, double* row, void* stream);

/* --- nets forward on row-major features (inference: colvar_model(), core.py:372-382,
 * 640-647).  out [B][n_out] where n_out = n_nets * d_L; upto_layer < n_layers stops a
 * single chain early (AutoEncoder encoder). */
int cvf_mlp_eval_rows(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, int64_t B, int upto_layer,
                      float* out, void* stream);

/* --- K6: Adam (torch.optim.Adam defaults as constructed at core.py:164: betas
 * (0.9,0.999), eps 1e-8, no weight decay, no amsgrad), one launch over the flat buffer.
 * step_count is a device int32 holding the number t of the CURRENT step (>= 1): it is advanced by the
 * gradient-producing call of the step (cvf_slab_reduce / cvf_ae_step), so a captured step replays
 * correctly without host involvement. */
int cvf_adam_step(float* theta, const float* grad, float* m, float* v, int64_t n, double lr, const float* lr_dev,
                  double beta1, double beta2, double eps, int32_t* step_count, const cvf_mlp_desc* mlp, float* packed,
                  void* stream);   /* lr_dev (may be NULL): device scalar overriding lr, as in cvf_adam_args */
int cvf_sgd_step(float* theta, const float* grad, int64_t n, double lr, const float* lr_dev, const cvf_mlp_desc* mlp,
                 float* packed, void* stream); /* mlp/packed: NULL, or the nets' desc + fragment buffer to refresh */

/* --- cross-rank sums of the data-parallel step (the reference has no multi-GPU path): RCCL all-reduces
 * (in place, SUM) issued on `stream`, for hosts that bind this library without PyTorch - one process per GPU, the caller's
 * current HIP device at cvf_comm_init is the rank's GPU.  Collective #1 = the fp64 batch sums between cvf_ef16_front /
 * cvf_ef_stats (loss_vec == NULL) and cvf_ef_loss; collective #2 = the fp32 flat gradient between cvf_slab_reduce
 * (adam == NULL) and cvf_adam_step.  Rank 0 calls cvf_comm_unique_id and hands the cvf_comm_unique_id_bytes() bytes to the
 * other ranks by any means (the shipped Python host: torch.distributed broadcast); RCCL is located at run time. */
int cvf_comm_unique_id_bytes(void);
int cvf_comm_unique_id(void* id_host);
int cvf_comm_init(void** comm, int rank, int world, const void* id_host);
int cvf_comm_allreduce_f64(void* comm, double* buf, int64_t n, void* stream);
int cvf_comm_allreduce_f32(void* comm, float* buf, int64_t n, void* stream);
int cvf_comm_destroy(void* comm);

/* --- the same two sums as a ONE-SHOT peer-to-peer reduce (csrc/p2p.hip): every rank writes its vector
 * into a slot of every peer's window (fine-grained device memory shared through HIP IPC handles: one xGMI hop, all links at once),
 * raises a flag there, waits for the `world` flags in its own window and adds the slots in RANK ORDER - every rank forms the same
 * sum bit for bit, whatever the arrival order.  One process per GPU.  Unlike the rest of this header cvf_p2p_create allocates (the
 * window, as a communicator does).  Sequence: every rank cvf_p2p_create -> exchange the cvf_p2p_handle_bytes()-byte handles by any
 * means, rank-major -> cvf_p2p_connect -> cvf_p2p_allreduce_* on a stream (in place, SUM; n * sizeof <= max_bytes; capturable:
 * the epoch lives on the device; all launches on one communicator must be ordered on ONE stream).  A peer whose data does not
 * arrive within the time-out (20 s; environment CVF_P2P_TIMEOUT_MS) does not hang the GPU: the kernel fills the result with NaN and
 * sets the communicator's error word, which cvf_p2p_error() returns (0 = none; a host-visible word - no device call, no
 * synchronisation: it reports what the kernels finished so far have found).  A host must read it wherever it reads results back
 * and treat a non-zero value as fatal for the communicator (the shipped host raises). */
int cvf_p2p_handle_bytes(void);
int cvf_p2p_create(void** comm, int rank, int world, int64_t max_bytes, void* handle_out_host);
int cvf_p2p_connect(void* comm, const void* all_handles_host);
int cvf_p2p_allreduce_f64(void* comm, double* buf, int64_t n, void* stream);
int cvf_p2p_allreduce_f32(void* comm, float* buf, int64_t n, void* stream);
int cvf_p2p_error(void* comm);
int cvf_p2p_destroy(void* comm);

/* --- the data-parallel step in FOUR launches (front, finish, backward, slab reduction - as the single-process step): the two
 * cross-rank sums folded into the launches on either side of them, over the same windows in their low-latency form (every 4-byte
 * payload travels as one 8-byte {payload, exchange number} word that the receiver polls: no flag, no fence, one xGMI hop;
 * csrc/cvf_p2p.hpp).  `p2p_comm`: a connected cvf_p2p communicator (max_bytes >= 4 * n_params).  Parallelises core.py:498-522.
 *  cvf_ef16_finish_dp : cvf_ef16_finish + collective #1 + cvf_ef_loss: the units' rows -> this rank's sums -> sum over ranks in
 *                       rank order (`stats`) -> loss_vec, coef; identical on every rank bit for bit.
 *  cvf_ef_stats_dp    : cvf_ef_stats + collective #1 + cvf_ef_loss (transfer-operator mode / shapes outside the fast layout).
 *  cvf_ef_loss_dp     : collective #1 + cvf_ef_loss on sums another launch left in `stats` (in place).
 *  cvf_slab_reduce_dp : cvf_slab_reduce + collective #2 + the optimiser step (adam != NULL) - every workgroup exchanges the
 *                       entries it has just summed and applies the identical Adam update; `grad` receives the global gradient.
 *                       Must follow one of the three calls above (or cvf_p2p_exchange_f64) of the same step: it carries that
 *                       exchange's number; a repeat without one in between sets the error word (bit 30).
 *  cvf_p2p_exchange_f64: any fp64 vector of at most 80 entries, in place, by the same low-latency exchange. */
int cvf_ef16_finish_dp(const cvf_ef_cfg* cfg, int64_t B, const double* scratch, double* stats, double* loss_vec, double* coef,
                       void* p2p_comm, void* stream);
int cvf_ef_stats_dp(const cvf_ef_cfg* cfg, int64_t B, const float* w, const float* y_tiled, const float* e_tiled,
                    const float* w_lag, const float* y_lag_tiled, double* scratch, double* stats, double* loss_vec,
                    double* coef, void* p2p_comm, void* stream);
int cvf_ef_loss_dp(const cvf_ef_cfg* cfg, double* stats, double* loss_vec, double* coef, void* p2p_comm, void* stream);
int cvf_slab_reduce_dp(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const cvf_adam_args* adam,
                       void* p2p_comm, void* stream);
int cvf_p2p_exchange_f64(void* comm, double* buf, int64_t n, void* stream);

/* --- measurement aid (no reference counterpart): plain streaming on this device, timed by bench.py in the same loop as
 * cvf_align_feature_fwd so that the alignment kernel's HBM fraction can be read against what the box delivers at that moment.
 * mode 0: dst <- src, one 16-byte piece per thread; mode 1: read-only sweep (dst receives one float per 2048 pieces).
 * dst holds cvf_probe_stream_out_floats(mode, n_float4) floats. */
int64_t cvf_probe_stream_out_floats(int mode, int64_t n_float4);
int cvf_probe_stream(int mode, float* dst, const float* src, int64_t n_float4, void* stream);

#ifdef __cplusplus
}
#endif
#endif
