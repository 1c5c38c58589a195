/*
 * pcr.h -- C ABI of libpcr_hip.so, the MI355X (gfx950) implementation of the siamese
 * point-cloud ReID hot path of bentherien/point-cloud-reid.
 *
 * Conventions (all entry points):
 *   - plain `extern "C"`, raw DEVICE pointers, sizes as int, a hipStream_t passed as void*;
 *   - outputs are caller-allocated (as in the reference's pybind11 wrappers, e.g.
 *     mmdet3d/ops/ball_query/src/ball_query.cpp:30-43: dims first, tensors after, outputs
 *     pre-allocated by the Python side);
 *   - the call only ENQUEUES work on `stream`: no allocation, no synchronisation;
 *   - returns a pcr_status (0 = ok).  Unlike the reference launchers, which fprintf + exit(-1)
 *     on a launch error (e.g. ball_query_cuda.cu:73-77), nothing here ever exits the process;
 *   - all float data is IEEE binary32, all index data int32, tensors contiguous, layouts as in
 *     the reference op they replace (cited per function).
 *
 * Section A replaces the reference's native extension modules one for one.
 * Section B are the fused model kernels, which have no native counterpart in the reference
 * (its model path is an unfused chain of ATen calls); each cites the Python it computes.
 */
#ifndef PCR_H_
#define PCR_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pcr_stream_t; /* hipStream_t */

enum pcr_status {
  PCR_OK = 0,
  PCR_ERR_INVALID = 1, /* bad size / null pointer / unsupported configuration */
  PCR_ERR_LAUNCH = 2   /* hipLaunchKernel reported an error (hipGetLastError) */
};

/* ABI version (bumped whenever a signature changes) and human-readable status. */
int pcr_abi_version(void);
#define PCR_PREC_F32 0
#define PCR_PREC_BF16X3 1
#define PCR_PREC_BF16 2
const char *pcr_status_string(int status);

/* ---------------------------------------------------------------- A. point ops ------- */

/* furthest_point_sampling_wrapper (ops/furthest_point_sample/src/furthest_point_sample.cpp:32-43,
 * kernel furthest_point_sample_cuda.cu:25-141).  xyz (B,N,3); temp (B,N) in/out scratch that the
 * caller fills with 1e10 (furthest_point_sample.py:29) and that holds the final min-distances on
 * return; idx (B,M) int32.  Start index 0; tie rule identical to the reference's per-thread scan +
 * halving merge tree for block = min(1024, 2^floor(log2 N)).  Requires 1 <= N < 2^22. */
int pcr_fps_f32(const float *xyz, float *temp, int *idx, int B, int N, int M, pcr_stream_t stream);

/* furthest_point_sampling_with_dist_wrapper (furthest_point_sample.cpp:45-57, kernel .cu:213-331).
 * dist (B,N,N) pairwise distance matrix instead of coordinates. */
int pcr_fps_dist_f32(const float *dist, float *temp, int *idx, int B, int N, int M,
                     pcr_stream_t stream);

/* calc_square_dist (ops/furthest_point_sample/utils.py:4-32): the (B,N,M) matrix the F-FPS / FS samplers hand to
 * pcr_fps_dist_f32 (points_sampler.py:121-157).  a (B,N,C), b (B,M,C) -> out[i][j] = (|a_i|^2 + |b_j|^2) - 2 <a_i,b_j>,
 * sums left to right over the channels, no fma (the reference leaves the order to its matmul); norm: sqrt(.)/C. */
int pcr_pairwise_sqdist_f32(const float *a, const float *b, float *out, int B, int N, int M, int C, int norm,
                            pcr_stream_t stream);

/* ball_query_wrapper (ops/ball_query/src/ball_query.cpp:30-43, kernel ball_query_cuda.cu:11-54).
 * centres (B,M,3), xyz (B,N,3) -> idx (B,M,K): the first K indices k (ascending) with
 * d2 == 0 || (min_r^2 <= d2 < max_r^2), padded with the first hit; rows with no hit are written
 * as zeros (the reference leaves its pre-zeroed buffer untouched; the result is the same). */
int pcr_ball_query_f32(const float *centres, const float *xyz, int *idx, int B, int N, int M,
                       float min_r, float max_r, int K, pcr_stream_t stream);

/* pcr_ball_query_f32 that also returns cnt (B,M): the number of genuine hits (<= K) of every row. */
int pcr_ball_query_cnt_f32(const float *centres, const float *xyz, int *idx, int *cnt, int B, int N, int M,
                           float min_r, float max_r, int K, pcr_stream_t stream);

/* pcr_ball_query_cnt_f32 that ALSO writes the compact row table the wave-autonomous ragged SA kernel reads
 * (pcr_sa_params.row_tab; ABI 12).  rows: pcr_ball_query_rows_floats(B, M, K) floats, 16-byte aligned, = for every
 * (cloud, run of 16 consecutive centres) a region of 16 K entries of 4 floats {bits of the neighbour index, dx, dy, dz}
 * (point - centre): centre after centre its first ceil2(max(cnt, 1)) rows of idx (odd counts and empty balls padded
 * with the row's first entry, exactly as idx pads), then zero entries up to the next multiple of 32 rows; the rest of
 * a region is not written.  idx may be NULL (not written).  Needs pcr_ball_query_rows_ok(N, K, min_r): N <= 1024,
 * K even, min_r == 0 (every ball-query layer of the reference's configs); PCR_ERR_INVALID otherwise. */
long pcr_ball_query_rows_floats(int B, int M, int K);
int pcr_ball_query_rows_ok(int N, int K, float min_r);
int pcr_ball_query_rows_f32(const float *centres, const float *xyz, int *idx, int *cnt, float *rows, int B, int N, int M,
                            float min_r, float max_r, int K, pcr_stream_t stream);

/* pcr_fps_f32 and pcr_ball_query_rows_f32 of the picked centres as ONE launch (round 5): a pick's distances to the cloud
 * are the ball query's distances of that centre, bit for bit, so the query costs one compare per point on top of the
 * sampling.  xyz (B,N,3), temp (B,N) as pcr_fps_f32 -> idx (B,M) the pick order, new_xyz (B,M,3) the centres' coordinates
 * (no gather launch), cnt (B,M) and rows exactly what pcr_ball_query_rows_f32(new_xyz, xyz, NULL, cnt, rows, .., 0, max_r,
 * K) writes.  Needs pcr_fps_ball_query_rows_ok(N, M, K): N <= 1024, 2 <= M <= N, K even.  (The reference runs the two
 * ops back to back: ops/pointnet_modules/point_sa_module.py:166-216 -> points_sampler.py:107-119, ball_query.py:14-47.) */
int pcr_fps_ball_query_rows_ok(int N, int M, int K);
int pcr_fps_ball_query_rows_f32(const float *xyz, float *temp, int *idx, float *new_xyz, int *cnt, float *rows, int B,
                                int N, int M, float max_r, int K, pcr_stream_t stream);

/* The model path's own (dormant) Python samplers / groupers, semantics of the PYTHON code rather than of the CUDA ops:
 *
 * farthest_point_sample (models/pointnet2_utils.py:116-137): first pick = start[b] (the reference draws it with
 * torch.randint; NULL = 0), running distance min(.), next pick = the maximum with ties to the LOWEST index
 * (torch.max), distance (dx^2+dy^2)+dz^2.  temp (B,N) is filled with 1e10 by the caller.  idx (B,M) int32.
 *
 * query_ball_point (:218-240): distances by square_distance's expanded form (-2<c,p> + |c|^2) + |p|^2 (:169-188),
 * a point is kept unless d > radius^2 (i.e. d <= r^2, where the CUDA op uses d < r^2), first K kept indices in index
 * order, padded with the first; a row without a hit is filled with N, as the reference's sort leaves it. */
int pcr_fps_py_f32(const float *xyz, float *temp, const int *start, int *idx, int B, int N, int M,
                   pcr_stream_t stream);
int pcr_query_ball_point_f32(const float *centres, const float *xyz, int *idx, int B, int N, int M, float radius,
                             int K, pcr_stream_t stream);

/* knn_wrapper (ops/knn/src/knn.cpp:28-41, kernel knn_cuda.cu:58-94).  xyz (B,N,3), centres (B,M,3)
 * -> idx (B,M,K) int32 and dist2 (B,M,K), ascending, produced by the same max-heap + heap-sort
 * sequence as the reference (so equal distances come out in the reference's order).
 * 1 <= K <= 100.  (knn.py:62 transposes idx to (B,K,M) on the Python side.) */
int pcr_knn_f32(const float *xyz, const float *centres, int *idx, float *dist2, int B, int N, int M,
                int K, pcr_stream_t stream);

/* gather_points_wrapper / gather_points_grad_wrapper (ops/gather_points/src/gather_points.cpp:28-52,
 * kernels gather_points_cuda.cu:8-26, :51-70).  feat (B,C,N), idx (B,M) -> out (B,C,M);
 * backward ACCUMULATES into grad_feat (B,C,N), which the caller zero-fills (gather_points.py:43). */
int pcr_gather_fwd_f32(const float *feat, const int *idx, float *out, int B, int C, int N, int M,
                       pcr_stream_t stream);
int pcr_gather_bwd_f32(const float *grad_out, const int *idx, float *grad_feat, int B, int C, int N,
                       int M, pcr_stream_t stream);

/* group_points forward / backward (ops/group_points/src/group_points.cpp:31-58, kernels
 * group_points_cuda.cu:56-79, :10-31).  feat (B,C,N), idx (B,S,K) -> out (B,C,S,K);
 * backward accumulates into the caller-zeroed grad_feat (B,C,N). */
int pcr_group_fwd_f32(const float *feat, const int *idx, float *out, int B, int C, int N, int S,
                      int K, pcr_stream_t stream);
int pcr_group_bwd_f32(const float *grad_out, const int *idx, float *grad_feat, int B, int C, int N,
                      int S, int K, pcr_stream_t stream);

/* three_nn_wrapper (ops/interpolate/src/interpolate.cpp:46-58, kernel three_nn_cuda.cu:11-65).
 * unknown (B,N,3), known (B,M,3) -> dist2 (B,N,3) SQUARED distances (three_nn.py:41 takes the
 * sqrt), idx (B,N,3).  Running bests are kept in double and compared with the float distance, as
 * in the reference. */
int pcr_three_nn_f32(const float *unknown, const float *known, float *dist2, int *idx, int B, int N,
                     int M, pcr_stream_t stream);

/* three_interpolate_wrapper / _grad_wrapper (interpolate.cpp:60-86, kernels
 * three_interpolate_cuda.cu:11-35, :61-84).  feat (B,C,M), idx/weight (B,N,3) -> out (B,C,N);
 * backward accumulates into the caller-zeroed grad_feat (B,C,M). */
int pcr_three_interp_fwd_f32(const float *feat, const int *idx, const float *weight, float *out,
                             int B, int C, int M, int N, pcr_stream_t stream);
int pcr_three_interp_bwd_f32(const float *grad_out, const int *idx, const float *weight,
                             float *grad_feat, int B, int C, int N, int M, pcr_stream_t stream);

/* ------------------------------------------------- B. fused model kernels ------------ */

/* Neighbour search of the "Point-Transformer" set-abstraction layers: centres are the first S
 * points of each cloud (random_point_sample, models/pointnet2_utils.py:139-149) and the K nearest
 * of all N points are selected (knn_point, :205-216).  The reference ranks an expanded-matmul
 * distance with an unstable argsort; this kernel selects by the direct squared distance
 * (dx*dx+dy*dy)+dz*dz with ties to the lower index and writes idx (B,S,K) in (distance, index)
 * order.  The consumer is a max over K, so only the K-SET matters (hard part 1).
 * xyz (B,N,3).  K <= N, K <= 64, N <= 16384. */
int pcr_knn_prefix_f32(const float *xyz, int *idx, int B, int N, int S, int K, pcr_stream_t stream);
/* The same search for TWO set-abstraction levels that query the same cloud (ABI 15; the Point-Transformer's first level
 * keeps all N points, so its second level -- S2 <= S centres, K2 >= K neighbours -- searches the cloud the first one
 * did): the K nearest of a query are the first K of its K2 nearest in (distance, index) order, so the first S2 queries are
 * ranked ONCE for K2 and write both lists.  idx (B,S,K) and idx2 (B,S2,K2) = what pcr_knn_prefix_f32(.., S, K) and
 * pcr_knn_prefix_f32(.., S2, K2) write, entry for entry (pointnet2_utils.py:139-149, 205-216 called by two layers,
 * backbone_net.py:50-81). */
int pcr_knn_prefix2_f32(const float *xyz, int *idx, int *idx2, int B, int N, int S, int K, int S2, int K2,
                        pcr_stream_t stream);

/* Packed weight image of one dense layer out = W x (W is (cout, cin) row-major as in
 * nn.Linear / 1x1 conv).  Returns the number of floats of the packed image for (cout, cin);
 * pcr_pack_weight_f32 writes it on the HOST (plain C, no GPU) so that a model can be packed at
 * load time and uploaded once.  The image is what the MFMA A-operand loads of every fused kernel
 * below read with one 16-byte load per lane. */
long pcr_packed_weight_floats(int cout, int cin);
int pcr_pack_weight_f32(const float *w, int cout, int cin, float *packed);
/* The same matrix as a bf16 image for the bf16 matrix core (v_mfma_f32_32x32x16_bf16): every weight is stored as
 * hi = bf16(w) and lo = bf16(w - hi) (round to nearest), laid out [ceil16(cin)/16][ceil32(cout)/32][hi, lo][64 lanes]
 * [8 bf16] so that a lane's A operand of one 16-channel step is one 16-byte load; element j of lane l (row l % 32, half
 * h = l / 32) of step s is column 16 s + 4 h + j (j < 4) | 16 s + 8 + 4 h + (j - 4): the order in which a 32 x 32
 * accumulator tile holds its rows, so that a layer's output can be stored already converted (csrc/tile_dense.h).  Kernels run in "bf16x3" mode (split
 * bf16: W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi, f32 accumulate, ~2^-17 relative per term) read both parts, in plain
 * "bf16" mode the hi part only.  Size in floats (4-byte units) / host-side pack. */
long pcr_packed_weight_bf16_floats(int cout, int cin);
int pcr_pack_weight_bf16x2_f32(const float *w, int cout, int cin, float *packed);

/* Grouped set-abstraction MLP: gather + edge/relative features + 3 x (1x1 conv -> per-channel
 * affine (folded eval-mode BatchNorm) -> ReLU) + max over the K neighbours, one pass, nothing but
 * the (B,C3,S) result written.
 *   mode 0 ("edge", PointNetSetAbstractionEdgeSA, models/pointnet2_utils.py:242-288, 333-357):
 *          rows = [xyz[idx]-centre (3), feat[centre] (D), feat[idx]-feat[centre] (D)], centre =
 *          point s (prefix sampling);
 *   mode 1 ("query-and-group", ops/group_points/group_points.py:94-118 with use_xyz=True, as used
 *          by PointSAModule, ops/pointnet_modules/point_sa_module.py:166-216):
 *          rows = [xyz[idx]-centre_xyz (3), feat[idx] (D)], centres given by centre_idx (B,S).
 * xyz (B,N,3); feat (B,D,N) or NULL when D == 0; idx (B,S,K); centre_idx (B,S) or NULL (=> s).
 * wp[l]: packed weights of layer l; scale[l]/shift[l]: (C_l) affine applied after the matmul
 * (conv bias and BatchNorm folded by the host).  out (B,C3,S). */
typedef struct pcr_sa_params {
  int mode, B, N, S, K, D;
  int c1, c2, c3;
  const float *xyz, *feat;
  const int *idx, *centre_idx;
  const float *wp[3], *scale[3], *shift[3];
  /* Optional decomposed first layer (fast path).  Layer 1 is linear in its input row, so
   *   W1 row = Wa dxyz + P[i] + Q[c],  P = Wf f (per point), Q = (Wc - Wf) f (edge mode only),
   * with Wa = W1[:, 0:3]; edge mode: Wc = W1[:, 3:3+D], Wf = W1[:, 3+D:3+2D]; query-and-group:
   * Wf = W1[:, 3:3+D], no Q.  wa is (c1,3) row-major; wpq is the PACKED image of the stacked
   * matrix [Wf ; Wc - Wf] ((2*c1, D), edge) or Wf ((c1, D), query-and-group); pq_ws is a caller
   * workspace of B*N*(2*c1 or c1) floats.  Leave wa NULL to force the generic kernel (wp[0]).
   * wa and wpq must be PRE-SCALED by scale[0] (row o times scale[0][o]); the fast path then
   * evaluates layer 1 as relu(wa dxyz + P[i] + Q[c] + shift[0]) and never reads scale[0].
   * Likewise wps[0], wps[1] are the packed images of diag(scale[1]) W2 and diag(scale[2]) W3, and
   * shift_pad[0], shift_pad[1] are shift[1], shift[2] zero-padded to a multiple of 32 floats: the
   * fast path seeds the MFMA accumulators with the shift and its epilogue is a bare ReLU. */
  const float *wa, *wpq;
  const float *wps[2], *shift_pad[2];
  /* Optional duplicate-free evaluation for ball-query groups (mode 1): cnt (B,S) = number of genuine hits
   * of each row of idx as returned by pcr_ball_query_cnt_f32 (entries [cnt,K) of a row repeat entry 0, and
   * a max over K ignores repeats), tile_ws = caller workspace of pcr_sa_tile_ws_ints(B,S,K,c2,c3) ints.  The kernel then runs the
   * MLP on ceil2(max(cnt,1)) rows per centre; the result is bit-identical to the K-row evaluation. */
  const int *cnt;
  int *tile_ws;
  float *pq_ws;
  int pq_ready; /* nonzero: pq_ws already holds the tables (caller ran pcr_dense_pm_f32 itself) */
  /* Layouts.  feat_point_major: feat is (B,N,D) instead of (B,D,N).  out_point_major: out is (B,S,c3) instead
   * of (B,c3,S): a centre's c3 channels are then one contiguous run (full-line stores; the (B,c3,S) form writes
   * 4-byte pieces S floats apart).  Both are pure layout choices; values are identical. */
  int feat_point_major, out_point_major;
  float *out;
  /* optional: (c1,3) dxyz weights of layer 1 (BatchNorm scale folded in, like wa) as a PACKED image
   * (pcr_pack_weight_f32 of the (c1,3) matrix): lets the persistent kernel run layer 1 on the matrix core too */
  const float *wa_packed;
  /* arithmetic of layers 2 and 3 (layer 1 and the tables stay f32): PCR_PREC_F32 = f32-input MFMA, exact fmaf chains
   * (wps); PCR_PREC_BF16X3 = split bf16, three bf16 MFMAs per product with f32 accumulation; PCR_PREC_BF16 = plain bf16
   * activations and weights, f32 accumulation (BASELINE config 2 as stated).  wps_bf[l]: pcr_pack_weight_bf16x2_f32
   * images of the matrices wps[l] holds.  Shapes the bf16 kernels do not cover run in f32. */
  int precision;
  const float *wps_bf[2];
  /* optional (ABI 11): pcr_pack_weight_f32 image of the (c1, 4) matrix [wa | shift[0]] -- the cout-split kernel feeds
   * (dx, dy, dz, 1) to the matrix core, so that layer 1's shift arrives with the coordinate term and costs no seed
   * reads.  Without it that kernel is not chosen. */
  const float *wa_shift_packed;
  /* optional (ABI 12): the ball query's row table of THIS launch's groups (pcr_ball_query_rows_f32, same B / S / K),
   * for ball-query layers with hit counts whose shape pcr_sa_uses_row_table accepts: the wave-autonomous ragged kernel
   * then reads a row's {neighbour, point - centre} with one load instead of chasing cnt -> idx -> xyz.  idx may be
   * NULL when it is given (cnt is still read).  Ignored by every other kernel. */
  const float *row_tab;
  /* optional (ABI 16): pcr_sa_claim_ws_ints(..) ints of scratch for the wave-autonomous K-row kernel on clouds of >= 1024
   * points: its waves then CLAIM their work items from counters in it (zeroed by the launch itself, on `stream`) instead
   * of walking a fixed stride, which keeps the waves of an XCD on consecutive items -- one or two clouds' tables live in
   * its L2 instead of three or four.  Same results bit for bit (an item's arithmetic does not depend on who runs it).
   * NULL, or a shape the query answers 0 for: fixed-stride items. */
  int *claim_ws;
  /* optional (ABI 17): nonzero = pq_ws was built by pcr_dense_pm_xyz_f32 (pq_ready must be set): the tables carry the
   * coordinate term of layer 1 and its shift -- P'[i] = P[i] + wa xyz[i], Q'[c] = Q[c] - wa xyz[c] + shift[0] -- so a row of
   * layer 1 is relu(P'[i] + Q'[c]): no coordinate loads, no layer-1 MFMAs, one add per element in the launch.  Only the
   * wave-autonomous K-row kernel reads such tables: set it ONLY when pcr_sa_tables_take_xyz answers 1 for the launch;
   * any other dispatch returns PCR_ERR_INVALID rather than evaluating the wrong formula. */
  int pq_has_xyz;
} pcr_sa_params;
int pcr_sa_mlp_f32(const pcr_sa_params *p, pcr_stream_t stream);
/* ints of pcr_sa_params.claim_ws a launch of this shape would use (0: it would not use any); shape-only */
long pcr_sa_claim_ws_ints(int c1, int c2, int c3, int K, int N, int precision);
/* ints of pcr_sa_params.tile_ws for the duplicate-free evaluation (tile lists + per-tile row tables) */
long pcr_sa_tile_ws_ints(int B, int S, int K, int c2, int c3);
/* 1: a launch of this shape WITHOUT hit counts (cnt NULL: all K rows of every group) also runs on the tile plan and
 * wants tile_ws (the cout-split kernel with register-resident weights, 128 / 128 / 256 in the bf16 modes): ragged and
 * K-row evaluation of such a layer then share one kernel, hence one arithmetic. */
int pcr_sa_krow_uses_tiles(int c1, int c2, int c3, int K, int precision);
/* 1: a ball-query layer of this shape with hit counts reads pcr_sa_params.row_tab when given (shape-only) */
int pcr_sa_uses_row_table(int c1, int c2, int c3, int K, int precision);

/* Per-point linear map with POINT-major output: x (B,cin,L) channel-major (or (B,L,cin) when x_point_major)
 * -> y (B,L,cout) = W x, wp packed (cout,cin), cout <= 1024 (a multiple of 4 beyond 256).  This is the table builder of the decomposed first
 * SA layer (pcr_sa_mlp_f32 runs it itself unless pq_ready is set). */
int pcr_dense_pm_f32(const float *x, const float *wp, float *y, int B, int cin, int cout, int L,
                     int x_point_major, pcr_stream_t stream);
/* the same map on the bf16 matrix core: wp_bf = pcr_pack_weight_bf16x2_f32 image of W, precision = PCR_PREC_BF16X3 or
 * PCR_PREC_BF16 */
int pcr_dense_pm_prec_f32(const float *x, const float *wp_bf, float *y, int B, int cin, int cout, int L,
                          int x_point_major, int precision, pcr_stream_t stream);
/* (ABI 17) the same tables WITH the coordinate term of the decomposed first layer (reference: the dxyz columns of
 * sample_and_group_edge's new_points, models/pointnet2_utils.py:242-288): y[b][l][o] = (W x)[o] + wxyz[o][3] +
 * wxyz[o][0] xyz[b][l][0] + wxyz[o][1] xyz[b][l][1] + wxyz[o][2] xyz[b][l][2], the coordinate products as f32 fmas on the
 * f32 sum (added z, y, x in that order), whatever `precision` the feature product W x runs in.  xyz (B,L,3), wxyz (cout,4)
 * row-major: the caller puts {+wa, 0} in the P rows and {-wa, shift} in the Q rows.  cout % 4 == 0.
 * q_rows / q_off: tokens l >= q_rows receive the couts [0, q_off) only, the rest of their rows is left unwritten -- the Q
 * half of an SA table is read for CENTRES only, which under prefix sampling (centre_idx == NULL) are the first S points:
 * q_rows = S rounded up to a multiple of 64, q_off = c1.  q_rows = L, q_off = cout: every token, every cout. */
int pcr_dense_pm_xyz_f32(const float *x, const float *wp_bf, const float *xyz, const float *wxyz, float *y, int B, int cin,
                         int cout, int L, int x_point_major, int precision, int q_rows, int q_off, pcr_stream_t stream);
/* shape-only: does the pcr_sa_mlp_f32 launch of this shape run on the wave-autonomous K-row kernel, i.e. may its tables be
 * built with pcr_dense_pm_xyz_f32 (pcr_sa_params.pq_has_xyz)?  Edge mode (0) with features, c1 == c2 == c3 in {32, 64, 128},
 * K a multiple of 16 that divides 32, 64 or 96, a bf16 precision. */
int pcr_sa_tables_take_xyz(int mode, int D, int c1, int c2, int c3, int K, int precision);

/* Linear-attention block shared by Self_Attention (models/pointnet2_utils.py:90-114), FP_SA
 * (:407-437) and corss_attention (models/attention.py:192-219), in two kernels.
 *
 * With h = relu(W0 xyz + b0) the reference's position encoding is pos = W2 h + b2; the HOST folds
 * W2/b2 into the projections (all products formed in fp64, rounded once to fp32):
 *   wq  = packed [Wq | Wq W2] (d, c1+d)  if q_pos  else packed Wq (d, c1);   bq = Wq b2 or 0
 *   wkv = packed [[Wk | kpos Wk W2] ; [Wv | Wv W2]] (2d, c2+d);   bkv = [kpos Wk b2 ; Wv b2]
 * where kpos = 1 for Self_Attention (keys carry the position encoding) and 0 otherwise.
 *
 * pcr_attn_kv_f32: per key-side cloud, K = elu(.)+1, V = (.)/Sk from one fused projection of
 *   [feat_k ; h]; accumulates KV_head = sum_s K V^T and ksum = sum_s K, folds the merge projection
 *   (wmerge (d,d) row-major) into KV and writes, per cloud, the packed (d,d) matrix
 *   M[o][dd] = sum_{v in head(dd)} Wm[o][v] KV[dd][v] followed by ksum (d).
 * pcr_attn_apply_f32: per query token, Q = elu(wq [x ; h] + bq)+1, Q' = Q Sk / (Q.ksum + 1e-6)
 *   per head, msg = LayerNorm(M Q'), feed-forward mlp2(relu(mlp0 [x ; msg])), LayerNorm, optional
 *   residual; cloud b reads the kv image of cloud kv_index[b] (NULL => b), which is how the siamese
 *   matching head pairs clouds without copying (ReIDNet.xcorr_eff, models/ReIDNet.py:231-247).
 *   Optional fused trailing 1x1 conv (Pointnet_Backbone.cov_final, models/backbone_net.py:89,124).
 * d_model in {32,64,96,128,256,512}; c2 % 8 == 0; q_pos requires c1 == c2 == d; residual requires cout == c1. */
typedef struct pcr_attn_params {
  int B, Lq, Sk;             /* clouds, query tokens per cloud, key tokens per cloud */
  int c1, c2, d, cout;       /* query feature dim, key feature dim, d_model, output dim */
  int nhead;
  int q_pos, residual;
  const float *feat_q, *xyz_q; /* (B,c1,Lq), (B,Lq,3) (xyz_q only read when q_pos) */
  const float *feat_k, *xyz_k; /* (B,c2,Sk), (B,Sk,3) */
  const int *kv_index;         /* (B) or NULL: apply: cloud b reads the kv image of cloud kv_index[b] */
  const int *q_index;          /* (B) or NULL: apply: cloud b takes its query tokens from cloud q_index[b]
                                  (gallery matching: B = number of (query, template) combinations) */
  const float *pos0_w, *pos0_b;   /* (d,3) row-major, (d) */
  const float *wq, *bq;           /* packed fused query projection, (d) */
  const float *wkv, *bkv;         /* packed fused key/value projection, (2d) */
  const float *wmerge;            /* (d,d) row-major */
  const float *wmlp0, *wmlp2;     /* packed (2d, c1+d), packed (cout, 2d) */
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const float *wfinal, *bfinal; int cfinal;       /* optional trailing conv: packed (cfinal,cout); bias zero-padded to a multiple of 32 */
  /* d_model 256 / 512 only (d % 64 == 0, head width a multiple of 64 and <= 256; the mul = 2 / 4 Point-Transformer
   * configs): the kv kernel splits a cloud over d/64 workgroups, band g owning rows [64g, 64g+64) of KV.
   * wkv_wide: d/64 packed images, image g = pcr_pack_weight_f32 of the (64 + dh, c2 + d) matrix made of rows
   * [64g, 64g+64) (K band) and [d + hd dh, d + (hd+1) dh) (V rows of the band's head hd) of the fused projection wkv;
   * bkv_wide: the same rows of bkv, (d/64, 64 + dh); wmerge_packed: packed image of wmerge (d,d).  NULL otherwise. */
  const float *wkv_wide, *bkv_wide, *wmerge_packed;
  float *kv;    /* workspace (B, pcr_attn_kv_floats(d)) */
  float *out;   /* (B, cfinal ? cfinal : cout, Lq) */
  /* precision != PCR_PREC_F32 and d <= 128: the dense phases of the apply kernel (Q, message, feed-forward, cov_final)
   * run as split bf16 (three bf16 MFMAs per product, f32 accumulate) on the pcr_pack_weight_bf16x2_f32 images of the same
   * matrices: wq_bf, wmlp0_bf, wmlp2_bf, wfinal_bf (NULL: f32).  The kv kernel then writes the per-cloud matrix M as a
   * bf16 image, so pcr_attn_kv_f32 and pcr_attn_apply_f32 must be called with the SAME precision. */
  int precision;
  const float *wq_bf, *wmlp0_bf, *wmlp2_bf, *wfinal_bf;
  /* pcr_attn_kv_f32, d <= 128: kv_splits > 1 splits every cloud's key tokens over that many workgroups (raw partial
   * matrices in kv_part, (B, kv_splits, d d + d) floats) and a second launch adds them in order and folds the merge
   * projection -- parallelism for batches of a few clouds; free on a grid that fills the chip.
   * pcr_attn_kv_splits(B, Sk, d) suggests a count from the launch shape alone (1 = the single-launch form; kv_part may
   * then be NULL): results do not depend on the batch size. */
  int kv_splits;
  float *kv_part;
  /* precision != PCR_PREC_F32, d = c2 = 64, Sk % 32 == 0 (the wave-autonomous kv kernel): pcr_pack_weight_bf16x2_f32
   * image of the fused K / V projection wkv -- the projection then runs as split bf16 too.  NULL: f32 projection. */
  const float *wkv_bf;
  /* c1 not a multiple of 16 (the first FP_SA block: c1 = 3): pcr_pack_weight_bf16x2_f32 image of mlp[0]'s weight with the
   * query-feature columns padded to a whole 16-channel step, [W[:, :c1] | 0 | W[:, c1:]] -- the wave-autonomous apply
   * kernel's contraction steps are 16 channels wide.  NULL: the tile kernel. */
  const float *wmlp0_bf_xpad;
  /* ABI 16 (round 6): pooled output.  NULL: the block output goes to `out` as before.  Non-NULL (only where
   * pcr_attn_apply_pool_ok says yes: the wave-autonomous d = c1 = cout = 64 form without q_pos / trailing conv, i.e. the
   * matching stages' corss_attention, models/attention.py:192-219): `out` is NOT written (may be NULL); instead every
   * virtual cloud leaves its per-channel maximum and SUM over its Lq tokens, pool_out (B, 2, 64) = [max | sum] -- what
   * get_pooled_feats 'both' (models/ReIDNet.py:526-534) needs of the block output, 512 bytes instead of 256 Lq. */
  float *pool_out;
} pcr_attn_params;
long pcr_attn_kv_floats(int d);
int pcr_attn_kv_splits(int B, int Sk, int d);
int pcr_attn_kv_f32(const pcr_attn_params *p, pcr_stream_t stream);
int pcr_attn_apply_f32(const pcr_attn_params *p, pcr_stream_t stream);
/* 1 if pcr_attn_apply_f32 honours p->pool_out for these parameters (a function of the launch SHAPE and the arithmetic
 * mode, never of B: a pair's result does not depend on the batch it travels in), else 0 */
int pcr_attn_apply_pool_ok(const pcr_attn_params *p);

/* Matching head tail: pool 'both' over the point-concatenated pair (get_pooled_feats,
 * models/ReIDNet.py:526-534: [max over 2L points, mean over 2L points]) followed by
 * LinearRes(2C,2C,GroupNorm) + Linear(2C,1) (models/lanegcn_nets.py:228-241, ReIDNet.py:455-457).
 * o (2P, C, L): cloud p and cloud p+P form pair p.  logits (P); pooled (P,2C) optional (may be NULL). */
typedef struct pcr_head_params {
  int P, C, L, groups;
  const float *o;
  const float *w1, *w2;            /* (2C,2C) row-major, NOT packed */
  const float *gn1_g, *gn1_b, *gn2_g, *gn2_b;
  const float *w_out, *b_out;      /* (1,2C), (1) */
  float *pooled, *logits;
  /* optional (ABI 12, appended): the TRANSPOSES of w1 / w2 ((2C,2C) row-major: w1t[i][o] = w1[o][i]).  With them the
   * matrix-vector products read 256 contiguous bytes per wave and step instead of 64 scattered 16-byte pieces (the
   * head of a 32 k-pair gallery launch was bound by exactly those reads); same products, same summation order. */
  const float *w1t, *w2t;
} pcr_head_params;
int pcr_pool_head_f32(const pcr_head_params *p, pcr_stream_t stream);

/* get_pooled_feats with pool_type='both' on its own (models/ReIDNet.py:529-532):
 * x (B,C,L) -> out (B,2C) = [max over L, mean over L]. */
int pcr_pool_both_f32(const float *x, float *out, int B, int C, int L, pcr_stream_t stream);

/* get_pooled_feats with pool_type='max' (models/ReIDNet.py:145,526-528; reid_pts_point-transformer_baseline.py):
 * nn.MaxPool1d(window) on the permuted (B,L,C) tensor = max over windows of `window` consecutive channels of every
 * point, floor mode.  x (B,C,L) -> out (B,L,C/window). */
int pcr_channel_max_f32(const float *x, float *out, int B, int C, int L, int window, pcr_stream_t stream);

/* Generic per-point dense layer y = act(scale * (W x) + shift) on channel-major tensors
 * x (B,cin,L) -> y (B,cout,L): 1x1 Conv1d / Linear (+ folded BatchNorm) of the PointNet encoder
 * (models/pointnet.py:27-45, 67-85, 103-127).  act: 0 none, 1 ReLU, 2 LeakyReLU(0.2) (dgcnn_orig.py:112-114).
 * wp packed. */
int pcr_dense_f32(const float *x, const float *wp, const float *scale, const float *shift, float *y,
                  int B, int cin, int cout, int L, int act, pcr_stream_t stream);
/* the same with x given point-major, (B,L,cin) */
int pcr_dense_xpm_f32(const float *x, const float *wp, const float *scale, const float *shift, float *y,
                      int B, int cin, int cout, int L, int act, pcr_stream_t stream);

/* Linear (no bias) -> GroupNorm [-> + res] [-> ReLU] in one launch: the three steps LinearRes repeats
 * (lanegcn_nets.py:228-241) on channel-major token tensors, y (B,cout,L) = [relu](GN(W x) * gamma + beta [+ res]).
 * GroupNorm is per token over groups of cout/groups consecutive channels (4, 8, 16 or 32 per group; anything else
 * returns PCR_ERR_INVALID and the caller runs pcr_dense_f32 + pcr_groupnorm_f32), biased variance, eps 1e-5;
 * res (optional) is (B,cout,L). */
int pcr_dense_gn_f32(const float *x, const float *wp, const float *gamma, const float *beta, const float *res,
                     float *y, int B, int cin, int cout, int L, int groups, int relu, pcr_stream_t stream);
/* The same two launches on the bf16 matrix core (round 4; ABI 11): wp_bf = pcr_pack_weight_bf16x2_f32 image of W,
 * precision = PCR_PREC_BF16X3 (split bf16, three MFMAs per product, f32 accumulate) or PCR_PREC_BF16.  Covered shapes:
 * pcr_dense_prec_ok(cin, cout, L) != 0 (cin a multiple of 64, cout a multiple of 32, channel-major x, shared weights);
 * anything else returns PCR_ERR_INVALID and belongs to the f32 launches.  Reference: the 1x1 convs / Linear layers of
 * models/pointnet.py:27-127, dgcnn_orig.py:147, lanegcn_nets.py:228-241 (LinearRes). */
int pcr_dense_prec_ok(int cin, int cout, int L);
int pcr_dense_prec_f32(const float *x, const float *wp_bf, const float *scale, const float *shift, float *y, int B, int cin,
                       int cout, int L, int act, int precision, pcr_stream_t stream);
int pcr_dense_gn_prec_f32(const float *x, const float *wp_bf, const float *gamma, const float *beta, const float *res,
                          float *y, int B, int cin, int cout, int L, int groups, int relu, int precision,
                          pcr_stream_t stream);
/* pcr_dense_prec_f32 for a POINT-major input x (B,L,cin) (ABI 12; pcr_dense_xpm_f32's shapes on the bf16 matrix core):
 * cin a multiple of 64, cout <= 128, the weight image within LDS (pcr_dense_xpm_prec_ok).  Reference: the Conv1d after
 * the last set-abstraction layer of models/pointnet2_ssg.py (PointNet2SSG.cov_final). */
int pcr_dense_xpm_prec_ok(int cin, int cout, int L);
int pcr_dense_xpm_prec_f32(const float *x, const float *wp_bf, const float *scale, const float *shift, float *y, int B,
                           int cin, int cout, int L, int act, int precision, pcr_stream_t stream);

/* ---- PointNet encoder pieces (models/pointnet.py:10-127) and LinearRes rows (lanegcn_nets.py:228-241) ---- */

/* (B,C,L) -> out (C,B) with out[c*B + b] = max over L: the global max pool of STN3d/STNkd
 * (pointnet.py:31,71), written channel-major with the clouds as tokens so that the fc layers that
 * follow are plain pcr_dense_f32 calls on a (1,C,B) tensor. */
int pcr_max_over_l_f32(const float *x, float *out, int B, int C, int L, pcr_stream_t stream);

/* pcr_dense_f32 followed by pcr_max_over_l_f32 as ONE launch that never writes the (B,cout,L) tensor (round 5): the conv3 +
 * BN + ReLU + max over the points of STN3d / STNkd (models/pointnet.py:27-33, 67-73).  out (cout,B) as pcr_max_over_l_f32
 * writes it; bit-equal to the two-launch form (a maximum does not depend on the order).  pcr_dense_max_ok: cout a multiple
 * of 256 and cin small enough for a 64-token tile of the whole input extent in 64 KB of LDS (cin <= 232). */
int pcr_dense_max_ok(int cin, int cout, int L);
int pcr_dense_max_f32(const float *x, const float *wp, const float *scale, const float *shift, float *out, int B, int cin,
                      int cout, int L, int act, pcr_stream_t stream);

/* t (1,k*k,B) = the fc3 output of an STN (+identity), entry [c*k + c2][b] = T_b[c][c2] -> per-cloud packed
 * weight images (B, packed(k,k)) of W_b = T_b^T; pcr_dense_bmm_f32(x, images) then equals
 * torch.bmm(x^T, T)^T (pointnet.py:110,118). */
int pcr_pack_bmm_f32(const float *t, float *wp_per_cloud, int B, int k, pcr_stream_t stream);
int pcr_dense_bmm_f32(const float *x, const float *wp_per_cloud, float *y, int B, int cin, int cout, int L,
                      pcr_stream_t stream);

/* GroupNorm over channel groups of every token of x (B,C,L) (nn.GroupNorm on (M,C) rows in the reference's
 * LinearRes, with M = B*L tokens), optional residual add, optional ReLU: y = [relu](GN(x) [+ res]). */
int pcr_groupnorm_f32(const float *x, const float *gamma, const float *beta, const float *res, float *y, int B,
                      int C, int L, int groups, int relu, pcr_stream_t stream);

/* ---- DGCNN EdgeConv (models/dgcnn_orig.py: knn :22-29, get_graph_feature :32-56, DGCNN.forward :127-152) ---- */

/* Feature-space kNN of every point among the points of its own cloud (replaces dgcnn_orig.knn, :22-29):
 * x (B,C,N) channel-major with batch stride x_bstride floats (0 = C*N), xx_ws = caller workspace of B*N floats
 * -> idx (B,N,K) int32, the K largest pd_ij = (2 <x_i,x_j> - |x_i|^2) - |x_j|^2 of row i, largest first, ties by
 * the lower index.  The reference leaves the order of summation of its matmul to the BLAS; this library fixes
 * <.,.> = fmaf chain over channels 0..C-1 and |.|^2 = left-to-right sum of rounded squares
 * (oracle/pcr_oracle.c:pcr_oracle_knn_feat).  K <= 64, K <= N <= 2048, C <= 512 (and the LDS budget: see
 * csrc/edge_kernels.hip). */
int pcr_knn_feat_f32(const float *x, float *xx_ws, int *idx, int B, int C, int N, int K, long x_bstride,
                     pcr_stream_t stream);

/* The gather + max + BatchNorm shift + LeakyReLU tail of one EdgeConv layer (dgcnn_orig.py:129-131 and the
 * like): ta, tb (B,N,Co) point-major tables A = (s.W1) f and Bt = (s.(W2-W1)) f built with pcr_dense_pm_f32
 * (s = folded BatchNorm scale, W = [W1 | W2] the (Co,2C) conv weight), idx (B,N,K) ->
 * out[b][c][i] = leaky_slope(max_j ta[b][idx[b][i][j]][c] + tb[b][i][c] + shift[c]), channel-major with batch
 * stride out_bstride floats (0 = Co*N); out2 (optional, may be NULL) receives the same values with its own
 * batch stride (the x1..x4 slices of the concatenated conv5 input, dgcnn_orig.py:145).  Co <= 256, K <= 64. */
int pcr_edge_max_f32(const float *ta, const float *tb, const int *idx, const float *shift, float slope, float *out,
                     long out_bstride, float *out2, long out2_bstride, int B, int N, int Co, int K,
                     pcr_stream_t stream);

/* The attention step of local_self_attention (models/attention.py:262-289): every point is ONE query token over its
 * K feature-space neighbours.  qkv (B,N,3C) point-major rows [q | k | v] = the three projections of
 * feat + pos_mlp(xyz) (built with pcr_dense_pm_f32), idx (B,N,K) from pcr_knn_feat_f32 ->
 * msg (B,C,N) channel-major, msg_i = sum_j a_ij v_j / (sum_j a_ij + eps), a_ij = <elu(q_i)+1, elu(k_j)+1> per head
 * (identical to the reference's LinearAttention with L = 1, S = K).  C <= 64, C / nhead a power of two. */
int pcr_local_attn_f32(const float *qkv, const int *idx, float *msg, int B, int N, int C, int K, int nhead, float eps,
                       pcr_stream_t stream);

/* ------------------------------------------------- C. training-mode kernels ------------ */
/* Forward with BatchNorm batch statistics and backward of the same path (the reference trains it with autograd over
 * unfused ATen ops: models/ReIDNet.py:586-634,694-738; pointnet2_utils.py:333-360; group_points_cuda.cu:10-31).
 * All tensors are (B, C, L) channel-major fp32; every reduction is two-stage in a fixed order (per-workgroup partials
 * + a reduce / finalize launch): no float atomics, gradients are bit-reproducible.  csrc/train_kernels.hip. */

/* device-side pcr_pack_weight_f32 (weights change every iteration): W (rows x cols, leading dimension ld) -> packed
 * image of W (transpose = 0), of W^T (transpose = 1), or both, W first (transpose = 2); packed holds
 * pcr_packed_weight_floats(cout, cin) floats per image */
int pcr_pack_weight_dev_f32(const float *w, int rows, int cols, int ld, int transpose, float *packed, pcr_stream_t stream);

/* workgroups a train-dense launch uses for (B, L) (a workgroup strides over tiles AND clouds): the partial buffers
 * below have this many entries */
int pcr_train_groups(int B, int L);
/* ... of a pcr_tdense_bwd_f32 launch that accumulates dW (cout x cin; cout = 0: no dW): wide layers on short clouds use
 * fewer workgroups, each accumulating over more clouds */
int pcr_train_groups_bwd(int B, int L, int cout, int cin);
/* ... of THE launch a parameter block describes (ABI 8): the narrow grouped-MLP layers (32 / 64 channels, L a multiple
 * of 32) run through wave-autonomous kernels with their own grid (csrc/train_stream_kernels.hip); every other launch
 * returns what the two functions above return.  Size stats / dstats / dwp / dbp with these. */
struct pcr_tdense_fwd;
struct pcr_tdense_bwd;
int pcr_tdense_fwd_groups(const struct pcr_tdense_fwd *p);
int pcr_tdense_bwd_groups(const struct pcr_tdense_bwd *p);
/* launch policy: train-dense launches with fewer than n 32-token blocks (B * L / 32) stay on the tile kernels
 * (default 8192); returns the previous value, n < 0 only reads it.  Process-wide; tests use it to run small shapes
 * through the wave-autonomous kernels. */
int pcr_set_stream_min_blocks(int n);

/* y = [relu](W f([x ; x2]) + bias [+ res]),  f(x) = [relu](isc x + ish) on the cin1 channels of x (the previous
 * layer's BatchNorm + ReLU, applied while the tile is loaded; isc NULL = identity).  stats (optional): partials
 * [groups][2][ceil32(cout)] of sum y and sum y^2 (before res / relu) for pcr_bn_fwd_finalize_f32.  cout <= 384. */
typedef struct pcr_tdense_fwd {
  int B, cin1, cin2, cout, L;
  const float *x, *x2;
  const float *isc, *ish;
  int in_relu;
  const float *wp, *bias;      /* packed (cout, cin1+cin2); bias zero-padded to ceil32(cout) or NULL */
  const float *res;
  int out_relu;
  float *y;
  float *stats;
  /* optional fused pooling of a grouped MLP's LAST layer (L = S pool_K rows per cloud): the max over the K rows of every
   * centre of relu(BatchNorm(y)) has its winner at the max of the RAW y where pool_gamma[c] >= 0 and at the min where
   * it is negative, so the launch can leave pool_ymax (B,cout,S) = the raw y at the winning row and pool_arg (B,cout,S)
   * = that row -- what pcr_sa_pool_fwd_f32 computes from a second pass over y.  Only the wave-autonomous kernels do
   * this (pool_K >= 32, L a multiple of lcm(32, pool_K)): pcr_tdense_fwd_pooled() says whether THIS launch will;
   * otherwise the fields are ignored and the caller runs pcr_sa_pool_fwd_f32. */
  int pool_K;
  const float *pool_gamma;
  float *pool_ymax;
  int *pool_arg;
} pcr_tdense_fwd;
int pcr_tdense_fwd_f32(const pcr_tdense_fwd *p, pcr_stream_t stream);
int pcr_tdense_fwd_pooled(const pcr_tdense_fwd *p);

/* Backward of that layer.  dy is formed while the tiles are loaded: dy_mode 0: dy = g; 1: dy = ka g + kb y + kc
 * (BatchNorm backward, constants from pcr_bn_bwd_finalize_f32; y = the layer's stored raw output); 2: dy = g [y > 0]
 * (layer stored after its ReLU); 3: as 1 with g = the gradient gp (B,cout,S) of the max-pooled output routed to row
 * argmax (B,cout,S) of every centre where pooled (B,cout,S) > 0 (L = S K; pooled NULL: gp is already zero there).
 * Outputs (each optional): dx / dx2 = W^T dy masked by f(x) > 0 when in_relu (wpT = packed W^T); dstats = partials
 * [groups][2][ceil32(cin1)] of sum dx and sum dx * (raw x) for the next BatchNorm backward (iinv = 1 / isc);
 * dwp = partials [groups][ceil32(cout)][ceil32(cin)] of dy f(x)^T, dbp = partials [groups][ceil32(cout)] of sum dy
 * (reduce with pcr_reduce_parts_f32).  cout <= 384, cin1 + cin2 <= 288. */
typedef struct pcr_tdense_bwd {
  int B, cin1, cin2, cout, L;
  const float *g, *y;
  int dy_mode;
  const float *ka, *kb, *kc;
  const int *argmax;
  const float *pooled;
  int K, S;
  const float *x, *x2;
  const float *isc, *ish, *iinv;
  int in_relu;
  const float *wpT;
  float *dx, *dx2, *dstats, *dwp, *dbp;
  long part_stride;   /* floats between consecutive workgroups' dwp / dbp partials; 0 = two dense arrays */
  /* optional (ABI 11): precision = PCR_PREC_BF16X3 with wpT_bf = the bf16 hi / lo image of W^T
   * (pcr_pack_weight_bf16_dev_f32, transpose = 1) runs the 128 x 128 layers' dx and dW as split bf16 on the bf16 matrix
   * core (three MFMAs per product, f32 accumulation); every other shape ignores both fields */
  int precision;
  const float *wpT_bf;
} pcr_tdense_bwd;
int pcr_tdense_bwd_f32(const pcr_tdense_bwd *p, pcr_stream_t stream);
/* device-side pcr_pack_weight_bf16x2_f32: the bf16 hi / lo image of W (rows x cols, leading dimension ld; transpose = 0)
 * or of W^T (transpose = 1); packed holds pcr_packed_weight_bf16_floats(cout, cin) floats */
int pcr_pack_weight_bf16_dev_f32(const float *w, int rows, int cols, int ld, int transpose, float *packed,
                                 pcr_stream_t stream);

/* Training-mode core of local_self_attention (attention.py:262-296): qkv (B,3C,N) channel-major = the fused q | k | v
 * projection of feat + pos(xyz) per POINT, idx (B,N,K) feature-space neighbours (pcr_knn_feat_f32).
 * fwd: msg (B,C,N), msg_i = sum_j a_ij v_j / (sum_j a_ij + eps), a_ij = <elu(q_i)+1, elu(k_j)+1> per head.
 * bwd: g (B,C,N) -> dq (C rows of N per cloud, clouds dq_bstride floats apart: a slice of the (B,3C,N) gradient) and
 * the per-edge gradients edge (B,2C,N,K) = [d k_j | d v_j contributions]; pcr_group_bwd_f32(edge, idx, .) folds them
 * onto the points (no float atomics).  C <= 64, C / nhead a power of two. */
int pcr_local_attn_train_fwd_f32(const float *qkv, const int *idx, float *msg, int B, int N, int C, int K, int nhead,
                                 float eps, pcr_stream_t stream);
int pcr_local_attn_train_bwd_f32(const float *qkv, const int *idx, const float *g, float *dq, long dq_bstride, float *edge,
                                 int B, int N, int C, int K, int nhead, float eps, pcr_stream_t stream);

/* out[r][c] = sum over p < nparts (increasing p) of part[p * stride + r * ld + c] */
int pcr_reduce_parts_f32(const float *part, int nparts, long stride, int rows, int cols, int ld, float *out,
                         pcr_stream_t stream);

/* n reductions of that kind in ONE launch (round 5): out (rows x cols, contiguous) = sum over p < nparts (increasing p) of
 * part[p stride + r ld + c].  `jobs` is a HOST array (passed by value in the kernel arguments, 64 per launch): the
 * regions of a backward launch's partial record are summed into compact per-parameter tensors (pcr_amd/train_ops.py:
 * reduce_regions). */
typedef struct pcr_reduce_job {
  const float *part;
  float *out;
  long stride;
  int nparts, rows, cols, ld;
} pcr_reduce_job;
int pcr_reduce_multi_f32(const pcr_reduce_job *jobs, int n, pcr_stream_t stream);

/* BatchNorm (training) from the statistics partials [nparts][2][ceil32(C)] over R rows: mean, biased variance ->
 * scale = gamma invstd, shift = beta - mean scale, inv_scale = 1 / scale; running statistics updated in place
 * (momentum, unbiased variance) when given.  nn.BatchNorm2d semantics (pointnet2_utils.py:353-355). */
typedef struct pcr_bn_fwd_fin {
  const float *part;
  int nparts, C;
  double R;
  const float *gamma, *beta;
  float eps, momentum;
  float *running_mean, *running_var;
  float *scale, *shift, *inv_scale, *mean, *invstd;
  /* optional: the partials are sums of (y - shift0[c * shift0_stride]) and of its square (a per-channel offset close
   * to the mean removes the cancellation of E[y^2] - mean^2); NULL = plain sums */
  const float *shift0;
  int shift0_stride;
} pcr_bn_fwd_fin;
int pcr_bn_fwd_finalize_f32(const pcr_bn_fwd_fin *p, pcr_stream_t stream);

/* BatchNorm backward constants from partials [nparts][2][ceil32(C)] of S1 = sum dyhat, S2 = sum dyhat * y:
 * dbeta = S1, dgamma = invstd (S2 - mean S1), and ka, kb, kc with dy = ka dyhat + kb y + kc */
typedef struct pcr_bn_bwd_fin {
  const float *part;
  int nparts, C;
  double R;
  const float *gamma, *mean, *invstd;
  float *ka, *kb, *kc, *dgamma, *dbeta;
  const float *centre;   /* optional (C): S2 was taken of dyhat * (y - centre[c]) (centre = mean: no cancellation); NULL = 0 */
} pcr_bn_bwd_fin;
int pcr_bn_bwd_finalize_f32(const pcr_bn_bwd_fin *p, pcr_stream_t stream);

/* First layer of the grouped edge MLP from per-point tables (training; csrc/train_sa_kernels.hip):
 * y[b][c][s K + k] = wa[c] . (xyz[idx] - xyz[s]) + bias[c] + tab[b][c][idx] + tab[b][c1 + c][s]   (tab NULL: no
 * point features), stats = partials [B][2][ceil32(c1)] of sum y, sum y^2.  Centres are the first S points. */
int pcr_sa_l1_fwd_f32(const float *xyz, const int *idx, const float *tab, const float *wa, const float *bias,
                      float *y, float *stats, int B, int N, int S, int K, int c1, pcr_stream_t stream);
/* its backward: dy = ka g + kb y + kc; dtab (B,2 c1,N) = [sum of dy over the rows that gathered each point ; sum over
 * the K rows of each centre (zero beyond S)], dwa = partials [B][c1][4] of (d wa, d bias).  No atomics: every
 * accumulator has one owner that adds in row order. */
int pcr_sa_l1_bwd_f32(const float *xyz, const int *idx, const float *g, const float *y, const float *ka,
                      const float *kb, const float *kc, float *dtab, float *dwa, int B, int N, int S, int K, int c1,
                      pcr_stream_t stream);
/* pooled[b][c][s] = max_k relu(scale[c] y[b][c][s K + k] + shift[c]), argmax = the first k attaining it, ymax (optional)
 * = the raw y at that row (what the backward's BatchNorm sums need) */
int pcr_sa_pool_fwd_f32(const float *y, const float *scale, const float *shift, float *pooled, int *argmax, float *ymax,
                        int B, int C, int S, int K, pcr_stream_t stream);
/* partials [B][2][ceil32(C)] of S1 = sum gp [pooled > 0], S2 = sum gp [pooled > 0] ymax for pcr_bn_bwd_finalize_f32;
 * gz (optional, (B,C,S)) = gp [pooled > 0], the routed gradient: pass it as g with pooled = NULL to dy_mode 3 */
int pcr_sa_pool_bwd_stats_f32(const float *gp, const float *pooled, const float *ymax, float *part, float *gz, int B, int C,
                              int S, pcr_stream_t stream);

/* LayerNorm / GroupNorm over the channels of every token of x (B,C,L) (G groups of C/G consecutive channels; LayerNorm:
 * G = 1), y = [relu]((x - mean) rstd gamma + beta [+ res]); mean / rstd (B,G,L) are kept for the backward.  Backward: dx and
 * partials [ceil(B L / 64)][2][C] (one row per 64-token wave) of (d gamma, d beta) for pcr_reduce_parts_f32.  csrc/train_attn_kernels.hip. */
int pcr_tnorm_fwd_f32(const float *x, const float *gamma, const float *beta, const float *res, float *y, float *mean,
                      float *rstd, int B, int C, int L, int G, float eps, int relu, pcr_stream_t stream);
/* y_relu: the forward output when relu was set (the gradient is masked by y > 0 first), else NULL; dres (optional): the
 * masked gradient = gradient of the residual input */
int pcr_tnorm_bwd_f32(const float *g, const float *x, const float *gamma, const float *mean, const float *rstd,
                      const float *y_relu, float *dx, float *dres, float *part, int B, int C, int L, int G,
                      pcr_stream_t stream);

/* Linear-attention core (LinearAttention.forward, models/pointnet2_utils.py:26-47), forward and backward, one workgroup
 * per (cloud, head): Q' = elu(q)+1, K' = elu(k)+1, V' = v/Sk, A = sum_s K'_s V'_s^T, ks = sum_s K'_s,
 * out_l = (Q'_l^T A) Sk / (Q'_l . ks + eps).  q / k / v (and dq / dk / dv) are (d, L) channel-major blocks per cloud
 * addressed with a batch stride, so slices of a fused (B,3d,L) projection need no copies.  d / H in {16, 32, 64}.
 * A (B,H,dh,dh) and ks (B,H,dh) are written by the forward and read by the backward. */
typedef struct pcr_linattn {
  int B, Lq, Sk, d, H;
  float eps;
  const float *q, *k, *v;
  long q_bs, k_bs, v_bs;
  float *out, *A, *ks;
  const float *dout;
  float *dq, *dk, *dv;
  long dq_bs, dk_bs, dv_bs;
  int kv_roll;                /* (ABI 13, appended) query cloud b reads the keys / values of cloud (b + kv_roll) % B and writes
                               * their gradients there: the matching stage's "halves swapped" without a rolled copy */
} pcr_linattn;
int pcr_linattn_fwd_f32(const pcr_linattn *p, pcr_stream_t stream);
int pcr_linattn_bwd_f32(const pcr_linattn *p, pcr_stream_t stream);

/* Fused per-token chain: the TAIL of an attention block in training mode (round 5, csrc/train_chain_kernels.hip) --
 * Self_Attention (models/pointnet2_utils.py:105-114), FP_SA (:428-437), corss_attention (models/attention.py:210-219):
 *   out = LN2(W2 relu(W0 [res ; LN1(Wm msg)])) [+ res]        msg (B,d,L) = the attention core's output, res (B,c1,L)
 * as ONE forward and ONE backward launch (the unfused graph: merge, norm, two dense layers, norm = 5 + 7 launches and
 * five partial-sum reductions).  A 64-token tile walks the chain inside LDS; the backward recomputes the chain for its
 * tile (bit-identically: same code) and keeps nothing but msg and res.  wm / w0 / w2 are pcr_pack_weight images of merge
 * (d,d), mlp[0] (hid, c1+d), mlp[2] (out,hid); w*T of their transposes; g1 b1 (d), g2 b2 (out) the LayerNorm affines;
 * residual needs out == c1.  Backward outputs: dmsg (B,d,L), dres (B,c1,L) and per-workgroup partial records
 * [pcr_attn_tail_groups][part_stride >= pcr_attn_tail_part_floats] =
 *   dWm [d][d] | dW0 [hid][ceil32(c1+d)] | dW2 [out][hid] | dgamma1 [d] | dbeta1 [d] | dgamma2 [out] | dbeta2 [out]
 * to be summed by pcr_reduce_parts_f32 (fixed order: gradients are bit-identical from run to run).
 * pcr_attn_tail_ok: is (d, c1, hid, out) one of the instantiated shapes (the reference's mul = 1 blocks with d <= 64)? */
typedef struct pcr_attn_tail {
  int B, L, d, c1, hid, out, residual;
  float eps;
  const float *msg, *res;
  const float *wm, *w0, *w2, *wmT, *w0T, *w2T;
  const float *g1, *b1, *g2, *b2;
  float *outp;                /* forward */
  const float *dout;          /* backward */
  float *dmsg, *dres, *parts;
  long part_stride;
  int precision;              /* (ABI 14, appended) arithmetic of the BACKWARD's matrix phases (dx, dW): PCR_PREC_F32 =
                               * f32-input MFMAs, wmT / w0T / w2T pcr_pack_weight images; PCR_PREC_BF16X3 = split bf16 on the
                               * bf16 matrix core, wmT / w0T / w2T pcr_pack_weight_bf16x2_f32 images (pcr_pack_weights_multi_f32
                               * kind 1), the f32 tile in LDS converted where it is consumed.  The forward chain and its
                               * recomputation keep the unfused launches' f32 fmaf chains: same ReLU masks, same values */
  int fwd_precision;          /* (ABI 14) PCR_PREC_BF16X3: the forward chain (and the recomputation, and then the backward
                               * whatever `precision` says) as split bf16 too, wm / w0 / w2 bf16 images -- an opt-in: a
                               * ReLU whose argument is within ~1e-5 of zero may open where the f32 graph keeps it shut */
} pcr_attn_tail;
int pcr_attn_tail_ok(int d, int c1, int hid, int out, int residual);
int pcr_attn_tail_part_floats(int d, int c1, int hid, int out);
int pcr_attn_tail_groups(const pcr_attn_tail *p);
int pcr_attn_tail_fwd_f32(const pcr_attn_tail *p, pcr_stream_t stream);
int pcr_attn_tail_bwd_f32(const pcr_attn_tail *p, pcr_stream_t stream);

/* Fused per-token chain: the HEAD of an attention block in training mode (csrc/train_chain_kernels.hip) -- position MLP,
 * residual add and the projections that read the result (models/pointnet2_utils.py:92-100 Self_Attention, :410-421 FP_SA;
 * models/attention.py:195-205 corss_attention):
 *   fp = x + P2 relu(P1 xyz + c1) + c2          x (B,c,L), xyz (B,3,L) channel-major, P1 (hd,3), P2 (c,hd)
 *   out (B, np d, L) = [W_0 s_0 ; ... ; W_{np-1} s_{np-1}],  s_j = fp if bit j of src is set, else x;  W_j (d,c)
 * (self block: q | k | v of fp, src = 7; cross / FP block: k of x | v of fp, src = 2) as one launch each way; the backward
 * recomputes the chain for its tile.  p1 / p2 / w[j] are pcr_pack_weight images, p2T / wT[j] of the transposes, c1 / c2
 * zero-padded to a multiple of 32 floats.  Backward outputs: dx (B,c,L) and per-workgroup partial records
 * [pcr_attn_head_groups][part_stride >= pcr_attn_head_part_floats] =
 *   dP1 [hd][32] | dP2 [c][hd] | dW_0 [d][c] | .. | dc1 [hd] | dc2 [c]        for pcr_reduce_parts_f32. */
typedef struct pcr_attn_head {
  int B, L, c, hd, d, np, src;
  const float *x, *xyz;
  const float *p1, *p2, *c1, *c2, *p2T;
  const float *w[3], *wT[3];
  float *outp;                /* forward */
  const float *dout;          /* backward */
  float *dx, *parts;
  long part_stride;
  int precision;              /* (ABI 14, appended) as pcr_attn_tail.precision: p2T, wT[] bf16 hi / lo images when
                               * PCR_PREC_BF16X3 */
  int fwd_precision;          /* as pcr_attn_tail.fwd_precision: p2, w[] bf16 images; p1 (three input channels) stays an f32
                               * image either way */
} pcr_attn_head;
int pcr_attn_head_ok(int c, int hd, int d, int np, int src);
int pcr_attn_head_part_floats(int c, int hd, int d, int np, int src);
int pcr_attn_head_groups(const pcr_attn_head *p);
int pcr_attn_head_fwd_f32(const pcr_attn_head *p, pcr_stream_t stream);
int pcr_attn_head_bwd_f32(const pcr_attn_head *p, pcr_stream_t stream);

/* Match-head pooling in training (get_pooled_feats 'both' over the point-concatenated pair, models/ReIDNet.py:529-532):
 * o (2P,C,L) -> pooled (P,2C) = [max, mean] over the 2L points of pair p (clouds p, p+P), arg (P,C) = position of the
 * maximum; backward: dout (2P,C,L) from g (P,2C). */
int pcr_pool_pair_fwd_f32(const float *o, float *pooled, int *arg, int P, int C, int L, pcr_stream_t stream);
int pcr_pool_pair_bwd_f32(const float *g, const int *arg, float *dout, int P, int C, int L, pcr_stream_t stream);
/* The same pooling for ONE tensor (match types that pool a single branch: `xcorr-baseline`, models/ReIDNet.py:258-264
 * with get_pooled_feats 'both' :529-532): o (P,C,L) -> pooled (P,2C) = [max over L, mean over L], arg (P,C); and its
 * backward dout (P,C,L). */
int pcr_pool_both_fwd_f32(const float *o, float *pooled, int *arg, int P, int C, int L, pcr_stream_t stream);
int pcr_pool_both_bwd_f32(const float *g, const int *arg, float *dout, int P, int C, int L, pcr_stream_t stream);
/* get_pooled_feats 'max' (models/ReIDNet.py:145, 526-528: nn.MaxPool1d(W) on the permuted (B,L,C) tensor) with the
 * winning channel kept for the backward: x (B,C,L) -> y (B,C/W,L), arg (B,C/W,L); backward dx (B,C,L). */
int pcr_channel_max_fwd_f32(const float *x, float *y, int *arg, int B, int C, int L, int W, pcr_stream_t stream);
int pcr_channel_max_bwd_f32(const float *g, const int *arg, float *dx, int B, int C, int L, int W, pcr_stream_t stream);

/* BatchNorm in batch-statistics mode as a stand-alone layer on (B,C,L) channel-major tensors (PointNet's BatchNorm1d
 * layers, models/pointnet.py:27-45, 103-127; csrc/train_bn_kernels.hip).
 * pcr_bn_sums_f32: partials [nparts][2][ceil32(C)] for pcr_bn_fwd_finalize_f32 (g NULL: sum y', sum y'^2) or
 *   pcr_bn_bwd_finalize_f32 (g given: sum g', sum g' y' with g' = g [scale y + shift > 0] when relu, else g), where
 *   y' = y - centre[c] (centre given), y - y[0][c][0] (centre NULL, centre_first != 0) or y.
 * pcr_bn_affine_f32: g NULL: out = [relu](a0 y + a1) (a0 = scale, a1 = shift); g given: out = a0 g' + a1 (y - centre[c])
 *   + a2, the gradient with respect to y: (ka, kb, kc) with centre NULL, or its centred form (ka, kb, -ka dbeta / R) with
 *   centre = the batch mean. */
int pcr_bn_sums_f32(const float *y, const float *g, const float *scale, const float *shift, int relu, float slope,
                    const float *centre, int centre_first, float *part, int nparts, int B, int C, int L,
                    pcr_stream_t stream);
int pcr_bn_affine_f32(const float *y, const float *g, const float *a0, const float *a1, const float *a2,
                      const float *scale, const float *shift, const float *centre, int relu, float slope, float *out,
                      int B, int C, int L, pcr_stream_t stream);
/* (relu != 0: the activation after the norm is z > 0 ? z : slope z -- ReLU with slope 0, LeakyReLU(0.2) with 0.2.)
 * EdgeConv tail in training mode (models/dgcnn_orig.py:127-143) on the materialised pre-activation y (B,C,N K):
 * pooled (B,C,N) = act(max_k (scale y + shift)), arg = the first k attaining it, yraw = y there; and the pooled gradient
 * routed back to its edge: g (B,C,N K) = gp act'(pooled) at k == arg, 0 elsewhere. */
int pcr_edge_pool_fwd_f32(const float *y, const float *scale, const float *shift, float slope, float *pooled, int *arg,
                          float *yraw, int B, int C, int N, int K, pcr_stream_t stream);
int pcr_edge_pool_route_f32(const float *gp, const float *pooled, const int *arg, float slope, float *g, int B, int C,
                            int N, int K, pcr_stream_t stream);
/* Per-cloud transforms (torch.bmm(x^T, T)^T, models/pointnet.py:109-111, 117-119): x (B,k,N), T (B,k,k) ->
 * y[b][j][n] = sum_i T[b][i][j] x[b][i][n]; transposed = 1 applies T^T instead (the backward's dx from dy);
 * pcr_bmm_dt_f32: dT[b][i][j] = sum_n x[b][i][n] dy[b][j][n].  k <= 128. */
int pcr_bmm_apply_f32(const float *x, const float *T, float *y, int B, int k, int N, int transposed, pcr_stream_t stream);
int pcr_bmm_dt_f32(const float *x, const float *dy, float *dT, int B, int k, int N, pcr_stream_t stream);

/* The packed images of MANY weights in one launch (a training step re-packs every weight after the update: ~76 small
 * launches otherwise).  descs (device): per tensor the contiguous row-major (rows x cols) matrix w and out, which
 * receives the image of W (pcr_packed... ceil8(cols) * ceil32(rows) floats) followed by the image of W^T
 * (ceil8(rows) * ceil32(cols) floats), as pcr_pack_weight_dev_f32(transpose = 2) lays them out.
 * cols == 0 marks a BIAS entry: w holds `rows` floats that are copied to the head of out, the caller's zero-padded
 * ceil32(rows) image (what the train-dense launches seed their accumulators from). */
typedef struct pcr_pack_desc {
  const float *w;
  float *out;
  int rows, cols;
  int kind;                   /* (ABI 14) 0: the f32 images above (or a bias); 1: the bf16 hi / lo images instead --
                               * pcr_pack_weight_bf16_dev_f32's image of W (pcr_packed_weight_bf16_floats(rows, cols)
                               * floats) followed by that of W^T (pcr_packed_weight_bf16_floats(cols, rows) floats) */
  int reserved;
} pcr_pack_desc;
int pcr_pack_weights_multi_f32(const pcr_pack_desc *descs_dev, int n, pcr_stream_t stream);

/* Parameter update of one iteration over EVERY parameter tensor in two launches: the global gradient norm of mmcv's
 * OptimizerHook(grad_clip=dict(max_norm, norm_type=2)) = torch.nn.utils.clip_grad_norm_, then torch.optim.AdamW's
 * step (configs_reid/_base_/schedules/cyclic_200e_lr3e-4.py:7-9; the cyclic lr / beta1 of lines 10-21 arrive as
 * per-tensor constants).  `tab` (device) lists the tensors; the launch is cut into chunks of pcr_opt_chunk() elements,
 * chunk i = elements [chunk_first[i], +chunk) of tensor chunk_tensor[i] (both device arrays).  A tensor whose g is
 * NULL is skipped (no gradient this iteration).  Per element, with g' = g * coef:
 *   p <- p * decay;  m <- m + one_m_beta1 (g' - m);  v <- beta2 v + one_m_beta2 g'^2;
 *   p <- p - step_size * m / (sqrt(v) / bc2_sqrt + eps)
 * where the host sets (in double, then rounds) decay = 1 - lr * weight_decay, step_size = lr / (1 - beta1^step),
 * bc2_sqrt = sqrt(1 - beta2^step), one_m_beta = 1 - beta.
 * pcr_grad_sumsq_f32 writes one sum of squares per chunk into part (n_chunks doubles); pcr_adamw_step_f32 with
 * part != NULL adds them in a fixed order, writes the norm to *grad_norm (optional) and, when max_norm > 0, uses
 * coef = min(1, max_norm / (norm + 1e-6)) and stores g' back into g; with part == NULL coef = 1.  No atomics: the
 * update is reproducible bit for bit. */
typedef struct pcr_opt_tensor {
  float *p, *g, *m, *v;
  long n;
  float step_size, bc2_sqrt, decay, one_m_beta1, beta2, one_m_beta2, eps, pad_;
} pcr_opt_tensor;
int pcr_opt_chunk(void);
int pcr_grad_sumsq_f32(const pcr_opt_tensor *tab, const int *chunk_tensor, const int *chunk_first, int n_chunks,
                       double *part, pcr_stream_t stream);
int pcr_adamw_step_f32(const pcr_opt_tensor *tab, const int *chunk_tensor, const int *chunk_first, int n_chunks,
                       const double *part, float max_norm, float *grad_norm, pcr_stream_t stream);

/* ---- measurement aid (bench.py; not on the hot path) ---- */

/* Sustained f32-MFMA probe: n_wg workgroups (one per CU: pass the CU count) each run iters * 16
 * v_mfma_f32_32x32x2_f32 per wave (= iters * 1024 matrix-pipe cycles) and write ticks[2 wg] = shader-clock ticks,
 * ticks[2 wg + 1] = constant-rate wall-clock ticks (pcr_wall_clock_khz) spent on them.  The clock the matrix core
 * ran at is iters * 1024 / (wall ticks / rate): what `roofline.clock_ghz` in the bench line reports. */
int pcr_wall_clock_khz(void);
/* PCR_PREC_* the matrix phases of the calling thread's LAST pcr_sa_mlp / pcr_dense_pm / pcr_attn_kv / pcr_attn_apply
 * launch really ran in (-1: none yet).  A requested precision is a request: shapes a bf16 unit does not instantiate run
 * f32, the tile kv kernel projects in f32 in every mode.  bench.py prices a launch against the peak of THIS arithmetic. */
int pcr_last_launch_arith(void);
int pcr_clock_probe(unsigned long long *ticks, int n_wg, int iters, pcr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCR_H_ */
