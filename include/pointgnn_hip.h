/* pointgnn_hip.h -- C ABI of libpointgnn_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the hot path of WeijingShi/Point-GNN.  The reference has
 * no FFI layer (everything is Python calling TensorFlow 1.15 / scikit-learn);
 * each entry point below names the reference interface it replaces
 * (file:line relative to the reference tree).  Host code (Python, ctypes) keeps
 * the reference's operator names and only moves pointers: see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter says `host`;
 *   - the caller owns every buffer; the library never frees or retains a
 *     pointer past the call; scratch comes from a caller-provided workspace
 *     sized by the matching *_workspace_bytes() query;
 *   - every entry takes a hipStream_t (as void*) and is asynchronous with
 *     respect to the host: no hidden synchronisation.  The only process-wide
 *     mutable state is the set of diagnostic knobs behind pgnn_set_tunable /
 *     pgnn_set_debug_buffer (end of this header): launch-shape and ablation
 *     switches for benchmarks and tests whose defaults never need changing.
 *     They are plain words read once per call; change them only while no
 *     other thread is inside the library.  Everything else is re-entrant
 *     across streams and threads;
 *   - return value: 0 = ok, negative = argument error (PGNN_E_*), positive =
 *     hipError_t of the failing runtime call.  pgnn_last_error() returns a
 *     thread-local message for the last non-zero return.  No C++ exception
 *     crosses the ABI;
 *   - indices are int32 on the wire (run.py:126-133, train.py:126-128), all
 *     floating point is fp32 unless stated;
 *   - must not be called in a forked child after HIP was initialised in the
 *     parent (the reference forks graph workers, train.py:430).
 */
#ifndef POINTGNN_HIP_H_
#define POINTGNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGNN_E_INVALID (-1)   /* bad argument (null pointer, negative size...) */
#define PGNN_E_WORKSPACE (-2) /* workspace too small                          */
#define PGNN_E_UNSUPPORTED (-3) /* shape outside the implemented range        */
#define PGNN_E_CAPACITY (-4)  /* output capacity too small                    */
#define PGNN_E_COMM (-5)      /* RCCL missing / a collective call failed       */

/* ---- library ---------------------------------------------------------- */
int pgnn_version(void);
const char *pgnn_last_error(void);
/* Sanity probe used by the Python loader: 0 iff `device_ptr` is a device
 * allocation visible to the HIP runtime this library is bound to. */
int pgnn_check_device_pointer(const void *device_ptr);

/* A stream whose kernels run on CUs [cu_first, cu_first + cu_count) only, or
 * (complement != 0) on all the others.  The frame pipeline builds the next
 * frame's graph on a few reserved CUs while the fused MFMA kernels of the
 * current frame own the rest: they fill a CU completely (VGPRs and LDS), so
 * without the split a kernel of another stream waits for their kernel
 * boundaries and then displaces one of their persistent workgroups.  Mask bits
 * are striped over the XCDs (bit i -> XCD i % 8): a contiguous range takes the
 * same number of CUs from every XCD.  *stream_out is a hipStream_t; destroy it
 * with pgnn_stream_destroy.  Every other entry takes such a stream like any
 * other. */
int pgnn_stream_create_cu_mask(int32_t cu_first, int32_t cu_count,
                               int32_t complement, void **stream_out);
int pgnn_stream_destroy(void *stream);

/* ---- scatter-max -------------------------------------------------------
 * Replaces graph_scatter_max_fn = tf.math.unsorted_segment_max
 * (models/gnn.py:106-109; call sites gnn.py:275-277, 362-365).
 * out[s, c] = max over rows r with seg_ids[r] == s of data[r, c]; a segment
 * with no rows gets the lowest finite float (TF semantics).  Ids outside
 * [0, num_segments) are ignored.  `ids_sorted` != 0 promises that seg_ids is
 * non-decreasing (what the radius-graph builder emits): whole-segment runs
 * are then written with plain stores instead of atomics.  Any order is
 * accepted when ids_sorted == 0.  ld_* are row strides in floats.
 * Algorithmic bytes: n_rows*n_cols*4 + n_rows*4 + num_segments*n_cols*4. */
int pgnn_scatter_max_f32(const float *data, int64_t ld_data,
                         const int32_t *seg_ids, int64_t n_rows,
                         int32_t n_cols, int32_t num_segments, float *out,
                         int64_t ld_out, int32_t ids_sorted, void *stream);

/* graph_scatter_sum_fn / graph_scatter_mean_fn = tf.math.unsorted_segment_sum /
 * unsorted_segment_mean (models/gnn.py:111-119; registered by the reference,
 * used by no shipped config).  Same arguments as pgnn_scatter_max_f32; `mean`
 * != 0 divides every segment by max(count, 1) (empty segments give 0, as in
 * TF) and needs `counts_ws`, int32[num_segments] of scratch.  Float atomic
 * adds: the summation order is unspecified, as on TensorFlow's GPU kernel. */
int pgnn_scatter_sum_f32(const float *data, int64_t ld_data,
                         const int32_t *seg_ids, int64_t n_rows, int32_t n_cols,
                         int32_t num_segments, float *out, int64_t ld_out,
                         int32_t mean, int32_t *counts_ws, void *stream);

/* ---- radius graph ------------------------------------------------------
 * Replaces gen_disjointed_rnn_local_graph_v3 (models/graph_gen.py:197-220):
 * for every centre, all points with float64 squared distance <= radius^2
 * (inclusive; scikit-learn ball-tree semantics).  Two phases, no hidden
 * sync:  count -> caller reads offsets[n_centers] and allocates -> fill.
 * `scale3` (host pointer to 3 doubles, may be NULL) is the reference's
 * optional `scale` pre-division (graph_gen.py:203-206).
 * Output rows are (point_idx, centre_idx), grouped by ascending centre.  */
size_t pgnn_radius_graph_workspace_bytes(int64_t n_points, int64_t n_centers);
int pgnn_radius_graph_count(const float *points, int64_t n_points,
                            const float *centers, int64_t n_centers,
                            double radius, const double *scale3_host,
                            void *workspace, size_t workspace_bytes,
                            int32_t *offsets /* [n_centers + 1] */,
                            void *stream);
int pgnn_radius_graph_fill(const float *points, int64_t n_points,
                           const float *centers, int64_t n_centers,
                           double radius, const double *scale3_host,
                           void *workspace, size_t workspace_bytes,
                           const int32_t *offsets /* from _count */,
                           int32_t *edges /* [capacity, 2] */,
                           int64_t capacity, void *stream);
/* The same two phases on FLOAT64 points / centres: the training data path
 * keeps the augmented cloud float64 through graph generation (train.py:88-90
 * -> graph_gen.py:187-194; the cast to float32 is train.py:124), and sklearn
 * searches float64 data as it is.  Same workspace query, same outputs. */
int pgnn_radius_graph_count_f64(const double *points, int64_t n_points,
                                const double *centers, int64_t n_centers,
                                double radius, const double *scale3_host,
                                void *workspace, size_t workspace_bytes,
                                int32_t *offsets /* [n_centers + 1] */,
                                void *stream);
int pgnn_radius_graph_fill_f64(const double *points, int64_t n_points,
                               const double *centers, int64_t n_centers,
                               double radius, const double *scale3_host,
                               void *workspace, size_t workspace_bytes,
                               const int32_t *offsets /* from _count */,
                               int32_t *edges /* [capacity, 2] */,
                               int64_t capacity, void *stream);
/* Capacity form of the same builder -- count, scan and fill in ONE call with
 * no size leaving the device (graph_gen.py:155-220 never waits for one: its
 * NumPy arrays carry their sizes; a device pipeline that reads K or E back
 * stalls the host once per level).  `points_cap` / `centers_cap` size the
 * launches and the workspace; when `n_points_dev` / `n_centers_dev` (device
 * int32, nullable) are given, min(*n_dev, cap) rows are valid -- this is how a
 * level takes the keypoints of pgnn_voxel_keypoints_* (whose count is
 * num_keypoints[0] on the device) as its points and centres.  Rows are written
 * up to `edge_capacity`; n_edges_dev[0] = rows written = min(E, capacity),
 * n_edges_dev[1] = E, the size required: the caller compares the two when it
 * reads the frame's results and re-runs with a larger buffer if they differ.
 * Same rows, same order as _count + _fill. */
size_t pgnn_radius_graph_dyn_workspace_bytes(int64_t points_cap,
                                             int64_t centers_cap);
int pgnn_radius_graph_dyn(const float *points, int64_t points_cap,
                          const int32_t *n_points_dev, const float *centers,
                          int64_t centers_cap, const int32_t *n_centers_dev,
                          double radius, const double *scale3_host,
                          void *workspace, size_t workspace_bytes,
                          int32_t *edges /* [edge_capacity, 2] */,
                          int64_t edge_capacity,
                          int32_t *n_edges_dev /* [2], device */, void *stream);
int pgnn_radius_graph_dyn_f64(const double *points, int64_t points_cap,
                              const int32_t *n_points_dev,
                              const double *centers, int64_t centers_cap,
                              const int32_t *n_centers_dev, double radius,
                              const double *scale3_host, void *workspace,
                              size_t workspace_bytes, int32_t *edges,
                              int64_t edge_capacity, int32_t *n_edges_dev,
                              void *stream);
/* The same builder in its two stages, for a caller that overlaps them with
 * other work: _grid needs the points only (cell keys, radix sort, bucket
 * bounds), _query the centres as well (count, scan, fill).  A frame's level-0
 * graph has the raw cloud as its points and the keypoints as its centres, so
 * its grid stage can run on a second stream BESIDE the keypoint selection and
 * only the query stage has to wait for it (the caller orders the two with its
 * own stream / event calls; both stages take the SAME workspace, capacities,
 * radius and scale).  _grid then _query on one stream is pgnn_radius_graph_dyn
 * exactly. */
int pgnn_radius_graph_dyn_grid(const float *points, int64_t points_cap,
                               const int32_t *n_points_dev, int64_t centers_cap,
                               double radius, const double *scale3_host,
                               void *workspace, size_t workspace_bytes,
                               void *stream);
int pgnn_radius_graph_dyn_grid_f64(const double *points, int64_t points_cap,
                                   const int32_t *n_points_dev,
                                   int64_t centers_cap, double radius,
                                   const double *scale3_host, void *workspace,
                                   size_t workspace_bytes, void *stream);
int pgnn_radius_graph_dyn_query(const float *points, int64_t points_cap,
                                const float *centers, int64_t centers_cap,
                                const int32_t *n_centers_dev, double radius,
                                const double *scale3_host, void *workspace,
                                size_t workspace_bytes, int32_t *edges,
                                int64_t edge_capacity, int32_t *n_edges_dev,
                                void *stream);
int pgnn_radius_graph_dyn_query_f64(const double *points, int64_t points_cap,
                                    const double *centers, int64_t centers_cap,
                                    const int32_t *n_centers_dev, double radius,
                                    const double *scale3_host, void *workspace,
                                    size_t workspace_bytes, int32_t *edges,
                                    int64_t edge_capacity, int32_t *n_edges_dev,
                                    void *stream);
/* Training-time fan-in cap (graph_gen.py:210-214, num_neighbors > 0): keeps a
 * uniformly random subset (without replacement) of `max_neighbors` edges for
 * every centre whose fan-in exceeds it (counter-based RNG keyed by `seed`,
 * centre and position).  Two phases like the builder: _count writes
 * new_offsets[n_centers + 1] from the uncapped CSR offsets; the caller reads
 * the total and allocates; _fill writes the surviving rows, same grouping. */
int pgnn_cap_neighbors_count(const int32_t *offsets, int64_t n_centers,
                             int32_t max_neighbors, int32_t *new_offsets,
                             void *stream);
int pgnn_cap_neighbors_fill(const int32_t *offsets, const int32_t *edges,
                            int64_t n_centers, int32_t max_neighbors,
                            uint64_t seed, const int32_t *new_offsets,
                            int32_t *new_edges, int64_t new_capacity,
                            void *stream);
/* The cap in capacity form (no host read), for a level built by
 * pgnn_radius_graph_dyn(_query)(_f64): takes that call's workspace (the CSR
 * offsets it left there; same points_cap / centers_cap), its edge rows and its
 * record n_edges_dev[2]; writes new_offsets[centers_cap + 1], the surviving
 * rows (same rows, same order as _count + _fill with this seed) up to
 * `new_capacity`, and the capped list's record n_new_dev[2] = {rows written,
 * rows required}.  If the UNCAPPED list overflowed its own capacity the capped
 * record reads {0, required}: the level is flagged the usual way (required >
 * written) and the caller rebuilds it. */
int pgnn_radius_graph_dyn_cap(const void *workspace, size_t workspace_bytes,
                              int64_t points_cap, int64_t centers_cap,
                              const int32_t *edges, int64_t edge_capacity,
                              const int32_t *n_edges_dev /* [2], device */,
                              int32_t max_neighbors, uint64_t seed,
                              int32_t *new_offsets, int32_t *new_edges,
                              int64_t new_capacity,
                              int32_t *n_new_dev /* [2], device */,
                              void *stream);

/* ---- keypoints ----------------------------------------------------------
 * 'center' mode = multi_layer_downsampling_select (graph_gen.py:49-90) for ONE
 * pooling level per call (the reference loops over arbitrary `levels`; every
 * shipped config has one downsampling level followed by equal scales, which
 * graph_gen.py:76-81 turns into copies.  These entries serve the FIRST
 * pooling level -- the search set is the voxelised cloud itself; the levels
 * above it take pgnn_voxel_keypoints_*_from below, and the Python mirror
 * pointgnn_amd.graph_gen loops over the levels like the reference):
 * open3d-0.7 voxel centroids (origin = min_bound - voxel/2,
 * float64 means in point order) followed by an exact float64 1-NN back to a
 * real point; exact distance ties are broken like scikit-learn's kd-tree
 * query does (see pgnn_kdtree_replica).  Keypoints are emitted in ascending
 * voxel-hash order.
 * 'random' mode = multi_layer_downsampling_random (graph_gen.py:92-153): one
 * uniformly chosen point per occupied voxel of the grid anchored at the
 * cloud minimum (+ `jitter3_host`, the reference's add_rnd3d origin shift,
 * in units of metres; NULL = none), RNG keyed by `seed`.
 * Capacity of both outputs is n_points rows.  num_keypoints: device int32[2],
 * [0] = K, [1] = tie-order status of the kd-tree replica ('center' only; 0 =
 * the reference's order -- libstdc++'s introselect INCLUDING its heap-select
 * fallback is replayed; non-zero = the cloud is outside what the replica
 * reproduces, exact 1-NN ties may then be broken differently from the
 * reference; the Python mirror raises on it).
 * Limits of 'center' mode: float32 points only (the replica's keys), at most
 * 524 288 points (PGNN_E_UNSUPPORTED beyond: node records are kept in LDS).
 * The tie rule replicated is scikit-learn >= 1.0's (std::nth_element in
 * sklearn/neighbors/_partition_nodes.pyx); the reference pins no version and a
 * 2020-era 0.22 partitions differently.                                     */
/* Replica of scikit-learn's KDTree(points, leaf_size=30) node order -- what
 * decides which of several exactly equidistant points
 * NearestNeighbors(algorithm='kd_tree').kneighbors returns (graph_gen.py:84-88)
 * and therefore the reference's keypoint in every 2-point voxel.
 * pgnn_voxel_keypoints_center builds it internally; this entry exposes the
 * arrays (device pointers) for tests against KDTree.get_arrays():
 *   idx_array   [n_points]      int32   (sklearn: intp)
 *   node_bounds [n_nodes][6]    float64 (lo x,y,z, hi x,y,z; sklearn stores
 *                                        [2][n_nodes][3])
 *   status      [1]             int32   0, or 1 if std::nth_element's
 *                                        heap-select fallback would have been
 *                                        taken (not replicated)
 * pgnn_kdtree_shape returns sklearn's n_levels / n_nodes for n_points (host
 * pointers).                                                              */
int pgnn_kdtree_shape(int64_t n_points, int32_t *n_levels, int32_t *n_nodes);
size_t pgnn_kdtree_workspace_bytes(int64_t n_points);
int pgnn_kdtree_replica(const float *points, int64_t n_points, void *workspace,
                        size_t workspace_bytes, int32_t *idx_array,
                        double *node_bounds, int32_t *status, void *stream);

size_t pgnn_keypoints_workspace_bytes(int64_t n_points);  /* either precision */
/* aux_stream (nullable): a second caller-owned stream.  When given, the
 * kd-tree replica (a latency-bound chain of ~10 dependent launches that only
 * needs the points) is issued there, forked from and joined back into `stream`
 * with two events, so it overlaps the voxel hashing; NULL = everything in
 * order on `stream`.  The results are identical. */
int pgnn_voxel_keypoints_center(const float *points, int64_t n_points,
                                double voxel_size, void *workspace,
                                size_t workspace_bytes,
                                int32_t *keypoint_indices, float *keypoint_xyz,
                                int32_t *num_keypoints, void *stream,
                                void *aux_stream);
int pgnn_voxel_keypoints_random(const float *points, int64_t n_points,
                                double voxel_size, const double *jitter3_host,
                                uint64_t seed, void *workspace,
                                size_t workspace_bytes,
                                int32_t *keypoint_indices, float *keypoint_xyz,
                                int32_t *num_keypoints, void *stream);

/* 'random' mode on a FLOAT64 cloud (train.py:88-90): NumPy's float64 `//` on
 * (points - min) [+ jitter] (graph_gen.py:123-128); keypoint_xyz is float64
 * (the reference returns rows of its float64 array).  */
int pgnn_voxel_keypoints_random_f64(const double *points, int64_t n_points,
                                    double voxel_size,
                                    const double *jitter3_host, uint64_t seed,
                                    void *workspace, size_t workspace_bytes,
                                    int32_t *keypoint_indices,
                                    double *keypoint_xyz,
                                    int32_t *num_keypoints, void *stream);

/* The second and later POOLING levels of a frame (graph_gen.py:49-90 and
 * :92-153 loop over arbitrary `levels`; every shipped config has one pooling
 * level, which the entries above serve):
 *   'center' (graph_gen.py:41-45, :78-88): the centroids are still those of
 *     the ORIGINAL cloud's voxels (`points`, at this level's voxel size), but
 *     the 1-NN search runs among `search_points` = the previous level's
 *     keypoint coordinates, with scikit-learn's kd-tree order over THAT set
 *     deciding exact ties.  keypoint_indices index search_points;
 *     keypoint_xyz = search_points[keypoint_indices].  num_keypoints as above
 *     ([1] = the tie-order status of the kd-tree replica of search_points).
 *   'random' (graph_gen.py:108-110, :121-150): `points` = the previous level's
 *     keypoints are voxelised on the grid anchored at the minimum of
 *     `origin_points` = the ORIGINAL cloud (n_origin = 0: of `points`).
 * Workspace: pgnn_keypoints_from_workspace_bytes(n_points, n_search) for
 * 'center', pgnn_keypoints_workspace_bytes(n_points) for 'random'.            */
size_t pgnn_keypoints_from_workspace_bytes(int64_t n_points, int64_t n_search);
int pgnn_voxel_keypoints_center_from(const float *points, int64_t n_points,
                                     const float *search_points,
                                     int64_t n_search, double voxel_size,
                                     void *workspace, size_t workspace_bytes,
                                     int32_t *keypoint_indices,
                                     float *keypoint_xyz,
                                     int32_t *num_keypoints, void *stream);
int pgnn_voxel_keypoints_random_from(const float *points, int64_t n_points,
                                     const float *origin_points,
                                     int64_t n_origin, double voxel_size,
                                     const double *jitter3_host, uint64_t seed,
                                     void *workspace, size_t workspace_bytes,
                                     int32_t *keypoint_indices,
                                     float *keypoint_xyz,
                                     int32_t *num_keypoints, void *stream);
int pgnn_voxel_keypoints_random_from_f64(
    const double *points, int64_t n_points, const double *origin_points,
    int64_t n_origin, double voxel_size, const double *jitter3_host,
    uint64_t seed, void *workspace, size_t workspace_bytes,
    int32_t *keypoint_indices, double *keypoint_xyz, int32_t *num_keypoints,
    void *stream);

/* ---- dense layers --------------------------------------------------------
 * A fully connected layer y = act(x @ W + b) (slim.fully_connected with
 * normalizer 'NONE', models/gnn.py:86-104 and :34-84) is held on the device in
 * MFMA-fragment order: pgnn_pack_fc() converts the reference's [k_in, n_out]
 * row-major weights + bias (HOST pointers) into `packed` (HOST, then copied to
 * the device by the caller).  Layout: kq = ceil(k_in/16) K-groups, nt =
 * ceil(n_out/16) column tiles; packed[((q*nt + t)*64 + lane)*4 + s] =
 * W[16q + 4*(lane>>4) + s][16t + (lane&15)] (zero outside), followed by
 * bias[16*nt] (zero padded).                                               */
size_t pgnn_packed_fc_floats(int32_t k_in, int32_t n_out);
int pgnn_pack_fc(const float *w_host, const float *b_host, int32_t k_in,
                 int32_t n_out, float *packed_host);

typedef struct pgnn_fc_layer {
  const float *packed; /* device: pgnn_pack_fc image                         */
  int32_t k_in;        /* logical input width (<= 16*kq)                     */
  int32_t n_out;       /* logical output width                               */
  int32_t relu_from;   /* ReLU on output columns >= relu_from; 0 = every
                          column, >= n_out = linear layer (is_logits)        */
} pgnn_fc_layer;

#define PGNN_MAX_LAYERS 8

/* ---- capacity form of the per-frame operators ------------------------------
 * The reference's graph builder hands NumPy arrays to the model
 * (run.py:219-222 -> 245-260): sizes travel with the arrays and nobody waits
 * for them.  On the device the keypoint count K and the edge counts exist
 * only after the builder's kernels ran; reading them back to size the next
 * launch stalls the host twice per frame.  The *_dyn entries below are the
 * operators of this section with every such count taken from DEVICE memory:
 * the ordinary size argument becomes the capacity (it sizes buffers and
 * grids), `dev` points at the int32 count (min(*dev, capacity) rows are
 * processed; the rows behind it are neither read nor written), and `hint` is
 * the count the host expects (e.g. the previous frame's): it only chooses
 * between kernels that give bit-identical results (weights-stationary vs
 * LDS-tile, 8-wave vs 4-wave rows kernel) and sizes strided grids, so a wrong
 * hint costs time, never correctness.  hint <= 0 means the capacity. */
typedef struct pgnn_dyn_count {
  const int32_t *dev; /* device int32: the actual count                     */
  int64_t hint;       /* expected value (host), <= 0: unknown               */
} pgnn_dyn_count;

/* Per-row MLP chain: y = MLP(concat(x[:, :nx], x2[:, :nx2])) (+ residual).
 * Covers multi_layer_neural_network_fn / multi_layer_fc_fn on per-vertex
 * inputs: PointSetPooling's output MLP (gnn.py:279-282), the auto-offset MLP
 * (gnn.py:341-346), the update MLP + residual (gnn.py:367-372) and, with a
 * block-diagonal packing, ClassAwarePredictor (gnn.py:133-163).  x2 / residual
 * may be NULL.  y receives 16*ceil(n_out_last/16) columns per row (zero
 * padded), so ld_y must be >= that.                                         */
int pgnn_mlp_fwd(const float *x, int64_t ld_x, int32_t nx, const float *x2,
                 int64_t ld_x2, int32_t nx2, int64_t n_rows,
                 const pgnn_fc_layer *layers_host, int32_t n_layers,
                 const float *residual, int64_t ld_res, float *y, int64_t ld_y,
                 void *stream);
int pgnn_mlp_fwd_dyn(const float *x, int64_t ld_x, int32_t nx, const float *x2,
                     int64_t ld_x2, int32_t nx2, int64_t rows_cap,
                     const pgnn_fc_layer *layers_host, int32_t n_layers,
                     const float *residual, int64_t ld_res, float *y,
                     int64_t ld_y, const pgnn_dyn_count *n_rows, void *stream);

/* Fused PointSetPooling front half (gnn.py:256-277): for every edge
 * (point, keypoint) build [features(point), xyz(point) - xyz(keypoint)], run
 * the point MLP (every layer ReLU) and scatter-max into the keypoint row.
 * Nothing of size E x C touches HBM.  `edges` must be grouped by ascending
 * keypoint when edges_sorted != 0.  out: [num_keypoints, ld_out], columns
 * 16*ceil(n_out/16) written (zero padded); keypoints without edges get
 * float lowest (TF unsorted_segment_max).                                   */
/* sched_ws (both fused entries; nullable): PGNN_SCHED_WS_INTS int32 on the
 * device, zero before the first use; the kernels leave them zero.  When given,
 * the last part of the row tiles (~12-15 %) is not part of the workgroups' /
 * waves' fixed ranges but a pool they take a tile or a small chunk at a time
 * through atomic counters when their own range is done: slack that absorbs
 * uneven ranges and a late start of some workgroups (kernels of other streams
 * occupying CUs when the grid starts).  Launches that may run concurrently
 * (different streams) need different sched_ws.  Results do not depend on it
 * (max is exact). */
#define PGNN_SCHED_WS_INTS 64
int pgnn_point_set_pooling_fwd(const float *point_features, int32_t n_feat,
                               const float *point_xyz,
                               const int32_t *keypoint_indices,
                               const int32_t *edges, int64_t n_edges,
                               int32_t num_keypoints,
                               const pgnn_fc_layer *layers_host,
                               int32_t n_layers, int32_t edges_sorted,
                               float *out, int64_t ld_out, int32_t *sched_ws,
                               void *stream);
int pgnn_point_set_pooling_fwd_dyn(const float *point_features, int32_t n_feat,
                                   const float *point_xyz,
                                   const int32_t *keypoint_indices,
                                   const int32_t *edges, int64_t edges_cap,
                                   int32_t keypoints_cap,
                                   const pgnn_fc_layer *layers_host,
                                   int32_t n_layers, int32_t edges_sorted,
                                   float *out, int64_t ld_out,
                                   int32_t *sched_ws,
                                   const pgnn_dyn_count *n_edges,
                                   const pgnn_dyn_count *num_keypoints,
                                   void *stream);

/* The same with a caller-provided workspace (gnn.py:256-277; the reference
 * materialises every layer's [E, C] tensor, this entry at most one of them).
 * Point MLPs whose last layer does not fit one CU's LDS -- ped_cyl's
 * 4-32-64-128-256-512 (configs/ped_cyl_auto_T3_trainval_config) -- run as TWO
 * launches: the chain up to the 256-wide hidden layer, its rows written to
 * `workspace` ([edges_cap, 256] floats), then the last layer weights-stationary
 * in four column groups with the segmented max (csrc/pool_split.h): 0.85 of
 * the fp32-MFMA peak against the one-launch LDS-tile kernel's 0.67, the maxima
 * bit-identical.  pgnn_point_set_pooling_workspace_bytes says how much
 * workspace a chain / edge count asks for (0: the one-launch kernels are used
 * and `workspace` may be NULL; edges_hint as pgnn_dyn_count.hint, 0 = none).
 * n_edges / num_keypoints: both NULL (host-sized form: edges_cap and
 * keypoints_cap are the counts) or both given (capacity form).  `workspace`:
 * 16-byte aligned, owned by the caller, free for reuse once the call's work on
 * `stream` is done (too small: PGNN_E_WORKSPACE, nothing launched); NULL or a
 * chain the split form does not cover: exactly pgnn_point_set_pooling_fwd(_dyn). */
int pgnn_point_set_pooling_workspace_bytes(const pgnn_fc_layer *layers_host,
                                           int32_t n_layers, int32_t n_feat,
                                           int64_t edges_cap,
                                           int64_t edges_hint, size_t *bytes);
int pgnn_point_set_pooling_fwd_ws(const float *point_features, int32_t n_feat,
                                  const float *point_xyz,
                                  const int32_t *keypoint_indices,
                                  const int32_t *edges, int64_t edges_cap,
                                  int32_t keypoints_cap,
                                  const pgnn_fc_layer *layers_host,
                                  int32_t n_layers, int32_t edges_sorted,
                                  float *out, int64_t ld_out, int32_t *sched_ws,
                                  const pgnn_dyn_count *n_edges,
                                  const pgnn_dyn_count *num_keypoints,
                                  void *workspace, size_t workspace_bytes,
                                  void *stream);

/* Training forward of the same: the fused kernel also writes the point MLP's
 * activations -- acts_host[0..3] (a HOST array of four DEVICE pointers):
 * [n_edges, 32], [n_edges, 64], [n_edges, 128] and [n_edges, ld_last] for the
 * shipped car chain 4-32-64-128-300 -- which the backward needs (the last one
 * against `out` gives the arg-max rows: rows and maxima come from the same
 * accumulators).  PGNN_E_UNSUPPORTED, having done nothing, for other chains
 * or fewer than ~65k edges: run pgnn_pool_features_fwd + pgnn_mlp_fwd per
 * layer + pgnn_scatter_max_f32 instead.  edges_sorted: bit 0 = grouped by
 * ascending dst; bit 1 = `out` already holds lowest() (the native trainer
 * fills it in the launch before: no fill launch inside the call); the same
 * two bits on pgnn_edge_mlp_scatter_max_rows_fwd.                            */
int pgnn_point_set_pooling_rows_fwd(const float *point_features, int32_t n_feat,
                                    const float *point_xyz,
                                    const int32_t *keypoint_indices,
                                    const int32_t *edges, int64_t n_edges,
                                    int32_t num_keypoints,
                                    const pgnn_fc_layer *layers_host,
                                    int32_t n_layers, int32_t edges_sorted,
                                    float *out, int64_t ld_out,
                                    float *const *acts_host, int64_t ld_last,
                                    void *stream);

/* Fused GraphNetAutoCenter edge stage (gnn.py:338-365).  The first edge layer
 * is linear before its ReLU, so with P = [h, x] @ W1 + b1 and Q = x' @ W1[C:]
 * (both per vertex, computed by pgnn_mlp_fwd / pgnn_offset_apply) the
 * per-edge hidden vector is ReLU(P[src] - Q[dst]); this entry gathers it,
 * applies the remaining edge layers (ReLU) and scatter-maxes into dst rows.
 * P, Q: [num_vertices, ld_pq] with ld_pq = 16*ceil(width/16), zero padded.
 * edges_sorted: bit 0 = edges sorted by dst; bit 1 = `out` already holds
 * float lowest() everywhere (pgnn_vertex_pre_edge_fwd filled it), skip the
 * fill.                                                                      */
int pgnn_edge_mlp_scatter_max_fwd(const float *P, const float *Q,
                                  int64_t ld_pq, int32_t width,
                                  const int32_t *edges, int64_t n_edges,
                                  int32_t num_vertices,
                                  const pgnn_fc_layer *layers_host,
                                  int32_t n_layers, int32_t edges_sorted,
                                  float *out, int64_t ld_out, int32_t *sched_ws,
                                  void *stream);
int pgnn_edge_mlp_scatter_max_fwd_dyn(const float *P, const float *Q,
                                      int64_t ld_pq, int32_t width,
                                      const int32_t *edges, int64_t edges_cap,
                                      int32_t vertices_cap,
                                      const pgnn_fc_layer *layers_host,
                                      int32_t n_layers, int32_t edges_sorted,
                                      float *out, int64_t ld_out,
                                      int32_t *sched_ws,
                                      const pgnn_dyn_count *n_edges,
                                      const pgnn_dyn_count *num_vertices,
                                      void *stream);

/* SECONDARY arithmetic for the same stage with ONE remaining edge layer of
 * 300x300 or 256x256 (csrc/edge_ws_bf16.h): the layer's product runs on the
 * bf16 matrix pipe with BOTH operands split exactly into three bf16 parts
 * (8 + 8 + 8 significand bits) and the six products of combined order <= 2
 * accumulated in fp32 -- the dropped terms are below 2^-24 of a product, one
 * fp32 rounding (the weights' parts are rounded to nearest, the gathered
 * operand's first part too, its residual is cut in two); results agree with
 * the fp32-MFMA entry to fp32 rounding noise (not bit for bit) and are no
 * further from a float64 evaluation or from the reference's TF graphs
 * (tests/test_gpu_bf16x3.py, the edge_arith-parametrised parity tests).  The
 * fp32 entry above stays the
 * default and the parity reference (gnn.py:355-365 is fp32 in the reference).
 * `image`: device copy of what pgnn_pack_fc_bf16x3 wrote for the layer
 * (pgnn_packed_fc_bf16x3_bytes of it); width = the layer's k_in, n_out its
 * width, relu_from as in pgnn_fc_layer.  n_edges / num_vertices: NULL = the
 * capacities are the counts; else the capacity form (counts on the device).
 * PGNN_E_UNSUPPORTED, having done nothing, for other shapes, fewer than
 * ~65k edges (tunable b16_force lifts that) or P / Q of 4 GiB and more (rows
 * are addressed with 32-bit byte offsets): run pgnn_edge_mlp_scatter_max_fwd. */
size_t pgnn_packed_fc_bf16x3_bytes(int32_t k_in, int32_t n_out);
int pgnn_pack_fc_bf16x3(const float *w_host, const float *b_host, int32_t k_in,
                        int32_t n_out, void *image_host);
int pgnn_edge_mlp_scatter_max_bf16x3_fwd(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t edges_cap, int32_t vertices_cap,
    const void *image, int32_t n_out, int32_t relu_from, int32_t edges_sorted,
    float *out, int64_t ld_out, const pgnn_dyn_count *n_edges,
    const pgnn_dyn_count *num_vertices, void *stream);

/* SECONDARY arithmetic 'f16x2' for the same stage and shapes
 * (csrc/edge_ws_f16.h): both operands of the layer's product are represented
 * by TWO fp16 values (x ~ x0 + x1' / 2^11, round to nearest, 22 significand
 * bits), three fp16 MFMAs per block accumulate in fp32 -- half the matrix
 * instructions of the bf16x3 entry, three column groups instead of four.  Not
 * exact: each operand is within 2^-22 of its fp32 value (fp32's own rounding
 * is 2^-24); the stage's distance to a float64 evaluation grows by a few per
 * cent over the fp32 entry's (tests: the edge_arith-parametrised parity tests,
 * same bars as bf16x3).  Range: the gathered operand is clamped at 65504, and
 * bit 0 of *status (device int32, nullable; the caller zeroes it) is raised
 * when one COULD have reached 32768 (an element of the first num_vertices
 * rows of P or Q at or above 16384 in magnitude, or not a number: checked in
 * the kernel's prologue) -- rerun the stage through the fp32 entry then.
 * `image`: device copy of what pgnn_pack_fc_f16x2 wrote (it refuses weights
 * of magnitude >= 32768 with PGNN_E_UNSUPPORTED).  Otherwise as the bf16x3
 * entry, PGNN_E_UNSUPPORTED included.                                        */
size_t pgnn_packed_fc_f16x2_bytes(int32_t k_in, int32_t n_out);
int pgnn_pack_fc_f16x2(const float *w_host, const float *b_host, int32_t k_in,
                       int32_t n_out, void *image_host);
int pgnn_edge_mlp_scatter_max_f16x2_fwd(
    const float *P, const float *Q, int64_t ld_pq, int32_t width,
    const int32_t *edges, int64_t edges_cap, int32_t vertices_cap,
    const void *image, int32_t n_out, int32_t relu_from, int32_t edges_sorted,
    float *out, int64_t ld_out, int32_t *status, const pgnn_dyn_count *n_edges,
    const pgnn_dyn_count *num_vertices, void *stream);

/* PointSetPooling (gnn.py:256-277) in the same SECONDARY arithmetic, for the
 * car point MLP 4 -> 32 -> 64 -> 128 -> 300 (csrc/pool_ws_f16.h): the three
 * narrow layers stay fp32 MFMA, the last layer's [E, 128] x [128, 300] product
 * -- 79 % of the stage's matrix time -- runs as three fp16 products per block
 * with both operands in two fp16 parts.  `layers` as for
 * pgnn_point_set_pooling_fwd (the last one's fp32 image is not read);
 * `last_image`: device copy of what pgnn_pack_fc_f16x2_acc wrote for the last
 * layer (the rows of W in the order a lane of the previous layer's fp32
 * accumulators holds them; same size as pgnn_packed_fc_f16x2_bytes);
 * `hidden_image` (nullable): the same for the 64 -> 128 layer below it, which
 * then runs in this arithmetic too (76 % of the remaining matrix time).  Counts
 * nullable (host-sized call) or both given (capacity form).  `status` as for
 * the edge entry.  Returns PGNN_E_UNSUPPORTED -- having done nothing -- for
 * other layer shapes or too few edges (tunable `b16_force` lifts the size
 * test): the caller takes pgnn_point_set_pooling_fwd(_dyn) then.            */
int pgnn_pack_fc_f16x2_acc(const float *w_host, const float *b_host,
                           int32_t k_in, int32_t n_out, void *image_host);
int pgnn_point_set_pooling_f16x2_fwd(
    const float *point_features, int32_t n_feat, const float *point_xyz,
    const int32_t *keypoint_indices, const int32_t *edges, int64_t edges_cap,
    int32_t keypoints_cap, const pgnn_fc_layer *layers, int32_t n_layers,
    const void *last_image, const void *hidden_image, int32_t edges_sorted,
    float *out, int64_t ld_out, int32_t *sched_ws, int32_t *status,
    const pgnn_dyn_count *n_edges, const pgnn_dyn_count *num_keypoints,
    void *stream);

/* Training forward of the same stage with ONE remaining edge layer: the fused
 * kernel also writes that layer's per-edge output rows [n_edges, ld_rows]
 * (the backward compares them with `out` to find the arg-max rows; rows and
 * maxima come from the same accumulators, so the winner reproduces `out` bit
 * for bit).  h1_out (nullable): the gathered hidden rows ReLU(P[src] -
 * Q[dst]) [n_edges, ld_pq] are written as well (= pgnn_edge_hidden_fwd).
 * Returns PGNN_E_UNSUPPORTED -- having done nothing -- when the
 * weights-stationary kernel does not apply (fewer than ~65k edges, layer
 * shapes other than 300x300 / 256x256): the caller then runs
 * pgnn_edge_hidden_fwd + pgnn_mlp_fwd + pgnn_scatter_max_f32.               */
int pgnn_edge_mlp_scatter_max_rows_fwd(const float *P, const float *Q,
                                       int64_t ld_pq, int32_t width,
                                       const int32_t *edges, int64_t n_edges,
                                       int32_t num_vertices,
                                       const pgnn_fc_layer *layer_host,
                                       int32_t edges_sorted, float *out,
                                       int64_t ld_out, float *rows_out,
                                       int64_t ld_rows, float *h1_out,
                                       void *stream);

/* x' = x + delta (gnn.py:346) and Q = x' @ Wx where Wx = the last 3 rows of
 * the first edge layer's weights (the rows that multiply the coordinate part
 * of the concat, gnn.py:350-352).  wx: device [3, ld_q] (zero padded).      */
int pgnn_offset_apply(const float *xyz, const float *delta, int64_t ld_delta,
                      int64_t n_rows, const float *wx, float *xyz_out,
                      float *Q, int64_t ld_q, void *stream);

/* Everything gnn.py:341-356 does per VERTEX before the per-edge work, in one
 * launch: delta = offset MLP(h[:, :c]) (n_offset_layers = 0: no auto offset,
 * delta = 0), Q = (xyz + delta) @ wx, P = [h[:, :c], xyz] @ W1 + b1
 * (p_layer: k_in = c + 3), and -- when agg is non-null -- the lowest() fill
 * of the [n_vertices, ld_agg] buffer the edge stage maxes into.  Equivalent
 * to pgnn_mlp_fwd (offset chain) + pgnn_offset_apply + pgnn_mlp_fwd (P) + fill
 * with identical arithmetic.  P, Q: [n_vertices, ld_pq], ld_pq = padded width
 * of p_layer's output; wx: device [3, ld_pq].                               */
int pgnn_vertex_pre_edge_fwd(const float *h, int64_t ld_h, int32_t c,
                             const float *xyz,
                             const pgnn_fc_layer *offset_layers_host,
                             int32_t n_offset_layers,
                             const pgnn_fc_layer *p_layer_host, const float *wx,
                             int64_t n_vertices, float *P, float *Q,
                             int64_t ld_pq, float *agg, int64_t ld_agg,
                             void *stream);
int pgnn_vertex_pre_edge_fwd_dyn(const float *h, int64_t ld_h, int32_t c,
                                 const float *xyz,
                                 const pgnn_fc_layer *offset_layers_host,
                                 int32_t n_offset_layers,
                                 const pgnn_fc_layer *p_layer_host,
                                 const float *wx, int64_t vertices_cap,
                                 float *P, float *Q, int64_t ld_pq, float *agg,
                                 int64_t ld_agg,
                                 const pgnn_dyn_count *n_vertices,
                                 void *stream);

/* Two per-vertex stages in ONE launch (small K: one 8-wave workgroup per
 * 16-row tile; PGNN_E_UNSUPPORTED -- having done nothing -- above ~32 rows per
 * CU or for chains wider than 320, the caller then runs the separate entries).
 * The reference evaluates them as separate TF ops on either side of an
 * operator boundary: the END of one layer -- PointSetPooling's output MLP
 * (gnn.py:279-283) or GraphNetAutoCenter's update MLP + residual
 * (gnn.py:367-372) -- and the START of the next GraphNetAutoCenter
 * (gnn.py:341-356: offset MLP, Q, P; see pgnn_vertex_pre_edge_fwd).
 *   y = front chain(x[:, :nx]) (+ residual)      -> [n_vertices, ld_y], always
 *                                                   written: it is the first
 *                                                   operator's output
 *   then pgnn_vertex_pre_edge_fwd on h = y.
 * Identical arithmetic to pgnn_mlp_fwd followed by pgnn_vertex_pre_edge_fwd
 * (the same MFMA sequence per output element): bit-identical results.
 * `agg` (the NEXT edge stage's aggregation buffer) must not alias `x`.       */
int pgnn_vertex_update_pre_edge_fwd(
    const float *x, int64_t ld_x, int32_t nx,
    const pgnn_fc_layer *front_layers_host, int32_t n_front,
    const float *residual, int64_t ld_res, float *y, int64_t ld_y, int32_t c,
    const float *xyz, const pgnn_fc_layer *offset_layers_host,
    int32_t n_offset_layers, const pgnn_fc_layer *p_layer_host, const float *wx,
    int64_t n_vertices, float *P, float *Q, int64_t ld_pq, float *agg,
    int64_t ld_agg, void *stream);
int pgnn_vertex_update_pre_edge_fwd_dyn(
    const float *x, int64_t ld_x, int32_t nx,
    const pgnn_fc_layer *front_layers_host, int32_t n_front,
    const float *residual, int64_t ld_res, float *y, int64_t ld_y, int32_t c,
    const float *xyz, const pgnn_fc_layer *offset_layers_host,
    int32_t n_offset_layers, const pgnn_fc_layer *p_layer_host, const float *wx,
    int64_t vertices_cap, float *P, float *Q, int64_t ld_pq, float *agg,
    int64_t ld_agg, const pgnn_dyn_count *n_vertices, void *stream);

/* The same for the model's last operator boundary: GraphNetAutoCenter's update
 * MLP + residual (gnn.py:367-372) followed by a second chain on its output --
 * the fused ClassAwarePredictor heads (gnn.py:133-163):
 *   y = front chain(x[:, :nx]) (+ residual)   -> [n_rows, ld_y], always written
 *   out = back chain(y[:, :ny])               -> [n_rows, ld_out]
 * = pgnn_mlp_fwd twice, bit-identical.  Same size limits as above.            */
int pgnn_mlp2_fwd(const float *x, int64_t ld_x, int32_t nx,
                  const pgnn_fc_layer *front_layers_host, int32_t n_front,
                  const float *residual, int64_t ld_res, float *y, int64_t ld_y,
                  int32_t ny, const pgnn_fc_layer *back_layers_host,
                  int32_t n_back, float *out, int64_t ld_out, int64_t n_rows,
                  void *stream);
int pgnn_mlp2_fwd_dyn(const float *x, int64_t ld_x, int32_t nx,
                      const pgnn_fc_layer *front_layers_host, int32_t n_front,
                      const float *residual, int64_t ld_res, float *y,
                      int64_t ld_y, int32_t ny,
                      const pgnn_fc_layer *back_layers_host, int32_t n_back,
                      float *out, int64_t ld_out, int64_t rows_cap,
                      const pgnn_dyn_count *rows, void *stream);

/* ---- training step (config 4: models.py:170-311, train.py:135-171,264-297,
 * 375-405, util/tf_util.py:3-43) ------------------------------------------
 * The backward pass needs the per-edge activations, so the training forward
 * materialises them with the primitives below (plus pgnn_mlp_fwd one layer at
 * a time and pgnn_scatter_max_f32); each primitive has an explicit adjoint.  */

/* Device-side pgnn_pack_fc (weights change every step).  transpose != 0 packs
 * W^T (the layer dX = dZ @ W^T of the backward pass; no bias).              */
int pgnn_pack_fc_device(const float *w, const float *b, int32_t k_in,
                        int32_t n_out, int32_t transpose, float *packed,
                        void *stream);

/* Every layer's device images refreshed in ONE launch (after pgnn_sgd_step):
 * `jobs_device` is a device array of n_jobs records
 *   struct { const float *w; const float *b; float *dst; int32_t k_in, n_out,
 *            kind, first_block, ld, reserved; }           (48 bytes each)
 * kind 0 = pgnn_pack_fc image of (w [k_in,n_out], b), 1 = image of w^T (no
 * bias), 2 = plain w^T as [n_out][16*ceil(k_in/16)] rows (zero padded; what
 * pgnn_segmax_fc_bwd_f32 reads), 3 = plain w as [k_in][16*ceil(n_out/16)]
 * rows (zero padded); 4 / 5 / 6 = strided block copy / block add / identity
 * (how the native trainer assembles its fused prediction-head matrices from
 * parameter blocks and hands their gradients back; `ld` = row stride of the
 * strided side).  A job covers ceil(elements / 256) blocks
 * of the launch; first_block is the running sum, total_blocks the grand
 * total.  The table is built once: the flat parameter buffer never moves.  */
int pgnn_pack_fc_many(const void *jobs_device, int32_t n_jobs,
                      int32_t total_blocks, void *stream);

/* H1[e] = ReLU(P[src(e)] - Q[dst(e)]) materialised, [n_edges, ld_pq].       */
int pgnn_edge_hidden_fwd(const float *P, const float *Q, int64_t ld_pq,
                         const int32_t *edges, int64_t n_edges, float *H1,
                         void *stream);
/* Adjoint: dP[src] += dH1, dQ[dst] -= dH1 (dH1 already masked by H1 > 0).
 * dP, dQ [n_vertices, ld] are zeroed by the call.                            */
int pgnn_edge_hidden_bwd(const float *dH1, int64_t ld, const int32_t *edges,
                         int64_t n_edges, int64_t n_vertices, float *dP,
                         float *dQ, void *stream);
/* batch_data (train.py:135-171): the frames of a batch merged into one
 * disjoint graph -- every array the concatenation of the frames' arrays,
 * keypoint indices and edge rows moved up by the points / centres of the
 * frames before them (`:150-160`) -- as ONE launch: job j copies n_words 4-byte
 * words from src to dst (device pointers; dst points INTO the merged array),
 * adding add0 to even and add1 to odd words as int32 (keypoint indices [k,1]:
 * add0 = add1 = points before; edge rows [E,2]: add0 = points before, add1 =
 * centres before; float arrays: 0, 0).  jobs_host is a HOST array. */
typedef struct pgnn_merge_job {
  const void *src;
  void *dst;
  int64_t n_words;
  int32_t add0, add1;
} pgnn_merge_job;
int pgnn_merge_rows(const pgnn_merge_job *jobs_host, int32_t n_jobs,
                    void *stream);
/* PointSetPooling edge features [f(src), xyz(src) - xyz(kp(dst)), 0...] as a
 * [n_edges, 16] matrix (gnn.py:256-267).                                     */
int pgnn_pool_features_fwd(const float *point_features, int32_t n_feat,
                           const float *point_xyz,
                           const int32_t *keypoint_indices,
                           const int32_t *edges, int64_t n_edges, float *F,
                           void *stream);
/* The same rows for any feature width, [n_edges, ld_f >= n_feat + 3] with zero
 * pad columns, from features with row stride ld_features: a PointSetPooling
 * above the first pooling level gathers the previous level's (300-wide)
 * vertex features, which the fused pooling kernel (n_feat <= 13) does not
 * take -- pgnn_pool_features_wide_fwd + pgnn_mlp_fwd + pgnn_scatter_max_f32
 * are the operator then (no shipped config).                                 */
int pgnn_pool_features_wide_fwd(const float *point_features,
                                int64_t ld_features, int32_t n_feat,
                                const float *point_xyz,
                                const int32_t *keypoint_indices,
                                const int32_t *edges, int64_t n_edges, float *F,
                                int64_t ld_f, void *stream);
/* dY[i] = (Y[i] > 0) ? dY[i] : 0 in place (tf ReluGrad), over `count` floats. */
int pgnn_relu_mask_mul(float *dY, const float *Y, int64_t count, void *stream);
/* Gradient of tf.math.unsorted_segment_max (TF's
 * _UnsortedSegmentMinOrMaxGrad): rows equal to their segment's max share the
 * segment's gradient equally.  relu_mask != 0 additionally zeroes rows whose
 * value is not > 0 (fused ReluGrad of the layer that produced `data`).
 * tie_count_ws: int32 [num_segments * n_cols] scratch.                       */
int pgnn_scatter_max_bwd_f32(const float *data, int64_t ld_data,
                             const int32_t *seg_ids, int64_t n_rows,
                             int32_t n_cols, int32_t num_segments,
                             const float *out, int64_t ld_out,
                             const float *grad_out, int64_t ld_grad_out,
                             int32_t *tie_count_ws, float *grad_data,
                             int64_t ld_grad_data, int32_t relu_mask,
                             void *stream);
/* Adjoint of  out = unsorted_segment_max(Y),  Y = ReLU(X W + b)  -- the last
 * per-edge layer and the aggregation of PointSetPooling (gnn.py:269-277) and
 * GraphNetAutoCenter (gnn.py:357-365) -- exploiting that the max passes its
 * gradient to one row per (segment, column): ~num_segments*n_cols non-zeros
 * among n_rows*n_cols, so neither dZ nor the two n_rows-sized GEMMs of the
 * dense adjoint are formed.  Same result as pgnn_scatter_max_bwd_f32
 * (relu_mask = 1: TF's tie rule, ReluGrad) followed by dX = dZ W^T (masked by
 * X > 0 when mask_x != 0: the ReluGrad of the layer that produced X) and
 * dW += X^T dZ, db += column sums of dZ, up to float32 summation order.
 *   Y [n_rows, ld_y], out [num_segments, ld_out]: 16-byte aligned rows padded
 *   to a multiple of 4 columns; X [n_rows, ld_x] (k_in features);
 *   WT = W^T as [n_cols][ld_wt] (pgnn_pack_fc_many kind 2);
 *   dX [n_rows, ld_dx] (nullable): dx_cols columns written (zeros beyond k_in);
 *   dW [k_in, n_cols] and db [n_cols] (nullable) are ACCUMULATED into.       */
size_t pgnn_segmax_fc_bwd_workspace_bytes(int64_t n_rows, int32_t n_cols,
                                          int32_t num_segments, int32_t k_in);
int pgnn_segmax_fc_bwd_f32(const float *Y, int64_t ld_y,
                           const int32_t *seg_ids, int64_t n_rows,
                           int32_t n_cols, int32_t num_segments,
                           const float *out, int64_t ld_out,
                           const float *grad_out, int64_t ld_go,
                           const float *X, int64_t ld_x, int32_t k_in,
                           const float *WT, int64_t ld_wt, float *dX,
                           int64_t ld_dx, int32_t dx_cols, int32_t mask_x,
                           float *dW, float *db, void *workspace,
                           size_t workspace_bytes, void *stream);
/* The GraphNetAutoCenter form (gnn.py:348-365): X = H1 = ReLU(P[src] - Q[dst])
 * [n_edges, ld_h1] feeds the layer, so the routed gradient is scattered
 * straight into dP[src] += g, dQ[dst] -= g (both [num_vertices, ld_pq], zeroed
 * by the call: pgnn_edge_hidden_bwd's outputs) and dH1 is never written.
 * edges [n_edges, 2] = (src, dst) grouped by dst for the in-register dQ runs
 * (any order is correct); dst_ids = its dst column, contiguous.  H1 may be
 * NULL (the fused training forward pgnn_edge_mlp_scatter_max_rows_fwd never
 * writes it): its rows are then recomputed from P and Q [num_vertices,
 * ld_pq] as ReLU(P[src] - Q[dst]).                                          */
int pgnn_edge_segmax_fc_bwd_f32(const float *Y, int64_t ld_y,
                                const int32_t *edges, const int32_t *dst_ids,
                                int64_t n_edges, int32_t n_cols,
                                int32_t num_vertices, const float *out,
                                int64_t ld_out, const float *grad_out,
                                int64_t ld_go, const float *H1, int64_t ld_h1,
                                const float *P, const float *Q, int32_t k_in,
                                const float *WT, int64_t ld_wt,
                                float *dP, float *dQ, int64_t ld_pq, float *dW,
                                float *db, void *workspace,
                                size_t workspace_bytes, void *stream);
/* dW [k_in, n_out] (= X^T dZ) and db [n_out] (= column sums of dZ; may be
 * NULL) of y = x W + b, deterministic (fixed-order slice reduction).
 * accumulate != 0 adds to dW/db.                                             */
size_t pgnn_weight_grad_workspace_bytes(int32_t k_in, int32_t n_out,
                                        int64_t n_rows);
int pgnn_weight_grad_f32(const float *X, int64_t ld_x, int32_t k_in,
                         const float *dZ, int64_t ld_dz, int32_t n_out,
                         int64_t n_rows, float *dW, float *db,
                         int32_t accumulate, void *workspace,
                         size_t workspace_bytes, void *stream);
/* Fused backward of PointSetPooling's narrow point-MLP layers for the shipped
 * car chain feat -> 32 -> 64 -> 128 (gnn.py:256-277 under tf.gradients): given
 * the gradient w.r.t. the third layer's pre-activation (what
 * pgnn_segmax_fc_bwd_f32 hands over for the 128 -> 300 layer), the weight and
 * bias gradients of all three layers in one pass over the E rows -- the two dX
 * products and ReLU masks in between happen in LDS.  feat [E,16] (k_in0 <= 15
 * columns used), act0 [E,32], act1 [E,64], dz2 [E,128]: contiguous rows;
 * w2t_packed / w1t_packed: the TRANSPOSED images (pgnn_pack_fc_many kind 1) of
 * the 64 -> 128 and 32 -> 64 layers.  Deterministic (fixed slice order).     */
size_t pgnn_pool_narrow_bwd_workspace_bytes(int64_t n_rows);
int pgnn_pool_narrow_bwd_f32(const float *feat, const float *act0,
                             const float *act1, const float *dz2,
                             int64_t n_rows, const float *w2t_packed,
                             const float *w1t_packed, int32_t k_in0, float *dW0,
                             float *db0, float *dW1, float *db1, float *dW2,
                             float *db2, int32_t accumulate, void *workspace,
                             size_t workspace_bytes, void *stream);
/* The same for MANY small layers in one launch pair (train.py:264-297 asks
 * tf.gradients for every variable at once; here the K-row layers of a step --
 * ~25 GEMMs of [k_in x K] x [K x n_out], K ~ 2 000 vertices -- would each be a
 * latency-bound launch plus a reduce: 0.7 ms of a 4.7 ms step).  jobs: a HOST
 * array; 1 <= n_out <= 320 per job; dW is [k_in, n_out] contiguous; the
 * outputs of different jobs must not overlap (they are written by one launch).  Fixed
 * slice partition and reduction order: deterministic, not bit-equal to
 * pgnn_weight_grad_f32 (other slice boundaries).                           */
typedef struct pgnn_wgrad_job {
  const float *X;   /* [n_rows, ld_x] layer input                           */
  int64_t ld_x;
  const float *dZ;  /* [n_rows, ld_dz] gradient w.r.t. the pre-activation    */
  int64_t ld_dz;
  int64_t n_rows;
  float *dW;        /* [k_in, n_out]                                         */
  float *db;        /* [n_out] or NULL                                       */
  int32_t k_in, n_out;
  int32_t accumulate, reserved;
} pgnn_wgrad_job;
size_t pgnn_weight_grad_many_workspace_bytes(const pgnn_wgrad_job *jobs_host,
                                             int32_t n_jobs);
int pgnn_weight_grad_many_f32(const pgnn_wgrad_job *jobs_host, int32_t n_jobs,
                              void *workspace, size_t workspace_bytes,
                              void *stream);
/* models.py:212-255 (cls 'softmax', loc 'huber_loss', delta 1):
 * sums4 (device, 4 doubles) = {sum_v CE_v, sum_v mean_7 huber_v*valid_v,
 * n_vertices, sum valid}; dlogits [n, nc] / dpred_box [n, nc, box_len]
 * receive the gradients of cls_grad_scale*sum CE + loc_grad_scale*sum loc
 * (the caller folds loss weights and the global 1/N, 1/N_valid of
 * train.py:264-288 into the two scales).                                     */
int pgnn_loss_fwd_bwd(const float *logits, int64_t ld_logits,
                      const int32_t *labels, const float *pred_box,
                      int32_t box_len, const float *gt_box, const float *valid,
                      int64_t n_vertices, int32_t num_classes,
                      float cls_grad_scale, float loc_grad_scale,
                      double *sums4, float *dlogits, float *dpred_box,
                      void *stream);
/* The same with the GLOBAL endpoint counts of train.py:268-284 on the device
 * (counts2 = {num_endpoint, num_valid_endpoint} as doubles, e.g. fresh out of
 * an all-reduce over the ranks): the kernel forms cls_loss_weight /
 * num_endpoint and loc_loss_weight / num_valid_endpoint itself (0 for a zero
 * count: tf.math.div_no_nan), so no host read sits between the forward and the
 * backward pass of a multi-rank step.                                         */
int pgnn_loss_fwd_bwd_counts(const float *logits, int64_t ld_logits,
                             const int32_t *labels, const float *pred_box,
                             int32_t box_len, const float *gt_box,
                             const float *valid, int64_t n_vertices,
                             int32_t num_classes, double cls_loss_weight,
                             double loc_loss_weight, const double *counts2,
                             double *sums4, float *dlogits, float *dpred_box,
                             void *stream);
/* The same pass with the other per-vertex classification losses of
 * models.py:210-228 and the per-class localisation weights of :240-246:
 * cls_kind 0 'softmax' (as above), 1 'focal_softmax' (models/loss.py:31-48:
 * (1 - p_label)^gamma * CE), 2 'focal_sigmoid' (models/loss.py:5-29: per-class
 * sigmoid CE * (1 - p_t)^gamma * (alpha | 1 - alpha), averaged over the classes
 * as well, models.py:229).  class_loc_weight (nullable, device float
 * [num_classes]): 'classwise_loc_loss_weight' of loc_loss_kwargs, applied to a
 * vertex's Huber terms by its label.  counts2 nullable: null = the host-scale
 * form (cls_grad_scale / loc_grad_scale used), else the _counts form.
 * gamma == 0 is plain CE / sigmoid CE (TF's pow(0, 0) = 1); for 0 < gamma < 1
 * the derivative's (1 - p_t)^(gamma - 1) is evaluated at max(1 - p_t, 1e-6).
 * 'top_k_softmax' / 'top_k_huber_loss': pgnn_loss_fwd_bwd_sel below.         */
int pgnn_loss_fwd_bwd_ex(const float *logits, int64_t ld_logits,
                         const int32_t *labels, const float *pred_box,
                         int32_t box_len, const float *gt_box,
                         const float *valid, int64_t n_vertices,
                         int32_t num_classes, float cls_grad_scale,
                         float loc_grad_scale, const double *counts2,
                         double cls_loss_weight, double loc_loss_weight,
                         int32_t cls_kind, float alpha, float gamma,
                         const float *class_loc_weight, double *sums4,
                         float *dlogits, float *dpred_box, void *stream);
/* The top-k variants (models.py:222-228 'top_k_softmax', :266-291
 * 'top_k_huber_loss') in three steps, all on the device:
 *   1. pgnn_loss_fwd_bwd_sel with point_cls / point_loc set (no gradients):
 *      the per-vertex classification loss and the per-vertex mean Huber loss
 *      (valid and class weights applied) -- what the reference hands to
 *      tf.math.top_k;
 *   2. pgnn_topk_mask_f32: mask[i] = 1 for the k largest values (equal values:
 *      lower index first, like tf.math.top_k), 0 elsewhere; k > n is an error
 *      as in TF;
 *   3. pgnn_loss_fwd_bwd_sel with select_cls / select_loc = those masks: a
 *      vertex outside a selection adds no loss and gets no gradient;
 *      sums4[3] counts the valid vertices INSIDE the loc selection
 *      (models.py:283-285); cls_sum_weight scales the classification sum and
 *      its gradient -- n / k turns the mean over n the callers divide by into
 *      the reference's mean over the k selected values (with towers /
 *      unify_copies, train.py:268-284: n_rank / k on every rank).
 * Every pointer after class_loc_weight is nullable; with all of them null and
 * cls_sum_weight 1 this is pgnn_loss_fwd_bwd_ex. */
int pgnn_loss_fwd_bwd_sel(const float *logits, int64_t ld_logits,
                          const int32_t *labels, const float *pred_box,
                          int32_t box_len, const float *gt_box,
                          const float *valid, int64_t n_vertices,
                          int32_t num_classes, float cls_grad_scale,
                          float loc_grad_scale, const double *counts2,
                          double cls_loss_weight, double loc_loss_weight,
                          int32_t cls_kind, float alpha, float gamma,
                          const float *class_loc_weight,
                          const float *select_cls, const float *select_loc,
                          float cls_sum_weight, float *point_cls,
                          float *point_loc, double *sums4, float *dlogits,
                          float *dpred_box, void *stream);
size_t pgnn_topk_mask_workspace_bytes(int64_t n);
int pgnn_topk_mask_f32(const float *values, int64_t n, int64_t k, float *mask,
                       void *workspace, size_t workspace_bytes, void *stream);
/* params -= lr * (grad_scale*grads + l1_scale*sign(params)*is_weight)
 * (GradientDescentOptimizer + slim.l1_regularizer on FC weights only).       */
int pgnn_sgd_step(float *params, const float *grads, const float *is_weight,
                  int64_t n, float lr, float grad_scale, float l1_scale,
                  void *stream);
/* The other optimizers train.py:380-391 offers, TF 1.x update rules on the same
 * flat buffers (gradient = grad_scale*grads + l1_scale*sign(params)*is_weight
 * as above; slot0 / slot1: optimizer state with the parameters' layout):
 *   kind 1 MomentumOptimizer (h0 = momentum, no Nesterov): slot0 = accumulator
 *   kind 2 RMSPropOptimizer (h0 = momentum, h1 = decay, h2 = epsilon; not
 *          centered): slot0 = ms (TF initialises it to ONES), slot1 = mom
 *   kind 3 AdamOptimizer (h0 = beta1, h1 = beta2, h2 = epsilon): slot0 = m,
 *          slot1 = v; `lr` is the caller's bias-corrected
 *          lr * sqrt(1 - beta2^t) / (1 - beta1^t), t = 1, 2, ...            */
int pgnn_optimizer_step(int32_t kind, float *params, const float *grads,
                        const float *is_weight, float *slot0, float *slot1,
                        int64_t n, float lr, float grad_scale, float l1_scale,
                        float h0, float h1, float h2, void *stream);
/* *out (device double) = sum |params| over is_weight entries (reg_loss/scale). */
int pgnn_l1_norm(const float *params, const float *is_weight, int64_t n,
                 double *out, void *stream);

/* ---- native training step (config 4) ----------------------------------------
 * What train.py builds with TF towers -- model.predict with saved
 * activations, tf.gradients of model.loss (models.py:79-163, 170-311;
 * train.py:225-297) -- as ONE object on the device: the whole forward and the
 * whole backward are two calls, every kernel launch inside is issued from C++.
 * (The Python mirror keeps a step composed of the primitives above; tests hold
 * the two to the same gradients.)
 *
 * The model is described by where each fully connected layer lives in the
 * caller's FLAT parameter buffer (reference variable order, [k_in, n_out]
 * row-major weights, then biases): offsets in floats.  Gradients go to a flat
 * buffer of the same layout and are ACCUMULATED (zero it per step).          */
#define PGNN_TRAIN_MAX_FC 8
#define PGNN_TRAIN_MAX_STAGES 8
#define PGNN_TRAIN_MAX_CLASSES 16
#define PGNN_TRAIN_MAX_LEVELS 4
typedef struct pgnn_train_fc {
  int64_t w_off, b_off; /* float offsets of weights / biases in the flat buffer */
  int32_t k_in, n_out;
} pgnn_train_fc;
typedef struct pgnn_train_stage {
  int32_t kind;        /* 0 = scatter_max_point_set_pooling (gnn.py:222-283),
                          1 = scatter_max_graph_auto_center_net (gnn.py:298-373) */
  int32_t graph_level; /* index into the batch's coords / keypoints / edges     */
  int32_t n_a, n_b, n_c, reserved;
  pgnn_train_fc a[PGNN_TRAIN_MAX_FC]; /* point MLP / edge MLP (all ReLU)        */
  pgnn_train_fc b[PGNN_TRAIN_MAX_FC]; /* output MLP (ReLU) / update MLP (last
                                         layer linear, + residual)             */
  pgnn_train_fc c[PGNN_TRAIN_MAX_FC]; /* auto-offset MLP, last layer linear;
                                         n_c = 0: auto_offset False            */
} pgnn_train_stage;
typedef struct pgnn_train_model {
  int32_t n_stages, num_classes, box_len, reserved;
  int64_t n_params;
  pgnn_train_stage stages[PGNN_TRAIN_MAX_STAGES];
  pgnn_train_fc cls[2];                          /* C -> 64 -> nc (gnn.py:145-152)  */
  pgnn_train_fc loc[PGNN_TRAIN_MAX_CLASSES][3];  /* per class C -> 64 -> 64 -> L   */
} pgnn_train_model;
typedef struct pgnn_train_batch { /* device pointers; what train.py's batch_data
                                     returns for this rank (train.py:135-171) */
  const float *input_v;           /* [n_vertices[0], n_feat]                   */
  int32_t n_feat, n_levels;
  int64_t n_vertices[PGNN_TRAIN_MAX_LEVELS + 1];
  const float *coords[PGNN_TRAIN_MAX_LEVELS + 1];   /* [n_vertices[l], 3]      */
  const int32_t *keypoints[PGNN_TRAIN_MAX_LEVELS];  /* [n_vertices[l + 1]]     */
  const int32_t *edges[PGNN_TRAIN_MAX_LEVELS];      /* [n_edges[l], 2]         */
  int64_t n_edges[PGNN_TRAIN_MAX_LEVELS];
  int32_t edges_sorted[PGNN_TRAIN_MAX_LEVELS];      /* grouped by ascending dst */
} pgnn_train_batch;

int pgnn_trainer_create(const pgnn_train_model *model_host, void **handle);
int pgnn_trainer_destroy(void *handle);
/* Device images of the weights (MFMA fragment order forward / transposed,
 * plain W^T of the sparse-adjoint layers, fused prediction heads) + the job
 * table that refreshes them: the caller allocates images_bytes once and binds
 * the flat buffers (they must not move afterwards).  bind packs once;
 * repack after every parameter update (one launch). */
size_t pgnn_trainer_images_bytes(void *handle);
int pgnn_trainer_bind(void *handle, float *params, float *grads, void *images,
                      size_t images_bytes, void *stream);
int pgnn_trainer_repack(void *handle, void *stream);
/* Scratch of one forward + backward on a batch of these sizes (saved
 * activations included). */
size_t pgnn_trainer_workspace_bytes(void *handle, const pgnn_train_batch *batch);
/* Forward with saved activations.  *logits / *pred_box point INTO the
 * workspace (valid until the next forward on it): logits [K, *ld_logits]
 * (num_classes columns used), pred_box [K, num_classes, box_len] contiguous. */
int pgnn_trainer_forward(void *handle, const pgnn_train_batch *batch,
                         void *workspace, size_t workspace_bytes,
                         const float **logits, int64_t *ld_logits,
                         const float **pred_box, void *stream);
/* Backward of the forward last run on `workspace`: dlogits [K, num_classes]
 * and dpred_box [K, num_classes, box_len] (what pgnn_loss_fwd_bwd wrote) ->
 * gradients accumulated into the bound flat gradient buffer. */
int pgnn_trainer_backward(void *handle, const pgnn_train_batch *batch,
                          void *workspace, size_t workspace_bytes,
                          const float *dlogits, const float *dpred_box,
                          void *stream);

/* Backward + the step's collectives in one call: pgnn_trainer_backward, then
 * -- when `comm` is not null -- pgnn_allreduce_step(comm, <bound gradient
 * buffer>, n_params, sums, n_sums) on the same stream: nothing of the host
 * between the last gradient kernel and the all-reduce's launch.  comm null:
 * exactly pgnn_trainer_backward. */
int pgnn_trainer_backward_sync(void *handle, const pgnn_train_batch *batch,
                               void *workspace, size_t workspace_bytes,
                               const float *dlogits, const float *dpred_box,
                               void *comm, double *sums, int64_t n_sums,
                               void *stream);

/* ---- collectives (config 4) ---------------------------------
 * One process per GPU.  Replaces the reference's in-process towers:
 * `average_gradients` (util/tf_util.py:3-43: concat the towers' gradients on a
 * new axis, reduce_mean over it) behind `unify_copies` (train.py:264-288:
 * every tower's cls / loc loss re-weighted by the GLOBAL endpoint counts) and
 * the single apply_gradients (train.py:397-405).  With every rank's loss
 * already divided by the global counts, the towers' mean x world = the SUM of
 * the ranks' gradients: one all-reduce(sum) of the flat fp32 gradient buffer
 * (1 489 609 floats = 5.96 MB for car_auto_T3; latency-bound on xGMI, hence
 * one call and no bucketing), and one of the counts / loss sums (float64).
 *
 * The communicator is RCCL's, bound at run time (dlopen of the RCCL image the
 * process already holds, else the system's librccl.so.1; PGNN_RCCL_LIB names
 * another).  Rank 0 calls pgnn_comm_unique_id and hands the
 * PGNN_COMM_ID_BYTES bytes to the other ranks by whatever the launcher offers
 * (a file, a torch.distributed store, MPI); then EVERY rank calls
 * pgnn_comm_init_rank (collective, blocking; uses the calling thread's current
 * HIP device).  The collectives are in place, asynchronous, ordered on
 * `stream` like every other entry; a world of one rank is valid (and still
 * enters RCCL).  Errors: PGNN_E_COMM with RCCL's message in pgnn_last_error. */
#define PGNN_COMM_ID_BYTES 128
int pgnn_comm_unique_id(void *id_host);
int pgnn_comm_init_rank(const void *id_host, int32_t world, int32_t rank,
                        void **comm_out);
int pgnn_comm_destroy(void *comm);
/* world / rank of `comm` (nullable: version only) and RCCL's version code
 * (e.g. 22606); each output nullable. */
int pgnn_comm_info(void *comm, int32_t *world, int32_t *rank,
                   int32_t *rccl_version);
/* Which RCCL image is bound (path or "<soname> (already loaded)"), or why none. */
const char *pgnn_comm_library(void);
/* 0, or PGNN_E_COMM with the asynchronous error RCCL recorded for `comm`. */
int pgnn_comm_async_error(void *comm);
/* buf[i] = sum over ranks of buf[i], in place. */
int pgnn_allreduce_sum_f32(void *comm, float *buf, int64_t n, void *stream);
int pgnn_allreduce_sum_f64(void *comm, double *buf, int64_t n, void *stream);
/* The step's two reductions as ONE RCCL group (one launch): the flat gradient
 * (fp32) and the loss sums / endpoint counts (float64).  Either may be empty. */
int pgnn_allreduce_step(void *comm, float *grads, int64_t n_grads, double *sums,
                        int64_t n_sums, void *stream);
/* buf of `root` -> every rank (initial weights of a run that was not resumed
 * from one checkpoint: the reference's towers share ONE set of variables,
 * train.py:225-227 under `reuse`). */
int pgnn_broadcast_f32(void *comm, float *buf, int64_t n, int32_t root,
                       void *stream);

/* ---- detection post-processing (the step right after
 * the path; run.py:264-326).  All arrays are device pointers.
 *
 * Box codec `classaware_all_class_box_encoding` (box_encoding.py:231-299), the
 * method of every shipped config.  cls_labels [n_rows] int32; xyz [n_rows,3];
 * boxes / encoded [n_rows, boxes_per_row, 7] = (x,y,z,l,h,w,yaw).  class_table
 * [n_table,5] rows = {l, h, w, yaw_offset, active}: row `label` describes how a
 * row with that label is (de)normalised -- the host builds it from label_map
 * and median_object_size_map (label -> median size, yaw_offset 0; label+1 ->
 * same size, yaw_offset pi/2).  Like the reference only box column 0 is class
 * scaled; the xyz offset applies to every column. */
int pgnn_box_decode_f32(const int32_t *cls_labels, const float *xyz,
                        const float *encoded, const float *class_table,
                        int32_t n_table, int64_t n_rows, int32_t boxes_per_row,
                        float *decoded, void *stream);
int pgnn_box_encode_f32(const int32_t *cls_labels, const float *xyz,
                        const float *boxes, const float *class_table,
                        int32_t n_table, int64_t n_rows, int32_t boxes_per_row,
                        float *encoded, void *stream);
/* run.py:266-290: candidates of probs [n_vertices, num_classes] = flat indices
 * p = k*num_classes + c with 0 < c < num_classes-1 and prob > 1/num_classes,
 * ascending (np.nonzero order).  out_label holds the merged class of
 * run.py:287-289 (2->1, 4->3, 6->5).  *out_count (device) receives the full
 * count even when it exceeds `capacity` (entries beyond it are dropped). */
int pgnn_detection_candidates(const float *probs, int64_t n_vertices,
                              int32_t num_classes, int32_t *out_index,
                              int32_t *out_label, int64_t capacity,
                              int32_t *out_count, void *stream);
/* The same for a capacity-form frame (the frame loop keeps frames in flight
 * and reads nothing between the GNN and this call): probs has n_vertices_cap
 * rows of which the first *n_vertices_dev (a device int32, the K of the
 * frame's count record) exist. */
int pgnn_detection_candidates_dyn(const float *probs, int64_t n_vertices_cap,
                                  const int32_t *n_vertices_dev,
                                  int32_t num_classes, int32_t *out_index,
                                  int32_t *out_label, int64_t capacity,
                                  int32_t *out_count, void *stream);
/* nms.py:241-300.  mode: 0 = nms_boxes_3d (plain; integer corners scaled by
 * appr_factor, nms.py:113-115), 1 = nms_boxes_3d_uncertainty (median merge +
 * score accumulation, what run.py uses), 2 = nms_boxes_3d_merge_only,
 * 3 = nms_boxes_3d_score_only; the overlap is overlapped_boxes_3d_fast_poly
 * (nms.py:64-88).  Boxes are ordered by descending score (ties: input order),
 * cut to top_k when top_k > 0, then scanned like the reference loop.  Outputs
 * (capacity n_boxes each) are the kept boxes in score order; attributes
 * (nullable -> arange) ride along; *out_count (device) = number kept. */
size_t pgnn_nms_workspace_bytes(int64_t n_boxes);
int pgnn_nms_boxes_3d(const int32_t *class_labels, const float *boxes_3d,
                      const float *scores, const int32_t *attributes,
                      int64_t n_boxes, double overlapped_thres, int32_t mode,
                      float appr_factor, int64_t top_k, void *workspace,
                      size_t workspace_bytes, int32_t *out_labels,
                      float *out_boxes, float *out_scores,
                      int32_t *out_attributes, int32_t *out_count,
                      void *stream);
/* overlapped_boxes_3d_fast_poly(single_box, box_list) on (x,y,z,l,h,w,yaw)
 * boxes (nms.py:64-88 after boxes_3d_to_corners :9-27); appr_factor > 0 applies
 * the integer corner rounding of nms.py:115.  overlap [n_boxes] float64. */
int pgnn_overlapped_boxes_3d(const float *single_box, const float *boxes_3d,
                             int64_t n_boxes, float appr_factor,
                             double *overlap, void *stream);
/* overlapped_boxes_3d(single_box, box_list) (nms.py:29-62), the cv2.fillPoly
 * raster overlap random_box_shift calls under every shipped train config
 * (preprocess.py:281-301): integer corner arrays [8,3] / [n_boxes,8,3] int32
 * (np.int32(appr_factor * boxes_3d_to_corners(.))), device pointers; overlap
 * [n_boxes] float64 = np.float32(intersection) / (union - intersection) with
 * pixel counts identical to cv2.fillPoly(LINE_8) + cv2.countNonZero on the
 * (z extent) x (x extent) buffers the reference allocates (OpenCV 4.2
 * drawing.cpp semantics: Bresenham outline after clipLine + 16.16 fixed-point
 * scan-line fill), evaluated in closed form per image row. */
int pgnn_overlapped_boxes_3d_raster(const int32_t *single_box_corners,
                                    const int32_t *box_corners, int64_t n_boxes,
                                    double *overlap, void *stream);

/* ---- KITTI frame ingest (the step right before the
 * path; dataset/kitti_dataset.py:587-609, 998-1006, 1036-1052, 666-689,
 * 990-996 = get_cam_points_in_image[_with_rgb] without the file reads).
 * velo_points [n,4] float32 (x,y,z,reflectance; the .bin layout) on the device;
 * velo_to_cam_3x4 (float32) and cam_to_image_3x3 (float64 = P2[:, :3]) are
 * HOST pointers, copied into the launch.  A point is kept when its camera-frame
 * z > 0.1 and its projection lies strictly inside (0,width) x (0,height);
 * kept points come out in scan order.  out_xyz [capacity,3]; out_attr
 * [capacity,attr_dim]: attr_dim 1 = reflectance, 4 = reflectance + r,g,b
 * sampled from image_bgr ([rows,cols,3] uint8, the cv2.imread layout; may be
 * null -> zeros).  *out_count (device) = number kept (also beyond capacity). */
/* ---- training targets (dataset/kitti_dataset.py
 * :143-162 sel_xyz_in_box3d, :1132-1284 assign_classaware_*_label_to_points;
 * train.py:100-130).  label_records [n_records,24] float64 on the device, one
 * row per ground-truth box in file order: normals[9] (rows wx,wy,wz), lower[3],
 * upper[3] (box3d_to_normals, :118-141, evaluated on the host), action
 * (0 = skip, 1 = object: class + box + valid 1, 2 = class only: valid 0),
 * class value, box[7] = (x3d,y3d,z3d,length,height,width,yaw wrapped into
 * (-pi/4, 3pi/4]).  A vertex strictly inside several boxes keeps what the
 * last one wrote, like the reference's slice assignments.  Outputs (each
 * nullable): cls_labels [n] int32 (0 = background), boxes_3d [n,7] float64,
 * valid_boxes [n] float32, owner [n] int32 = index of the last box containing
 * the vertex or -1 (with one action-1 record: the sel_xyz_in_box3d mask). */
int pgnn_assign_box_labels(const float *xyz, int64_t n_points,
                           const double *label_records, int32_t n_records,
                           int32_t *cls_labels, double *boxes_3d,
                           float *valid_boxes, int32_t *owner, void *stream);
/* Float64 vertices (train.py:100-118 passes the float64 vertex_coord_list of
 * the augmented cloud): np.matmul(xyz_f64, normals.T) on the exact values. */
int pgnn_assign_box_labels_f64(const double *xyz, int64_t n_points,
                               const double *label_records, int32_t n_records,
                               int32_t *cls_labels, double *boxes_3d,
                               float *valid_boxes, int32_t *owner,
                               void *stream);
/* pgnn_box_encode_f32 evaluated in float64 on float64 boxes and a float64
 * class table, rounded to float32 once -- what train.py:120-130 computes
 * (`box_encoding_fn(cls_labels, xyz, boxes_3d_f64, label_map).astype(f32)`). */
int pgnn_box_encode_f64(const int32_t *cls_labels, const float *xyz,
                        const double *boxes, const double *class_table,
                        int32_t n_table, int64_t n_rows, int32_t boxes_per_row,
                        float *encoded, void *stream);
/* ... with float64 vertices as well (train.py:120-122 in the training path). */
int pgnn_box_encode_f64_xyz64(const int32_t *cls_labels, const double *xyz,
                              const double *boxes, const double *class_table,
                              int32_t n_table, int64_t n_rows,
                              int32_t boxes_per_row, float *encoded,
                              void *stream);

/* Point-wise parts of the training augmentations (models/preprocess.py:44-78
 * random_rotation_all / random_flip_all, :239-326 random_box_shift) on a
 * float64 [n,3] device buffer (the reference's cloud is float64 from the first
 * `xyz.dot(R.T)` until train.py:124).  rot_3x3 / shift_3 / box_record_24 are
 * HOST pointers.  pgnn_points_affine_f64: xyz <- xyz @ rot^T + shift for the
 * points with select[i] != 0 (select null: all; rot null: identity).
 * pgnn_points_in_box_f64: inside[i] (nullable) = 1 when point i is strictly
 * inside the box record (layout of pgnn_assign_box_labels), *count (nullable,
 * device) = number of inside points with exclude[i] == 0 (exclude nullable). */
int pgnn_points_affine_f64(double *xyz, int64_t n_points, const double *rot_3x3,
                           const double *shift_3, const int32_t *select,
                           void *stream);
int pgnn_points_in_box_f64(const double *xyz, int64_t n_points,
                           const double *box_record_24, const int32_t *exclude,
                           int32_t *inside, int32_t *count, void *stream);

/* Streaming classification metrics of the training / evaluation loops
 * (train.py:301-368, eval.py:176-245): per class tf.metrics.recall and
 * tf.metrics.precision of argmax(probs) against the labels, and
 * tf.metrics.auc(labels == c, probs[:, c], num_thresholds, curve='PR',
 * summation_method='careful_interpolation').  `state` is a caller-owned device
 * buffer of pgnn_metrics_state_bytes() bytes, zeroed by the caller to reset
 * (the reference re-initialises its local variables per epoch); it holds int64
 * counters only, so states of several ranks add up with one integer
 * all-reduce.  pgnn_metrics_update folds one batch in; pgnn_metrics_compute
 * writes out[c*3 + {0,1,2}] = recall, precision, PR-AUC of class c (float32
 * arithmetic as TensorFlow's). */
size_t pgnn_metrics_state_bytes(int32_t n_classes, int32_t n_thresholds);
int pgnn_metrics_update(const float *probs, int64_t ld_probs,
                        const int32_t *labels, int64_t n_rows,
                        int32_t n_classes, int32_t n_thresholds, void *state,
                        void *stream);
int pgnn_metrics_compute(const void *state, int32_t n_classes,
                         int32_t n_thresholds, float *out, void *stream);

size_t pgnn_kitti_ingest_workspace_bytes(int64_t n_points);
int pgnn_kitti_cam_points_in_image(
    const float *velo_points, int64_t n_points, const float *velo_to_cam_3x4,
    const double *cam_to_image_3x3, double image_width, double image_height,
    const uint8_t *image_bgr, int64_t image_rows, int64_t image_cols,
    void *workspace, size_t workspace_bytes, float *out_xyz, float *out_attr,
    int32_t attr_dim, int64_t capacity, int32_t *out_count, void *stream);

/* ---- diagnostics (not part of the reference-facing surface) -------------- */
/* Process-wide knobs for benchmarks and tests; see "Conventions".  Keys:
 *   launch shape   scatter_rows_per_wave, scatter_nt, mlp_blocks_per_cu,
 *                  edge_msub, pool_msub, mlp_pool_pct, wgrad_wg_target,
 *                  ws_xcds, ws_prio, ws_pool_pct, ws_chunk, ws_balance, ws_reserve
 *   graph builder  graph_max_wgs (cap on the workgroups of one builder launch;
 *                  every builder kernel strides over its work), graph_lds_pad
 *                  (bytes of dynamic LDS every builder launch asks for without
 *                  using them: such a workgroup cannot be placed beside one of
 *                  the weights-stationary kernels); with ws_reserve they put
 *                  the builder on CUs the fused kernels leave free (measured:
 *                  no net gain; default 0 = off)
 *   kernel choice  b16_force (1: pgnn_edge_mlp_scatter_max_bf16x3_fwd also
 *                  takes lists of a few tiles, where it otherwise answers
 *                  PGNN_E_UNSUPPORTED: the parity tests on small fixtures;
 *                  likewise the f16x2 edge and pooling entries), f16_pool (0:
 *                  pgnn_point_set_pooling_f16x2_fwd declines, i.e. 'f16x2'
 *                  models pool in fp32; 2: it ignores hidden_image: same-box
 *                  A/Bs; default 1),
 *                  mlp_debug bits 2048 / 4096 (edge stage: LDS-tile kernel /
 *                  weights-stationary kernel), 8192 / 16384 (pooling stage),
 *                  1024 (pooling hidden layers through the LDS tile), 32 / 128
 *                  (scatter-max epilogue forms), 512 (4-wave small-row kernel)
 *   ablations      NOT in this library: mlp_debug bits 1 / 2 / 4 (drop gather
 *                  loads / last GEMM / epilogue: wrong results, timing only)
 *                  and graph_debug bit 1 (skip the kd-tree replica) exist only
 *                  in diagnostic builds (-DPGNN_DIAG, tools/build_variant.py);
 *                  the default build rejects them with PGNN_E_INVALID and
 *                  compiles their code out
 * Every key accepted here leaves results bit-identical (tested).
 * Returns 0, or PGNN_E_INVALID for an unknown key / value out of range.     */
int pgnn_set_tunable(const char *key, int value);
/* Device buffer for per-tile cycle stamps of the fused kernels
 * (tools/tile_timeline.py, tools/ws_timeline.py); NULL disables.           */
int pgnn_set_debug_buffer(void *device_ptr);

#ifdef __cplusplus
}
#endif
#endif /* POINTGNN_HIP_H_ */
