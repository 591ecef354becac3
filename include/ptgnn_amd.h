/*
 * ptgnn_amd -- MI355X (gfx950 / CDNA4) message-passing core for microsoft/ptgnn.
 *
 * C ABI of libptgnn_amd.so.  The reference (100 % Python) has no FFI of its own for this path;
 * its device work is delegated to `torch` and the third-party `torch_scatter` wheel.  Each entry
 * point below names the reference call site(s) (paths relative to the ptgnn checkout) whose
 * device work it replaces.  INTEGRATION.md shows the ctypes binding a ptgnn maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - the library never allocates, frees or retains user-visible memory: outputs and
 *     workspaces are caller-owned (in the Python host they are torch tensors, so the caching
 *     allocator, streams and graph capture keep working);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); no entry point
 *     synchronises the host;
 *   - return value: 0 = ok, <0 = PTGNN_AMD_E*; ptgnn_amd_last_error() returns a thread-local
 *     message for the last failure on the calling thread;
 *   - fp32 row-major matrices; a leading dimension `ld*` is counted in floats;
 *   - entry points are re-entrant: any number of host threads may call them concurrently, on the
 *     same or on different streams.  What the library keeps per process, and how it is guarded:
 *       * an immutable device-property cache and the per-(kernel, device) dynamic-LDS attribute
 *         (hipFuncSetAttribute, set once under a mutex; not legal inside a stream capture, so the
 *         first use of a streaming kernel has to happen outside one -- inside, the call falls back to
 *         the tile kernel or, where there is none, returns EUNSUPPORTED);
 *       * side streams + fork / join events of the aggregation (plans of >= 2 M edges with hub rows):
 *         one set per (device, caller stream), created on first use outside a capture, looked up and
 *         used under mutexes -- two caller streams never share an event; at most 64 sets, beyond that
 *         the launches stay on the caller's stream;
 *       * launch counters (ptgnn_amd_launch_count): relaxed atomics, never read by the library;
 *       * two TEST / DEVELOPER switches, process-wide on purpose (a torch backward runs on autograd worker threads,
 *         which have to see the setting of the thread that built the graph): ptgnn_amd_set_gemm_mode (which of two
 *         kernel families with identical bits serves the dense blocks) and ptgnn_amd_set_plan_path (which record
 *         format the plan build uses at small sizes).  RESULTS NEVER DEPEND ON EITHER -- they exist so that the
 *         parity tests can drive every kernel family / sort path on every shape; relaxed atomics, safe to flip
 *         between calls, not meant for production code (two models in one process simply share the default:
 *         dispatch by shape).  Environment variables
 *         (PTGNN_AMD_*) are read per call or once at first use; they are developer / test switches.
 *     Caller-owned state with an ownership rule: the plan build's `control` block and the hub
 *     `hub_tickets` counters are ZERO AT REST and belong to the launches of ONE stream at a time --
 *     give every stream that builds plans / aggregates concurrently its own.
 */
#ifndef PTGNN_AMD_H_
#define PTGNN_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden; the declarations of this header are its ONLY exported
 * symbols (`nm -D --defined-only libptgnn_amd.so` lists ptgnn_amd_* and nothing else). */
#pragma GCC visibility push(default)

/* 101: ptgnn_amd_shard_index gained `bad_index_count` (round 4, mid-signature); the fused aggregation + node-update
 * entry point ptgnn_amd_gather_update_f32 was added (round 5).  102: ptgnn_amd_weighted_pool*_f32 (round 6).
 * Callers check ptgnn_amd_version() >= the version their header was compiled against (the Python host does,
 * ptgnn_amd/_lib.py). */
#define PTGNN_AMD_VERSION 102 /* 0.1.2 */

enum {
  PTGNN_AMD_OK = 0,
  PTGNN_AMD_EINVAL = -1,       /* bad argument (null pointer, negative size, bad enum)         */
  PTGNN_AMD_EUNSUPPORTED = -2, /* shape outside what the kernels were built for                 */
  PTGNN_AMD_EHIP = -3,         /* a HIP runtime call or launch failed                           */
  PTGNN_AMD_EWORKSPACE = -4,   /* workspace too small                                           */
  PTGNN_AMD_ERANGE = -5        /* an index was out of [0, num_nodes) (only from *_validate)     */
};

/* reduce: the `aggregation_fn` strings accepted by AbstractMessagePassingLayer._aggregate_messages
 * (ptgnn/neuralmodels/gnn/messagepassing/abstractmessagepassing.py:38-50 -> torch_scatter.scatter):
 * "sum"/"add" = 0, "mean" = 1, "max" = 2, "min" = 3.  Empty segments yield 0 for every mode. */
enum { PTGNN_AMD_SUM = 0, PTGNN_AMD_MEAN = 1, PTGNN_AMD_MAX = 2, PTGNN_AMD_MIN = 3 };

/* Row epilogue fused into the aggregation kernel (MlpMessagePassingLayer,
 * mlpmessagepassing.py:114-117 + :56-58): message activation GELU (exact erf) and LayerNorm. */
enum { PTGNN_AMD_EPI_NONE = 0, PTGNN_AMD_EPI_GELU = 1, PTGNN_AMD_EPI_LAYERNORM = 2,
       PTGNN_AMD_EPI_GELU_LAYERNORM = 3 };

/* activation fused into ptgnn_amd_linear_f32 */
enum { PTGNN_AMD_ACT_NONE = 0, PTGNN_AMD_ACT_TANH = 1, PTGNN_AMD_ACT_RELU = 2 };

int ptgnn_amd_version(void);
const char *ptgnn_amd_last_error(void);

/* TEST / DEVELOPER switch (see "Conventions"): kernel family of the dense blocks (ptgnn_amd_linear_f32,
 * ptgnn_amd_gru_cell*_f32, ptgnn_amd_edge_linear_f32) -- i.e. of what replaces nn.Linear / nn.GRUCell at
 * gatedmessagepassing.py:57-69 and mlpmessagepassing.py:96-117.  Process-wide; initial value from the
 * environment variable PTGNN_AMD_GEMM (default 1).
 *   0  128 x 128 tile kernels, exact fp32 MFMA
 *   1  streaming weight-stationary kernels, exact fp32 MFMA (v_mfma_f32_32x32x2_f32: an fmaf chain)
 * Both produce the same bits (one K accumulation order); shapes the streaming kernels do not tile run on mode 0.
 * (Mode 2 of rounds 2-4 -- f32 emulated by a 3 x bf16 operand split -- was removed in round 5; it answers EINVAL.) */
int ptgnn_amd_set_gemm_mode(int mode);
int ptgnn_amd_get_gemm_mode(void);

/* ------------------------------------------------------------------------------------------
 * Which kernel family served a call is decided per call from shape, size and GEMM mode (the streaming
 * kernels take widths that are multiples of 64 and, for `linear`, enough rows to fill the chip).
 * ptgnn_amd_launch_count(id) returns how many launches of family `id` this process has made
 * (host-side relaxed counters; tests take differences around a call to assert the dispatch);
 * ptgnn_amd_launch_name(id) names it (NULL past the last id).
 * PTGNN_AMD_FORCE_STREAM=1 in the environment (read per call) lifts the SIZE thresholds of the
 * streaming `linear` dispatch, so that small reference fixtures can be replayed on it.
 * ---------------------------------------------------------------------------------------- */
enum {
  PTGNN_AMD_KERNEL_STREAM_LINEAR = 0,
  PTGNN_AMD_KERNEL_STREAM_LINEAR_RING,
  PTGNN_AMD_KERNEL_STREAM_GRU,
  PTGNN_AMD_KERNEL_STREAM_GRU_RING,
  PTGNN_AMD_KERNEL_STREAM_EDGE,
  PTGNN_AMD_KERNEL_STREAM_EDGE_SHARED,
  PTGNN_AMD_KERNEL_STREAM_EDGE_V2,
  PTGNN_AMD_KERNEL_WGRAD_STREAM,
  PTGNN_AMD_KERNEL_TILE_LINEAR,
  PTGNN_AMD_KERNEL_TILE_GRU,
  PTGNN_AMD_KERNEL_TILE_EDGE,
  PTGNN_AMD_KERNEL_TILE_WGRAD,
  PTGNN_AMD_KERNEL_GATHER_UPDATE,
  PTGNN_AMD_KERNEL_COUNT_
};
int64_t ptgnn_amd_launch_count(int kernel_id);
const char *ptgnn_amd_launch_name(int kernel_id);

/* ------------------------------------------------------------------------------------------
 * Graph plan: merge the per-edge-type adjacency lists of one minibatch into ONE
 * destination-sorted CSR that all L message-passing layers of a forward reuse.
 *
 * Replaces: `torch.cat([adj[1] ...])` + the per-type Python loop + the unsorted scatter index
 *   (gatedmessagepassing.py:46,50-64; mlpmessagepassing.py:81-112) -- the adjacency is the same
 *   for every layer (graphneuralnetwork.py:122-131), so the sort is paid once per batch.
 *
 * Input : num_types pairs of int64 device arrays (exactly the tensors ptgnn's
 *         GraphNeuralNetworkModel.finalize_minibatch produces, graphneuralnetwork.py:461-467),
 *         given as HOST arrays of device pointers + HOST array of lengths.
 * Output: rowptr[num_nodes+1]; for CSR slot i (in-edges of node v occupy
 *         rowptr[v]..rowptr[v+1], in the reference's message order = type-major, then edge
 *         order, i.e. the sort is STABLE so fp32 sums fold in the same order as the CPU path):
 *           col[i]  = (src << type_bits) | edge_type     (type_bits = ceil(log2(num_types)))
 *           perm[i] = position of the edge in the type-major concatenation (nullable)
 * `num_src_rows` = number of rows of the table the sources index (0 = num_nodes); it exceeds
 * num_nodes for a dst-range shard whose sources include halo rows (ptgnn_amd/sharded.py).
 * Requires num_src_rows << type_bits < 2^31 and num_edges < 2^31, else EUNSUPPORTED.
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_csr_workspace_bytes(int64_t num_edges, int64_t num_nodes);
size_t ptgnn_amd_csr_control_bytes(void);
/* Test / A-B knob of the plan build (process-wide; results never depend on it): 0 = pick the record format by
 * size (default), 1 = force the 12-byte records + wide buckets of the large-graph path, 2 = additionally force
 * one LSD pre-pass (the > 21 row-bit path) -- so the parity tests can drive every path at sizes the numpy
 * argsort oracle finishes in seconds. */
int ptgnn_amd_set_plan_path(int path);
int ptgnn_amd_type_bits(int32_t num_types);
int ptgnn_amd_csr_build(const int64_t *const *src_per_type, /* host [num_types] of device ptrs */
                        const int64_t *const *dst_per_type, /* host [num_types] of device ptrs */
                        const int64_t *edges_per_type,      /* host [num_types]                */
                        int32_t num_types, int64_t num_nodes, int64_t num_src_rows,
                        int swap_src_dst, /* 0: rows = dst, col = (src << type_bits) | type          *
                                           * 1: rows = src, col = (dst << type_bits) | type          *
                                           * 2: rows = src * num_types + type, col = dst; pass       *
                                           *    num_nodes = source rows * num_types (backward plan)  */
                        int32_t *rowptr, int32_t *col, int32_t *perm /* nullable */,
                        int32_t *max_degree /* nullable device scalar: longest row */,
                        int32_t hub_threshold /* 0 = no hub list */,
                        int32_t *hub_entries /* nullable: int32 [2 * ceil(E/1024)][2] (chunk,row) */,
                        int32_t *hub_count /* nullable device scalar: number of pairs */,
                        int32_t *bad_index_count /* nullable device scalar, ACCUMULATED (never reset here):   *
                                                  * ids outside [0, num_nodes) / [0, num_src_rows) -- the     *
                                                  * reference device-asserts on those in F.embedding          *
                                                  * (gatedmessagepassing.py:54-56); here they are clamped to *
                                                  * row 0 so nothing is read or written out of bounds, and   *
                                                  * counted so the host can raise                            */,
                        void *control /* nullable: ptgnn_amd_csr_control_bytes() of device memory that is ZERO  *
                                       * AT REST: zero-filled once by the caller, used by the builds of ONE   *
                                       * stream at a time, left zero-filled by every build (digit totals and  *
                                       * tile counters of the three-launch build).  Null: the library zeroes  *
                                       * a block inside `workspace` with one extra memset per build          */,
                        void *workspace, size_t workspace_bytes, void *stream);

/* The hub (chunk, row) list of an existing rowptr (plans that are not built by ptgnn_amd_csr_build,
 * e.g. pooling over the already sorted node_to_graph_idx).  *hub_count must be 0 on entry. */
int ptgnn_amd_hub_list(const int32_t *rowptr, int64_t num_rows, int32_t hub_threshold,
                       int32_t *hub_entries, int32_t *hub_count, void *stream);

/* Optional debug aid (the reference performs no range check either): counts indices outside
 * [0, num_nodes) into *bad_count (device int32, caller zeroes it). */
int ptgnn_amd_validate_indices(const int64_t *idx, int64_t n, int64_t num_nodes,
                               int32_t *bad_count, void *stream);

/* ------------------------------------------------------------------------------------------
 * Index bookkeeping of a destination-range shard (north star: "large batched graphs shard by destination-node
 * range across up to 8 GPUs"; SURVEY.md 8e, 8f-3 "halo send lists").  The reference has no counterpart -- its only
 * multi-GPU mode is whole-batch data parallelism (distributedtrainer.py:250-297); the tensors it starts from are
 * the per-type int64 adjacency lists of finalize_minibatch (graphneuralnetwork.py:461-467), here in GLOBAL node ids
 * and restricted to the edges whose destination this rank owns ([lo, hi)).
 *   local_dst[e] = dst - lo;  local_src[e] = src - lo for an own source, else n_local + the source's slot in the
 *   sorted list of DISTINCT remote sources `need_ids` (ascending global ids = grouped by owner, because the
 *   ranges bounds[0] <= ... <= bounds[world] are ordered by rank); edges in the type-major order of the lists.
 *   stats (device int64 [world + 2 + num_types], zeroed here): [0, world) halo rows per owner |
 *   [world] edges with a remote source | [world + 1] halo rows in all | then own-source edges per edge type.
 * need_capacity >= min(num_edges, total_nodes - (hi - lo)).  Nothing synchronises with the host.
 * A source id outside [0, total_nodes) is clamped into the id space and COUNTED in *bad_index_count (after the remap
 * it would be an ordinary own / halo row: the plan build's range guard cannot see it any more); a destination
 * outside [lo, hi) leaves as a local id outside [0, hi - lo), which ptgnn_amd_csr_build counts.
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_shard_index_workspace_bytes(int64_t total_nodes);
int ptgnn_amd_shard_index(const int64_t *const *src_per_type, const int64_t *const *dst_per_type,
                          const int64_t *edges_per_type, int32_t num_types, int64_t lo, int64_t hi,
                          const int64_t *bounds /* device [world + 1] */, int32_t world, int64_t total_nodes,
                          int64_t *local_src, int64_t *local_dst, int64_t *need_ids, int64_t need_capacity,
                          int64_t *stats,
                          int32_t *bad_index_count /* nullable; device int32, NOT zeroed here: += number of
                                                    * global source ids outside [0, total_nodes) (they are
                                                    * clamped; the reference device-asserts on them) */, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Unique (edge type, source) pairs of a forward plan.  A GGNN message is W_t . x[src]
 * (gatedmessagepassing.py:52-58: index_select of the source states, then the type's bias-free Linear): edges of one
 * type that leave the same node carry the SAME message row, so the grouped per-edge GEMM only has to produce one row
 * per pair that occurs and the aggregation reads it through slot_row.  Integer bookkeeping over the plan's col array
 * (ptgnn_amd_csr_build, mode 0); the values of the layer do not change.
 *   unique_src [capacity >= min(num_edges, num_src_rows * num_types)]: source node of every message row -- rows are
 *       type-major, ascending in the source id inside a type (so unique_src + the prefix of counts IS the adjacency
 *       input of ptgnn_amd_edge_linear_f32 for the de-duplicated launch);
 *   counts (device int64 [num_types + 1]): rows per edge type, [num_types] = rows in all;
 *   slot_row (int32 [num_edges]): message row of every CSR slot (the `col` argument of ptgnn_amd_gather_reduce_f32
 *       with type_bits = 0, where the per-edge form passes the plan's perm).
 * Nothing synchronises with the host.
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_unique_sources_workspace_bytes(int64_t num_src_rows, int32_t num_types);
size_t ptgnn_amd_edge_table_bytes(void);
int ptgnn_amd_unique_sources(const int32_t *col, int64_t num_edges, int32_t type_bits, int32_t num_types,
                             int64_t num_src_rows, int32_t *slot_row, int64_t *unique_src, int64_t capacity,
                             int64_t *counts, void *edge_table /* nullable; ptgnn_amd_edge_table_bytes() */,
                             void *workspace, size_t workspace_bytes, void *stream);
/* The grouped per-edge GEMM of ptgnn_amd_edge_linear_f32 (no target-state half) over the message rows of
 * ptgnn_amd_unique_sources: msg[r] = act(W_t x[unique_src[r]]) for the rows r of edge type t.  How many rows each type
 * has is only known on the device, so the launch geometry (rows, 32-row units and workgroups per edge type) is read from
 * `edge_table`, which ptgnn_amd_unique_sources filled on the same stream -- the host never waits for the counts.
 * msg must hold min(num_edges, num_src_rows * num_types) rows.  PTGNN_AMD_EUNSUPPORTED for shapes outside the streaming
 * edge GEMM (ptgnn_amd_edge_linear_shared_supported == 0: the caller keeps the per-edge form). */
int ptgnn_amd_edge_linear_shared_supported(int32_t state_dim, int32_t msg_dim, int32_t num_types);
int ptgnn_amd_edge_linear_shared_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                     const void *edge_table, const float *const *w_per_type, int32_t num_types,
                                     int32_t msg_dim, int act, float *msg, int64_t ld_msg, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused gather -> (+ destination term) -> segment reduce -> (row epilogue).
 *
 *   out[v, :] = EPI( REDUCE_{i in rowptr[v]..rowptr[v+1]}  Ysrc[src_i, t_i*M : (t_i+1)*M]
 *                                                         (+ Ydst[v,   t_i*M : (t_i+1)*M]) )
 *   with (src_i, t_i) unpacked from col[i].
 *
 * Replaces, for one layer: F.embedding gathers, torch.cat, the per-type Linear outputs being
 *   materialised as [E, M], torch.cat of messages/targets and torch_scatter.scatter
 *   (gatedmessagepassing.py:54-68; mlpmessagepassing.py:88-112; abstractmessagepassing.py:44-50),
 *   using  Linear_t(x_src) == (X W_t^T)[src]  (bias-free Linear commutes with the row gather),
 *   and for MLP-MP with target state  W_t [x_u ; x_v] = W_t^s x_u + W_t^d x_v.
 *
 *   ysrc : [num_src_rows, ld_y] ; block t of M columns holds X W_t^T.  With type_bits == 0 and
 *          ld_y == M this is also the plain segment reduce of a materialised message matrix
 *          (col = perm), i.e. the torch_scatter seam itself.
 *   ydst : nullable, same column layout with its own leading dimension, indexed by the
 *          DESTINATION node.
 *   argout: nullable int32 [num_nodes, M]; for max/min the winning CSR slot, -1 if empty
 *          (backward routing; torch_scatter's arg_out).
 *   ln_gamma/ln_beta: LayerNorm affine parameters (EPI_*LAYERNORM only).
 *
 * Hub rows (power-law graphs): rows with more than `hub_threshold` in-edges (0 = never; otherwise
 * >= 2048 and equal to the threshold given to ptgnn_amd_csr_build) are split over 1024-slot chunks:
 * one small extra launch walks the plan's (chunk, row) list `hub_entries`/`hub_count`; the last
 * chunk of a hub to arrive folds the chunk partials in order and applies the epilogue, so one
 * 10^5-edge destination does not serialise on one lane group.  Needs `hub_ws` of
 * ptgnn_amd_hub_workspace_bytes(num_edges, msg_dim, argout != 0) bytes (scratch) and `hub_tickets`,
 * int32[ptgnn_amd_hub_ticket_count(num_edges, msg_dim)] that is ZERO on entry (the kernel leaves it
 * zero again, so one zeroed buffer per plan AND STREAM serves every call on that stream: two streams
 * aggregating over one plan concurrently need a buffer each); with either NULL every
 * row takes the serial path.  Deterministic; a hub row's fp32 fold order differs from the serial
 * order (max/min and argout are exact).
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_hub_workspace_bytes(int64_t num_edges, int32_t msg_dim, int with_arg);
int64_t ptgnn_amd_hub_ticket_count(int64_t num_edges, int32_t msg_dim);
int ptgnn_amd_gather_reduce_f32(const float *ysrc, int64_t ld_ysrc,
                                const float *ydst /* nullable */, int64_t ld_ydst,
                                const int32_t *rowptr, const int32_t *col,
                                int32_t type_bits, int64_t num_nodes, int32_t msg_dim,
                                int reduce, int epilogue, const float *ln_gamma,
                                const float *ln_beta, float ln_eps, float *out, int64_t ld_out,
                                int32_t *argout /* nullable */, int64_t num_edges,
                                int32_t hub_threshold, const int32_t *hub_entries /* nullable */,
                                const int32_t *hub_count /* nullable */, void *hub_ws /* nullable */,
                                size_t hub_ws_bytes, int32_t *hub_tickets /* nullable */,
                                void *stream);
/* The same aggregation for the destination rows [row_begin, row_end) only (every pointer and count keeps its
 * whole-plan meaning; rows outside the range are neither read nor written, hub rows outside it are skipped).  A
 * layer's aggregation -> node update chain (gatedmessagepassing.py:63-69) is row-wise after the aggregation, so a
 * host MAY pipeline it over row ranges (the update of one range on one stream, the aggregation of the next on
 * another): ptgnn_amd.ops.aggregate_gru does, opt-in (PTGNN_AMD_AGG_PIPELINE=<ranges>; off by default -- it measured
 * slower than the unsplit pair on MI355X, profiles/r05_notes.md).  Launches over one plan must not overlap EACH OTHER
 * in time (they share `hub_tickets`). */
int ptgnn_amd_gather_reduce_rows_f32(const float *ysrc, int64_t ld_ysrc,
                                     const float *ydst /* nullable */, int64_t ld_ydst,
                                     const int32_t *rowptr, const int32_t *col,
                                     int32_t type_bits, int64_t num_nodes, int32_t msg_dim,
                                     int reduce, int epilogue, const float *ln_gamma,
                                     const float *ln_beta, float ln_eps, float *out, int64_t ld_out,
                                     int32_t *argout /* nullable */, int64_t num_edges,
                                     int32_t hub_threshold, const int32_t *hub_entries /* nullable */,
                                     const int32_t *hub_count /* nullable */, void *hub_ws /* nullable */,
                                     size_t hub_ws_bytes, int32_t *hub_tickets /* nullable */,
                                     int64_t row_begin, int64_t row_end, void *stream);

/* ------------------------------------------------------------------------------------------
 * MlpMessagePassingLayer, aggregation AND node update in one launch (mlpmessagepassing.py:107-117 with the update
 * network of :56-66):   out[v, :] = act(W . EPI(reduce over the CSR slots i of row v of msg[col[i] >> type_bits, :]) + b)
 * i.e. ptgnn_amd_gather_reduce_f32 (no destination term, no argout) followed by ptgnn_amd_linear_f32, bit for bit,
 * without the [num_nodes, msg_dim] aggregate ever existing in memory.  msg_dim must be 64 (the README's default
 * architecture; BASELINE config 4), out_dim a multiple of 32 up to 128: ptgnn_amd_gather_update_supported.  Every row
 * folds serially in slot order (no hub / long-row launches): meant for minibatch-sized plans.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_gather_update_supported(int32_t msg_dim, int32_t out_dim);
int ptgnn_amd_gather_update_f32(const float *msg, int64_t ld_msg, const int32_t *rowptr, const int32_t *col,
                                int32_t type_bits, int64_t num_nodes, int32_t msg_dim, int reduce, int epilogue,
                                const float *ln_gamma /* nullable */, const float *ln_beta /* nullable */, float ln_eps,
                                const float *w /* [out_dim, msg_dim] */, const float *bias /* nullable */,
                                int32_t out_dim, int act, float *out, int64_t ld_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * reduce = "mul" of the scatter seam: abstractmessagepassing.py:44-50 hands the aggregation name to
 * torch_scatter.scatter, whose reduce set also holds "mul" (no shipped ptgnn configuration uses it).
 *   out[v, :] = product over the CSR slots i of row v of msg[perm[i], :], folded in slot order (= edge
 *   order: the plan's sort is stable); rows without in-edges are 1, like torch_scatter's scatter_mul
 *   (Reducer<MUL>::init(), no masked fill).  msg rows in the type-major message order, `perm` and
 *   `rowptr` from ptgnn_amd_csr_build.  Deterministic; a plain HBM-bound kernel (hub rows are serial).
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_segment_mul_f32(const float *msg, int64_t ld_msg, const int32_t *rowptr, const int32_t *perm,
                              int64_t num_rows, int64_t num_edges, int32_t dim, float *out, int64_t ld_out,
                              void *stream);

/* Backward of the max/min aggregation w.r.t. the message table, over the BACKWARD plan (rows =
 * src * T + type, col = dst, built by ptgnn_amd_csr_build mode 2):
 *   out[r, c] = sum_{i in row r}  [ arg[col_i, c] == slot_of[i] ] * grad[col_i, c]
 * arg = the forward call's argout (winning forward CSR slot per (dst node, feature)); slot_of[i] = the
 * forward slot of backward slot i.  This is torch_scatter's "gradient flows to arg_out only"
 * rule (abstractmessagepassing.py:38-50 through torch_scatter's autograd) without materialising the
 * [E, M] per-edge gradient. */
int ptgnn_amd_gather_reduce_masked_f32(const float *grad, int64_t ld_grad, const int32_t *arg,
                                       const int32_t *rowptr, const int32_t *col,
                                       const int32_t *slot_of, int64_t num_rows, int32_t msg_dim,
                                       float *out, int64_t ld_out, int64_t num_edges,
                                       int32_t hub_threshold, const int32_t *hub_entries,
                                       const int32_t *hub_count, void *hub_ws /* nullable */,
                                       size_t hub_ws_bytes, int32_t *hub_tickets /* nullable */,
                                       void *stream);

/* ------------------------------------------------------------------------------------------
 * y = act(x W^T + b) on the matrix cores with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Linear at gatedmessagepassing.py:20-23,57-61 (all T edge-type weights stacked
 * into one [T*M, H] matrix => ONE wide GEMM instead of T ragged ones), mlp.py:54-75,
 * mlpmessagepassing.py:60-63 (Linear + Tanh), residuallayers.py:112-116.
 *   x [rows, k] (ld_x), w [n_out, k] row-major contiguous (nn.Linear layout), bias nullable.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_linear_f32(const float *x, int64_t rows, int32_t k, int64_t ld_x, const float *w,
                         int32_t n_out, const float *bias, int act, float *y, int64_t ld_y,
                         void *stream);
/* y = act(x W^T + b) + addend, the add folded into the GEMM's store epilogue (training: the GRU cell's backward
 * `d_h + d_gh W_hh`, which torch ran as a separate pass over [N, H] per layer).  Streaming shapes only (K % 64 == 0,
 * n_out % 32 == 0, 16-byte aligned rows): others return PTGNN_AMD_EUNSUPPORTED without touching `y`. */
int ptgnn_amd_linear_add_f32(const float *x, int64_t rows, int32_t k, int64_t ld_x, const float *w, int32_t n_out,
                             const float *bias /* nullable */, int act, const float *addend, int64_t ld_add, float *y,
                             int64_t ld_y, void *stream);

/* ------------------------------------------------------------------------------------------
 * Minibatch assembly on the device (GraphNeuralNetworkModel.extend_minibatch_with / finalize_minibatch,
 * graphneuralnetwork.py:386-493: per-graph `adj + nodes_in_mb_so_far`, np.concatenate, int64 tensors,
 * the node_to_graph_idx generator at :440-443): one launch turns the RAW per-graph int32 arrays
 * (one staging buffer) into every int64 index tensor of the minibatch.
 *   out[i] = (i < n_in ? in[i] : 0) + seg_add[s]   for  seg_start[s] <= i < seg_start[s+1]
 *   in int32 [n_in] device; seg_start int64 [num_segments + 1] (seg_start[0] = 0, last = n_out),
 *   seg_add int64 [num_segments]; out int64 [n_out].  Elements past n_in are "fill" segments
 *   (node_to_graph_idx, reference_node_graph_idx).  Bit-exact integer work.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_batch_offsets_i64(const int32_t *in, int64_t n_in, const int64_t *seg_start,
                                const int64_t *seg_add, int32_t num_segments, int64_t n_out,
                                int64_t *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-edge message GEMM, every edge type in ONE launch (grouped GEMM with gathered A rows):
 *   msg[off_t + e, :] = act( [ x[src_t[e], :] ; x[dst_t[e], :] (if dst_per_type) ] W_t^T ),  e < E_t
 * rows in the reference's message order (type-major, then edge order) = the matrix
 * `torch.cat(all_messages)` of gatedmessagepassing.py:50-64 / mlpmessagepassing.py:81-108, without the
 * F.embedding outputs, the torch.cat copies and the T per-type launches.  Cheaper than the per-node
 * pre-transform when E < N*T (many sparse edge types).  Feed the result to
 * ptgnn_amd_gather_reduce_f32 with col = perm, type_bits = 0.
 *   src/dst_per_type: HOST arrays of int64 device arrays (ptgnn's adjacency tensors); dst nullable.
 *   w_per_type: HOST array of device pointers to the per-type nn.Linear weights
 *               [msg_dim, state_dim * (dst ? 2 : 1)], row-major contiguous.
 * Requires state_dim % 32 == 0, msg_dim % 4 == 0, 16-byte aligned rows (else EUNSUPPORTED).
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_edge_linear_f32(const float *x, int64_t ld_x,
        int64_t num_rows /* rows of x: gathered node ids are clamped into [0, num_rows) */,
        int32_t state_dim,
                              const int64_t *const *src_per_type,
                              const int64_t *const *dst_per_type /* nullable */,
                              const int64_t *edges_per_type, const float *const *w_per_type,
                              int32_t num_types, int32_t msg_dim, int act, float *msg,
                              int64_t ld_msg, void *stream);

/* The same launch with per-edge FEATURE rows behind the gathered halves (layers built with
 * `edge_feature_dimension` / `features_dimension`): gatedmessagepassing.py:57-61
 * `edge_transformation_layer(cat([edge_source_states, features], -1))`, mlpmessagepassing.py:90-98
 * `cat([message_input, features], -1)`, the features coming from graphneuralnetwork.py:162-186:
 *   msg[off_t + e, :] = act( [ x[src_t[e]] ; x[dst_t[e]] (if dst) ; feat_t[e, :feat_dim] ] W_t^T )
 * without materialising the [E, state_dim (+ state_dim) + feat_dim] input matrix.
 *   feat_per_type: HOST array of device pointers, feat_t = [E_t, feat_dim] with row stride ld_feat;
 *   w_per_type[t]: [msg_dim, state_dim * (dst ? 2 : 1) + feat_dim] row-major contiguous.
 * Requires feat_dim > 0, feat_dim % 4 == 0, ld_feat % 4 == 0 and 16-byte aligned feature rows (callers zero-pad odd
 * widths, and the weight columns with them: the padded products are exact zeros) next to the requirements above. */
int ptgnn_amd_edge_linear_feat_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                   const int64_t *const *src_per_type,
                                   const int64_t *const *dst_per_type /* nullable */,
                                   const float *const *feat_per_type, int64_t ld_feat, int32_t feat_dim,
                                   const int64_t *edges_per_type, const float *const *w_per_type,
                                   int32_t num_types, int32_t msg_dim, int act, float *msg,
                                   int64_t ld_msg, void *stream);

/* ------------------------------------------------------------------------------------------
 * Training variants of the per-edge message GEMM (GGNN, gatedmessagepassing.py:57-61:
 * `edge_transformation_layer(self.__dropout(cat([edge_source_states, features])))`).
 *
 * ptgnn_amd_edge_linear_dropout_f32: as ptgnn_amd_edge_linear_f32 (no target-state half, no
 * activation) with nn.Dropout(p) folded in.  The keep mask is a counter-based hash of
 * (dropout_seed, global message row, forward input column) -- see ptgnn_amd/csrc/dense_common.h --
 * so no [E, H] mask tensor exists and the backward kernels regenerate it:
 *   dropout_mode 1: mask the gathered INPUT rows        msg = (mask * x[src] / (1-p)) W_t^T   (forward)
 *   dropout_mode 2: mask the OUTPUT rows                out = (x[src] W_t^T) * mask / (1-p)
 *                   (backward w.r.t. the gathered input: x = d_msg, src = identity index,
 *                    w_per_type = W_t^T [state_dim_fwd, msg_dim_fwd]; here msg_dim = forward state_dim)
 *   dropout_mode 0 or dropout_p == 0: plain ptgnn_amd_edge_linear_f32.
 * The mask is Bernoulli(1 - p') per element with p' = round(p * 65536) / 65536.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_edge_linear_dropout_f32(const float *x, int64_t ld_x,
        int64_t num_rows /* rows of x: gathered node ids are clamped into [0, num_rows) */,
        int32_t state_dim,
                                      const int64_t *const *src_per_type,
                                      const int64_t *edges_per_type, const float *const *w_per_type,
                                      int32_t num_types, int32_t msg_dim, float *msg, int64_t ld_msg,
                                      int dropout_mode, float dropout_p, uint64_t dropout_seed,
                                      void *stream);

/* ------------------------------------------------------------------------------------------
 * The same per-edge dropout with the keep mask as ONE BIT per element (round 4).  Evaluating the hash
 * inside the forward, input-gradient and weight-gradient GEMMs costs ~55 vector instructions per 16
 * MFMAs, and on gfx950 fp32 MFMA shares the vector lanes with the VALU; as bits the mask costs one
 * dword load per 32 columns and 3-4 instructions per element.
 *
 * ptgnn_amd_dropout_bitmask: bits[row * (width / 32) + c], bit b = keep flag of column 32 c + b of message
 *   row `row`, from the SAME hash of (dropout_seed, row, column) the *_dropout_f32 / *_weight_grad_f32
 *   entry points evaluate -- the two families are interchangeable and bit-identical.  width % 32 == 0;
 *   `bits` holds ptgnn_amd_dropout_bitmask_bytes(rows, width) bytes.  One mask serves the forward, the
 *   input gradient and the weight gradient of one layer call (E * H / 8 bytes; the gathered [E, H] input
 *   it replaces is 32 x that).
 * ptgnn_amd_edge_linear_masked_f32: ptgnn_amd_edge_linear_dropout_f32 (modes 1 and 2, same argument
 *   meaning) with `mask_bits` in place of the seed; runs on the fence-free streaming kernel.  Shapes:
 *   ptgnn_amd_edge_linear_masked_supported(state_dim, msg_dim, mode) (state_dim in {64, 128, 256},
 *   msg_dim in {64, 128, 256}); anything else returns EUNSUPPORTED -- use the *_dropout_f32 form there.
 * ptgnn_amd_edge_weight_grad_masked_f32: ptgnn_amd_edge_weight_grad_f32 (no target-state half) with
 *   `mask_bits`; shapes: ptgnn_amd_edge_weight_grad_masked_supported (state_dim % 128 == 0, msg_dim % 32 == 0).
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_dropout_bitmask_bytes(int64_t rows, int32_t width);
int ptgnn_amd_dropout_bitmask(int64_t rows, int32_t width, float dropout_p, uint64_t dropout_seed,
                              uint32_t *bits, void *stream);
int ptgnn_amd_edge_linear_masked_supported(int32_t state_dim, int32_t msg_dim, int dropout_mode);
int ptgnn_amd_edge_linear_masked_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                     const int64_t *const *src_per_type, const int64_t *edges_per_type,
                                     const float *const *w_per_type, int32_t num_types, int32_t msg_dim,
                                     float *msg, int64_t ld_msg, int dropout_mode, float dropout_p,
                                     const uint32_t *mask_bits, void *stream);
int ptgnn_amd_edge_weight_grad_masked_supported(int32_t state_dim, int32_t msg_dim);
int ptgnn_amd_edge_weight_grad_masked_f32(const float *x, int64_t ld_x, int64_t num_rows, int32_t state_dim,
                                          const int64_t *const *src_per_type, const int64_t *edges_per_type,
                                          const float *grad_msg, int64_t ld_grad_msg, int32_t num_types,
                                          int32_t msg_dim, float dropout_p, const uint32_t *mask_bits,
                                          float *grad_w, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Weight gradient of the per-edge message Linear for every edge type in one call (autograd of the
 * nn.Linear at gatedmessagepassing.py:20-23,57-61 / the MLP output layer at mlpmessagepassing.py:96-98):
 *   grad_w[t, m, k] = sum_{e < E_t} grad_msg[off_t + e, m] * in_t[e, k],
 *   in_t[e, :] = [ x[src_t[e], :] ; x[dst_t[e], :] (if dst_per_type) ]  (* dropout mask if dropout_p > 0)
 *   grad_w: [num_types, msg_dim, in_dim] contiguous, in_dim = state_dim * (dst ? 2 : 1); types without
 *           edges get zeros.  Deterministic: chunk partials on fp32 MFMA, then an ordered reduction.
 *   workspace: ptgnn_amd_edge_wgrad_workspace_bytes(total edges, num_types, msg_dim, in_dim) bytes.
 * Requires state_dim % 4 == 0, msg_dim % 4 == 0, 16-byte aligned rows (else EUNSUPPORTED).
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_edge_wgrad_workspace_bytes(int64_t num_edges, int32_t num_types, int32_t msg_dim,
                                            int32_t in_dim);
int ptgnn_amd_edge_weight_grad_f32(const float *x, int64_t ld_x,
        int64_t num_rows /* rows of x: gathered node ids are clamped into [0, num_rows) */,
        int32_t state_dim,
                                   const int64_t *const *src_per_type,
                                   const int64_t *const *dst_per_type /* nullable */,
                                   const int64_t *edges_per_type, const float *grad_msg,
                                   int64_t ld_grad_msg, int32_t num_types, int32_t msg_dim,
                                   float dropout_p, uint64_t dropout_seed, float *grad_w,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* Dense form: grad_w [n_out, k] = grad_y^T . x over `rows` rows, and (grad_b non-null) the bias
 * gradient grad_b [n_out] = column sums of grad_y from the same pass (autograd of nn.Linear /
 * nn.GRUCell parameters, gatedmessagepassing.py:25, mlpmessagepassing.py:60-63).  Workspace:
 * ptgnn_amd_edge_wgrad_workspace_bytes(rows, 1, n_out, k). */
int ptgnn_amd_linear_weight_grad_f32(const float *x, int64_t ld_x, int32_t k, const float *grad_y,
                                     int64_t ld_grad_y, int64_t rows, int32_t n_out, float *grad_w,
                                     float *grad_b /* nullable */, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Training form of the fused GRU cell (autograd of nn.GRUCell, gatedmessagepassing.py:25,69).
 * ptgnn_amd_gru_cell_train_f32: as ptgnn_amd_gru_cell_f32, additionally writes
 *   gates [n, 4*hd] = [ r | z | n | gh_n ]  (gh_n = W_hn h + b_hn) for the backward.
 * ptgnn_amd_gru_cell_backward_gates_f32: the gate math's backward,
 *   d_gi [n, 3hd], d_gh [n, 3hd] (gradients of the two gate pre-activations, gate order r, z, n) and
 *   d_h [n, hd] = grad_out * z (the direct path); the GEMM halves follow with ptgnn_amd_linear_f32 /
 *   ptgnn_amd_linear_weight_grad_f32.  Requires hd % 4 == 0 and 16-byte aligned rows.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_gru_cell_train_f32(const float *a, int64_t ld_a, const float *h, int64_t ld_h,
                                 const float *w_ih, const float *w_hh, const float *b_ih,
                                 const float *b_hh, int64_t n, int32_t m, int32_t hd, float *out,
                                 int64_t ld_out, float *gates, void *stream);
int ptgnn_amd_gru_cell_backward_gates_f32(const float *grad_out, int64_t ld_grad_out, const float *gates,
                                          const float *h, int64_t ld_h, int64_t n, int32_t hd,
                                          float *d_gi, float *d_gh, float *d_h, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the segment reduce w.r.t. the message matrix (the autograd torch_scatter supplies
 * behind abstractmessagepassing.py:44-50):
 *   out[perm[s], :] = grad[slot_row[s], :]                                   sum / mean (pre-scaled)
 *   out[perm[s], c] = arg[slot_row[s], c] == s ? grad[slot_row[s], c] : 0    max / min
 *   slot_row int32 [num_slots]: destination row of CSR slot s (rowptr expanded); arg: the argout of
 *   ptgnn_amd_gather_reduce_f32 ([num_rows, dim], nullable); out [num_slots, dim] message order.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_segment_spread_f32(const float *grad, int64_t ld_grad, const int32_t *arg /* nullable */,
                                 const int32_t *slot_row, const int32_t *perm, int64_t num_slots,
                                 int32_t dim, float *out, int64_t ld_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Row epilogue of the MLP message-passing layer as a differentiable pair (the training step; in inference the
 * same arithmetic is the `epilogue` of ptgnn_amd_gather_reduce_f32):
 *   y = LayerNorm(GELU(x))   with `flags` = PTGNN_AMD_EPI_* naming which of the two apply
 * Replaces: nn.GELU() -> nn.LayerNorm(message_dimension) at mlpmessagepassing.py:20,44-47,114-116 and their autograd
 *   (exact-erf GELU; LayerNorm eps / affine as given; two-pass mean / variance), x, y, grad_* [rows, dim] fp32,
 *   dim <= 512.  The forward equals the fused inference epilogue bit for bit.
 * Backward: grad_x [rows, dim]; grad_gamma / grad_beta [dim] (LayerNorm only; OVERWRITTEN, deterministic: per-
 *   workgroup partial rows in `workspace` (ptgnn_amd_row_epilogue_workspace_bytes) summed in a fixed order).
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_row_epilogue_f32(const float *x, int64_t ld_x, int64_t rows, int32_t dim, int32_t flags,
                               const float *ln_gamma /* nullable without LayerNorm */,
                               const float *ln_beta /* nullable without LayerNorm */, float ln_eps, float *y,
                               int64_t ld_y, void *stream);
size_t ptgnn_amd_row_epilogue_workspace_bytes(int64_t rows, int32_t dim);

/* Backward of the node update's activation + dropout (mlpmessagepassing.py:62-66: Linear -> Tanh -> Dropout) in one
 * pass: out[i] = grad[i] * (keep[i] ? scale : 0) * act'(y[i]) with y = the activation's OUTPUT (tanh': 1 - y^2,
 * relu': y > 0, none: 1), keep = the dropout's boolean mask (one byte per element; NULL = no dropout), scale = 1/(1-p).
 * Contiguous arrays of n elements, n % 4 == 0, 16-byte aligned (else EUNSUPPORTED). */
int ptgnn_amd_act_dropout_backward_f32(const float *grad, const float *y, const uint8_t *keep, float scale, int act,
                                       int64_t n, float *out, void *stream);
int ptgnn_amd_row_epilogue_backward_f32(const float *x, int64_t ld_x, const float *grad_y, int64_t ld_gy,
                                        int64_t rows, int32_t dim, int32_t flags, const float *ln_gamma,
                                        float ln_eps, float *grad_x, int64_t ld_gx,
                                        float *grad_gamma /* nullable without LayerNorm */,
                                        float *grad_beta /* nullable without LayerNorm */, void *workspace,
                                        size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * h' = GRUCell(a, h)  (gate order r, z, n), gatedmessagepassing.py:25,69.
 *   a [n, m], h [n, hd], w_ih [3hd, m], w_hh [3hd, hd], b_ih/b_hh [3hd], out [n, hd].
 *   Gate GEMMs run on fp32 MFMA with the gate non-linearities fused in the epilogue; no
 *   [n, 3hd] intermediates reach HBM.
 * ---------------------------------------------------------------------------------------- */
int ptgnn_amd_gru_cell_f32(const float *a, int64_t ld_a, const float *h, int64_t ld_h,
                           const float *w_ih, const float *w_hh, const float *b_ih,
                           const float *b_hh, int64_t n, int32_t m, int32_t hd, float *out,
                           int64_t ld_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Weighted-sum pooling of elements into samples (nodes into graphs):
 *   out[g, :] = sum_{i : map[i] == g} sigmoid(x[i, :] . w) * x[i, :]
 * Replaces: WeightedSumVarSizedElementReduce.forward, ptgnn/neuralmodels/reduceops/varsizedsummary.py:68-81 --
 *   nn.Linear(D, 1, bias=False) (a gemv), torch.sigmoid, the broadcast multiply that materialises [N, D] and
 *   torch_scatter.scatter_sum -- as used by GruGlobalStateUpdate (globalgraphexchange.py:37-45) in the VarMisuse GGNN
 *   stack (varmisuse/train.py:87-92).  One pass over x; no float atomics: a segment is cut into 128-row chunks counted
 *   from its own start, the chunk partials are added in chunk order -- a fixed function of the segment's rows and their
 *   order, wherever the segment sits in the batch (not the reference's serial fold order: fp32 rounding only).
 *   x [num_elements, dim], w [dim], rowptr int32 [num_segments + 1] / perm int32 [num_elements]: the stable plan of the
 *   element -> sample map (ptgnn_amd_csr_build over (map, map); perm[s] = element of plan slot s), out [num_segments, dim].
 *   dim <= 1024.  workspace: ptgnn_amd_weighted_pool_workspace_bytes.
 * Backward: grad_x [num_elements, dim] and grad_w [dim] (both OVERWRITTEN; grad_w deterministic) from
 *   grad_out [num_segments, dim] and the int64 map itself.
 * ---------------------------------------------------------------------------------------- */
size_t ptgnn_amd_weighted_pool_workspace_bytes(int64_t num_segments, int64_t num_elements, int32_t dim);
int ptgnn_amd_weighted_pool_f32(const float *x, int64_t ld_x, const float *w, const int32_t *rowptr,
                                const int32_t *perm, int64_t num_segments, int64_t num_elements, int32_t dim,
                                float *out, int64_t ld_out, void *workspace, size_t workspace_bytes, void *stream);
size_t ptgnn_amd_weighted_pool_backward_workspace_bytes(int64_t num_elements, int32_t dim);
int ptgnn_amd_weighted_pool_backward_f32(const float *x, int64_t ld_x, const float *w, const int64_t *map,
                                         const float *grad_out, int64_t ld_go, int64_t num_elements, int32_t dim,
                                         float *grad_x, int64_t ld_gx, float *grad_w, void *workspace,
                                         size_t workspace_bytes, void *stream);

/* Row gather out[i, :] = x[idx[i], :] (F.embedding at gatedmessagepassing.py:54-56,
 * mlpmessagepassing.py:88,91) for the general per-edge path (edge features / training dropout /
 * mlp_hidden_layers > 0) and for task heads (output_node_representations[node_idx_references]). */
int ptgnn_amd_gather_rows_f32(const float *x, int64_t ld_x, const int64_t *idx, int64_t n_idx,
                              int32_t dim, float *out, int64_t ld_out, void *stream);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* PTGNN_AMD_H_ */
