/*
 * tnco_hip.h -- C ABI of libtnco_hip.so: batched simulated annealing of
 * tensor-network contraction trees on one MI355X (gfx950).
 *
 * Drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to the google-research/tnco checkout).  The reference exposes
 * ONE replica per pybind11 object and is stepped from Python
 * (tnco/app/infinite_memory/sa.py:199-209); this ABI carries MANY replicas per
 * handle and a whole beta schedule per call.  Plain pointers and sizes only;
 * every input is copied at create time and every output is copied into
 * caller-owned buffers (the reference takes its ctree by value and clones the
 * cost model, include/tnco/optimize/infinite_memory/optimizer.hpp:61-67).
 *
 * Threading: a handle is bound to one device and one stream and is not
 * thread-safe; distinct handles are independent.
 *
 * Status codes: 0 ok; TNCO_HIP_EINVAL -> ValueError in the Python shim
 * (std::invalid_argument / std::domain_error in the reference,
 * include/tnco/ctree.hpp:48-50, infinite_memory/optimizer.hpp:77-87);
 * TNCO_HIP_ERUNTIME -> RuntimeError; TNCO_HIP_ENOTIMPL -> NotImplementedError.
 * tnco_hip_last_error() returns the message of the last failing call on the
 * calling thread.
 *
 * Three families of entry points:
 *   the operator   tnco_hip_create / _run / _run_fw / _sync / _get_* / _set_* / _validate / _best / _destroy ...:
 *                  what the reference's Optimizer_<cost>[_<width>] objects do, batched;
 *   helpers        tnco_hip_random_trees / _greedy_trees[_device] / _linear_paths* (the host code either side of the
 *                  path), tnco_hip_comm_* (the one collective of a multi-GPU launch), device queries;
 *   diagnostics    tnco_hip_diag_*: counters, timers and internals the tests, bench.py and tools/ read.  They have
 *                  no reference counterpart and are NOT part of the drop-in contract (a binding of the reference
 *                  needs none of them; INTEGRATION.md's stub uses none).
 */
#ifndef TNCO_HIP_H_
#define TNCO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNCO_HIP_OK 0
#define TNCO_HIP_EINVAL 1
#define TNCO_HIP_ERUNTIME 2
#define TNCO_HIP_ENOTIMPL 3

/* Acceptance rule, include/tnco/optimize/prob/{base,greedy,mh}.hpp. */
#define TNCO_HIP_PROB_BASE 0
#define TNCO_HIP_PROB_GREEDY 1
#define TNCO_HIP_PROB_MH 2

/* cost_type / width_type, include/tnco/globals.hpp:81-117 (float64 / float32
 * only; long double and float1024 are CPU-only in the reference). */
#define TNCO_HIP_F64 0
#define TNCO_HIP_F32 1

typedef struct tnco_hip_ctx* tnco_hip_handle;

/*
 * Problem description = the arguments of
 *   tnco_core.ContractionTree(nodes, inds, dims, check_shared_inds)
 *     (include/tnco/ctree.hpp:59-94),
 *   SimpleCostModel / SimpleSparseIndsCostModel
 *     (include/tnco/optimize/infinite_memory/cost_model/simple.hpp:57-88,
 *      simple_sparse_inds.hpp:51-93), and
 *   Optimizer_<cost>(ctree, cmodel, seed=, disable_shared_inds=)
 *     (include/tnco/optimize/infinite_memory/optimizer.hpp:61-88,
 *      include/tnco/optimize/optimizer.hpp:57-82)
 * for n_replicas replicas at once.  N = 2*n_leaves-1, W = ceil(n_inds/64)
 * (at least 1); bit p of word p/64 of a mask <-> index position p.
 */
typedef struct tnco_hip_desc {
  int32_t n_leaves;            /* leaves occupy node slots [0, n_leaves), root is slot N-1 */
  int32_t n_inds;
  int64_t n_replicas;
  const uint64_t* leaf_masks;  /* [n_leaves][W]; identical for every replica */
  const uint64_t* output_mask; /* [W] legs of the result tensor, or NULL (none) */
  const int32_t* links;        /* per replica [3][N]: left[N], right[N], parent[N]; null = -1.  Host memory,
                                  or device memory of `device` (tnco_hip_greedy_trees_device) */
  int64_t links_stride;        /* int32 elements between replicas; 0 = one tree shared by all */
  const uint64_t* node_masks;  /* optional per replica [N][W] legs of EVERY node (the reference
                                  passes them in); NULL = derive them on the device from the
                                  leaves (rule of tnco/ctree.py:163-189) */
  int64_t node_masks_stride;   /* uint64 elements between replicas; 0 = shared */
  uint64_t dim_uniform;        /* used when dims == NULL */
  const uint64_t* dims;        /* [n_inds] or NULL; all-equal collapses to uniform (ctree.hpp:79-89) */
  const uint64_t* sparse_mask; /* [W] or NULL -> SimpleCostModel */
  uint64_t n_projs;            /* > 0 when sparse_mask != NULL */
  int32_t cost_dtype;          /* TNCO_HIP_F64 | TNCO_HIP_F32 */
  int32_t disable_shared_inds;
  const uint32_t* seeds;       /* [n_replicas] std::mt19937 seeds (already reduced mod 2^32) */
  int32_t device;              /* HIP device ordinal */
  int32_t width_dtype;         /* TNCO_HIP_F32 (the reference's default width_type) | TNCO_HIP_F64 */
  /* Finite width (include/tnco/optimize/finite_width/greedy/optimizer.hpp:72-115 and
   * finite_width/cost_model/simple.hpp:89-95): active when max_width is finite and >= 0; pass
   * NAN / INFINITY for the infinite-memory optimizer. */
  double max_width;
  uint64_t max_number_new_slices; /* finite_width/greedy/optimizer.hpp:226-321 (the reference's app
                                     always passes 0; any value is implemented) */
  const uint64_t* skip_slices;    /* [W] or NULL */
  const uint64_t* slices;         /* [W] per row: the `slices` constructor argument (no initial slicing, no
                                     PRNG draw), or NULL = greedy initial slicing (draws from the PRNG) */
  /* Restoring a batch -- the remaining constructor arguments, which Optimizer.__reduce__ round-trips
   * (tnco/optimize/infinite_memory/optimizer.py:243-245, finite_width/optimizer.py:343-346;
   * include/tnco/optimize/optimizer.hpp:57-77, infinite_memory/optimizer.hpp:61-88,
   * finite_width/greedy/optimizer.hpp:72-115).  As in the reference the caches are rebuilt from the
   * trees and min_total_cost = get_cost(min_ctree[, min_slices]).  All optional (zero / NULL). */
  const int32_t* min_links;       /* per replica [3][N] like `links`: min_ctree.  NULL = the current tree.  Host memory */
  int64_t min_links_stride;       /* int32 elements between replicas; 0 = one tree shared by all */
  const uint64_t* min_slices;     /* [W] per row: min_slices.  NULL = slices */
  int64_t slices_stride;          /* uint64 elements between the replicas' rows of `slices` and of `min_slices`;
                                     0 = one row shared by all */
  const uint32_t* prng_states;    /* [n_replicas][625]: prng_state of every replica (624 words + position, the
                                     numbers of the reference's string seed, optimize/optimizer.hpp:68-71);
                                     replaces `seeds`, which may then be NULL */
} tnco_hip_desc;

/* Replaces the Optimizer_<cost> constructor for a batch: copies inputs, builds
 * CostCache / HyperCache (include/tnco/optimize/infinite_memory/utils.hpp:22-116)
 * and validates every tree (ctree.hpp:101-152).  EINVAL with "Contraction is
 * not valid." / "Precision is too low." as the reference throws. */
int tnco_hip_create(const tnco_hip_desc* desc, tnco_hip_handle* out);

/* Replaces the Python step loop `for beta in betas: prob.beta = beta;
 * opt.update(prob)` (tnco/app/infinite_memory/sa.py:199-209 over
 * Optimizer::update, infinite_memory/optimizer.hpp:90-221): n_steps sweeps on
 * every replica, sweep k using betas[k].  The call copies `betas`, enqueues its
 * kernels and returns without waiting -- neither for them nor for the previous
 * call's (calls queue up on the device); any getter synchronises.  A handle whose
 * replicas do not fill whole rounds of resident workgroups runs every step as two
 * concurrent launches over half of the replicas each (streams of its own, forked
 * from / joined into the handle's stream): same results, no idle tail. */
int tnco_hip_run(tnco_hip_handle h, int prob_kind, const double* betas, int64_t n_steps);

/* Finite-width twin: `opt.update(prob, update_slices=(n % update_slices_every == 0))` for
 * n = step_offset, step_offset + 1, ... (tnco/app/finite_width/sa.py:219-233 over
 * finite_width/greedy/optimizer.hpp:117-390).  update_slices_every <= 0: never re-slice. */
int tnco_hip_run_fw(tnco_hip_handle h, int prob_kind, const double* betas, int64_t n_steps,
                    int64_t update_slices_every, int64_t step_offset);

/* slices / min_slices properties (finite_width/greedy/optimizer.hpp:66-67) of one replica, [W]
 * words each; either may be NULL. */
int tnco_hip_get_slices(tnco_hip_handle h, int64_t replica, uint64_t* slices, uint64_t* min_slices);

/* The same for k replicas at once: slices / min_slices [k][W] (either may be NULL). */
int tnco_hip_get_slices_many(tnco_hip_handle h, int64_t k, const int64_t* replicas, uint64_t* slices,
                             uint64_t* min_slices);

/* Diagnostics of the LAST re-slice of every replica (no reference counterpart; tests pick the replicas whose
 * re-slice took a rare path and compare exactly those with the oracle): how[r] = 1 the cost cache was re-priced
 * (fw_wave_kernel), 0 it was rebuilt in full (or the replica has no slices); n_changed[r] = indices
 * by which the proposed slices differed from the current ones (-1: not recorded -- the single-kernel form, or more
 * than the re-pricing handles).  Either may be NULL.  EINVAL for a handle without re-pricing. */
int tnco_hip_diag_reslice_info(tnco_hip_handle h, int32_t* how, int32_t* n_changed);

/* Diagnostics (no reference counterpart): what the re-slices of this handle did since it was created, in replica
 * re-slices (one replica at the end of one re-slicing sweep, greedy/optimizer.hpp:359-376).  out8[0] launched in
 * the re-priced form (one wavefront per replica); out8[1] of those, left to the full rebuild; why: out8[2] more
 * too-wide tensors / a deeper tree / more candidate legs than the wavefront form lists, out8[3] more changed indices
 * than it re-prices or an index it cannot place, out8[4] a cost outside a double's powers of two; out8[5] launched
 * in the walk + full-rebuild form; out8[6..7] reserved (0).  bench.py reports out8[1] / out8[0]. */
int tnco_hip_diag_fw_stats(tnco_hip_handle h, int64_t* out8);

int tnco_hip_sync(tnco_hip_handle h);

/* total_cost / min_total_cost properties (optimizer.hpp:253-257) of every
 * replica as raw doubles (the reference prints them to a 6-digit Decimal,
 * optimizer.hpp:278-289; callers wanting that format it host-side).
 * Either pointer may be NULL. */
int tnco_hip_get_costs(tnco_hip_handle h, double* total_cost, double* min_total_cost);

/* ctree / min_ctree read-only properties (optimize/optimizer.hpp:207-208) of
 * one replica.  which: 0 current, 1 best-so-far.  masks ([N][W]) may be NULL. */
int tnco_hip_get_tree(tnco_hip_handle h, int64_t replica, int which, int32_t* left,
                      int32_t* right, int32_t* parent, uint64_t* masks);

/* CostCache / HyperCache contents of one replica (what is_valid() rebuilds and
 * compares, optimizer.hpp:223-251); ccost/partial [N], hyper [N][W]; any may
 * be NULL.  The device does not store a HyperCache: hyper[p] = legs(p) & legs(c0) & legs(c1)
 * (infinite_memory/utils.hpp:82-91) is a function of the legs, derived by the kernels where
 * they need it and here on the host. */
int tnco_hip_get_caches(tnco_hip_handle h, int64_t replica, double* ccost, double* partial,
                        uint64_t* hyper);

/* is_valid(atol) for every replica, recomputed from scratch on the device
 * (optimizer.hpp:223-251).  n_bad = number of replicas failing; first_bad = one
 * of them or -1. */
int tnco_hip_validate(tnco_hip_handle h, double atol, int64_t* n_bad, int64_t* first_bad);

/* prng_state property / string-seed constructor (optimize/optimizer.hpp:68-71,
 * 191-195): 624 state words followed by the position, exactly the numbers
 * `oss << std::mt19937` prints. */
int tnco_hip_get_prng(tnco_hip_handle h, int64_t replica, uint32_t* state625);
int tnco_hip_set_prng(tnco_hip_handle h, int64_t replica, const uint32_t* state625);
/* The same for k replicas in one call: states [k][625]; replicas == NULL means replicas 0 .. k-1. */
int tnco_hip_get_prng_many(tnco_hip_handle h, int64_t k, const int64_t* replicas, uint32_t* states625);
int tnco_hip_set_prng_many(tnco_hip_handle h, int64_t k, const int64_t* replicas, const uint32_t* states625);

/* The k replicas of lowest min_total_cost, ascending, ties by replica id
 * (replaces `sorted(results)` of tnco/app/infinite_memory/sa.py:257 for the
 * head of the list).  k <= 2048: selected on the device (per-block bitonic sort + merge passes),
 * only the k pairs are copied back; longer heads sort the replica records on the host. */
int tnco_hip_best(tnco_hip_handle h, int64_t k, double* costs, int64_t* replicas);
/* min over the replicas of min_total_cost, reduced on the device into *device_dst_f64 (device
 * memory, e.g. a torch tensor's storage): the operand of the RCCL all-reduce(min) that replaces
 * the `sorted(results)[0]` of sa.py:257 across GPUs -- no host round trip. */
int tnco_hip_min_cost_device(tnco_hip_handle h, void* device_dst_f64);
/* ctree / min_ctree (optimize/optimizer.hpp:207-208) of k replicas in ONE buffer, and
 * get_contraction (include/tnco/utils.hpp:53-71) of each, computed on the device:
 * links [k][3][N] (left, right, parent); contraction [k][n_leaves-1][3] = (child0, child1, node)
 * per internal node in the post-order of utils.hpp:34-51, or NULL. */
int tnco_hip_get_trees(tnco_hip_handle h, int64_t k, const int64_t* replicas, int which, int32_t* links,
                       int32_t* contraction);
/* ContractionTree.path() (tnco/ctree.py:350-388) for k contractions of one component (host code):
 * tensors_pos[nc] = ascending positions of the component's tensors among all n_tensors
 * (`_tensors_pos`, ctree.py:128-132); contraction as tnco_hip_get_trees returns it; paths
 * [k][nc-1][2] linear (einsum) format, operand positions in the order (child0, child1). */
int tnco_hip_linear_paths(int32_t n_tensors, int32_t nc, const int32_t* tensors_pos, int64_t k,
                          const int32_t* contraction, int32_t* paths, int32_t n_threads);
/* merge_contraction_paths (tnco/utils/tn.py:334-401) for k results (host code): triples [k][steps][3]
 * = the components' contractions concatenated, in SSA form over all tensors (leaves 0..n_tensors-1,
 * step s creates id n_tensors + s); paths [k][steps][2], each pair sorted. */
int tnco_hip_linear_paths_ssa(int32_t n_tensors, int32_t steps, int64_t k, const int32_t* triples,
                              int32_t* paths, int32_t n_threads);

/* Work counters summed over replicas: move evaluations (iterations of the
 * while loop at optimizer.hpp:117-192), accepted moves, best-tree updates, and
 * moves whose (D, E) order was drawn at random (optimize/optimizer.hpp:135-141).
 * Any pointer may be NULL. */
int tnco_hip_diag_counters(tnco_hip_handle h, uint64_t* moves, uint64_t* accepted,
                          uint64_t* improved, uint64_t* random_picks);
/* Number of whole-tree copies taken for min_ctree (summed over replicas).  The
 * reference copies the tree on EVERY improvement (optimizer.hpp:198-201); this
 * build journals rotations and copies only after a journal overflow. */
int tnco_hip_diag_full_copies(tnco_hip_handle h, uint64_t* n);
/* Per replica move counter ([n_replicas]). */
int tnco_hip_diag_moves(tnco_hip_handle h, uint64_t* moves_per_replica);
/* Diagnostic (no reference counterpart): shader cycles per stage of the sweep
 * loop summed over replicas -- [mt19937, state branches, landing fence, store
 * phase, loop iterations].  All zero unless the library was built with
 * -DTNCO_PROFILE (`make -C tnco_amd/csrc profile`, tools/stage_cycles.py); with a finite-width
 * handle the slots are [too-wide counts, post-order, get_slices, rebuild + commit, re-slices]. */
int tnco_hip_diag_stage_cycles(tnco_hip_handle h, uint64_t* out5);

/* Device time of the kernels launched by tnco_hip_run / tnco_hip_run_fw since the last reset (HIP
 * events on the handle's streams), and the number of schedule chunks launched (one per call unless the
 * schedule is very long).  A handle that splits its steps over two streams (tnco_hip_diag_launch_groups() > 1)
 * reports the time from the first launch after the reset to the end of the last one: its concurrent
 * launches overlap, their durations do not add up. */
int tnco_hip_diag_kernel_time(tnco_hip_handle h, double* ms, int64_t* launches, int reset);
/* The same time split by kernel: [0] sa_run_kernel (Optimizer::update, infinite memory),
 * [1] the moves of the finite-width optimizer (finite_width/greedy/optimizer.hpp:130-331), [2] its re-slice
 * (:359-389: get_slices, the cost cache rebuilt or re-priced, the end of the sweep), [3] what orders the too-wide
 * tensors for get_slices (greedy/utils.hpp:62: fw_walk2_kernel; 0 when the whole re-slice of a
 * replica is one wavefront of fw_wave_kernel: counted under [2]); launches4 = launches of each -- for a finite-width
 * handle that runs its halves on two streams, launches PER STREAM (every event is one launch on each stream; the counts
 * are whole numbers because both streams launch the same sequence).  Either array ([4]) may be NULL. */
int tnco_hip_diag_kernel_times(tnco_hip_handle h, double* ms4, int64_t* launches4, int reset);

/* Concurrent launches a step of tnco_hip_run is split into (1, or 2 when there are more workgroups than resident ones: see
 * tnco_hip_run).  0: the handle keeps its trees in LDS during a launch (csrc/sa_small.h; the fast cost path only: uniform
 * power-of-two dims, float64, no sparse legs, no best trees handed in) -- without hyper-indices trees of up to 64 leaves and
 * 128 indices whatever the number of replicas, up to 128 leaves while the CUs hold the replicas in two rounds of 32 per CU, and
 * any tree of up to 1024 indices (at most 32 per tensor) when the whole batch fits the CUs' LDS at once (512 leaves: 512
 * replicas).  There are no workgroups of node blocks to split.  Same results either way, bit for bit. */
int tnco_hip_diag_launch_groups(tnco_hip_handle h);

/* Bytes of device memory held by the handle. */
int64_t tnco_hip_diag_device_bytes(tnco_hip_handle h);

/* Use an existing hipStream_t (e.g. torch's current stream) instead of the
 * handle's own. NULL restores the private stream. */
int tnco_hip_set_stream(tnco_hip_handle h, void* hip_stream);

void tnco_hip_destroy(tnco_hip_handle h);
/* The device memory of a destroyed handle (blocks of 1 MB and more) is kept for the next tnco_hip_create
 * of the process -- a job that builds optimizer after optimizer asks for the same sizes again, and
 * hipMalloc right after hipFree of ~10 GB sporadically took seconds (csrc/dev_cache.h; bounded by
 * TNCO_HIP_CACHE_MB, default min(an eighth of the device's memory, 32 GB), 0: off; emptied when ANY device
 * allocation of this library fails, which is then retried).  Other allocators of the process (PyTorch,
 * RCCL) cannot reach into it: CALL tnco_hip_release_cached BEFORE HANDING THE GPU TO ANOTHER ALLOCATOR.
 * tnco_hip_release_cached gives everything back now; tnco_hip_diag_cached_bytes: how much is held. */
void tnco_hip_release_cached(void);
uint64_t tnco_hip_diag_cached_bytes(void);

/* Host helper replacing the per-run call
 * get_random_contraction_path(...) + ContractionTree(path, ...)
 * (tnco/app/infinite_memory/sa.py:173-190, tnco/utils/tn.py:109-273,
 * tnco/ctree.py:108-251) for a batch: one seeded random initial tree per
 * replica for ONE connected component, written as links [R][3][N].
 * holders_off[n_inds+1] / holders[] = CSR list of the (ascending) leaves
 * holding each index.  Algorithm documented in tnco_amd/ctree.py
 * random_contraction (random Kruskal over a seeded index permutation). */
int tnco_hip_random_trees(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                          const int32_t* holders, int64_t n_replicas, const uint32_t* seeds,
                          int32_t* links_out, int32_t n_threads);

/* The same batch as the REFERENCE draws it (tnco/utils/tn.py:189-230): CPython's
 * Random(seed).shuffle of the component's tensors, then opt_einsum's greedy path finder with every
 * dimension 2 (restated: opt_einsum is not pinned by the reference and absent here -- "parity
 * unpinned"; spec in tnco_amd/ctree.py greedy_contraction).  output_mask ([W] or NULL): the output
 * indices held by at most one tensor (tn.py:175-178 drops the others).  draws ([n_replicas] or
 * NULL): in = 32-bit outputs of Random(seed) already consumed by earlier components of the same run
 * (the reference shares one generator over the components, tn.py:163), out = consumed after this
 * one. */
int tnco_hip_greedy_trees(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                          const int32_t* holders, const uint64_t* output_mask, int64_t n_replicas,
                          const uint32_t* seeds, uint64_t* draws, int32_t* links_out,
                          int32_t n_threads);

/* The same trees drawn ON THE DEVICE (csrc/greedy_device.hip: CPython's generator one lane per tree,
 * the greedy path finder one wavefront per tree -- over a multigraph held in LDS where the network has no
 * hyper-index, over index sets in memory otherwise); the trees come back in links_out (host memory).
 * Networks outside the kernel's limits (tnco_hip_diag_greedy_device_supported == 0: more than 2040
 * indices or 2000 tensors, an index held by more than 6 tensors) and single trees that end in outer
 * products are done by tnco_hip_greedy_trees on n_threads host threads: same result either way.
 * links_out (host, may be NULL) receives the trees; links_device (may be NULL) receives a pointer to
 * the same [n_replicas][3][N] array in DEVICE memory, which tnco_hip_create takes as tnco_hip_desc.links
 * without a copy or a host-side check (it validates the trees on the device).  That memory belongs
 * to the library: it stays valid until the next tnco_hip_greedy_trees_device call or
 * tnco_hip_greedy_device_release(). */
int tnco_hip_greedy_trees_device(int32_t device, int32_t n_leaves, int32_t n_inds, const int32_t* holders_off,
                                 const int32_t* holders, const uint64_t* output_mask, int64_t n_replicas,
                                 const uint32_t* seeds, uint64_t* draws, int32_t* links_out,
                                 int32_t** links_device, int32_t n_threads);
/* device -> host copy of such a buffer */
int tnco_hip_copy_to_host(void* dst, const void* device_src, uint64_t bytes);
int tnco_hip_diag_greedy_device_supported(int32_t n_leaves, int32_t n_inds, const int32_t* holders_off);
/* diagnostics: the 36-bit key of the cost 2^a - 2^b - 2^c by which the device generator orders its
 * candidates (larger cost <=> larger key; a, b, c <= 2040) */
uint64_t tnco_hip_diag_greedy_cost_key(int32_t a, int32_t b, int32_t c);
/* the device memory tnco_hip_greedy_trees_device keeps between calls (one block, re-used) is freed */
void tnco_hip_greedy_device_release(void);
/* diagnostics: trees of the last device call that the host version did (-1: the whole batch) */
int64_t tnco_hip_diag_greedy_device_redone(void);

/*
 * The exchange between the GPUs of one node -- replaces what little tnco/parallel.py:330-341 moves between its worker
 * processes (the results; here: the best cost, the heads of the result lists).  One process per GPU; RCCL over xGMI,
 * loaded with dlopen from the ROCm installation, i.e. on this library's own HIP runtime.  Rendezvous: rank 0 calls
 * tnco_hip_comm_unique_id and hands the 128 bytes to the other ranks by any means (tnco_amd/parallel.py: over TCP,
 * on the side channel the ranks share); then every rank calls tnco_hip_comm_init (collective).
 */
typedef struct tnco_hip_comm_s* tnco_hip_comm;
int tnco_hip_comm_unique_id(uint8_t* id128);
int tnco_hip_comm_init(int rank, int world, const uint8_t* id128, int device, tnco_hip_comm* out);
void tnco_hip_comm_destroy(tnco_hip_comm c);
/* min over the ranks of the best min_total_cost (replaces `sorted(results)[0]`, sa.py:257, across GPUs): the handle's
 * replicas are reduced on the device straight into the operand of ncclAllReduce(ncclMin) -- 8 bytes over xGMI, no host
 * round trip before the collective.  h == NULL: `local` is this rank's value instead. */
int tnco_hip_comm_allreduce_min(tnco_hip_comm c, tnco_hip_handle h, double local, double* out_min);
/* every rank's `bytes` bytes to every rank, in rank order: recv holds world * bytes */
int tnco_hip_comm_allgather(tnco_hip_comm c, const void* send, void* recv, uint64_t bytes);
int tnco_hip_comm_barrier(tnco_hip_comm c);
const char* tnco_hip_comm_last_error(void);

/* "name|pci ...|uuid ...|N CUs" of a device: what a multi-GPU bench line lists per rank */
int tnco_hip_device_name(int device, char* buf, int cap);
int tnco_hip_device_count(void);
const char* tnco_hip_last_error(void);
const char* tnco_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TNCO_HIP_H_ */
