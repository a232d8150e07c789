/*
 * rn_potgnn.h -- C ABI of the MI355X (gfx950) PotGNN polarizability evaluator.
 *
 * This is the drop-in boundary for ONE path of wolearyc/ramannoodle: batched evaluation
 * of the PotGNN polarizability model.  The reference has no FFI of its own (it is pure
 * Python); each entry point below names the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, no torch types; all pointers are caller-owned unless stated otherwise.
 *   - every function returns RN_OK (0) or a negative rn_status; rn_potgnn_last_error()
 *     gives a human-readable message for the most recent failure on that handle
 *     (or the most recent rn_potgnn_create failure when the handle is NULL).
 *   - Threading: a handle owns mutable state (device workspaces, streams, the forward tape, the
 *     last error string), so it is NOT re-entrant; every entry point that takes a handle holds the
 *     handle's own lock for the duration of the call, i.e. calls on ONE handle from several threads
 *     are serialised by the library, and calls on DISTINCT handles (also on one device) run
 *     concurrently.  The asynchronous pair rn_potgnn_calc_polarizabilities_async / rn_potgnn_wait
 *     keeps its ordering guarantees per handle.  rn_potgnn_last_error() returns a pointer into the
 *     handle: read it before the next call on that handle from another thread.  The reference's
 *     PolarizabilityModel is single-threaded and synchronous (abstract.py:10-29).
 *     The small getters / setters (config_flags, kernel_times, set_profiling, train_row_count, set_stat_reducer) and the
 *     argument checks that read handle state take the same lock.
 *   - no C++ exception crosses this boundary: every failure is a negative rn_status.
 *   - rn_potgnn_forward* / rn_potgnn_train_forward* evaluate in float32 (the *_f64 entries in float64).  A training
 *     forward's tape lives in the handle's workspace until its backward: an evaluation, Jacobian or another training forward
 *     on the same handle in between voids it, and the backward then fails with RN_ERR_INVALID_ARGUMENT.
 *   - the role-specialised EdgeBlock bounds its internal spin waits; a wait that runs out is reported as a launch failure
 *     by the first entry point that synchronises (evaluation with synchronize != 0, rn_potgnn_wait, the training entries,
 *     the Jacobian, rn_potgnn_adam_step), and the rows the affected workgroup stored are NaN, so an entry point that does
 *     not synchronise never hands back plausible numbers from such a launch.
 *   - "host" entry points take host buffers and include PCIe transfers;
 *     "device" entry points take device (HBM) pointers and a hipStream_t (as void*).
 */
#ifndef RN_POTGNN_H
#define RN_POTGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rn_status {
  RN_OK = 0,
  RN_ERR_INVALID_ARGUMENT = -1, /* mapped to ValueError by the Python wrapper          */
  RN_ERR_UNSUPPORTED = -2,      /* e.g. embedding size > 128                            */
  RN_ERR_NO_DEVICE = -3,        /* no gfx950 device / HIP runtime failure at init       */
  RN_ERR_HIP = -4,              /* a HIP call failed; see rn_potgnn_last_error          */
  RN_ERR_OUT_OF_MEMORY = -5
} rn_status;

typedef struct rn_potgnn rn_potgnn; /* opaque */

/*
 * Model description handed to rn_potgnn_create.  Replaces the state that
 * PotGNN.__init__ builds (ramannoodle/pmodel/torch/_gnn.py:484-539).
 */
typedef struct rn_potgnn_config {
  int32_t num_atoms;            /* N                                                    */
  int32_t num_edges;            /* E  directed edges of the frozen reference graph      */
  int32_t num_atom_types;       /* K  rows of the Embedding (_gnn.py:503,509)           */
  int32_t size_node_embedding;  /* Fn (1..128)                                          */
  int32_t size_edge_embedding;  /* Fe (1..128)                                          */
  int32_t num_message_passes;   /* P                                                    */
  double gauss_coefficient;     /* -0.5/(mu1-mu0)^2 as the reference computed it
                                   (_gnn.py:64); the mu grid itself is the
                                   "_edge_embedding.offset" buffer inside `weights`     */
  int32_t max_chunk_structures; /* 0 = choose automatically; else structures per device
                                   work chunk (workspace is sized from it)              */
  int32_t device;               /* HIP device ordinal                                   */
} rn_potgnn_config;

/*
 * One-time neighbour search of the reference structure on the device; replaces the pair
 * test of _radius_graph_pbc (ramannoodle/pmodel/torch/_utils.py:118-137): float32
 * minimum-image distances, adjacency[a*N + b] = (dist <= cutoff && a != b) as host uint8.
 * Compacting the flags row-major gives the edge list sorted by (a, b).
 */
int rn_potgnn_radius_graph(const double *lattice, const double *positions, int32_t num_atoms,
                           double cutoff, int device, uint8_t *adjacency);

/* Number of floats rn_potgnn_create expects in `weights` for this configuration. */
size_t rn_potgnn_weight_count(const rn_potgnn_config *cfg);

/*
 * Create an evaluator.
 *   edge_a, edge_b  int32[E]  reference-graph edges (a -> b), sorted by (a, b), exactly
 *                             rows 1 and 2 of PotGNN._ref_edge_indexes
 *                             (_gnn.py:492-496, _utils.py:137).  The edge triplets
 *                             (_utils.py:153-168) are a pure function of this list and
 *                             are enumerated on the device; see
 *                             rn_potgnn_debug_triplets.
 *   atom_types      int32[N]  _atom_type_map[atomic_numbers] (_gnn.py:502-506,557)
 *   lattice         f64[3*3]  row-major, rows are lattice vectors (Angstrom)
 *   weights         f32[rn_potgnn_weight_count]  the floating-point entries of
 *                             PotGNN.state_dict() concatenated in state_dict order,
 *                             each tensor row-major in its native torch layout
 *                             (SURVEY.md section 8b lists the keys; the integer
 *                             "num_batches_tracked" entry is skipped)
 *   mean, stddev    f64[3*3]  _mean_polarizability / _stddev_polarizability
 */
int rn_potgnn_create(const rn_potgnn_config *cfg, const int32_t *edge_a,
                     const int32_t *edge_b, const int32_t *atom_types,
                     const double *lattice, const float *weights, size_t num_weights,
                     const double *mean, const double *stddev, rn_potgnn **out);

/* Releases device memory and streams owned by the handle.  NULL is allowed. */
void rn_potgnn_destroy(rn_potgnn *h);

/*
 * Replaces PotGNN.calc_polarizabilities (_gnn.py:667-721).
 *   positions  host f64[S*N*3]  fractional coordinates, C-contiguous, not modified
 *   alpha      host f64[S*3*3]  de-standardised symmetric tensors (alpha*sigma + mu)
 * S may be 0.  Includes H2D/D2H copies (pipelined: see rn_potgnn_calc_polarizabilities_to_device).
 */
int rn_potgnn_calc_polarizabilities(rn_potgnn *h, const double *positions, int64_t S,
                                    double *alpha);

/*
 * The same evaluation (float32 arithmetic, host float64 positions) with the result left on the device: d_alpha device
 * f64[S*9].  The positions are cast to float32 while they are staged into page-locked memory -- the reference casts them
 * before any arithmetic (_gnn.py:709), so the results are bit-identical to a float64 upload at half the PCIe bytes -- and
 * go through one work chunk at a time, cast / copy / kernels of consecutive chunks overlapped (rn_potgnn_calc_polarizabilities
 * does the same and then copies the result down).  Returns once everything is enqueued; `stream` (a hipStream_t, NULL = the null stream) is
 * made to wait for the evaluation, so work the caller enqueues on it afterwards -- the all-gather of a sharded run,
 * dynamics/_trajectory.py:71-90 across ranks -- sees the finished d_alpha.  `positions` may be reused on return.
 */
int rn_potgnn_calc_polarizabilities_to_device(rn_potgnn *h, const double *positions, int64_t S, double *d_alpha,
                                              void *stream);

/*
 * The same evaluation with every kernel instantiated for float64 -- what the reference computes
 * when torch's default dtype is float64 (calc_polarizabilities casts lattice and positions to
 * torch.get_default_dtype(), _gnn.py:705-710, and the modules were built in that dtype).  The
 * float32 master weights are widened exactly; geometry, projections, LayerNorms, gates and the
 * readout run in double.  Roughly 8x slower than the float32 path (no matrix cores, half the
 * frames per launch); meant for validation and for callers that need more than float32 carries.
 */
int rn_potgnn_calc_polarizabilities_f64(rn_potgnn *h, const double *positions, int64_t S,
                                        double *alpha);

/*
 * Pipelined host entry for streamed trajectories: the call enqueues the
 * host-to-device copy of `positions` on the handle's copy stream, the evaluation behind it and the
 * device-to-host copy of the result, and returns without waiting, so that the copy of block k+1
 * (and whatever the caller does meanwhile, e.g. parsing block k+2) overlaps the evaluation of
 * block k.  Two staging slots: a third call first waits for the first one.  `positions` must stay
 * untouched and `alpha` unread until rn_potgnn_wait returns; both should be page-locked
 * (rn_host_buffer_alloc) -- pageable memory works but makes the copies synchronous.
 */
int rn_potgnn_calc_polarizabilities_async(rn_potgnn *h, const double *positions, int64_t S,
                                          double *alpha);
/* Blocks until every rn_potgnn_calc_polarizabilities_async call issued so far has finished. */
int rn_potgnn_wait(rn_potgnn *h);

/* Page-locked host memory for the pipelined entry (hipHostMalloc / hipHostFree). */
int rn_host_buffer_alloc(size_t bytes, int device, void **out);
void rn_host_buffer_free(void *p);

/*
 * Same computation on device-resident buffers (what bench.py times):
 *   d_positions device f64[S*N*3];  d_alpha device f64[S*9] or NULL;
 *   d_vec6 device f32[S*6] or NULL -- standardised (xx,yy,zz,xy,xz,yz), i.e. the value
 *   of PotGNN.forward in eval mode (_gnn.py:617-665).
 * Work is enqueued on `stream` (a hipStream_t; NULL = the null stream) and, when
 * `synchronize` is non-zero, waited for before returning.
 */
int rn_potgnn_forward_device(rn_potgnn *h, const double *d_positions, int64_t S,
                             double *d_alpha, float *d_vec6, void *stream,
                             int synchronize);

/*
 * The device-resident entry with every kernel instantiated for float64 (see
 * rn_potgnn_calc_polarizabilities_f64): d_positions device f64[S*N*3] -> d_alpha device f64[S*9].
 * Used by the sharded phonon path, whose finite differences need float64 and whose results stay in
 * HBM until the all-gather.
 */
int rn_potgnn_forward_device_f64(rn_potgnn *h, const double *d_positions, int64_t S, double *d_alpha,
                                 void *stream, int synchronize);

/*
 * Replaces PotGNN.forward for host callers (_gnn.py:617-665, eval mode):
 * host f64 positions -> host f32[S*6] standardised 6-vectors.
 */
int rn_potgnn_forward(rn_potgnn *h, const double *positions, int64_t S, float *vec6);

/*
 * PotGNN.forward(lattice[S,3,3], atomic_numbers, positions[S,N,3]) with a lattice PER SAMPLE
 * (_gnn.py:603-611: the minimum-image displacements of sample s are mapped to Cartesian with
 * lattice[s]; the graph topology stays the reference structure's): host f64 lattices[S*9]
 * (row-major, rows are lattice vectors), host f64 positions -> host f32[S*6], eval mode.
 */
int rn_potgnn_forward_lattices(rn_potgnn *h, const double *lattices, const double *positions,
                               int64_t S, float *vec6);

/*
 * PotGNN.forward(lattice[S,3,3], atomic_numbers[S,N], positions[S,N,3]) with BOTH per-sample
 * inputs (_gnn.py:617-665): `lattices` host f64[S*9] or NULL (the reference structure's lattice
 * for every sample); `atom_types` host int32[S*N] or NULL (the reference structure's species) --
 * the atom TYPE of every (sample, atom), i.e. PotGNN._atom_type_map[atomic_numbers]
 * (_convert_to_atom_type, _gnn.py:541-557), which selects the row of the node-embedding table
 * (_gnn.py:642-643).  Types outside [0, num_atom_types) are refused (RN_ERR_INVALID_ARGUMENT): the
 * reference's Embedding raises for them.  Graph topology stays the reference structure's.
 * host f64 positions -> host f32[S*6] standardised 6-vectors, eval mode.
 */
int rn_potgnn_forward_samples(rn_potgnn *h, const double *lattices, const int32_t *atom_types,
                              const double *positions, int64_t S, float *vec6);

/*
 * PotGNN.forward in evaluation mode as the reference computes it when the model was built under
 * torch.set_default_dtype(torch.float64): its parameters are then float64 and forward computes in their dtype
 * (_gnn.py:617-665, 493-494).  Every kernel instantiated for double (the float32 master weights widened exactly);
 * lattices f64[S*9] or NULL, atom_types int32[S*N] or NULL as for rn_potgnn_forward_samples; vec6 host f64[S*6],
 * standardised (xx,yy,zz,xy,xz,yz).
 */
int rn_potgnn_forward_samples_f64(rn_potgnn *h, const double *lattices, const int32_t *atom_types,
                                  const double *positions, int64_t S, double *vec6);

/*
 * The same on DEVICE buffers (PotGNN.forward with CUDA tensors: the reference returns its result on the device its
 * inputs live on, _gnn.py:617-665, test/tests/torch/test_gnn.py:130-160): d_lattices device f32[S*9] or NULL,
 * d_atom_types device int32[S*N] or NULL (already validated by the caller: entries in [0, num_atom_types)),
 * d_positions device f64[S*N*3] -> d_vec6 device f32[S*6]; work is enqueued on `stream`.
 */
int rn_potgnn_forward_samples_device(rn_potgnn *h, const float *d_lattices, const int32_t *d_atom_types,
                                     const double *d_positions, int64_t S, float *d_vec6, void *stream, int synchronize);

/*
 * Replaces the finite-difference loop of Phonons.get_raman_spectrum
 * (ramannoodle/dynamics/_phonon.py:93-106) with ONE batched evaluation of the 2M
 * displaced cells in double precision on the device:
 *   raman[m] = (alpha(ref + delta*d_m) - alpha(ref - delta*d_m)) / delta
 * (divides by delta, not 2*delta -- as the reference does).
 *   ref_positions host f64[N*3], displacements host f64[M*N*3], raman host f64[M*9].
 */
int rn_potgnn_raman_tensors(rn_potgnn *h, const double *ref_positions,
                            const double *displacements, int64_t M, double delta,
                            double *raman);

/*
 * Jacobian of the standardised 6-vector (PotGNN.forward, eval mode) with respect to the
 * fractional coordinates at one structure, by reverse-mode differentiation on the device:
 *   jac[k][n][c] = d vec6_k / d x_{n,c},   jac is host f64[6*N*3].
 * use_float64 != 0 evaluates forward and reverse passes in double precision.
 */
int rn_potgnn_alpha_jacobian(rn_potgnn *h, const double *positions, int use_float64, double *jac);

/*
 * Analytic counterpart of rn_potgnn_raman_tensors -- the d(alpha)/d(r) x phonon-eigenvector
 * contraction: raman[m] = 2 * (d alpha / d r)|_ref . d_m   (factor 2: the reference divides
 * its +-delta difference by delta, ramannoodle/dynamics/_phonon.py:106).  One forward and
 * one reverse pass instead of 2M forward passes; differs from the finite difference by
 * O(delta^2).
 */
int rn_potgnn_raman_tensors_analytic(rn_potgnn *h, const double *ref_positions,
                                     const double *displacements, int64_t M, double *raman);

/* ------------------------------------------------------------------ training
 * Replaces the forward/backward of one optimisation step of train_single_epoch
 * (ramannoodle/pmodel/torch/_train.py:63-76); the optimiser itself stays with the caller
 * (torch.optim works on the host copies of the parameters).
 */

/* Replace the parameters of an existing evaluator (same layout as rn_potgnn_create). */
int rn_potgnn_set_weights(rn_potgnn *h, const float *weights, size_t num_weights);

/*
 * PotGNN.forward in TRAINING mode (_gnn.py:617-665 with BatchNorm1d using the statistics of
 * all S*E rows of this batch, _gnn.py:534) on S <= max_chunk_structures frames; keeps the
 * tape for rn_potgnn_train_backward.  Returns the standardised 6-vectors and the batch
 * mean / biased variance of the BatchNorm input (host f32[Fe] each) so that the caller can
 * update running_mean / running_var as torch does.
 */
int rn_potgnn_train_forward(rn_potgnn *h, const double *positions, int64_t S, float *vec6,
                            float *batch_mean, float *batch_var);

/*
 * Gradient of a scalar loss with respect to every parameter, given dL/dvec6 (host f32[S*6])
 * for the batch of the preceding rn_potgnn_train_forward.  `grads` (host f32, length
 * rn_potgnn_weight_count) has the layout of `weights`; entries of buffers
 * (Gaussian offsets, running statistics) are zero.
 */
int rn_potgnn_train_backward(rn_potgnn *h, const float *dvec6, float *grads);

/*
 * rn_potgnn_train_forward with a lattice and / or atom types PER SAMPLE (training-mode PotGNN.forward accepts any
 * lattice[S,3,3] and atomic_numbers[S,N], _gnn.py:603-611, 541-557): `lattices` host f64[S*9] or NULL, `atom_types` host
 * int32[S*N] or NULL, as for rn_potgnn_forward_samples.  The following rn_potgnn_train_backward(_device) differentiates
 * that forward (the embedding gradient is summed per sample's atom types).  _f64: the float64 validation leg.
 */
int rn_potgnn_train_forward_samples(rn_potgnn *h, const double *lattices, const int32_t *atom_types, const double *positions,
                                    int64_t S, float *vec6, float *batch_mean, float *batch_var);
int rn_potgnn_train_forward_samples_f64(rn_potgnn *h, const double *lattices, const int32_t *atom_types,
                                        const double *positions, int64_t S, double *vec6, double *batch_mean,
                                        double *batch_var);

/*
 * Device-resident optimisation (_train.py:63-76 without host round trips).  With device training
 * enabled, rn_potgnn_train_forward also updates the BatchNorm running statistics where the weights
 * live (torch semantics: momentum 0.1, unbiased variance); rn_potgnn_train_backward_device leaves the
 * parameter gradients in HBM (packed layout; rn_potgnn_gradient_buffer exposes the float32 buffer so
 * that data-parallel ranks can all-reduce it in place over RCCL), and rn_potgnn_adam_step applies
 * torch.optim.Adam (no amsgrad; weight_decay is added to the gradient; `step` counts from 1) to the
 * parameters in HBM and recomputes what depends on them.  A step uploads dL/dvec6 (24 B per structure)
 * and downloads S*6 outputs and 2*Fe statistics.  rn_potgnn_get_weights returns the current values
 * in the layout of rn_potgnn_create's `weights` (state_dict order, buffers included).
 */
int rn_potgnn_set_device_training(rn_potgnn *h, int enabled);
int rn_potgnn_train_backward_device(rn_potgnn *h, const float *dvec6);
/*
 * The same step with nothing crossing PCIe (round 4: train_single_epoch moves a batch to the device and takes the loss
 * there, _train.py:51-75): positions device f64[S*N*3], d_lattices device f32[S*9] or NULL, d_atom_types device int32[S*N]
 * or NULL (validated by the caller) -> d_vec6 device f32[S*6]; then the cotangents d_dvec6 device f32[S*6].  Both need
 * device-resident training (rn_potgnn_set_device_training), order their work after `stream` and make `stream` wait for
 * it; neither synchronises.  The gradients stay in HBM for rn_potgnn_adam_step.
 */
int rn_potgnn_train_forward_samples_device(rn_potgnn *h, const float *d_lattices, const int32_t *d_atom_types,
                                           const double *d_positions, int64_t S, float *d_vec6, void *stream);
int rn_potgnn_train_backward_samples_device(rn_potgnn *h, const float *d_dvec6, void *stream);
int rn_potgnn_gradient_buffer(rn_potgnn *h, void **device_ptr, size_t *count);
int rn_potgnn_adam_step(rn_potgnn *h, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int64_t step);
int rn_potgnn_get_weights(rn_potgnn *h, float *weights, size_t num_weights);

/*
 * The same step evaluated in float64 on the device (every kernel of the forward and of the
 * reverse pass has a double instantiation): what the float32 gradients are validated against,
 * next to float64 autograd through the oracle.  vec6 / batch_mean / batch_var / dvec6 / grads
 * are host float64 arrays with the layouts of the float32 entry points.
 */
int rn_potgnn_train_forward_f64(rn_potgnn *h, const double *positions, int64_t S, double *vec6,
                                double *batch_mean, double *batch_var);
int rn_potgnn_train_backward_f64(rn_potgnn *h, const double *dvec6, double *grads);

/*
 * Data-parallel training (gradient all-reduce plus all-reduced BatchNorm batch
 * statistics for single-device parity).  With a reducer installed, rn_potgnn_train_forward /
 * _backward hand it the float64 column sums of the readout BatchNorm (forward: sum z, sum z^2
 * and the row count; backward: sum dy, sum dy*zhat) and continue with what it leaves in
 * `values`: the element-wise SUM over all ranks.  Every rank must take the same number of
 * training steps.  The parameter gradients returned by rn_potgnn_train_backward are those of
 * this rank's rows; averaging them over ranks (what DistributedDataParallel does) gives the
 * gradient of the mean loss over the global batch.  fn == NULL removes the reducer.
 * rn_potgnn_train_row_count: rows the last train_forward's statistics covered (all ranks),
 * for the unbiased running variance.
 */
typedef int (*rn_potgnn_reduce_fn)(double *values, int64_t count, void *ctx);
int rn_potgnn_set_stat_reducer(rn_potgnn *h, rn_potgnn_reduce_fn fn, void *ctx);
double rn_potgnn_train_row_count(const rn_potgnn *h);

/* ------------------------------------------------------------------ introspection */

/*
 * On-device reduction of a polarizability time series to the unpolarised MD Raman spectrum
 * before the laser / Bose-Einstein corrections: replaces the diff, the seven
 * autocorrelations and the seven FFTs of MDRamanSpectrum.measure (ramannoodle/spectrum/_raman.py:
 * 282-297 via calc_signal_spectrum, spectrum/utils.py:76-124) by one batched forward FFT, one
 * weighted power spectrum, one inverse FFT and one length-(S-1) FFT in float64 (hipFFT, loaded
 * on first use: RN_ERR_UNSUPPORTED if it cannot be).  alpha: host float64[S][3][3];
 * intensities: host float64[num_bins] with num_bins = ceil((S-1)/2) - 1, the non-negative
 * frequencies of fftfreq(S-1) without the zero bin (45 a^2 + 7 g^2, same scale as the reference).
 */
int rn_md_raman_intensities(const double *alpha, int64_t S, int device, double *intensities,
                            int64_t num_bins);
/*
 * The same reduction for a time series that is already in HBM (the output of
 * rn_potgnn_forward_device, produced on `stream`): d_alpha is a device float64[S][3][3]; only the
 * num_bins intensities travel to the host (intensities: host float64[num_bins]).  hipFFT plans and
 * work buffers are cached per (device, S) by both entry points.
 */
int rn_md_raman_intensities_device(const double *d_alpha, int64_t S, int device, double *intensities,
                                   int64_t num_bins, void *stream);

/* Introspection: bit 0 = the fused EdgeBlock kernel is in use (float32, Fn and Fe padded to
 * 64); bit 1 = every pass takes the folded-LayerNorm-scale triplet loop; bit 2 = the fused
 * kernels' matrix products run as split-f16 MFMA (default; RN_POTGNN_MFMA=f32 at create time
 * selects the exact-f32 MFMA); bit 3 = the narrow-width kernels (one lane per row; Fn, Fe <= 16 in
 * an instantiated pair, e.g. the documented Fn = 5, Fe = 14) are in use; bit 4 = split-f16 was
 * requested but the range guard refused it (a non-finite weight, or readout hidden activations
 * that the weights allow beyond 3e4): the exact-f32 MFMA instantiations run instead.  Weight
 * matrices of any finite scale are fine: each is prescaled by a power of two into f16's range;
 * bit 7 = the library was built with -DRN_EXPERIMENTS=1 and carries the opt-in round-3 experiment kernels
 * (experiments/ at the repository root); only then can bit 5 (RN_POTGNN_EDGE2=1 at create time: frame-pipelined EdgeBlock,
 * edge_block2_kernel + edge_c2_kernel) or bit 6 (RN_POTGNN_EDGE3=1: twelve-wave EdgeBlock, edge_block3_kernel +
 * edge_c2_kernel) be set.  The product build ignores those knobs.
 * bit 8 = every pass of a float32 evaluation takes the role-specialised fused EdgeBlock (edge_block_ps_kernel,
 * csrc/kernels_edge_ps.hip: producer waves + consumer waves in one 768-thread workgroup per CU; needs bits 0-2;
 * RN_POTGNN_EDGE_PS=0 at create time keeps the per-frame kernel of bit 0).
 * bit 9 = float32 evaluations take the atom-owning fused NodeBlock (node_block_atom_kernel, csrc/kernels_node_atom.hip:
 * tiles of 16 atoms, round r = their r-th in-edges, the gate on the MFMA accumulators; needs bits 0 and 2 and in-degrees
 * even enough that it pays; RN_POTGNN_NODE_ATOM=0 at create time keeps node_block_fused_kernel).
 * bit 10 = float32 evaluations keep the edge embedding in HBM as split-f16 operand pairs ([f16 hi x8][f16 lo x8] per eight
 * columns, hi = f16(x), lo = f16(x - hi): the MFMA operand itself, same 256 B per row) between the geometry kernel, the
 * EdgeBlocks, the NodeBlocks and the readout; needs bits 8 and 9; RN_POTGNN_PAIR_ROWS=0 at create time keeps float32 rows. */
int rn_potgnn_config_flags(const rn_potgnn *h);

/*
 * Host-only (no device is touched): the schedule check the library runs per atom tile before it lets the role-specialised
 * EdgeBlock (csrc/kernels_edge_ps.hip) take a graph.  rb[i], re[i] = first and end source row of destination i of the tile
 * (destinations sorted by atom, rows counted within the tile); back = rounds that may still read the ring when a step
 * rewrites it (2 .. 5 in handles the library creates; 1 .. 8 accepted here); ring_tiles = the ring's capacity in 16-row tiles (8; 7 for the GRAM instantiation).  Returns 1 when
 * the producers' schedule holds (never more than two source tiles per step, no ring slot rewritten under a round in
 * flight), 0 when not, a negative rn_status on bad arguments; *window (optional) = the most tiles one round's source rows span.
 */
int rn_potgnn_debug_ps_schedule(const int32_t *rb, const int32_t *re, int32_t num_destinations, int32_t back,
                                int32_t ring_tiles, int32_t *window);

/* Number of edge triplets T of the frozen graph. */
int64_t rn_potgnn_num_triplets(const rn_potgnn *h);

/*
 * Writes the triplet index arrays exactly as the device kernels enumerate them, in
 * the order and meaning of the 7-tuple BatchTriplets holds (_utils.py:161-168):
 * idx_i, idx_j, idx_k, slot5 (= PyG idx_kj), slot6 (= PyG idx_ji); each host int32[T].
 * Produced by a device kernel that shares the enumeration code with the aggregation
 * kernel, so tests can check bit-exact index parity.
 */
int rn_potgnn_debug_triplets(rn_potgnn *h, int32_t *idx_i, int32_t *idx_j,
                             int32_t *idx_k, int32_t *slot5, int32_t *slot6);

/*
 * Copies one intermediate of the most recent evaluation's LAST chunk to the host,
 * un-padded: stage 0 = unit vectors+distance [rows,4], 1 = node embedding after pass
 * `index` (0 = initial) [S_c*N,Fn], 2 = edge embedding after pass `index` [S_c*E,Fe]
 * (only the most recent pass and pass 0... see DESIGN.md: intermediates are kept only
 * when the handle was created with RN_POTGNN_KEEP_STAGES=1 in the environment),
 * 3 = readout embedding [S_c*E,12].  Returns rows written via *rows.
 */
int rn_potgnn_debug_stage(rn_potgnn *h, int stage, int index, float *out,
                          size_t out_capacity, int64_t *rows, int64_t *cols);

/* Per-kernel device time of the most recent evaluation (HIP events on the handle's
 * stream), accumulated over chunks; names[i] are static strings.  Enabled with
 * rn_potgnn_set_profiling(h, 1).  Returns the number of entries written (<= cap). */
int rn_potgnn_set_profiling(rn_potgnn *h, int enabled);
int rn_potgnn_kernel_times(rn_potgnn *h, const char **names, double *millis,
                           int64_t *launches, int cap);

const char *rn_potgnn_last_error(const rn_potgnn *h);
const char *rn_potgnn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RN_POTGNN_H */
