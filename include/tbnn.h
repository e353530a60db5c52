/*
 * tbnn.h -- C ABI of the MI355X-native HMC sampler for dense Bayesian neural
 * networks (drop-in for the HMC hot path of alpha-davidson/TensorBNN).
 *
 * The reference has no FFI boundary of its own: its hot path is Python calling
 * TensorFlow-Probability.  Each entry point below
 * names the reference interface (file:line under /root/reference) it replaces.
 * A reference maintainer binds these with ctypes; the stub is shown in
 * INTEGRATION.md and shipped as tensorbnn_amd/_native.py.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, <0 on error
 *     (message via tbnn_last_error(), thread-local).
 *   - caller owns all host buffers, the library owns all device buffers.
 *   - one handle = one chain = one HIP device + one HIP stream.  A handle is
 *     not thread-safe; distinct handles are.
 *   - every call returns after its stream work has completed.
 *   - there is NO CPU fallback: tbnn_create fails when no gfx950 device is
 *     visible.  tbnn_adapter_* is host-only C++ (the reference's adapter is
 *     host-side TF-eager code as well) and works without a GPU.
 *
 * State-vector contract:
 *   theta = concat over dense layers of ( W row-major [out,in], b [out] )
 *   eta   = concat over dense layers of ( loc_w, g_w, loc_b, g_b ), then
 *           sqrt(sd) when the likelihood is TBNN_LIK_GAUSSIAN.
 */
#ifndef TBNN_H
#define TBNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBNN_MAX_LAYERS 16
#define TBNN_ABI_VERSION 3   /* 3: tbnn_lint_status, tbnn_last_transition_path; 2: tbnn_hmc_step_each / _run_each / tbnn_hyper_step_each, tbnn_debug_fused_burst, multi-chain handles (tbnn_create_multi: buffers of set/get_state, hmc_step/run, hyper_step are [chains][...]), tbnn_build_id, tbnn_comm_count, tbnn_hyper_probs_many, tbnn_debug_momentum */

/* activation layer that follows a dense layer
 * (tensorBNN/activationFunctions.py:27-63) */
enum { TBNN_ACT_NONE = 0, TBNN_ACT_RELU = 1, TBNN_ACT_TANH = 2, TBNN_ACT_SIGMOID = 3,
       TBNN_ACT_EXP = 4 /* activationFunctions.py:14-24 */, TBNN_ACT_ELU = 5 /* :66-76 */ };
/* prior family of a dense layer: CauchyDenseLayer (= DenseLayer) layer.py:101,
 * GaussianDenseLayer layer.py:282 */
enum { TBNN_PRIOR_CAUCHY = 0, TBNN_PRIOR_GAUSSIAN = 1 };
/* likelihood.py:63 (Gaussian), :136 (FixedGaussian), :205 (Bernoulli) */
enum { TBNN_LIK_GAUSSIAN = 0, TBNN_LIK_FIXED_GAUSSIAN = 1, TBNN_LIK_BERNOULLI = 2 };

/* kernel selection for the forward+backward pass.  AUTO: a fused shape-specialised MFMA kernel where one covers the network
 * (built in or registered at run time), else the layered run-time-shape MFMA kernels (any architecture); FAST: a fused kernel or
 * an error; GENERIC: the thread-per-row kernel (on-device cross-check) */
enum { TBNN_KERNEL_AUTO = 0, TBNN_KERNEL_GENERIC = 1, TBNN_KERNEL_FAST = 2 };

typedef struct tbnn_layer_desc {
    int32_t in_dim;   /* layer.py:110 inputDims  */
    int32_t out_dim;  /* layer.py:111 outputDims */
    int32_t act;      /* TBNN_ACT_*   */
    int32_t prior;    /* TBNN_PRIOR_* */
} tbnn_layer_desc;

typedef struct tbnn_net_desc {
    int32_t n_layers;
    const tbnn_layer_desc* layers;
    int32_t likelihood;  /* TBNN_LIK_* */
    float fixed_sd;      /* FixedGaussianLikelihood(sd=) likelihood.py:138-141 */
    int32_t kernel;      /* TBNN_KERNEL_* */
    int32_t reserved;
} tbnn_net_desc;

/* per-transition results: what network.train prints/threads through
 * (network.py:410-411 acceptRate; :596-600 prints; paramAdapter.py:219-222 SJD) */
typedef struct tbnn_step_out {
    int32_t accepted;          /* Metropolis decision */
    int32_t n_leapfrog;        /* leapfrog steps executed (= L) */
    float log_accept_ratio;    /* TFP log_accept_ratio (non-finite -> -inf) */
    float accept_prob;         /* lar<0 ? exp(lar) : 1   network.py:410-411 */
    double logp_old;           /* target log-prob at the start state */
    double logp_new;           /* target log-prob at the proposal */
    double kinetic_old;        /* 1/2 |p0|^2 */
    double kinetic_new;        /* 1/2 |p_L|^2 */
    double sjd;                /* sum (new-old)^2 over theta (0 when rejected) */
    float device_us;           /* hipEvent time of the whole transition */
    float fwdbwd_us;           /* mean hipEvent duration of the profiled fused fwd+bwd launches (0: profiling off) */
} tbnn_step_out;

typedef struct tbnn_ctx* tbnn_handle;

const char* tbnn_last_error(void);
int tbnn_abi_version(void);
/* hash of the sources (kernel headers, translation units, this header, compiler flags) the loaded library was built from;
 * profiles/rocprof_kernel_us.json and pmc_traffic.json carry the id of the library they were measured on, and bench.py
 * quotes them only when it matches */
const char* tbnn_build_id(void);
/* what the build-time MFMA hazard check (tensorbnn_amd/hazard_lint.py, through checked_compile.py) did to each kernel unit of the loaded library:
 * "<unit>: listing checked (<n> repaired: <rule>:<count> ..); disassembly clean; ..." -- a library cannot be built without the check */
const char* tbnn_lint_status(void);
/* number of visible HIP devices (<0: error) */
int tbnn_device_count(void);

/* network.__init__/add/setupMCMC state on the device (network.py:19-58,
 * :173-191).  seed/chain_id key the Philox4x32-10 chain RNG that replaces
 * tf.random.set_seed(50) (network.py:562). */
int tbnn_create(const tbnn_net_desc* desc, int device, uint64_t seed, uint32_t chain_id,
                tbnn_handle* out);
/* Several independent chains of one network on ONE device behind one handle (round 4).  SURVEY 8(e) puts one chain on each
 * GPU; small problems -- the reference's own examples: Examples/trainRegression.py:33 has 11 rows -- leave most of a GPU idle
 * and are bound by launch latency, so the per-chain kernels of all n_chains chains run as ONE launch each (gridDim.y = chain).
 * Chain c is bit for bit the chain tbnn_create(desc, device, seed, chain_id + c) would be, at the same step size and leapfrog
 * count for all (they advance in lockstep).  On such a handle
 *   tbnn_set_state / tbnn_get_state take [n_chains][P] floats, tbnn_set_hypers / tbnn_get_hypers [n_chains][H];
 *   tbnn_hmc_run fills outs[chain][epoch], tbnn_hmc_step and tbnn_hyper_step out[chain] (no injected draws, no trace);
 *   tbnn_set_data[_device], tbnn_set_validation, tbnn_forward / tbnn_predict / tbnn_metrics with an explicit theta,
 *   tbnn_forward_many and tbnn_hyper_probs_many work as usual; everything that addresses ONE chain's state (tbnn_logp_grad,
 *   tbnn_hyper_logp_grad, theta == NULL predictions, tbnn_gather_samples, tbnn_set_row_shard) returns an error. */
int tbnn_create_multi(const tbnn_net_desc* desc, int device, uint64_t seed, uint32_t chain_id, int32_t n_chains,
                      tbnn_handle* out);
int tbnn_chain_count(tbnn_handle h);  /* 1 for tbnn_create */
int tbnn_destroy(tbnn_handle h);
int tbnn_param_count(tbnn_handle h);  /* P */
int tbnn_hyper_count(tbnn_handle h);  /* H */
/* name of the kernel variant in use ("fast3<...>", "mid<...>", "wide<...>", "layered<...>", "generic") */
const char* tbnn_kernel_name(tbnn_handle h);
/* which kernels ran the leapfrog steps of the LAST transition: "per-step" (a fused pass + k_update per step) or "trajectory" (small narrow
 * problems: the L steps in one launch, kernels_traj.hpp); "none" before the first.  The choice does not depend on tbnn_set_profiling; a traced
 * transition (trace_logp) always takes the per-step kernels, whose sums run in another order: not bit-equal to the untraced one. */
const char* tbnn_last_transition_path(tbnn_handle h);

/* trainX / trainY staging, network.py:41-45.  X [n,d_in] row-major, Y [n,d_out]. */
int tbnn_set_data(tbnn_handle h, const float* X, const float* Y, int64_t n);
/* same, from device pointers (e.g. a torch tensor's data_ptr on this device) */
int tbnn_set_data_device(tbnn_handle h, const float* dX, const float* dY, int64_t n);

/* network.states / network.hyperStates, network.py:53-56 */
int tbnn_set_state(tbnn_handle h, const float* theta);
int tbnn_get_state(tbnn_handle h, float* theta);
int tbnn_set_hypers(tbnn_handle h, const float* eta);
int tbnn_get_hypers(tbnn_handle h, float* eta);

/* the target closure calculateProbs (network.py:370-392) and its gradient
 * (TF autodiff inside TFP).  theta/eta may be NULL = use the
 * handle's current state.  stat (optional) receives the data-term statistic:
 * sum((y-f)^2) for the Gaussian likelihoods, the log-likelihood for Bernoulli. */
int tbnn_logp_grad(tbnn_handle h, const float* theta, const float* eta, double* logp,
                   float* grad, double* stat);

/* network.predict, network.py:141-171: out is [d_out, n] like the reference. */
int tbnn_forward(tbnn_handle h, const float* theta, const float* X, int64_t n, float* out);

/* One weight transition = InnerStepMain (network.py:368-412):
 * tfp.mcmc.HamiltonianMonteCarlo + sample_chain(num_results=1), call sites
 * network.py:394-408.  p0 (P floats) / log_u (1 float) may be NULL (device
 * Philox draw) or injected for parity tests.  trace_logp (optional, L+1
 * doubles) receives the target log-prob at q_0..q_L. */
int tbnn_hmc_step(tbnn_handle h, float eps, int32_t L, const float* p0, const float* log_u,
                  tbnn_step_out* out, double* trace_logp);

/* n_epochs transitions back to back with fixed (eps, L) and no host
 * round-trip in between (adapter bypassed).  outs: n_epochs records. */
int tbnn_hmc_run(tbnn_handle h, float eps, int32_t L, int32_t n_epochs, tbnn_step_out* outs);

/* Multi-chain handles with EVERY chain at its own step size and leapfrog count -- the reference runs one paramAdapter per chain
 * (network.py:221-235, :603-607), so C chains of it are C schedules.  eps, L: n_chains values.  The chains advance in lockstep for
 * max_c L[c] steps; chain c takes its closing half kick at step L[c] and is skipped from then on (its blocks of the fused pass exit
 * at once), so chain c is bit for bit tbnn_create(.., chain_id + c) driven with (eps[c], L[c]).  out[c].n_leapfrog = L[c].
 * tbnn_hmc_run_each: n_epochs such transitions back to back, outs[chain][epoch].  (On a one-chain handle: arrays of one.) */
int tbnn_hmc_step_each(tbnn_handle h, const float* eps, const int32_t* L, tbnn_step_out* out);
int tbnn_hmc_run_each(tbnn_handle h, const float* eps, const int32_t* L, int32_t n_epochs, tbnn_step_out* outs);
/* the hyper transition of every chain at its own step size eps_h[c] (one dual averaging per chain, network.py:457-469) */
int tbnn_hyper_step_each(tbnn_handle h, const float* eps_h, int32_t L_h, tbnn_step_out* out);

/* One hyper-parameter transition = the HMC part of InnerStepHyper
 * (network.py:414-456); dual averaging (:457-469) stays with the caller. */
int tbnn_hyper_step(tbnn_handle h, float eps_h, int32_t L_h, const float* p0, const float* log_u,
                    tbnn_step_out* out);
/* the hyper target (network.py:416-440) and its gradient w.r.t. eta */
int tbnn_hyper_logp_grad(tbnn_handle h, const float* eta, double* logp, float* grad);

/* checkpoint export: writes theta (P floats) then eta (H floats) to a device
 * buffer of P+H floats (feeds the RCCL all-gather at sample time). */
int tbnn_export_sample_device(tbnn_handle h, float* d_out);

/* the chain RNG, exposed for tests: n standard normals / one log-uniform for
 * (epoch, purpose) exactly as tbnn_hmc_step would draw them. */
int tbnn_debug_draw(tbnn_handle h, uint32_t epoch, uint32_t purpose, int32_t n, float* out_normals,
                    float* out_log_u);
/* diagnostic (tests/test_gpu_properties.py): the momentum the last trajectory of tbnn_hmc_step ended with (P floats: p_L after the
 * closing half kick, kept whether or not the proposal was accepted).  With it a test can run the integrator backwards: from q_L
 * with p0 = -p_L the same L steps must return to q_0 (tfp's leapfrog is time-reversible; call sites network.py:394-408). */
int tbnn_debug_momentum(tbnn_handle h, float* p_out);
/* epoch counter that keys the RNG (incremented by every tbnn_hmc_step) */
int tbnn_set_epoch(tbnn_handle h, uint32_t epoch);
/* record a hipEvent pair around every stride-th fused fwd+bwd launch on the
 * chain's stream (fills fwdbwd_us); stride <= 0 turns it off */
int tbnn_set_profiling(tbnn_handle h, int stride);
/* diagnostic (tools/stamps.py): launches the narrow fused kernel three times at the current state and returns the
 * in-kernel stamps of workgroup 0 of the last launch: out16[0..7] = 100 MHz wall clock at start, prologue done, first
 * tile done, tile loop done, end, cooperative tail done, staging done, tile slabs done; out16[8..15] = the shader clock at
 * the same points.  Narrow ahead-of-time kernels only; the chain state is not touched. */
int tbnn_debug_stamps(tbnn_handle h, uint64_t* out16);
/* measurement (bench.py's roofline): `reps` fused forward+backward passes of the CURRENT state back to back on the chain's stream between ONE
 * pair of events; *us_per_pass = elapsed / reps.  (An event pair around a single launch, tbnn_set_profiling, carries ~3 us of the pair's own
 * cost: 45.4 us where rocprofv3 sees 42.6 at configs[1].)  The chain state is not touched.  Reference: the gradient evaluation of one leapfrog
 * step, network.py:394-408. */
int tbnn_debug_fused_burst(tbnn_handle h, int32_t reps, float* us_per_pass);

/* ---- predictions and metrics over the staged rows: no host traffic but the result.
 * network.__init__ stages the validation set next to the training set (network.py:47-51). ---- */
int tbnn_set_validation(tbnn_handle h, const float* X, const float* Y, int64_t n);
/* network.predict(train=True/False) (network.py:141-171): which = 0 training rows, 1 validation rows;
 * theta NULL = current state; out: host [d_out, n] or NULL (predictions stay on the device). */
int tbnn_predict(tbnn_handle h, int which, const float* theta, float* out);
/* Ensemble prediction, predictor.predict (predictor.py:132-155): the forward pass of m saved
 * networks, theta_i = thetas + i * theta_stride (theta_stride >= P floats), over the same rows.  X NULL: the staged
 * rows selected by `which` (0 training, 1 validation); else n host rows [n, d_in].  out: host [m][d_out][n].
 * Narrow shapes run one batched launch of the forward-only MFMA kernel (grid.y = network). */
int tbnn_forward_many(tbnn_handle h, const float* thetas, int32_t m, int64_t theta_stride, int which, const float* X,
                      int64_t n, float* out);
/* metrics.py:30-141 in one pass over the predictions: with p = f*sd+mean, r = y*sd+mean (exp() of either on
 * request: scaleExp; SquaredError leaves the validation predictions un-exponentiated, metrics.py:44-47)
 *   out3[0] = mean (p-r)^2            SquaredError
 *   out3[1] = mean 100*|(p-r)/r|      PercentError
 *   out3[2] = mean |r - round(p)|     1 - Accuracy */
int tbnn_metrics(tbnn_handle h, int which, const float* theta, float mean, float sd, int exp_pred, int exp_real,
                 double out3[3]);

/* predictor.trainProbs / reweight (predictor.py:157-273): for m saved networks, the sum over the dense
 * layers of calculateHyperProbs(hypers, tensors) (layer.py:199-242, :379-422) in ONE launch (one workgroup per network).
 * theta_i = thetas + i * theta_stride (>= P floats), eta_i = etas + i * eta_stride (>= 4 * layers floats: loc_w, g_w, loc_b,
 * g_b per dense layer); priors: `layers` TBNN_PRIOR_* values to judge the layers under (reweight loads another architecture)
 * or NULL = the chain's own.  out: m doubles (sums in fp64). */
int tbnn_hyper_probs_many(tbnn_handle h, const int32_t* priors, const float* thetas, int64_t theta_stride,
                          const float* etas, int64_t eta_stride, int32_t m, double* out);

/* ---- kernels for shapes outside the ahead-of-time registries.  The shape-specialised MFMA kernels are
 * C++ templates; tensorbnn_amd/jit.py instantiates them for a given network with hipcc at run time
 * (cached .so), and this call hands the result to the library: later tbnn_create calls with
 * TBNN_KERNEL_AUTO / _FAST for that shape use it.  (The reference gets the same effect from
 * tf.function tracing + XLA, network.py:359-362.) ---- */
int tbnn_register_kernel_lib(const char* path);
/* 0: no fused kernel covers the shape (KERNEL_AUTO then runs the layered run-time-shape MFMA kernels); 1 / 2: ahead-of-time narrow / wide MFMA kernels; 3: a registered library */
int tbnn_fused_kernel_available(const tbnn_net_desc* desc);

/* ---- RCCL over xGMI.  librccl.so is resolved with dlopen on the first
 * tbnn_comm_* call: single-GPU use never loads it.  One communicator per chain handle; every
 * collective is enqueued on the chain's own stream (no host sync inside a transition). ---- */
#define TBNN_COMM_ID_BYTES 128
typedef struct tbnn_comm* tbnn_comm_handle;
/* rank 0 draws the id (ncclGetUniqueId); the host hands it to the other ranks (torch.distributed
 * broadcast, MPI, a file ...) */
int tbnn_comm_unique_id(unsigned char id[TBNN_COMM_ID_BYTES]);
/* ncclCommInitRank on the chain's device; collective over all `world` ranks */
int tbnn_comm_create(tbnn_handle h, int world, int rank, const unsigned char id[TBNN_COMM_ID_BYTES],
                     tbnn_comm_handle* out);
/* ranks in the communicator as the collective library reports them (ncclCommCount); < 0: error */
int tbnn_comm_count(tbnn_comm_handle c);
int tbnn_comm_destroy(tbnn_comm_handle c);
/* checkpoint-time gather (network.py:610-663 writes one chain; with N chains every rank ends up
 * with all N samples): all-gather of (theta, eta) = P+H floats per rank.  d_out: device buffer of
 * world*(P+H) floats or NULL (library-owned buffer); host_out: world*(P+H) floats or NULL. */
int tbnn_gather_samples(tbnn_handle h, tbnn_comm_handle c, float* d_out, float* host_out);
/* row-sharded single chain: the ranks of `c` hold disjoint row blocks of (X, Y) and identical
 * theta / eta / seed / chain_id; the data-term gradient (P floats) and the likelihood statistic are
 * all-reduced after every fused pass, so every rank takes the same leapfrog trajectory and the same
 * Metropolis decision.  n_total = rows over all ranks (normaliser of the Gaussian likelihood).
 * c == NULL switches back to the unsharded chain. */
int tbnn_set_row_shard(tbnn_handle h, tbnn_comm_handle c, int64_t n_total);

/* ---- (eps, L) adapter: paramAdapter (tensorBNN/paramAdapter.py:11-292), host C++ ---- */
typedef struct tbnn_adapter* tbnn_adapter_handle;
/* paramAdapter.__init__ :39-93 (k = burnin/averagingSteps, network.py:229-230) */
int tbnn_adapter_create(float e1, int32_t L1, float el, float eu, int32_t eNumber, int32_t Ll,
                        int32_t Lu, int32_t lStep, int32_t m, double k, float a, float delta,
                        int32_t randomSteps, uint64_t seed, tbnn_adapter_handle* out);
int tbnn_adapter_destroy(tbnn_adapter_handle a);
/* paramAdapter.update :199-292.  state: P floats (the new theta).  inject_u
 * (<0: draw) replaces tf.random.uniform :232; inject_e/inject_l (<0: draw)
 * replace random.choice :283-284 with grid indices.  Outputs next (eps, L). */
int tbnn_adapter_update(tbnn_adapter_handle a, const float* state, int32_t P, float inject_u,
                        int32_t inject_e, int32_t inject_l, float* eps_out, int32_t* L_out,
                        float* sjd_out);

#ifdef __cplusplus
}
#endif
#endif /* TBNN_H */
