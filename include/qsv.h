/*
 * libqsv -- MI355X (gfx950) statevector + Pauli-expectation backend for the QUEASARS circuit-evaluation path.
 *
 * C ABI.  Plain pointers and sizes only; no Python, torch or C++ types cross this boundary.
 *
 * The reference (DLR-RB/QUEASARS, pure Python) has no FFI for this path: its boundary is the Python protocol
 *     BaseCircuitEvaluator.evaluate_circuits(circuits, parameter_values) -> list[float]   and   .n_qubits
 *     (reference: queasars/circuit_evaluation/circuit_evaluation.py:62-87)
 * whose implementations hand (circuit, operator, parameter values) "pubs" to a Qiskit primitive
 *     OperatorCircuitEvaluator.evaluate_circuits        (circuit_evaluation.py:200-215, estimator branch)
 *     OperatorSamplerCircuitEvaluator.evaluate_circuits (circuit_evaluation.py:147-157, sampler branch)
 *     measure_quasi_distributions                       (circuit_evaluation.py:29-59)
 * Each entry point below says which of those calls it replaces.  INTEGRATION.md shows the ctypes stub a
 * QUEASARS maintainer would add.
 *
 * Conventions
 *   - little endian qubits: qubit q is bit q of a basis-state index (reference: queasars/utility/pauli_strings.py:38-40)
 *   - a Pauli term is (x_mask, z_mask, coeff): factor on qubit q is I/X/Z/Y for (x,z) bit pair 00/10/01/11
 *   - gates: id, u(theta,phi,lam) and cu3(theta,phi,lam) with Qiskit's matrix definitions
 *     (reference: queasars/minimum_eigensolvers/evqe/quantum_circuit/quantum_gate.py:78-79, :96-102, :157-165)
 *   - every function returns 0 on success, a negative QSV_E_* code otherwise; qsv_last_error() gives text
 *   - the caller owns every host array it passes in or receives results in; the library owns all device memory
 *   - a handle may be used from several host threads; calls on one handle are serialised internally
 */
#ifndef QSV_H
#define QSV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qsv_handle qsv_t;

enum { QSV_OP_ID = 0, QSV_OP_U = 1, QSV_OP_CU3 = 2 };
enum { QSV_F64 = 0, QSV_F32 = 1 };
enum { QSV_NO_CONTROL = 0xFF };

enum {
    QSV_OK = 0,
    QSV_E_ARG = -1,      /* bad argument */
    QSV_E_DEVICE = -2,   /* HIP runtime error (no device, out of memory, launch failure) */
    QSV_E_STATE = -3,    /* call order (e.g. expectation requested before an operator was set) */
    QSV_E_UNSUPPORTED = -4
};

/* One decomposed circuit instruction.  An angle is params[p_x] when p_x >= 0, else the literal. 40 bytes. */
typedef struct qsv_op {
    uint8_t kind;    /* QSV_OP_* */
    uint8_t target;  /* qubit the 2x2 matrix acts on */
    uint8_t control; /* control qubit for cu3, QSV_NO_CONTROL otherwise */
    uint8_t flags;   /* reserved, 0 */
    int32_t p_theta, p_phi, p_lambda;
    double theta, phi, lambda;
} qsv_op;

/* Tuning knobs of the pass scheduler (0 = library default). */
typedef struct qsv_plan_config {
    int32_t tile_bits; /* k: qubits resident on chip per pass (per workgroup tile of 2^k amplitudes) */
    int32_t reg_bits;  /* r: qubits held in each thread's registers at a time (2^r amplitudes per thread) */
    int32_t low_bits;  /* c: lowest qubits always kept in the tile so global accesses stay coalesced */
    int32_t group;     /* circuits evaluated per launch group (0 = size the group to the Infinity Cache) */
    int32_t exchange;  /* LDS transpose: 1 whole complex element per access, 2 re/im planes both resident,
                          3 re then im through one plane buffer (half the LDS); fp32 always uses 1 */
} qsv_plan_config;

/* Counters of the most recent qsv_eval_* call (timings need qsv_set_profiling(h, 1)). */
typedef struct qsv_profile {
    uint64_t n_evals;          /* circuit evaluations in the call */
    uint64_t n_pass_launches;  /* launches of the gate-pass kernel */
    uint64_t n_state_passes;   /* sum over launches of states swept (launch x circuits in its group) */
    uint64_t n_gates;          /* non-identity gates applied */
    uint64_t state_bytes;      /* algorithmic state bytes of the gate-pass launches: 16 * 2^n per state and direction a
                                  fused pass design has to move (pass 0 only writes, a fused last pass only reads) */
    double pass_ms;            /* device time of the gate-pass launches, summed over pushes (HIP events on the stream
                                  each push runs on; pushes on the two streams overlap, so this can exceed wall time) */
    double expect_ms;          /* device time of expectation / reduction kernels */
    double total_ms;           /* device time of the whole call, first launch to last */
    double pass_window_ms;     /* wall-clock window from the first gate-pass launch to the end of the last one */
    uint64_t moved_bytes;      /* state bytes the launches really moved: less than state_bytes when a compact first pass
                                  replaced the state round trip between the first two passes by a small table */
    /* The kernels of the hot path; with profiling on, every launch is bracketed by HIP events on the stream it runs
       on.  [0] = the synthesising first pass of the gate-pass kernel (writes only; also the one-tile virtual circuits
       of split evaluations), [1] = every later pass (the last one fuses the diagonal expectation and then only
       reads), [2] = the contraction kernel of split evaluations (reads the diagonal table once, forms the amplitudes
       from two small tables).  bytes = algorithmic state bytes at the pass's own price (16 * 2^n per state and
       direction it has to move; for [2] what the contraction reads per state: the diagonal table, 8 * 2^n, and the two
       side tables), moved = what a launch really moves (compact tables instead of states; = bytes for [2]), flops = 24
       per amplitude pair a pass updates (4 multiplications + 10 fused multiply-adds), for [2] 8 J + 5 per amplitude
       (J product terms). */
    uint64_t kernel_launches[3];
    double kernel_ms[3];
    uint64_t kernel_bytes[3];
    uint64_t kernel_moved_bytes[3];
    double kernel_flops[3];
    uint64_t kernel_states[3]; /* states swept, summed over launches */
} qsv_profile;

/* ---- lifetime ---------------------------------------------------------------------------------- */

/* Create an evaluator for n_qubits on HIP device `device`.  Replaces constructing a Qiskit primitive. */
int qsv_create(int n_qubits, int dtype, int device, const qsv_plan_config* cfg /* may be NULL */, qsv_t** out);
void qsv_destroy(qsv_t* h);
/* Text of the last error on this handle (or of the last failed qsv_create when h is NULL). */
const char* qsv_last_error(const qsv_t* h);
/* Launch on an existing HIP stream (hipStream_t passed as void*); NULL = the library's own stream. */
int qsv_set_stream(qsv_t* h, void* hip_stream);
int qsv_n_qubits(const qsv_t* h);

/* ---- operator ---------------------------------------------------------------------------------- */

/*
 * Set the observable H = sum_k (coeff_re[k] + i coeff_im[k]) P_k.
 * Replaces passing `operator` in each pub (circuit_evaluation.py:204-208).  An operator whose terms are all
 * I/Z takes the diagonal fast path (one table D[i] = sum_k c_k (-1)^popcount(i & z_k) built once on device).
 */
int qsv_set_operator(qsv_t* h, int n_terms, const uint64_t* x_mask, const uint64_t* z_mask,
                     const double* coeff_re, const double* coeff_im);

/* ---- circuits ---------------------------------------------------------------------------------- */

/* Register a circuit structure once; evaluations then send only parameter values. */
int qsv_circuit_create(qsv_t* h, int n_ops, const qsv_op* ops, int n_params, int* out_circuit_id);
/* The same for many structures at once (circuit i = ops[op_offsets[i] .. op_offsets[i+1]), n_params[i] parameters):
 * the pass scheduler runs on several host threads.  What a generation of EVQE needs after topological search or layer
 * removal, when a whole population of new structures arrives together (reference: one fresh QuantumCircuit per
 * individual, queasars/minimum_eigensolvers/evqe/evolutionary_algorithm/selection.py:75-82). */
int qsv_circuits_create(qsv_t* h, int n_circuits, const int64_t* op_offsets, const qsv_op* ops, const int* n_params,
                        int* out_circuit_ids);
int qsv_circuit_destroy(qsv_t* h, int circuit_id);

/*
 * KEPT STATES.  A layer search evaluates one circuit over and over with only one layer's angles changing
 * (reference: optimize_layer_of_individual binds every other layer, mutation.py:57-59,
 * individual.py:288-322 get_partially_parameterized_quantum_circuit): everything in front of that layer is the same state
 * in every evaluation.  qsv_prefix_create runs n_states (circuit, parameter vector) pairs from |0..0> ONCE and keeps their
 * final states resident (2^n amplitudes each); a circuit registered with qsv_circuit(s)_create_on_prefix(es) starts from
 * such a state instead of |0..0> and is evaluated by every qsv_eval_* entry point like any other circuit id (qsv_statevector
 * too; the sampling entry points refuse it).  Unsplittable (deep) individuals then cost the passes of the layers from the
 * searched one on, not of the whole circuit.  A kept state lives until qsv_prefix_destroy AND the last circuit registered on
 * it is destroyed; its memory is reused afterwards.
 */
int qsv_prefix_create(qsv_t* h, int n_states, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                      int* out_prefix_ids);
int qsv_prefix_destroy(qsv_t* h, int n_states, const int* prefix_ids);
/* Kept states alive on the handle (held by the caller or by a circuit). */
int qsv_prefix_count(const qsv_t* h);
int qsv_circuit_create_on_prefix(qsv_t* h, int prefix_id, int n_ops, const qsv_op* ops, int n_params, int* out_circuit_id);
int qsv_circuits_create_on_prefixes(qsv_t* h, int n_circuits, const int64_t* op_offsets, const qsv_op* ops, const int* n_params,
                                    const int* prefix_ids, int* out_circuit_ids);

/*
 * Which way an expectation value of a registered circuit goes under the operator set now, and about what it costs: what a
 * scheduler needs to deal individuals of unequal depth to several GPUs (the reference balances dynamically, one future per
 * individual on a pool: selection.py:75-82, mutation.py:206-216), and what decides whether a layer search is worth a kept
 * state.  microseconds: GPU time per evaluation inside a full launch, from the measured figures of DESIGN.md (an estimate:
 * only ratios matter to its users).
 */
enum { QSV_ROUTE_ONE_TILE = 0, QSV_ROUTE_SPLIT_ONE_LAUNCH = 1, QSV_ROUTE_SPLIT = 2, QSV_ROUTE_PASSES = 3 };
typedef struct qsv_circuit_cost_t {
    int32_t route;         /* QSV_ROUTE_* */
    int32_t n_keys;        /* split routes: cut keys (2^keys product terms) */
    int32_t n_passes;      /* gate passes (split routes: of the longer virtual circuit) */
    int32_t on_kept_state; /* the circuit continues a kept state */
    double microseconds;
} qsv_circuit_cost_t;
int qsv_circuit_cost(qsv_t* h, int circuit_id, qsv_circuit_cost_t* out);

/*
 * Expectation values real(<psi_i|H|psi_i>) of n_evals (circuit, parameter vector) pairs, |psi_i> prepared from
 * |0..0>.  params holds the vectors back to back, vector i at params[param_offsets[i] .. param_offsets[i+1]).
 * Replaces `estimator.run(pubs, precision=0).result()` + `real(res.data.evs)` (circuit_evaluation.py:210-215).
 */
int qsv_eval_circuits(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets,
                      const double* params, double* out_expectations);

/*
 * ONE evaluation, merged with the evaluations other host threads ask for at the same time: the first caller collects
 * the requests that arrive within `window_us` microseconds (0 = library default; it stops earlier once as many
 * callers as last time have arrived, or nobody new comes), runs them as one batch and every caller gets its value.
 * This is the native form of the reference's BatchingMutexPrimitiveJobRunner (a 0.1 s collection window in front of a
 * primitive that is not thread safe, queasars/circuit_evaluation/mutex_primitives.py:67-199) for its calling pattern:
 * population_size threads, one circuit per call (selection.py:75-82, mutation.py:63-75).
 */
int qsv_eval_coalesced(qsv_t* h, int circuit_id, const double* params, int n_params, double window_us,
                       double* out_expectation);

/*
 * Streaming form of qsv_eval_circuits, for callers whose parameter vectors become available (or are converted)
 * piecemeal: qsv_eval_begin lays the batch out, each qsv_eval_push ships the packed parameter values of evaluations
 * [first, first+count) and launches them asynchronously, qsv_eval_end waits and returns all results.  Pushes must be
 * in order and contiguous; a push of any size is accepted (at most qsv_group_size() of its evaluations run side by
 * side in one launch).  The handle is locked from begin to end; end must be called even after a failed push.
 */
int qsv_eval_begin(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_counts);
int qsv_eval_push(qsv_t* h, int first, int count, const double* values);
/* Where the library keeps the parameter values of evaluations [first, first + count) of the open batch until their kernels
 * have read them (pinned host memory, sum of their param_counts doubles, back to back): a caller that writes them THERE and
 * passes the same pointer to qsv_eval_push saves the library's copy (84 KB per population of the benchmark: 5 us of a 72 us
 * step).  Valid until that push; evaluations must still be pushed in order. */
int qsv_eval_staging(qsv_t* h, int first, int count, double** values);
/* qsv_eval_push for parameter values that ALREADY LIVE IN DEVICE MEMORY (this handle's GPU; an optimiser that runs on the
 * device, a torch tensor): `device_values` points at the first value of evaluation `first`, the values of the push packed
 * back to back by the counts given to qsv_eval_begin -- a row-major matrix of equal rows is such a packing when every
 * evaluation declares the row length as its count (a circuit takes the first n_params values of its row).  Nothing is copied
 * and nothing crosses PCIe: the kernels read the values where they are, so they must stay unchanged until qsv_eval_end has
 * returned (for a batch that does not wait, qsv_eval_set_output: until its work is complete).  `ready_event`: a hipEvent_t
 * after which the values are complete -- every stream of the handle waits for it --, or NULL when they already are (the
 * caller synchronised, or wrote them on the handle's stream, qsv_set_stream).  Pushes of both kinds may be mixed in a batch.
 * (An evaluation that declares more than 1024 values -- rows padded that far -- is prepared without the LDS copy of its vector:
 * the same result to the last bits, not bit for bit.) */
int qsv_eval_push_device(qsv_t* h, int first, int count, const double* device_values, void* ready_event);
int qsv_eval_end(qsv_t* h, double* out_expectations);
/*
 * Results into DEVICE memory (n_evals doubles, this handle's GPU), for a caller that feeds them to something on the
 * device -- the fitness all-gather of a population sharded over several GPUs (one process per GPU; the reference hands
 * every individual to a worker of its own, evolutionary_algorithm/selection.py:75-85).  Call between qsv_eval_begin and
 * the first push.  qsv_eval_end(h, NULL) then returns WITHOUT waiting: the results are complete once the work
 * enqueued so far on the handle's stream (qsv_set_stream) is, so whatever the caller enqueues on that stream next sees
 * them, and one synchronisation at the end of ITS chain replaces the library's.  With a non-NULL pointer qsv_eval_end
 * waits and also copies the results to the host.  The next call on the handle waits for an unfinished batch first.
 */
int qsv_eval_set_output(qsv_t* h, double* device_out);
/*
 * The caller has SEEN every result of the last batch that ended without waiting (qsv_eval_set_output into memory the host can
 * read -- a node's shared fitness table: a result is an evaluation's last store): nothing of that batch is still running,
 * and the next call on the handle need not wait for the streams before it reuses the staging buffers (20 us per step of a
 * caller that hands its parameter values over as host arrays).  Saying so about results that have not all arrived is the
 * caller's error.
 */
int qsv_eval_results_seen(qsv_t* h);
/*
 * The optimiser's share of one iteration of R lock-step SPSA runs as ONE launch on the handle's stream, for a parameter search
 * whose state lives in device memory (evqe/device_search.py; the reference runs one qiskit_algorithms SPSA per individual on a
 * worker thread, mutation.py:28-89): with qsv_eval_push_device and qsv_eval_set_output an iteration is this launch plus the
 * evaluation's, and the host waits for neither.  All pointers are device memory of the handle's GPU, row-major, rows of
 * `width` doubles (a run shorter than the widest is padded with zero signs).
 *   accept  (values != NULL): values[2r], values[2r + 1] = f(x_r + eps delta_r), f(x_r - eps delta_r) measured with
 *           delta_accept; update = (f+ - f-) / (2 eps) * delta, divided by its norm if trust_region and the norm exceeds 1, times
 *           lr; x_r -= update for runs with active[r] != 0; iterations[r] counts them; a run stops (active[r] = 0) at maxiter,
 *           at 2 * iterations >= maxfev (maxfev >= 0), or by the reference's SPSATerminationChecker rule over `window` =
 *           allowed_consecutive_violations + 1 relative changes of 0.5 (f+ + f-) below min_rel (window = 0: no rule;
 *           previous / n_values / changes are its state: zeros, zeros, +inf before the first call).
 *   propose (delta_propose != NULL): points[2r] = x_r + eps delta, points[2r + 1] = x_r - eps delta.
 * Products and sums are rounded one by one as the host's NumPy expressions are; the norm is a fixed-order sum of its own.
 */
typedef struct qsv_spsa_step_args {
    int32_t n_runs, width;
    double* x;
    uint8_t* active;
    int64_t* iterations;
    const double* delta_accept;
    const double* values;
    const double* delta_propose;
    double* points;
    double eps, lr;
    int32_t trust_region, maxiter;
    int32_t window;
    int32_t reserved;
    double min_rel;
    int64_t maxfev;
    double* previous;
    int64_t* n_values;
    double* changes;
} qsv_spsa_step_args;
int qsv_spsa_step(qsv_t* h, const qsv_spsa_step_args* args);
/* How many pushes the open batch is best delivered in (1 or 2): measurement-backed advice, any number works. */
int qsv_eval_suggested_pushes(const qsv_t* h);
/* Launch-group size of the handle (evaluations whose states are resident at the same time). */
int qsv_group_size(const qsv_t* h);

/* Same, with the op lists passed inline (circuit i = ops[op_offsets[i] .. op_offsets[i+1])). */
int qsv_eval_batch(qsv_t* h, int n_evals, const int64_t* op_offsets, const qsv_op* ops,
                   const int64_t* param_offsets, const double* params, double* out_expectations);

/* Final state of one circuit as interleaved (re, im) doubles, 2 * 2^n values (debug / parity checks). */
int qsv_statevector(qsv_t* h, int circuit_id, const double* params, int n_params, double* out_re_im);

/* |amplitude|^2 of every basis state, 2^n doubles.  Exact counterpart of the sampler's distribution. */
int qsv_probabilities(qsv_t* h, int circuit_id, const double* params, int n_params, double* out_probs);

/*
 * Draw `shots` basis states from the circuit's output distribution on the device (seeded inverse-CDF sampling).
 * Replaces `sampler.run(pubs, shots)` + `get_counts()` in measure_quasi_distributions (circuit_evaluation.py:50-59).
 * The draw is a function of (seed, circuit, parameters) on a given handle configuration; a circuit that has a split form
 * is sampled from its two side tables (no 2^n probabilities are formed), which walks the index space in another order
 * than the probabilities-based sampler: same distribution, different samples for the same seed.
 */
int qsv_sample(qsv_t* h, int circuit_id, const double* params, int n_params, int shots, uint64_t seed,
               uint64_t* out_states);

/*
 * The same for a batch: evaluation i draws `shots` samples into out_states[i * shots ..] from its own random stream
 * derived from (seed, i).  When the operator set on the handle is diagonal (I/Z terms only) and out_values is not
 * NULL, out_values[i * shots + s] receives the operator's value on that sample, sum_k c_k (-1)^popcount(state & z_k)
 * -- what `_evaluate_sparsepauli` computes per measured state in get_expectation_with_operator
 * (reference: queasars/circuit_evaluation/expectation_calculation.py:64-66) -- so the host only sorts for the CVaR.
 */
int qsv_sample_batch(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                     int shots, uint64_t seed, uint64_t* out_states, double* out_values /* may be NULL */);

/*
 * Sampling and the CVaR in one call: out_cvar[i] = CVaR_alpha of the operator's values on evaluation i's `shots` samples
 * (the samples qsv_sample_batch draws for the same seed), i.e. the mean of the lowest alpha * shots sample values with
 * the boundary sample weighted fractionally -- what get_expectation_with_operator / _get_expectation compute from the
 * measured distribution (reference: queasars/circuit_evaluation/expectation_calculation.py:14-69; alpha = 1: the plain
 * mean).  Needs a diagonal operator on the handle, 0 < alpha <= 1 and shots <= 4096; the samples are sorted on the
 * device and never cross PCIe.
 */
int qsv_sample_cvar_batch(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                          int shots, uint64_t seed, double alpha, double* out_cvar);

/*
 * The sampler branch without sampling noise: out_cvar[i] = CVaR_alpha of the operator's values under the EXACT distribution
 * |amplitude|^2 of evaluation i -- the value get_expectation_with_operator / _get_expectation would return for a measured
 * distribution that equals the exact one (reference: queasars/circuit_evaluation/expectation_calculation.py:14-32, :55-69),
 * including the loop's stopping rule numpy.isclose(gathered, alpha) and, for states of equal value, the index order a stable
 * sort leaves them in.  Needs a diagonal operator on the handle, 0 < alpha <= 1 and at most 28 qubits; deterministic.
 * (For alpha = 1 this is the expectation value qsv_eval_circuits computes.)  Circuits that have a split form are read from
 * their two side tables, the others from the probabilities their last gate pass writes.
 */
int qsv_exact_cvar_batch(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                         double alpha, double* out_cvar);

/* ---- sharded populations on one node ------------------------------------------------------------- */

/*
 * The waiting part of one step through a node's shared fitness table (queasars_amd/distributed.py, _NodeTable; reference:
 * the executor's futures of selection.py:75-85 -- here every rank's values land in a table all ranks map).  `own`: this rank's
 * slot (count doubles that the rank marked with QSV_TABLE_SENTINEL before it started its evaluation, whose kernels store
 * the values there); `done`: the ranks' step counters, `stride` 64-bit words apart.  Spins until no sentinel is left in the
 * slot, publishes done[rank * stride] = step, spins until every rank's counter has reached `step`.  Needs no handle and no
 * GPU.  Returns 0; 1 / 2 when the slot / the counters were not there after budget_us microseconds (nothing published in
 * case 1): the caller decides how to go on waiting.
 */
#define QSV_TABLE_SENTINEL 0x7FF8C0DEC0DE0001ull
int qsv_fitness_table_wait(const volatile uint64_t* own, int count, volatile int64_t* done, int stride, int world, int rank,
                           int64_t step, int budget_us);

/* ---- measurement support ----------------------------------------------------------------------- */

/*
 * Switches of a handle, for measurements and tests (the defaults are the measured best).  name / value:
 *   "split" 0|1          weakly entangled circuits run as two virtual circuits (csrc/split.hpp); 1 needs a handle that was
 *                        created with splitting on.  Applies to circuits registered afterwards.
 *   "factor" 0|1         split evaluations use the factorised expectation kernels instead of the contraction sweep
 *   "fused_factor" 0|1   ... inside the launch that runs their virtual circuits, where a circuit qualifies (one launch per push)
 *   "fused_lds_table" 0|1 ... and there a side hands its state to its Gram matrices through LDS where it fits (up to twelve
 *                        virtual qubits as the state would lie in memory; three-key sides of thirteen as padded rows read in
 *                        place) instead of through its slot: the same sums in the same order, the same bits
 *   "sides_r3" 0|1       ... at 20 qubits and below (12-qubit tiles, fp64) the sides of that route are planned with EIGHT amplitudes per
 *                        thread instead of the handle's sixteen -- twice the waves per side: a gate phase is one wave's issue time over
 *                        its own amplitudes --, and a side of thirteen virtual qubits runs as TWO workgroups, one per value of its last
 *                        key qubit, that trade half rows through memory (where its plan does not leave that qubit outside the tile:
 *                        as two tiles swept by one workgroup).  Other orders of the sums (1e-10 apart).  Applies to circuits
 *                        registered afterwards.
 *   "side_diag" 0|1      ... and reads its values of D from a table of its own -- entry x of a side = D[x deposited in the side's
 *                        qubits], one run, filled when the circuit's plan is uploaded -- instead of gathering them from D, one
 *                        cache line per value: the same values, the same bits
 *   "split_max_keys" 0..5 most cut keys of a split form (default 5; four and five: quadratic operators only).  Applies to
 *                        circuits registered afterwards.
 *   "chain_stream" 0|1   in a push that holds split evaluations of both kinds (finished by the launch that runs their
 *                        virtual circuits / with launches of their own) the second kind runs on the second lane's stream
 *   "poll_results" 0|1   a waiting qsv_eval_end watches the (pinned) result buffer instead of the streams: every result is one
 *                        8-byte store, visible about 5 us before the stream's completion signal (diagonal operators)
 *   "repeat_layout" 0|1  a batch with the previous batch's circuit ids and counts, nothing registered or set in between,
 *                        keeps that batch's layout (an optimiser's next iteration)
 *   "split_sampling" 0|1 split circuits are sampled from their side tables
 *   "streams" 1..4       HIP streams the pushes of a batch cycle over (at most as many as were created with the handle)
 * Returns QSV_E_ARG for an unknown name or a value out of range.
 */
int qsv_set_option(qsv_t* h, const char* name, int value);

int qsv_set_profiling(qsv_t* h, int enabled);
int qsv_get_profile(const qsv_t* h, qsv_profile* out);

/*
 * Roofline microbenchmark: apply one u (control < 0) or cu3 gate with the given angles
 * to the resident 2^n state `reps` times and report the average device time per sweep in milliseconds.
 */
int qsv_bench_gate(qsv_t* h, int target, int control, double theta, double phi, double lambda, int reps,
                   double* out_ms_per_sweep);

/*
 * Same for an arbitrary bound gate list (every angle literal): the ops are scheduled into passes WITHOUT folding
 * and applied read-modify-write to the resident state `reps` times.  Reports milliseconds per repetition and the
 * number of passes one repetition takes.
 */
int qsv_bench_ops(qsv_t* h, int n_ops, const qsv_op* ops, int reps, double* out_ms_per_rep, int* out_n_passes);

/*
 * Build (without touching any device) the pass plan the scheduler produces for a circuit and copy its encoded
 * words out; *n_words receives the size needed.  Used by the CPU-side tests to check the scheduler.
 */
int qsv_plan_build(int n_qubits, int dtype, int n_ops, const qsv_op* ops, const qsv_plan_config* cfg,
                   uint32_t* out_words, size_t capacity_words, size_t* n_words);

/*
 * How the scheduler would split a circuit into two virtual circuits (csrc/split.hpp), without touching any device.
 * Returns the number of cut keys K (>= 0), or -1 when the circuit keeps its ordinary plan; an argument error is
 * QSV_E_ARG - 100.  mask_a receives the qubits of side A (the rest is side B).  The virtual circuits come back as op
 * lists on popcount(side) + K qubits (side qubits in ascending order, then the keys): kind QSV_OP_U / QSV_OP_CU3,
 * angles as in the input, except that p_theta = -2 / -3 / -4 stands for the fixed matrix diag(1, 0) / [[1,0],[1,0]] / X.
 * The final states a_kappa, b_kappa of the two circuits give  psi[i] = sum_kappa a[kappa, i|A] * b[kappa, i|B].
 * Used by the CPU-side tests.
 */
int qsv_split_describe(int n_qubits, int n_ops, const qsv_op* ops, int max_side, uint64_t* mask_a, qsv_op* ops_a,
                       int capacity_a, int* n_ops_a, qsv_op* ops_b, int capacity_b, int* n_ops_b);

/* Library version string. */
const char* qsv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QSV_H */
