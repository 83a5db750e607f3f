// C ABI of libqsv (include/qsv.h): handle, device memory, plan cache, launch sequencing.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/qsv.h"
#include "kernels.hpp"
#include "plan.hpp"
#include "split.hpp"

using namespace qsv;

namespace {

// what a result slot holds until its evaluation's kernel writes it: a NaN whose payload no arithmetic produces
constexpr uint64_t kResultSentinel = 0x7ff8dead5e4717e1ull;
constexpr int kPollMicros = 200;

// Error text is kept per calling thread (and per handle it belongs to): the thread that received a failing return code
// is the one that asks for the text, and it must not race with another thread's failure on the same handle.
constexpr size_t kInlineCacheLimit = 4096;  // structures qsv_eval_batch keeps registered between calls

thread_local std::string g_create_error;
thread_local std::string g_handle_error;
thread_local const void* g_handle_error_owner = nullptr;

// A circuit the scheduler could split (split.hpp): the plans of the two virtual circuits and the contraction's index
// maps follow the ordinary plan in `plan.words` (offsets relative to it, like everything in a plan).
struct SplitInfo {
    bool ok = false;
    int n_keys = 0;
    PlanStats stats[2];
    int t[2] = {0, 0};              // thread bits of the two plans
    int outer[2] = {0, 0};          // qubits outside a tile (0: the virtual circuit is one tile)
    int n_virtual[2] = {0, 0};      // qubits of the virtual circuits
    uint32_t off_side[2] = {0, 0};  // word offsets of the side plans
    uint32_t off_block = 0;         // ... and of the split block (kernels.hpp)
    int tile_bits[2] = {0, 0};      // tile of the two plans (a side's tile may be larger than the handle's, see build_circuit)
    bool fused = false;             // both virtual circuits are one pass (kernels.hpp kEvalFused)
    bool halves = false;            // ... three keys, thirteen virtual qubits a side: two workgroups per side (kernels.hpp kEvalHalves)
    int side_r = 0;                 // amplitudes per thread of the side plans, as log2 (the handle's, or 3: build_circuit)
};

struct Circuit {
    int n_params = 0;
    int n_gates = 0;                // non-identity ops
    CircuitPlan plan;               // words: everything of this circuit that lives in the device arena; stats: of the
                                    // ordinary multi-pass plan, once it exists
    SplitInfo split;
    // The ordinary plan of a circuit that has a split form is scheduled when something first needs it (a general
    // operator, a state read-out, sampling): a population of fresh structures under a diagonal operator never does,
    // and its scheduling was half the cost of registering a structure.
    bool has_plan = false;
    uint32_t off_plan = 0;          // word offset of the ordinary plan inside plan.words
    bool fold = true;
    std::vector<GateIn> gates;      // (kept until the ordinary plan is built)
    std::vector<AngleSource> angles;
    bool uploaded = false;
    bool staged = false;            // scratch flag of upload_plans (a circuit may appear several times in a batch)
    uint32_t plan_base = 0;         // word offset in the device arena
    int prefix_id = -1;             // >= 0: the circuit continues that kept state (qsv_circuit_create_on_prefix) instead of |0..0>
};

// A kept state (qsv_prefix_create): one slot of the handle's prefix buffer, alive while the caller holds it or a circuit
// continues it.
struct PrefixState {
    uint32_t slot = 0;
    int refs = 0;          // circuits registered on it
    bool released = false; // the caller has let go (qsv_prefix_destroy): the slot is recycled once refs == 0
};

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace

// Persistent host threads for scheduling many circuit structures at once (a generation of EVQE brings up to a
// population of new structures): spawning threads per call would cost more than the plans.
class WorkerPool {
public:
    explicit WorkerPool(unsigned n_threads) {
        for (unsigned i = 0; i < n_threads; ++i) threads_.emplace_back([this]() { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        wake_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    // run fn(0 .. count-1), the calling thread takes part; returns when every index is done
    void run(size_t count, const std::function<void(size_t)>& fn) {
        if (count == 0) return;
        {
            std::lock_guard<std::mutex> lock(mu_);
            fn_ = &fn;
            count_ = count;
            next_.store(0);
            pending_ = count;
            ++generation_;
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= count_) return;
            (*fn_)(i);
            std::lock_guard<std::mutex> lock(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                wake_.wait(lock, [&]() { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                if (!fn_) continue;
            }
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable wake_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    std::atomic<size_t> count_{0};
    size_t pending_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;
};

struct qsv_handle {
    int n = 0, dtype = 0, device = 0;
    PlanConfig cfg;
    Geometry geo;
    int group = 1;
    size_t amp_bytes = 16;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // Second stream for the streaming evaluation: consecutive pushes alternate between the two, so that the
    // compute-bound first pass of one push runs beside the memory-bound later passes of the other.
    std::vector<hipStream_t> side_streams;  // (own, non-blocking); pushes cycle over `stream` and these
    int n_lane_streams = 0;  // the first so many of side_streams are lanes of pushes
    int aux_stream = -1;  // index in side_streams of the stream that is never a push's lane: in a batch that mixes split and
                          // ordinary evaluations the ordinary ones run there, beside the split ones (eval_push).  Created
                          // when a batch first needs it: a process has few hardware queues (four by default), and streams
                          // beyond them share queues -- with two handles alive, a third stream per handle made the two
                          // pushes of a 256-evaluation step run one after the other (262 -> 358 us).
    int chain_stream = -1;  // index in side_streams of the stream that takes, in a push that holds both kinds, the split
                            // evaluations that need launches of their own (virtual circuits, Gram matrices, combination) --
                            // beside the one-launch ones on the push's lane instead of in front of them (eval_push)
    // Counts everything that could make the layout of an earlier batch stale: registrations, operators, options, every full
    // batch_layout, a reallocated staging buffer.  A batch whose ids and counts are the previous batch's, with nothing counted
    // in between, is that batch again (an optimiser's next iteration over the same population): eval_begin keeps its layout.
    uint64_t epoch = 1;
    bool repeat_enabled = true;
    bool chain_enabled = true;
    bool fused_lds_table = true;  // one-launch route: small sides hand their state to the Gram matrices through LDS (kModeFusedLdsTable)
    int n_cus = 256;
    bool poll_results = true;  // a waiting end of a batch watches the (pinned) result buffer instead of the streams: the last
                               // workgroups' stores are visible about 5 us before hipStreamSynchronize returns (eval_end)
    hipStream_t work = nullptr;  // stream of the push being issued (null: `stream`)
    hipEvent_t ev_join = nullptr;
    int n_streams = 2;           // streams a batch cycles over (QSV_STREAMS, 1 .. 4)
    bool split_enabled = true;   // weakly entangled circuits run as two virtual circuits + a contraction (split.hpp)
    bool split_sampling = true;  // ... and are sampled from their two side tables (kernels.hpp: launch_split_sample)
    bool factor_enabled = true;  // ... and, under a quadratic diagonal operator, need no sweep over the 2^n indices at all
                                 // (kernels.hpp: launch_factor)
    bool fused_factor = true;    // ... in the launch that runs the virtual circuits, where the circuit qualifies (kEvalFused)
    int split_max_keys = 5;      // most cut keys of a split form: four and five (16 / 32 product terms, quadratic operators
                                 // only) -- their chain of launches (virtual circuits of up to 16 qubits, Gram matrices of up to
                                 // 1024 entries) runs beside the one-launch evaluations of the push on the second lane's stream
                                 // (eval_push).  QSV_SPLIT_MAX_KEYS / qsv_set_option "split_max_keys" = 3: the round-2 limit
    bool quadratic = false;      // the operator is diagonal and every term has at most two Z factors
    DeviceBuffer d_quad;         // its couplings as an n x n matrix
    DeviceBuffer d_fterms;       // a general operator's terms as a plain list (kernels.hpp: launch_factor_terms)
    uint32_t n_fterms = 0;
    DeviceBuffer d_fpart;        // ... and that kernel's partial sums
    DeviceBuffer d_factor;       // launch_factor's partial Gram matrices, one region per side-table slot
    DeviceBuffer d_factor_count; // kModeFusedFactor: one counter per side-table slot (each fused evaluation adds two)
    DeviceBuffer d_factor_big;   // launch_factor_big's partial Gram matrices (four and five keys), allocated on first need
    DeviceBuffer d_factor_big_count;  // ... and their workgroup counters (zeroed once)
    uint32_t stream_mode = 0;    // kModeStreaming when a state is larger than the Infinity Cache (256 MiB), else 0
    mutable std::mutex mu;
    std::atomic<std::thread::id> batch_owner{};  // thread that holds `mu` between qsv_eval_begin and qsv_eval_end

    // operator
    int n_terms = 0;
    bool diagonal = false;        // every term is I/Z: the whole expectation is fused into the last gate pass
    bool has_diag_part = false;   // some terms are I/Z (their diagonal table exists)
    int n_groups = 0;             // x-mask groups of the off-diagonal terms
    DeviceBuffer d_z, d_cre, d_diag, d_term_partials, d_groups, d_term_odd;
    DeviceBuffer d_order, d_sorted;  // the basis states in ascending order of the diagonal operator's value, and the values
    bool order_valid = false;        // (sort.hip; built when the exact-probability CVaR first needs them)
    int pauli_nb = 0;

    // circuits
    std::unordered_map<int, Circuit> circuits;
    int next_circuit_id = 1;
    // kept states (qsv_prefix_create): slots of one buffer that grows by doubling
    std::unordered_map<int, PrefixState> prefixes;
    int next_prefix_id = 1;
    DeviceBuffer d_prefix;
    // the sides' own tables of D (kernels.hpp kSplitSideDiag): filled when a split circuit's plan is uploaded, gone with the arena
    DeviceBuffer d_sdiag;
    size_t sdiag_used = 0;             // doubles handed out
    bool side_diag = true;             // (QSV_SIDE_DIAG=0: the sums gather from D itself, as before round 4)
    bool sides_r3 = true;              // one-launch route at 20 qubits: sides of up to twelve virtual qubits planned with EIGHT amplitudes per thread
                                       // (a gate phase is one wave's issue time over its own amplitudes: twice the waves, half the time),
                                       // three-key sides of thirteen as two workgroups each (kEvalHalves); circuits registered afterwards
    size_t prefix_slots = 0;           // capacity
    std::vector<uint32_t> prefix_free; // recycled slots
    size_t prefix_used = 0;            // slots handed out so far (below capacity)
    std::unordered_map<std::string, int> inline_cache;
    std::unique_ptr<WorkerPool> pool;  // created on first use (qsv_circuits_create, qsv_eval_batch)
    std::mutex pool_mu;

    // qsv_eval_coalesced: requests of concurrent callers waiting to be merged into one batch
    struct CoalesceRequest {
        int circuit_id = 0;
        const double* params = nullptr;
        int n_params = 0;
        double value = 0.0;
        int rc = QSV_OK;
        std::string err;
        bool done = false;
    };
    std::mutex cq_mu;
    std::condition_variable cq_cv;
    std::vector<CoalesceRequest*> cq;
    std::atomic<size_t> cq_count{0};   // = cq.size(), readable without the lock (the collecting caller spins on it)
    bool cq_collecting = false;        // some caller is collecting or evaluating a batch
    size_t cq_expected = 0;            // size of recent batches: how many callers to expect
    uint64_t cq_batches = 0;           // batches evaluated so far (diagnostic)

    // device memory
    DeviceBuffer d_arena;  // plans (uint32 words)
    size_t arena_used_words = 0;
    DeviceBuffer d_states;
    DeviceBuffer d_wtab;      // compact tables of pass 0 (plan.hpp COMPACT), one per state slot
    uint64_t wtab_stride = 0; // amplitudes per slot
    DeviceBuffer d_side;      // final states of the virtual circuits of split evaluations (split.hpp): two halves per slot
    uint64_t side_stride = 0; // amplitudes per slot
    int side_slots = 0;       // a split evaluation needs no state, so many more of them than `group` run side by side
    DeviceBuffer d_batch;     // [EvalDesc x B][parameter vectors]
    DeviceBuffer d_mats;      // per evaluation: gate matrices in schedule order + product-state factors
    int tiles_per_block = 1;        // pass 0 (and the only pass of a small circuit)
    int tiles_per_block_later = 1;  // passes 1.. (a multiple of tiles_per_block)
    DeviceBuffer d_partials;  // [B][blocks_per_state]
    DeviceBuffer d_out;       // [B]
    DeviceBuffer d_scratch;   // probabilities / converted state
    void* h_batch = nullptr;  // pinned
    size_t h_batch_bytes = 0;
    // The same bytes in DEVICE memory, written by the host directly over the PCIe BAR (large-BAR systems: every MI355X host):
    // what the kernels read -- descriptors and parameter vectors -- is then a local read instead of a trip over PCIe per
    // dependent load (scripts/ubench/bar_write.hip: 32 KiB written in 0.9 us; one wave reading them back: 28 us against 91
    // from pinned host memory).  batch_ship copies each push's ranges across and fences; kernels get these pointers.
    void* d_ship = nullptr;
    bool bar_ship = false;
    std::vector<char> ship_shadow;  // what d_ship's descriptor regions hold (host copy, for the compare)
    uint32_t* h_stage = nullptr;  // pinned staging buffer for plan uploads
    size_t h_stage_words = 0;
    double* h_out = nullptr;  // pinned
    bool repeat_device_descs = true;  // (QSV_REPEAT_DESCS=0: a repeated batch reads its descriptors from pinned memory again)
    const double* dev_params_checked = nullptr;  // the last qsv_eval_push_device pointer found to be this device's memory
    double* out_target = nullptr;  // qsv_eval_set_output: device memory the open batch's results go to instead of h_out
    bool async_pending = false;    // a batch ended without waiting (qsv_eval_end with a device output): the staging buffers
                                   // may still be read by its kernels
    void* h_samples = nullptr;  // pinned: sampled states (and their operator values) of one qsv_sample_batch call
    size_t h_samples_bytes = 0;
    size_t h_out_count = 0;

    // streaming evaluation (qsv_eval_begin / push / end)
    struct Batch {
        bool open = false;
        std::vector<Circuit*> circs;
        std::vector<uint32_t> param_base;  // per evaluation, in doubles
        std::vector<uint32_t> n_params;
        size_t desc_bytes = 0;
        size_t pushed = 0;                 // evaluations launched so far
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pass_events, exp_events;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> launch_events[3];  // per launch: [0] first pass, [1] later passes, [2] contraction
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        bool split_any = false; // some evaluation of the batch runs split: the descriptor array has a second region
        std::vector<char> split;  // per evaluation
        bool cont_any = false;  // some evaluation continues a kept state (Circuit::prefix_id): a push puts those last
        size_t snap_n_cont = 0;
        std::vector<uint32_t> eval_at;  // descriptor position -> evaluation (a push puts its split evaluations first)
        int ways = 1;           // streams this batch cycles over
        unsigned used_mask = 0; // side streams (bit i = side_streams[i]) with work of this batch in flight
        bool aux_plain = false; // the batch's ordinary evaluations run on the auxiliary stream (eval_begin)
        bool chain_now = false; // this push: the split evaluations with launches of their own go to the chain stream (eval_push)
        bool chain_crossed = false; // some push of this batch put its chain on the other lane's stream (eval_push)
        bool sentinels = false; // the result buffer was filled with kResultSentinel before the first push (eval_begin)
        // the batch as qsv_eval_begin was called (kept for the next call's comparison), and what the last COMPLETED batch left
        // in place: valid while snap_epoch == the handle's epoch
        std::vector<int> cur_ids, snap_ids;
        std::vector<int64_t> cur_counts, snap_counts;
        uint64_t snap_epoch = 0;
        size_t snap_n_split = 0;
        bool have_ids = false;     // cur_ids / cur_counts describe THIS batch (it came through qsv_eval_begin)
        bool repeat = false;       // this batch reuses the previous batch's layout (descriptors as its one push left them)
        bool whole_push = false;   // this batch was pushed in one piece
        size_t aux_count = 0;   // ... how many of them have been pushed (their state slots cycle over the whole group)

        size_t n_pushes = 0;
        const double* dev_params = nullptr;  // this push's parameter values live in device memory (qsv_eval_push_device):
                                             // where evaluation 0's values would be -- descriptors index it like the staging buffer
    } batch;
    std::unique_lock<std::mutex> batch_lock;  // held from begin to end

    // profiling
    bool profiling = false;
    bool stamping = false;  // per-launch events are recorded (inside qsv_eval_push of a profiled batch only)
    qsv_profile prof{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
};

namespace {

int fail(qsv_t* h, int code, const std::string& msg) {
    if (h) {
        g_handle_error = msg;
        g_handle_error_owner = h;
    } else {
        g_create_error = msg;
    }
    return code;
}

inline hipStream_t ws(const qsv_t* h) { return h->work ? h->work : h->stream; }

#define QSV_HIP(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail((h), QSV_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

// Nothing on either stream may still be using a buffer that is about to be replaced.
hipError_t sync_streams(qsv_t* h) {
    hipError_t e = h->stream ? hipStreamSynchronize(h->stream) : hipSuccess;
    for (hipStream_t st : h->side_streams)
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

int ensure(qsv_t* h, DeviceBuffer& b, size_t bytes) {
    if (b.bytes >= bytes && b.ptr) return QSV_OK;
    if (b.ptr) {
        QSV_HIP(h, sync_streams(h));
        QSV_HIP(h, hipFree(b.ptr));
        b.ptr = nullptr;
        b.bytes = 0;
    }
    size_t want = std::max(bytes, size_t(256));
    QSV_HIP(h, hipMalloc(&b.ptr, want));
    b.bytes = want;
    return QSV_OK;
}

int validate_ops(qsv_t* h, int n, int n_ops, const qsv_op* ops, int n_params) {
    if (n_ops < 0 || (n_ops > 0 && !ops)) return fail(h, QSV_E_ARG, "ops is null");
    for (int i = 0; i < n_ops; ++i) {
        const qsv_op& o = ops[i];
        if (o.kind > QSV_OP_CU3) return fail(h, QSV_E_ARG, "unknown op kind at op " + std::to_string(i));
        if (o.target >= n) return fail(h, QSV_E_ARG, "target qubit out of range at op " + std::to_string(i));
        if (o.kind == QSV_OP_CU3 && (o.control >= n || o.control == o.target))
            return fail(h, QSV_E_ARG, "bad control qubit at op " + std::to_string(i));
        if (o.kind != QSV_OP_ID)
            for (int32_t p : {o.p_theta, o.p_phi, o.p_lambda})
                if (p >= n_params || p < -1)  // (below -1: the scheduler's own fixed matrices, split.hpp)
                    return fail(h, QSV_E_ARG, "parameter index out of range at op " + std::to_string(i));
    }
    return QSV_OK;
}

std::vector<GateIn> gates_of(const qsv_op* ops, int n_ops, std::vector<AngleSource>* angles) {
    std::vector<GateIn> gates;
    angles->clear();
    for (int i = 0; i < n_ops; ++i) {
        angles->push_back(AngleSource{ops[i].p_theta, ops[i].p_phi, ops[i].p_lambda, ops[i].theta, ops[i].phi,
                                      ops[i].lambda});
        if (ops[i].kind == QSV_OP_ID) continue;
        GateIn g;
        g.target = ops[i].target;
        g.control = ops[i].kind == QSV_OP_CU3 ? int(ops[i].control) : -1;
        g.op = i;
        gates.push_back(g);
    }
    return gates;
}

PlanConfig resolve_config(const qsv_plan_config* cfg, int dtype, int n_qubits) {
    PlanConfig pc;
    pc.amp_bytes = dtype == QSV_F64 ? 16 : 8;
    // Tile geometry by size (fp64; measured on MI355X with the EVQE benchmark family, scripts/geometry_sweep.sh): 16 amplitudes
    // per thread halve the per-amplitude cost of decoding the plan from n = 20 on, and 13 tile qubits save a pass
    // from n = 21 on (n = 24: 3 -> 2.25 passes on average, +33 % evaluations per second).
    // (single precision likewise since its round loop is generated assembly too, round 4: n = 20, L = 8 60 k -> 81 k evals/s with
    // 16 amplitudes per thread, n = 24 3.0 k -> 3.8 k with 13-qubit tiles, profiles/r04_fp32.txt)
    if (n_qubits >= 20) {
        pc.reg_bits = 4;
        pc.tile_bits = n_qubits >= 21 ? 13 : 12;
    }
    // Runs in memory.  The load and store layouts of a multi-gate pass keep only the lowest `lane_bits` tile qubits on the low
    // lanes (everything else goes where the first round's targets want it): a wave's access is 64 / 2^lane_bits separate runs
    // of 2^lane_bits amplitudes.  Two bits are 64-byte runs in double precision -- half an L2 line (128 bytes on this part) --
    // and 32-byte runs in single precision.  Measured in round 4 (profiles/r04_fp32.txt): with THREE the later pass of config
    // 5's genome goes from 0.55 to 0.68 of 8 TB/s at n = 28 (fp64) and from 0.36 to 0.57 (fp32), whole deep evaluations at
    // n = 26 / 28 gain 10 - 20 %; at n <= 24 in double precision the pass gains as much as the lost free qubit costs in
    // passes (n = 24: L = 8 +5 %, L = 4 -5 %).  So: three wherever the state is beyond the Infinity Cache, and in single
    // precision from 20 qubits on.
    if ((size_t(1) << n_qubits) * size_t(pc.amp_bytes) > (size_t(256) << 20) || (dtype != QSV_F64 && n_qubits >= 20))
        pc.low_bits = pc.lane_bits = 3;
    if (dtype != QSV_F64) pc.xmode = 0;  // fp32: one 8-byte complex element per LDS access
    if (const char* e = getenv("QSV_XMODE")) pc.xmode = atoi(e);
    if (const char* e = getenv("QSV_TILE_BITS")) pc.tile_bits = atoi(e);
    if (const char* e = getenv("QSV_REG_BITS")) pc.reg_bits = atoi(e);
    if (const char* e = getenv("QSV_LOW_BITS")) pc.low_bits = atoi(e);
    if (const char* e = getenv("QSV_LANE_BITS")) pc.lane_bits = atoi(e);
    if (const char* e = getenv("QSV_FOLD")) pc.fold = atoi(e) != 0;
    if (const char* e = getenv("QSV_COMPACT")) pc.compact = atoi(e) != 0;
    if (const char* e = getenv("QSV_SWAPS")) pc.swaps = atoi(e) != 0;
    // multiplexed gates (plan.hpp FUSION): the assembly round loops of both precisions take their entries at full speed (fp32
    // since round 4: RoundLoopF32's one body serves products of matrices as well)
    pc.fuse = true;
    if (const char* e = getenv("QSV_FUSE")) pc.fuse = atoi(e) != 0;
    if (const char* e = getenv("QSV_RETRIES")) pc.retries = atoi(e);
    if (cfg) {
        if (cfg->tile_bits > 0) pc.tile_bits = cfg->tile_bits;
        if (cfg->reg_bits > 0) pc.reg_bits = cfg->reg_bits;
        if (cfg->low_bits > 0) pc.low_bits = cfg->low_bits;
        if (cfg->exchange > 0) pc.xmode = cfg->exchange - 1;
    }
    if (dtype != QSV_F64 && pc.xmode != 0) pc.xmode = 0;
    pc.elem_bytes = (dtype == QSV_F64 && pc.xmode == 0) ? 16 : 8;
    return pc;
}

// The ordinary (multi-pass) plan of a circuit.  With THREE low lane bits instead of two a pass over a state runs faster (128-byte
// runs, resolve_config) but has one free tile qubit fewer, which costs some circuits a pass: where the handle's default is two
// (double precision up to 24 qubits) the circuit is scheduled both ways and takes three when that costs no pass -- three in four
// deep individuals (n = 24, L = 8: 24 of 32; L = 4: 31 of 32).  QSV_LOW_BITS / QSV_LANE_BITS / QSV_AUTO_LANE=0 switch it off.
CircuitPlan build_ordinary_plan(const qsv_t* h, const std::vector<GateIn>& gates, const std::vector<AngleSource>& angles, const PlanConfig& pc) {
    CircuitPlan two = build_plan(h->n, gates, angles, pc);
    static const bool automatic = !getenv("QSV_LOW_BITS") && !getenv("QSV_LANE_BITS") && !(getenv("QSV_AUTO_LANE") && atoi(getenv("QSV_AUTO_LANE")) == 0);
    if (!automatic || pc.low_bits != 2 || pc.lane_bits != 2 || h->geo.blocks_per_state < 2 || two.stats.n_passes < 1) return two;
    PlanConfig wide = pc;
    wide.low_bits = wide.lane_bits = 3;
    try {
        CircuitPlan three = build_plan(h->n, gates, angles, wide);
        if (three.stats.n_passes <= two.stats.n_passes) return three;
    } catch (const std::exception&) {
        // (the narrower plan stands)
    }
    return two;
}

// Validate an op list and schedule it.  Touches only immutable parts of the handle (n, cfg), so it needs no lock and
// several threads may build plans at the same time.
int build_circuit(qsv_t* h, int n_ops, const qsv_op* ops, int n_params, bool fold, Circuit* out, std::string* err) {
    std::string local;
    {
        // validate_ops reports through fail(): keep its text
        int rc = validate_ops(h, h->n, n_ops, ops, n_params);
        if (rc) {
            if (err) *err = g_handle_error;
            return rc;
        }
    }
    out->n_params = n_params;
    std::vector<AngleSource> angles;
    std::vector<GateIn> gates = gates_of(ops, n_ops, &angles);
    out->n_gates = int(gates.size());
    out->fold = fold;
    try {
        PlanConfig pc = h->cfg;
        pc.fold = pc.fold && fold;
        // (with QSV_FACTOR=0 not under a general operator: every evaluation then needs the state, i.e. the ordinary plan; a
        // circuit registered now and evaluated under a diagonal operator later simply takes the ordinary path)
        const bool state_needed = h->n_terms > 0 && !h->diagonal && !h->factor_enabled;
        if (h->split_enabled && !state_needed && pc.fold && h->n > h->geo.k && h->n <= 28) {  // (28: the contraction's 32-bit byte offsets into D)
            // a virtual circuit may be a few qubits larger than a tile (it then takes the pass kernel a few passes over up
            // to sixteen tiles: nothing next to the 2^n indices of the contraction)
            // ... but one tile each is what to look for first: no second pass, one workgroup per virtual circuit
            // (with 16 amplitudes per thread a side's tile may be one qubit larger than the handle's, below: sides that
            // fit THAT are the second choice)
            const int side_tile = std::max(h->geo.k, std::min(h->geo.r + 9, int(kMaxTileBits)));
            std::vector<int> limits{h->geo.k};
            if (side_tile > h->geo.k && side_tile < h->n) limits.push_back(side_tile);
            // (every size up to tile + 4 is a stage of its own since round 4: the first limit that has a partition wins, and with
            // the stages two qubits apart config 5's genome -- 14 + 14 qubits and two keys under limit 16 -- was taken as 13 + 15
            // under limit 17 on handles with 13-qubit tiles, a quarter more rows for the term kernel)
            for (int extra = 1; extra <= kSideExtraBits; ++extra) {
                const int limit = std::min(h->geo.k + extra, h->n - 1);
                if (limit > limits.back()) limits.push_back(limit);
            }
            SplitCircuits sc = find_split(h->n, gates, angles, limits, h->split_max_keys);
            if (sc.ok && std::max(sc.n_side[0], sc.n_side[1]) > kSideMaxOwnBits) sc.ok = false;
            if (sc.ok) {
                SplitInfo& sp = out->split;
                std::vector<uint32_t>& w = out->plan.words;
                bool fits = true;
                // The one-launch route's sides at 20 qubits, where a gate phase is what ONE WAVE takes to issue a gate over its own
                // amplitudes (HISTORY round 4, item 14): planned with eight amplitudes per thread instead of sixteen -- twice the waves
                // -- if both are at most twelve virtual qubits (one tile of 512 threads) and stay one pass; three keys and thirteen
                // virtual qubits a side: over 12-qubit tiles the scheduler leaves a qubit that is never a target outside the tile --
                // a key qubit; if that is the LAST one on both sides and one pass does, each side runs as two workgroups
                // (kEvalHalves).  Each form is tried and kept only if it holds; else the handle's own geometry, as before.
                const bool r3_handle = h->sides_r3 && h->dtype == QSV_F64 && h->geo.k == 12 && h->geo.r == 4 && sc.n_keys <= 3;
                // (half sides: every thirteen-qubit side of the circuit, if each leaves its LAST key qubit outside the tile)
                bool try_halves = r3_handle && sc.n_keys >= 1 && std::max(sc.n_side[0], sc.n_side[1]) + sc.n_keys == kFusedLdsRowsBits &&
                                  !getenv("QSV_NO_HALF_SIDES");  // (tests: the swept form, which a plan that keeps its last key inside the tile falls back to)
                // (thirteen virtual qubits otherwise: two 12-qubit tiles swept by the side's one workgroup -- as long as there is a key
                // qubit to leave outside the tile and one pass does)
                const int most_virtual = std::max(sc.n_side[0], sc.n_side[1]) + sc.n_keys;
                bool try_r3 = try_halves || (r3_handle && (most_virtual <= 12 || (most_virtual == 13 && sc.n_keys >= 1)));
                const size_t words_before = w.size();
                for (bool planned = false; !planned;) {  // (at most three times: half sides, swept tiles, the handle's own geometry)
                planned = true;
                for (int s = 0; s < 2 && fits && planned; ++s) {
                    PlanConfig side = pc;
                    sp.n_virtual[s] = sc.n_side[s] + sc.n_keys;
                    // (a virtual circuit one qubit larger than the handle's tile still fits ONE workgroup when that may have
                    // 512 threads: n = 20 keeps 12-qubit tiles of 256 threads for its states, but a three-key circuit's
                    // 13-qubit sides then stay one tile and one pass -- and with that in the one-launch path, kEvalFused)
                    side.tile_bits = std::min(sp.n_virtual[s], side_tile);
                    side.reg_bits = try_r3 ? 3 : h->geo.r;
                    side.compact = false;  // (the compact-table buffer is where a side's state lives)
                    fits = side.tile_bits > side.reg_bits;
                    if (!fits) break;
                    // A side of ONE tile is synthesised, not loaded, and its final state goes to LDS or to a small table that only
                    // its own workgroup reads back: no layout of it has to keep the low qubits on the low lanes -- the first round
                    // may target them, and the swaps that brought them home at the end (up to four of a side's ten) are not made
                    // (a zero-key circuit of the benchmark alone: 34.1 -> 32.5 us).
                    if (sp.n_virtual[s] <= side.tile_bits) side.lane_bits = 0;
                    if (try_halves && sp.n_virtual[s] == kFusedLdsRowsBits) {  // (the same goes for a half side: its state stays in LDS)
                        side.tile_bits = 12;
                        side.lane_bits = 0;
                    } else if (try_r3) {
                        side.tile_bits = std::min(sp.n_virtual[s], 12);
                    }
                    if (const char* env = getenv("QSV_SIDE_LANE_BITS")) side.lane_bits = atoi(env);  // (measurements)
                    const CircuitPlan p = build_plan(sp.n_virtual[s], sc.gates[s], sc.angles[s], side);
                    sp.stats[s] = p.stats;
                    sp.t[s] = side.tile_bits - side.reg_bits;
                    sp.outer[s] = sp.n_virtual[s] - side.tile_bits;
                    sp.tile_bits[s] = side.tile_bits;
                    sp.off_side[s] = uint32_t(w.size());
                    w.insert(w.end(), p.words.begin(), p.words.end());
                    if (try_r3) {
                        bool ok = p.stats.n_passes == 1;
                        if (ok && try_halves && sp.n_virtual[s] == kFusedLdsRowsBits) {  // (the tile's qubits are 0 .. 11: the last key qubit, 12, is the tile number)
                            const uint32_t* c0 = p.words.data();
                            const uint32_t* p0 = c0 + c0[kCircuitHeaderWords];
                            bool last_outside = true;
                            for (uint32_t j = 0; j < 12 && last_outside; ++j) last_outside = p0[kPassHeaderWords + j] != uint32_t(kFusedLdsRowsBits - 1);
                            if (!last_outside) {  // (two tiles swept by the side's one workgroup, then: planned anew with that form's lanes)
                                try_halves = false;
                                planned = false;
                            }
                        }
                        if (!ok) {  // (as before: the handle's geometry, one tile)
                            try_halves = try_r3 = false;
                            planned = false;
                        }
                        if (!planned) w.resize(words_before);
                    }
                }
                }
                sp.halves = fits && try_halves;
                sp.side_r = try_r3 ? 3 : h->geo.r;
                // the contraction kernel's view of the circuit (kernels.hpp, split block): ready-made pieces of the two
                // table indices for the lanes, the wave index, a thread's own five bits and the chunk number
                const int wave_bits = h->geo.t > 6 ? h->geo.t - 6 : 0;
                const int loop0 = 6 + wave_bits, chunk0 = loop0 + kSplitLoopBits;
                fits = fits && h->n >= chunk0 && h->n - chunk0 <= 14;
                if (fits) {
                    int own_a = 0;  // how many of a thread's own bits belong to side A
                    for (int b = 0; b < kSplitLoopBits; ++b) own_a += int(sc.mask[0] >> (loop0 + b) & 1u);
                    const bool swap_xy = own_a > kSplitMaxLoopX;  // X = the side with fewer of them
                    const int sx = swap_xy ? 1 : 0;
                    const int loop_x = swap_xy ? kSplitLoopBits - own_a : own_a;
                    // bit of index position p in the table index of the side it belongs to
                    std::vector<uint32_t> colx(size_t(h->n), 0), coly(size_t(h->n), 0);
                    {
                        int cx_ = 0, cy_ = 0;
                        for (int q = 0; q < h->n; ++q) {
                            if (sc.mask[sx] >> q & 1u)
                                colx[size_t(q)] = 1u << cx_++;
                            else
                                coly[size_t(q)] = 1u << cy_++;
                        }
                    }
                    sp.off_block = uint32_t(w.size());
                    w.resize(w.size() + kSplitBlockWords, 0);
                    uint32_t* blk = w.data() + sp.off_block;
                    blk[0] = uint32_t(sc.n_keys);
                    blk[1] = uint32_t(sc.n_side[sx]);
                    blk[2] = uint32_t(sc.n_side[1 - sx]);
                    blk[3] = (swap_xy ? 1u : 0u) | uint32_t(loop_x) << 8;
                    blk[kSplitMaskX] = uint32_t(sc.mask[sx]);
                    blk[kSplitMaskY] = uint32_t(sc.mask[1 - sx]);
                    blk[kSplitSideDiag] = blk[kSplitSideDiag + 1] = kNoSideDiag;  // (upload_plans names the tables)
                    {
                        int at = 0;
                        for (int which = 0; which < 2; ++which)
                            for (int b = 0; b < kSplitLoopBits; ++b) {
                                const int q = loop0 + b;
                                const bool is_x = sc.mask[sx] >> q & 1u;
                                if (is_x != (which == 0)) continue;
                                blk[kSplitLoopCols + at] = is_x ? colx[size_t(q)] : coly[size_t(q)];
                                blk[kSplitLoopPos + at] = 1u << q;
                                ++at;
                            }
                    }
                    // (entry v of a table = entry v without its lowest set bit | that bit's own pieces: one step per entry)
                    auto fill = [&](uint32_t* table, uint32_t entries, int first_bit) {
                        table[0] = table[1] = 0;
                        for (uint32_t v = 1; v < entries; ++v) {
                            const uint32_t low = v & (0u - v), rest = v ^ low;
                            const int q = first_bit + __builtin_ctz(low);
                            table[2 * v] = table[2 * rest] | (q < h->n ? colx[size_t(q)] : 0u);
                            table[2 * v + 1] = table[2 * rest + 1] | (q < h->n ? coly[size_t(q)] : 0u);
                        }
                    };
                    fill(blk + kSplitLaneTable, 64, 0);
                    fill(blk + kSplitWaveTable, 1u << wave_bits, 6);
                    fill(blk + kSplitChunkLow, 128, chunk0);
                    fill(blk + kSplitChunkHigh, 128, chunk0 + 7);
                    sp.n_keys = sc.n_keys;
                    sp.ok = true;
                    // (the one-launch path exists in the 16-amplitudes-per-thread instantiation: its 128 registers hold the
                    // Gram matrices' accumulators, the 80 of the 8-amplitude one do not)
                    // (a side of several tiles is swept by its ONE workgroup, tile after tile, as long as it is one pass)
                    // A property of the circuit and the handle, never of the batch: the two routes add in different orders.
                    // Measured (profiles/r03_fused_factor.txt): the one-launch route wins where a side's Gram matrices are a
                    // few blocks per wave -- 20-qubit registers, every population size -- and loses from 22 qubits on, where
                    // a side's table is 2^11 .. 2^13 rows for the four to eight waves of its one workgroup.
                    sp.fused = h->geo.r == 4 && h->geo.k == 12 && sp.n_keys <= 3 && ((sp.outer[0] == 0 && sp.outer[1] == 0) || sp.halves || (sp.side_r == 3 && sp.outer[0] <= 1 && sp.outer[1] <= 1)) &&
                               sp.stats[0].n_passes == 1 && sp.stats[1].n_passes == 1;
                    if (const char* env = getenv("QSV_FUSED_MAX_KEYS")) sp.fused = sp.fused && sp.n_keys <= atoi(env);  // (measurements)
                    if (getenv("QSV_SPLIT_DEBUG"))
                        fprintf(stderr, "split: keys %d sides %d+%d virtual %d/%d tiles 2^%d/2^%d passes %d/%d fused %d\n", sp.n_keys, sc.n_side[0],
                                sc.n_side[1], sp.n_virtual[0], sp.n_virtual[1], sp.outer[0], sp.outer[1], sp.stats[0].n_passes,
                                sp.stats[1].n_passes, int(sp.fused));
                } else {
                    w.resize(sp.off_side[0] ? sp.off_side[0] : w.size());
                    sp = SplitInfo{};
                }
            }
        }
        if (out->split.ok) {
            out->gates = std::move(gates);
            out->angles = std::move(angles);
        } else {
            const CircuitPlan p = build_ordinary_plan(h, gates, angles, pc);
            out->off_plan = uint32_t(out->plan.words.size());
            out->plan.words.insert(out->plan.words.end(), p.words.begin(), p.words.end());
            out->plan.stats = p.stats;
            out->has_plan = true;
        }
    } catch (const std::exception& e) {
        if (err) *err = std::string("plan: ") + e.what();
        return QSV_E_ARG;
    }
    return QSV_OK;
}

// The ordinary plan of a circuit registered in split form, on first need (caller holds the handle).
int ensure_plan(qsv_t* h, Circuit& c) {
    if (c.has_plan) return QSV_OK;
    try {
        PlanConfig pc = h->cfg;
        pc.fold = pc.fold && c.fold;
        const CircuitPlan p = build_ordinary_plan(h, c.gates, c.angles, pc);
        c.off_plan = uint32_t(c.plan.words.size());
        c.plan.words.insert(c.plan.words.end(), p.words.begin(), p.words.end());
        c.plan.stats = p.stats;
    } catch (const std::exception& e) {
        return fail(h, QSV_E_ARG, std::string("plan: ") + e.what());
    }
    c.has_plan = true;
    c.uploaded = false;  // (the arena copy, if any, lacks the new words)
    c.gates.clear();
    c.gates.shrink_to_fit();
    c.angles.clear();
    c.angles.shrink_to_fit();
    return QSV_OK;
}

// Schedule `count` structures, in parallel when there are several.  build(i) fills circuits[i]; the first failure is
// reported.  Needs no handle lock (build_circuit reads only immutable parts of the handle).
struct BuiltCircuit {
    Circuit circuit;
    std::string err;
    int rc = QSV_OK;
};
void build_many(qsv_t* h, size_t count, const std::function<void(size_t, BuiltCircuit&)>& build, std::vector<BuiltCircuit>& out) {
    out.resize(count);
    if (count < 4) {
        for (size_t i = 0; i < count; ++i) build(i, out[i]);
        return;
    }
    WorkerPool* pool;
    {
        std::lock_guard<std::mutex> lock(h->pool_mu);
        if (!h->pool) {
            unsigned hw = std::max(2u, std::min(16u, std::thread::hardware_concurrency()));
            if (const char* env = getenv("QSV_BUILD_THREADS")) hw = std::max(2u, std::min(64u, unsigned(atoi(env))));
            h->pool.reset(new WorkerPool(hw - 1));
        }
        pool = h->pool.get();
    }
    // one job at a time per pool: callers on several threads take turns
    static std::mutex job_mu;
    std::lock_guard<std::mutex> job(job_mu);
    pool->run(count, [&](size_t i) { build(i, out[i]); });
}

// a circuit that continued kept state `id` is gone (caller holds h->mu)
void prefix_unref(qsv_t* h, int id) {
    auto it = h->prefixes.find(id);
    if (it == h->prefixes.end()) return;
    if (--it->second.refs <= 0 && it->second.released) {
        h->prefix_free.push_back(it->second.slot);
        h->prefixes.erase(it);
    }
}

// (caller holds h->mu)
int insert_circuit(qsv_t* h, Circuit&& c) {
    const int id = h->next_circuit_id++;
    h->circuits.emplace(id, std::move(c));
    return id;
}

int register_circuit(qsv_t* h, int n_ops, const qsv_op* ops, int n_params, int* out_id, bool fold = true) {
    Circuit c;
    std::string err;
    int rc = build_circuit(h, n_ops, ops, n_params, fold, &c, &err);
    if (rc) return fail(h, rc, err);
    *out_id = insert_circuit(h, std::move(c));
    return QSV_OK;
}

// Make the plans of `circs` resident in the device arena.  New plans are staged back to back in one pinned host buffer
// and shipped with ONE copy per batch (a population of fresh structures used to cost one copy + one stream
// synchronisation per circuit).  When the arena is full it is rebuilt from the circuits that are still registered: the
// space of destroyed circuits is reclaimed, and the arena only grows when the live plans really need more room.
int upload_plans(qsv_t* h, const std::vector<Circuit*>& circs) {
    std::vector<Circuit*> fresh;
    size_t need = 0, need_d = 0;
    // (doubles of the two sides' own tables of D: split circuits whose expectation comes from Gram sums -- a quadratic diagonal operator)
    auto side_diag_doubles = [&](const Circuit* c) -> size_t {
        if (!h->side_diag || !h->d_diag.ptr || !c->split.ok || !(h->diagonal && h->quadratic && h->d_quad.ptr)) return 0;
        const uint32_t* blk = c->plan.words.data() + c->split.off_block;
        return (size_t(1) << blk[1]) + (size_t(1) << blk[2]);
    };
    for (Circuit* c : circs)
        if (!c->uploaded && !c->staged) {
            c->staged = true;
            fresh.push_back(c);
            need += c->plan.words.size();
            need_d += side_diag_doubles(c);
        }
    for (Circuit* c : fresh) c->staged = false;
    if (fresh.empty()) return QSV_OK;
    const size_t cap = h->d_arena.bytes / 4;
    const bool tables_full = h->sdiag_used + need_d > h->d_sdiag.bytes / 8;
    if (h->arena_used_words + need > cap || tables_full) {
        // rebuild: everything still registered that is part of this batch goes in again; plans of other live circuits
        // are re-uploaded when they are next used
        QSV_HIP(h, sync_streams(h));
        for (auto& kv : h->circuits) kv.second.uploaded = false;
        fresh.clear();
        need = need_d = 0;
        for (Circuit* c : circs)
            if (!c->staged) {
                c->staged = true;
                fresh.push_back(c);
                need += c->plan.words.size();
                need_d += side_diag_doubles(c);
            }
        for (Circuit* c : fresh) c->staged = false;
        h->arena_used_words = 0;
        h->sdiag_used = 0;
        if (need_d > h->d_sdiag.bytes / 8 || (tables_full && h->d_sdiag.bytes < (size_t(256) << 20))) {
            // (it was full, or never there: twice the room up to 256 MiB, so that a run that keeps registering structures -- an
            // EVQE run does, every generation -- comes here, and uploads everything again, ever more rarely)
            const size_t new_doubles = std::max(std::max(need_d * 2, h->d_sdiag.bytes / 4), size_t(1) << 19);
            if (h->d_sdiag.ptr) QSV_HIP(h, hipFree(h->d_sdiag.ptr));
            h->d_sdiag = DeviceBuffer{};
            QSV_HIP(h, hipMalloc(&h->d_sdiag.ptr, new_doubles * 8));
            h->d_sdiag.bytes = new_doubles * 8;
        }
        if (getenv("QSV_ARENA_DEBUG"))
            fprintf(stderr, "plan arena: rebuilt for a batch of %zu plans, %zu words of %zu%s\n", fresh.size(), need, cap, need > cap ? " (grows)" : "");
        if (need > cap) {
            const size_t new_cap = std::max(need * 2, size_t(1) << 20);
            if (h->d_arena.ptr) QSV_HIP(h, hipFree(h->d_arena.ptr));
            h->d_arena = DeviceBuffer{};
            QSV_HIP(h, hipMalloc(&h->d_arena.ptr, new_cap * 4));
            h->d_arena.bytes = new_cap * 4;
        }
    }
    if (h->h_stage_words < need) {
        if (h->h_stage) {
            QSV_HIP(h, sync_streams(h));
            QSV_HIP(h, hipHostFree(h->h_stage));
            h->h_stage = nullptr;
            h->h_stage_words = 0;
        }
        const size_t want = std::max(need * 2, size_t(1) << 16);
        QSV_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), want * 4, hipHostMallocDefault));
        h->h_stage_words = want;
    } else {
        // the previous batch's copy out of the staging buffer must be complete before it is overwritten
        QSV_HIP(h, hipStreamSynchronize(h->stream));
    }
    size_t cur = 0;
    std::vector<SideDiagJob> jobs;
    for (Circuit* c : fresh) {
        if (c->split.ok) {
            // the sides' own tables of D: named in the split block, filled behind the copy below
            uint32_t* blk = c->plan.words.data() + c->split.off_block;
            blk[kSplitSideDiag] = blk[kSplitSideDiag + 1] = kNoSideDiag;
            if (side_diag_doubles(c))
                for (uint32_t xy = 0; xy < 2; ++xy) {
                    blk[kSplitSideDiag + xy] = uint32_t(h->sdiag_used);
                    jobs.push_back(SideDiagJob{uint32_t(h->sdiag_used), blk[kSplitMaskX + xy], blk[1 + xy]});
                    h->sdiag_used += size_t(1) << blk[1 + xy];
                }
        }
        std::memcpy(h->h_stage + cur, c->plan.words.data(), c->plan.words.size() * 4);
        c->plan_base = uint32_t(h->arena_used_words + cur);
        cur += c->plan.words.size();
        c->uploaded = true;
    }
    QSV_HIP(h, hipMemcpyAsync(static_cast<uint32_t*>(h->d_arena.ptr) + h->arena_used_words, h->h_stage, need * 4,
                              hipMemcpyHostToDevice, h->stream));
    h->arena_used_words += need;
    if (getenv("QSV_ARENA_DEBUG"))
        fprintf(stderr, "plan arena: %zu fresh plans, %zu side tables of D (%zu doubles used of %zu), diag %p\n", fresh.size(), jobs.size(), h->sdiag_used,
                h->d_sdiag.bytes / 8, h->d_diag.ptr);
    for (size_t first = 0; first < jobs.size(); first += size_t(kSideDiagJobsPerLaunch)) {
        SideDiagJobs batch{};
        const int n_jobs = int(std::min(jobs.size() - first, size_t(kSideDiagJobsPerLaunch)));
        std::copy(jobs.begin() + long(first), jobs.begin() + long(first) + n_jobs, batch.job);
        QSV_HIP(h, launch_side_diag(static_cast<const double*>(h->d_diag.ptr), static_cast<double*>(h->d_sdiag.ptr), batch, n_jobs, h->stream));
    }
    // launches on the second stream read the arena too: they wait for this copy
    if (!h->side_streams.empty()) {
        QSV_HIP(h, hipEventRecord(h->ev_join, h->stream));
        for (hipStream_t st : h->side_streams) QSV_HIP(h, hipStreamWaitEvent(st, h->ev_join, 0));
    }
    return QSV_OK;
}

int ensure_host_batch(qsv_t* h, size_t bytes) {
    if (h->h_batch_bytes >= bytes) return QSV_OK;
    if (h->h_batch) {
        QSV_HIP(h, sync_streams(h));
        QSV_HIP(h, hipHostFree(h->h_batch));
        h->h_batch = nullptr;
    }
    size_t want = std::max(bytes * 2, size_t(1) << 16);
    h->epoch += 1;
    QSV_HIP(h, hipHostMalloc(&h->h_batch, want, hipHostMallocDefault));
    h->h_batch_bytes = want;
    if (h->bar_ship) {
        if (h->d_ship) QSV_HIP(h, hipFree(h->d_ship));
        h->d_ship = nullptr;
        h->ship_shadow.clear();
        QSV_HIP(h, hipMalloc(&h->d_ship, want));
    }
    return QSV_OK;
}

// where the kernels read a batch's descriptors and parameters from
const void* ship_base(const qsv_t* h) { return h->bar_ship ? h->d_ship : h->h_batch; }
// where the kernels read the descriptors of the current batch from: a batch that repeats the one before it (same layout, kept:
// qsv_eval_begin) finds them in the device copy that batch's kernels left -- one PCIe round trip less at the start of every
// workgroup
const EvalDesc* descs_base(const qsv_t* h) {
    if (h->batch.repeat && !h->bar_ship && h->repeat_device_descs) return static_cast<const EvalDesc*>(h->d_batch.ptr);
    return static_cast<const EvalDesc*>(ship_base(h));
}
// where the kernels of the current push read parameter values from: the pinned staging buffer, or the caller's device memory
const double* params_base(const qsv_t* h) {
    if (h->batch.dev_params) return h->batch.dev_params;
    return reinterpret_cast<const double*>(static_cast<const char*>(h->h_batch) + h->batch.desc_bytes);
}

int ensure_host_out(qsv_t* h, size_t count) {
    if (h->h_out_count >= count) return QSV_OK;
    if (h->h_out) {
        QSV_HIP(h, sync_streams(h));
        QSV_HIP(h, hipHostFree(h->h_out));
        h->h_out = nullptr;
    }
    size_t want = std::max(count * 2, size_t(256));
    QSV_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_out), want * sizeof(double), hipHostMallocDefault));
    h->h_out_count = want;
    return QSV_OK;
}

struct EventPair {
    hipEvent_t a, b;
};

// ---- batches ---------------------------------------------------------------------------------------------
// A batch is laid out once (descriptors, parameter offsets, matrix regions), then evaluations are PUSHED in
// group-aligned slices: each push ships that slice's parameter values, turns them into matrices on the device and
// launches the slice's gate passes, all asynchronously, so the host can prepare the next slice meanwhile.

// doubles of an evaluation's matrix region(s): one for the ordinary plan, or one per virtual circuit when it runs split
size_t mat_doubles_of(const qsv_t* h, const Circuit& c, bool split, int side) {
    if (!split)
        return mat_region_doubles(uint32_t(c.plan.stats.n_real_gates), uint32_t(h->n), h->geo.t, h->n - h->geo.k,
                                  c.plan.stats.n_passes);
    return mat_region_doubles(uint32_t(c.split.stats[side].n_real_gates), uint32_t(c.split.n_virtual[side]), c.split.t[side],
                              c.split.outer[side], c.split.stats[side].n_passes);
}

// allow_split: the caller only wants <D> of the final states (fused diagonal expectation), so a circuit that has a
// split form (split.hpp) may run as its two virtual circuits + the contraction kernel.
// max_keys: split forms with more cut keys than the caller's kernels take (four and five keys: the factorised expectation
// under a quadratic operator only) are not used; such a circuit runs its ordinary plan.
int batch_layout(qsv_t* h, const std::vector<Circuit*>& circs, const std::vector<int64_t>& n_params, bool allow_split = false,
                 int max_keys = 3) {
    auto splits = [&](const Circuit* c) { return allow_split && c->split.ok && c->split.n_keys <= max_keys; };
    qsv_handle::Batch& b = h->batch;
    const size_t n_evals = circs.size();
    int rc;
    h->epoch += 1;  // (whatever an earlier batch left in the staging buffer is overwritten below)
    b.repeat = false;
    if (h->async_pending) {  // the kernels of a batch that ended without waiting read the staging buffers written below
        QSV_HIP(h, sync_streams(h));
        h->async_pending = false;
    }
    {
        // ordinary plans that are needed now and do not exist yet (circuits registered in split form): scheduled on the
        // host's worker threads when there are several
        std::vector<Circuit*> missing;
        for (Circuit* c : circs)
            if (!splits(c) && !c->has_plan && std::find(missing.begin(), missing.end(), c) == missing.end())
                missing.push_back(c);
        if (missing.size() >= 4) {
            std::vector<BuiltCircuit> built;
            build_many(h, missing.size(), [&](size_t i, BuiltCircuit& out) {
                Circuit& c = *missing[i];
                try {
                    PlanConfig pc = h->cfg;
                    pc.fold = pc.fold && c.fold;
                    out.circuit.plan = build_ordinary_plan(h, c.gates, c.angles, pc);
                } catch (const std::exception& e) {
                    out.rc = QSV_E_ARG;
                    out.err = std::string("plan: ") + e.what();
                }
            }, built);
            for (size_t i = 0; i < missing.size(); ++i) {
                if (built[i].rc) return fail(h, built[i].rc, built[i].err);
                Circuit& c = *missing[i];
                const CircuitPlan& p = built[i].circuit.plan;
                c.off_plan = uint32_t(c.plan.words.size());
                c.plan.words.insert(c.plan.words.end(), p.words.begin(), p.words.end());
                c.plan.stats = p.stats;
                c.has_plan = true;
                c.uploaded = false;
                c.gates.clear();
                c.angles.clear();
            }
        } else {
            for (Circuit* c : missing)
                if ((rc = ensure_plan(h, *c))) return rc;
        }
    }
    if ((rc = upload_plans(h, circs))) return rc;
    size_t total_params = 0, total_mats = 0;
    b.split.assign(n_evals, 0);
    b.split_any = false;
    b.cont_any = false;
    b.eval_at.resize(n_evals);
    for (size_t i = 0; i < n_evals; ++i) b.eval_at[i] = uint32_t(i);
    for (size_t i = 0; i < n_evals; ++i) {
        if (n_params[i] < circs[i]->n_params)
            return fail(h, QSV_E_ARG, "circuit needs " + std::to_string(circs[i]->n_params) + " parameter values, got " +
                                          std::to_string(n_params[i]));
        total_params += size_t(n_params[i]);
        const bool split = splits(circs[i]);
        b.split[i] = split;
        b.split_any |= split;
        total_mats += mat_doubles_of(h, *circs[i], split, 0) + (split ? mat_doubles_of(h, *circs[i], true, 1) : 0);
    }
    if (total_params >= (size_t(1) << 31) || total_mats >= (size_t(1) << 31))
        return fail(h, QSV_E_ARG, "batch too large");
    b.desc_bytes = ((sizeof(EvalDesc) * n_evals * (b.split_any ? 2 : 1) + 63) / 64) * 64;
    const size_t bytes = b.desc_bytes + (total_params + 1) * sizeof(double);
    if ((rc = ensure_host_batch(h, bytes))) return rc;
    if ((rc = ensure(h, h->d_batch, bytes))) return rc;
    if ((rc = ensure(h, h->d_mats, total_mats * sizeof(double)))) return rc;
    EvalDesc* hd = static_cast<EvalDesc*>(h->h_batch);
    b.circs = circs;
    b.param_base.resize(n_evals);
    b.n_params.resize(n_evals);
    size_t pcur = 0, mcur = 0;
    for (size_t i = 0; i < n_evals; ++i) {
        const Circuit& c = *circs[i];
        const uint32_t slot = uint32_t(i % size_t(h->group));
        if (!b.split[i]) {
            uint32_t flags = 0, kept_slot = 0;
            if (c.prefix_id >= 0) {  // (alive: a circuit registered on a kept state holds it)
                flags = kEvalPrefix;
                kept_slot = h->prefixes.find(c.prefix_id)->second.slot;
                b.cont_any = true;
            }
            hd[i] = EvalDesc{c.plan_base + c.off_plan, uint32_t(mcur), slot, uint32_t(i), uint32_t(pcur), uint32_t(n_params[i]), flags, kept_slot};
            mcur += mat_doubles_of(h, c, false, 0);
            if (b.split_any) hd[n_evals + i] = EvalDesc{0, 0, slot, uint32_t(i), 0, 0, kEvalNull, 0};
        } else {
            // one descriptor per virtual circuit: side A here, side B in the second region
            for (int s = 0; s < 2; ++s) {
                hd[size_t(s) * n_evals + i] =
                    EvalDesc{c.plan_base + c.split.off_side[s], uint32_t(mcur), slot, uint32_t(i), uint32_t(pcur),
                             uint32_t(n_params[i]), kEvalSide | (s ? kEvalSideB : 0u) | (c.split.fused ? kEvalFused : 0u) | (c.split.fused && c.split.halves ? kEvalHalves : 0u),
                             c.plan_base + c.split.off_block};
                mcur += mat_doubles_of(h, c, true, s);
            }
        }
        b.param_base[i] = uint32_t(pcur);
        b.n_params[i] = uint32_t(n_params[i]);
        pcur += size_t(n_params[i]);
        h->prof.n_gates += uint64_t(c.n_gates);
    }
    // (no copy here: prepare_kernel reads descriptors and parameters from this pinned buffer and writes the device
    // copy of the descriptors itself)
    b.pushed = 0;
    return QSV_OK;
}

const EvalDesc* batch_evals(const qsv_t* h) { return static_cast<const EvalDesc*>(h->d_batch.ptr); }

// Ship the parameter values of evaluations [first, first+count) (packed back to back in `values`) and prepare
// their matrices.
// (The first n_fused evaluations of the range are split evaluations: the pass kernel prepares their virtual circuits
// itself, kModeFusedPrepare.)
int batch_ship(qsv_t* h, size_t first, size_t count, const double* values, size_t n_fused = 0) {
    qsv_handle::Batch& b = h->batch;
    if (count == 0) return QSV_OK;
    const size_t p0 = b.param_base[first];
    const size_t p1 = size_t(b.param_base[first + count - 1]) + b.n_params[first + count - 1];
    double* hp = reinterpret_cast<double*>(static_cast<char*>(h->h_batch) + b.desc_bytes);
    if (p1 > p0 && !b.dev_params && values != hp + p0) std::memcpy(hp + p0, values, (p1 - p0) * sizeof(double));  // (qsv_eval_staging: in place)
    if (h->bar_ship) {
        // the push's descriptors (both regions of a batch with split evaluations) into the device copy: plain stores through
        // the write-combining BAR mapping, fenced before any launch that reads them is queued -- and only what differs from
        // what the copy already holds (a population evaluated again: nothing; the copy costs 0.5 us per 2 KiB, the compare
        // nothing).  The parameters stay in pinned memory: 30 KiB through the BAR cost the host what the kernel saves.
        char* dst = static_cast<char*>(h->d_ship);
        const char* src = static_cast<const char*>(h->h_batch);
        const size_t P = b.circs.size();
        if (h->ship_shadow.size() < b.desc_bytes) h->ship_shadow.assign(b.desc_bytes, char(0xff));
        bool wrote = false;
        for (size_t region = 0; region < (b.split_any ? 2u : 1u); ++region) {
            const size_t at = (region * P + first) * sizeof(EvalDesc), bytes = count * sizeof(EvalDesc);
            if (at + bytes > b.desc_bytes || std::memcmp(h->ship_shadow.data() + at, src + at, bytes) == 0) continue;
            std::memcpy(dst + at, src + at, bytes);
            std::memcpy(h->ship_shadow.data() + at, src + at, bytes);
            wrote = true;
        }
        if (wrote) _mm_sfence();
    }
    const EvalDesc* host_evals = descs_base(h);
    const double* ship_params = params_base(h);
    if (count > n_fused)
        QSV_HIP(h, launch_prepare(static_cast<const uint32_t*>(h->d_arena.ptr), host_evals + first + n_fused,
                                  static_cast<EvalDesc*>(h->d_batch.ptr) + first + n_fused, ship_params,
                                  static_cast<double*>(h->d_mats.ptr), int(count - n_fused), ws(h), 1, 0, h->dtype));
    return QSV_OK;
}

// One workgroup per evaluation (the register fits one tile) under a diagonal operator: the pass kernel prepares the
// evaluation itself and writes the expectation value straight to the result buffer -- one launch per push instead of three.
bool single_workgroup_path(const qsv_t* h, uint32_t mode) {
    return h->geo.blocks_per_state == 1 && h->diagonal && (mode & kModeFinalDiag) && !(mode & kModeFinalStore) &&
           (mode & kModeSynthFirst) && !getenv("QSV_NO_DIRECT");
}

unsigned chunks_per_state(const qsv_t* h) {
    const unsigned tpb = unsigned(std::min<uint32_t>(uint32_t(h->tiles_per_block), h->geo.blocks_per_state));
    return h->geo.blocks_per_state / tpb;
}

// partial sums the fused last pass leaves per state: one per wave of every workgroup
unsigned partials_per_state(const qsv_t* h) { return chunks_per_state(h) * unsigned(h->geo.threads_launch / 64); }

hipError_t stamp(qsv_t* h, std::vector<std::pair<hipEvent_t, hipEvent_t>>& list, bool begin) {
    if (!h->profiling) return hipSuccess;
    if (begin) {
        std::pair<hipEvent_t, hipEvent_t> p{nullptr, nullptr};
        hipError_t e = hipEventCreate(&p.first);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&p.second);
        if (e != hipSuccess) return e;
        list.push_back(p);
        return hipEventRecord(p.first, ws(h));
    }
    return hipEventRecord(list.back().second, ws(h));
}

// Split evaluations need no contraction sweep under a quadratic diagonal operator (kernels.hpp: launch_factor).
bool factor_path(const qsv_t* h) { return h->factor_enabled && h->diagonal && h->quadratic && h->d_quad.ptr != nullptr && h->d_factor.ptr != nullptr; }

// ... and none under a general operator (kernels.hpp: launch_factor_terms).
bool factor_terms_path(const qsv_t* h) { return h->factor_enabled && !h->diagonal && h->n_fterms > 0 && h->d_side.ptr != nullptr; }

// Split evaluations flagged kEvalFused are finished by the launch that runs their virtual circuits (quadratic operator).
bool fused_route(const qsv_t* h) { return factor_path(h) && h->d_factor_count.ptr != nullptr && h->fused_factor; }

// Workgroups of a one-launch evaluation that do not leave at once: one per side, two for a half side (kernels.hpp kEvalHalves).
size_t working_workgroups(const SplitInfo& sp) {
    return 2u + (sp.halves ? size_t(sp.n_virtual[0] == kFusedLdsRowsBits) + size_t(sp.n_virtual[1] == kFusedLdsRowsBits) : 0u);
}

// Run the gate passes of evaluations [first, first+count) of the current batch (one launch group).
int run_group(qsv_t* h, const std::vector<Circuit*>& circs, size_t first, size_t count, uint32_t mode) {
    const qsv_handle::Batch& b = h->batch;
    const bool batch_split = b.split_any && b.split.size() == circs.size();
    const bool reordered = (b.split_any || b.cont_any) && b.split.size() == circs.size();
    // descriptor position -> evaluation (eval_push put the push's split evaluations first, those on kept states last)
    auto eval_of = [&](size_t pos) { return reordered ? size_t(b.eval_at[pos]) : pos; };
    size_t n_split = 0, n_cont = 0;
    int max_passes = 0;
    for (size_t i = 0; i < count; ++i) {
        const size_t e = eval_of(first + i);
        if (batch_split && b.split[e]) {
            n_split = i + 1;  // (split evaluations lead each push, so also each group of it)
        } else {
            max_passes = std::max(max_passes, circs[e]->plan.stats.n_passes);
            n_cont += circs[e]->prefix_id >= 0 ? 1 : 0;
        }
    }
    // Evaluations that continue a kept state read it in their first pass (the later-pass instantiation of the kernel, no
    // synthesis): a launch group holds only such evaluations, or none
    if (n_cont > 0) {
        if (n_cont != count) return fail(h, QSV_E_STATE, "internal: a launch group mixes evaluations on kept states with others");
        mode &= ~uint32_t(kModeSynthFirst | kModeFusedPrepare | kModeDirectResult);
    }
    const bool any_split = n_split > 0;
    const size_t n_plain = count - n_split;
    PassArgs a{};
    a.plan = static_cast<const uint32_t*>(h->d_arena.ptr);
    a.mats = static_cast<const double*>(h->d_mats.ptr);
    a.evals = batch_evals(h) + first;
    a.states = h->d_states.ptr;
    a.wtab = h->d_wtab.ptr;
    a.wtab_stride = h->wtab_stride;
    a.diag = static_cast<const double*>(h->d_diag.ptr);
    a.partials = static_cast<double*>(h->d_partials.ptr);
    a.state_stride = uint64_t(1) << h->n;
    a.mode = mode | h->stream_mode;
    a.prefix_states = h->d_prefix.ptr;
    a.side_diag = static_cast<const double*>(h->d_sdiag.ptr);
    {
        static const uint32_t dephase = getenv("QSV_DEPHASE") ? uint32_t(atoi(getenv("QSV_DEPHASE"))) : 0u;
        a.dephase = dephase;
    }
    // Pass 0 and the later passes have their own grids (tiles per workgroup): a compact pass 0 has few tiles and wants
    // them spread, a later pass sweeps all of them and amortises its set-up over more.  Which grid an evaluation's
    // LAST pass runs on depends on its own pass count only, so its partial sums are laid out (and added) the same
    // way in any batch.
    const unsigned chunks = chunks_per_state(h);
    const unsigned tpb_later = unsigned(std::min<uint32_t>(uint32_t(h->tiles_per_block_later), h->geo.blocks_per_state));
    const unsigned chunks_later = std::max(1u, std::min(chunks, h->geo.blocks_per_state / std::max(1u, tpb_later)));
    a.partial_chunks = chunks;
    a.region_stride = uint32_t(circs.size());
    const uint64_t sweep = (uint64_t(1) << h->n) * h->amp_bytes;
    // of the split evaluations (they lead the group) the first n_unfused need launches of their own after the virtual
    // circuits; the others are finished by the launch that runs theirs (kEvalFused, under a quadratic operator only)
    size_t n_unfused = n_split;
    const bool fuse_ok = fused_route(h) && !(mode & kModeSidesOnly);
    if (fuse_ok) {
        n_unfused = 0;
        while (n_unfused < n_split && !circs[eval_of(first + n_unfused)]->split.fused) ++n_unfused;
    }
    const bool two_chains = b.chain_now && h->chain_stream != -1 && n_unfused > 0 && n_split > n_unfused && factor_path(h);
    // (the chain's stream: the second lane's, or -- for a push on the second lane -- the handle's own, which `work` = null means)
    hipStream_t const chain_st = h->chain_stream >= 0 ? h->side_streams[size_t(h->chain_stream)] : nullptr;
    if (any_split) {
        if (n_plain > 0) return fail(h, QSV_E_STATE, "internal: a launch group mixes split and ordinary evaluations");
        a.wtab = h->d_side.ptr;  // (the side tables' own slots)
        a.wtab_stride = h->side_stride;
        a.quad = static_cast<const double*>(h->d_quad.ptr);
        a.factor_scratch = static_cast<double*>(h->d_factor.ptr);
        a.factor_counters = static_cast<uint32_t*>(h->d_factor_count.ptr);
        a.n_full = uint32_t(h->n);
        a.result_out = h->out_target ? h->out_target : h->h_out;
        a.tiles_per_block = 1;
        a.host_params = params_base(h);
        a.mats_out = static_cast<double*>(h->d_mats.ptr);
        // threads and LDS of a launch over split evaluations [lo, hi) of the group: a side's tile may be larger than the
        // handle's (build_circuit)
        auto shape_of = [&](size_t lo, size_t hi, int* threads, size_t* lds, int* passes, unsigned* tiles) {
            int tile = 0, t = 0;
            *passes = 1;
            *tiles = 1;
            for (size_t i = lo; i < hi; ++i) {
                const SplitInfo& sp = circs[eval_of(first + i)]->split;
                for (int s = 0; s < 2; ++s) {
                    tile = std::max(tile, sp.tile_bits[s]);
                    t = std::max(t, sp.t[s]);
                    *passes = std::max(*passes, sp.stats[s].n_passes);
                    *tiles = std::max(*tiles, 1u << sp.outer[s]);
                }
            }
            *threads = std::max(64, 1 << t);
            *lds = (size_t(1) << tile) * h->amp_bytes / (h->cfg.xmode == 2 ? 2 : 1);
        };
        auto at_least_four_waves = [](int threads) { return std::max(256, threads); };  // (the fused factor tail's Gram waves)
        auto launch_sides = [&](size_t lo, size_t hi, uint32_t extra_mode, int r) -> int {
            int threads, passes;
            size_t lds;
            unsigned tiles;
            shape_of(lo, hi, &threads, &lds, &passes, &tiles);
            if (extra_mode & kModeFusedFactor) threads = at_least_four_waves(threads);
            // (fused: ONE workgroup per side, one tile -- or, a half side, one of its two tiles -- then on to the side's Gram matrices)
            bool any_halves = false;
            if (extra_mode & kModeFusedFactor)
                for (size_t i = lo; i < hi; ++i) any_halves |= circs[eval_of(first + i)]->split.halves;
            // (a side's one workgroup sweeps its tiles itself, whatever the grid's width: pass_kernel)
            a.tiles_per_block = (extra_mode & kModeFusedFactor) ? tiles : 1u;
            const unsigned grid_x = (extra_mode & kModeFusedFactor) ? (any_halves ? 2u : 1u) : tiles;
            // (pass 0 prepares the virtual circuits' matrices and tables itself: no prepare launch ran for them)
            a.evals = batch_evals(h) + first + lo;
            a.host_evals = descs_base(h) + first + lo;
            a.evals_out = static_cast<EvalDesc*>(h->d_batch.ptr) + first + lo;
            for (int p = 0; p < passes; ++p) {
                a.pass_index = uint32_t(p);
                a.mode = ((p == 0 ? (mode | kModeFusedPrepare | extra_mode) : mode) & ~uint32_t(kModeSidesOnly)) | h->stream_mode;
                const int kind = p == 0 ? 0 : 1;
                size_t need = lds;
                if (p == 0) need = std::max(need, kFusedPrepareLdsBytes);
                if (p == 0 && (extra_mode & kModeFusedFactor)) need = std::max(need, kFusedFactorLdsBytes);
                // at most one workgroup per CU in flight anyway (two per evaluation): room for the sides' states in LDS
                size_t working = 0;
                for (size_t i = lo; i < hi; ++i) working += working_workgroups(circs[eval_of(first + i)]->split);
                if (p == 0 && (extra_mode & kModeFusedFactor) && h->dtype == QSV_F64 && (h->fused_lds_table || any_halves) && working <= size_t(h->n_cus)) {
                    a.mode |= kModeFusedLdsTable;
                    need = std::max(need, kFusedLdsTableEnd);
                }
                if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[kind], true));
                QSV_HIP(h, launch_pass(h->dtype, r, h->cfg.xmode, dim3(grid_x, unsigned(hi - lo), 2), threads, need, ws(h), a));
                if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[kind], false));
                h->prof.n_pass_launches += 1;
                h->prof.kernel_launches[kind] += 1;
            }
            return QSV_OK;
        };
        int rc2;
        // (both kinds in one group: the ones with launches of their own start first, on the chain stream where the push
        // has one -- theirs is the longer chain)
        hipStream_t const lane_of_group = h->work;
        if (two_chains) h->work = chain_st;
        // (a launch is one instantiation of the kernel: within each kind the sides with eight amplitudes per thread come first,
        // order_split_first; half sides need their states in LDS, i.e. at most one workgroup per compute unit: four per evaluation)
        auto launch_sides_by_r = [&](size_t lo, size_t hi, uint32_t extra_mode) -> int {
            // (runs of evaluations whose sides have the same number of amplitudes per thread: order_split_first puts the eight-
            // amplitude ones first within each kind, but a range may hold both kinds -- every split evaluation under an operator
            // that has no one-launch route)
            for (size_t at = lo; at < hi;) {
                const int r = circs[eval_of(first + at)]->split.side_r == 3 ? 3 : h->geo.r;
                size_t end = at;
                bool halves_here = false;
                while (end < hi && (circs[eval_of(first + end)]->split.side_r == 3 ? 3 : h->geo.r) == r) {
                    halves_here |= circs[eval_of(first + end)]->split.halves;
                    ++end;
                }
                // (half sides need the launch's states in LDS: at most one working workgroup per compute unit)
                for (size_t from = at; from < end;) {
                    size_t to = end;
                    if (halves_here && (extra_mode & kModeFusedFactor)) {
                        size_t wg = 0;
                        for (to = from; to < end && wg + working_workgroups(circs[eval_of(first + to)]->split) <= size_t(h->n_cus); ++to)
                            wg += working_workgroups(circs[eval_of(first + to)]->split);
                        if (to == from) to = from + 1;
                    }
                    const int rc3 = launch_sides(from, to, extra_mode, r);
                    if (rc3) return rc3;
                    from = to;
                }
                at = end;
            }
            return QSV_OK;
        };
        if (n_unfused > 0 && (rc2 = launch_sides_by_r(0, n_unfused, 0u))) return rc2;
        h->work = lane_of_group;
        // (the one-launch kind: as one launch -- unless the group is too large for one with the sides' states in LDS and holds half
        // sides, which lead the range: those go first, to the chain stream where the push has one free, the others follow as before)
        size_t n_halved = 0, all_working = 0;
        while (n_unfused + n_halved < n_split && circs[eval_of(first + n_unfused + n_halved)]->split.halves) ++n_halved;
        for (size_t i = n_unfused; i < n_split; ++i) all_working += working_workgroups(circs[eval_of(first + i)]->split);
        if (n_halved > 0 && n_unfused + n_halved < n_split && all_working > size_t(h->n_cus)) {
            // (too many for one launch with the sides' states in LDS, which half sides need: those few lead the range and get a
            // launch of their own -- on the chain stream where the push has one free --, the others follow as one launch, their
            // states handed over through their slots as in any launch of that size.  Measured, 128 evaluations with one half-sided
            // circuit: 56 us per step; cut into launches of one working workgroup per compute unit each: 75)
            if (b.chain_now && h->chain_stream != -1 && n_unfused == 0) h->work = chain_st;
            if ((rc2 = launch_sides_by_r(n_unfused, n_unfused + n_halved, kModeFusedFactor))) return rc2;
            h->work = lane_of_group;
            if ((rc2 = launch_sides_by_r(n_unfused + n_halved, n_split, kModeFusedFactor))) return rc2;
        } else if (n_split > n_unfused && (rc2 = launch_sides_by_r(n_unfused, n_split, kModeFusedFactor))) return rc2;
        a.mode = mode | h->stream_mode;
        // what the side circuits move: they synthesise their input and write their final states
        for (size_t i = 0; i < n_split; ++i) {
            const SplitInfo& sp = circs[eval_of(first + i)]->split;
            const uint64_t bytes = ((uint64_t(1) << sp.n_virtual[0]) + (uint64_t(1) << sp.n_virtual[1])) * h->amp_bytes;
            h->prof.state_bytes += bytes;
            h->prof.moved_bytes += bytes;
            h->prof.kernel_bytes[0] += bytes;
            h->prof.kernel_moved_bytes[0] += bytes;
        }
        // (the fused ones: what their Gram matrices read and compute is part of that launch)
        for (size_t i = n_unfused; i < n_split; ++i) {
            const SplitInfo& sp = circs[eval_of(first + i)]->split;
            for (int s = 0; s < 2; ++s) {
                const int side_bits = sp.n_virtual[s] - sp.n_keys;
                const uint64_t moved = (uint64_t(1) << sp.n_virtual[s]) * h->amp_bytes + (uint64_t(8) << side_bits);
                h->prof.kernel_bytes[0] += moved;
                h->prof.kernel_moved_bytes[0] += moved;
                h->prof.kernel_flops[0] += double(uint64_t(1) << side_bits) * double(1u << (2 * sp.n_keys)) * (side_bits / 2.0 + 5.0) * 2.0;
                h->prof.kernel_states[0] += 1;
                if (!sp.stats[s].pass_pairs.empty()) h->prof.kernel_flops[0] += 24.0 * sp.stats[s].pass_pairs[0];
            }
            h->prof.kernel_states[2] += 1;
        }
    }
    a.evals = batch_evals(h) + first + n_split;
    const bool direct = mode & kModeDirectResult;  // (eval_push decides, for the whole push)
    const bool fused = direct || (mode & kModeFusedPrepare);  // (... or the sampler path, for one-tile registers)
    if (mode & kModeFinalProbs) a.partials = static_cast<double*>(h->d_scratch.ptr);  // [slot][2^n] probabilities
    if (fused) {
        a.mode |= kModeFusedPrepare;
        a.host_evals = descs_base(h) + first + n_split;
        a.evals_out = static_cast<EvalDesc*>(h->d_batch.ptr) + first + n_split;
        a.host_params = params_base(h);
        a.mats_out = static_cast<double*>(h->d_mats.ptr);
        a.result_out = h->out_target ? h->out_target : h->h_out;
    }
    for (int p = 0; p < max_passes && n_plain > 0; ++p) {
        const unsigned chunks_p = p == 0 ? chunks : chunks_later;
        a.tiles_per_block = (h->geo.blocks_per_state + chunks_p - 1) / chunks_p;
        dim3 grid(chunks_p, unsigned(n_plain));
        a.pass_index = uint32_t(p);
        {
            static const bool tile_major = getenv("QSV_TILE_MAJOR") && atoi(getenv("QSV_TILE_MAJOR")) != 0;
            const bool swap = tile_major && !(p == 0 && (mode & kModeSynthFirst)) && n_plain > 1;
            a.mode = swap ? (a.mode | kModeTileMajor) : (a.mode & ~uint32_t(kModeTileMajor));
            if (swap) grid = dim3(unsigned(n_plain), chunks_p);
        }
        const int kind = (p == 0 && (mode & kModeSynthFirst)) ? 0 : 1;  // which instantiation of the kernel runs
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[kind], true));
        QSV_HIP(h, launch_pass(h->dtype, h->geo.r, h->cfg.xmode, grid, h->geo.threads_launch,
                               std::max(h->geo.lds_bytes, fused ? kFusedPrepareLdsBytes : size_t(0)), ws(h), a));
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[kind], false));
        h->prof.n_pass_launches += 1;
        h->prof.n_state_passes += n_plain;
        h->prof.kernel_launches[kind] += 1;
        // Algorithmic state bytes at this pass's own price: every pass reads and writes the state once, except that a
        // synthesising pass 0 does not read and a fused last pass does not write.  What it really moves is less when
        // a compact pass 0 writes, and pass 1 reads, a table of 2^cb tiles instead of the state.
        for (size_t i = n_split; i < count; ++i) {
            const PlanStats& st = circs[eval_of(first + i)]->plan.stats;
            if (p >= st.n_passes) continue;
            const bool reads = !(p == 0 && (mode & kModeSynthFirst));
            const bool writes = !(p + 1 == st.n_passes && !(mode & kModeFinalStore));
            const uint64_t table = (mode & kModeSynthFirst) && st.compact_bits >= 0
                                       ? (uint64_t(1) << (st.compact_bits + h->geo.k)) * h->amp_bytes
                                       : sweep;
            const uint64_t alg = (reads ? sweep : 0) + (writes ? sweep : 0);
            const uint64_t moved = (reads ? (p == 1 ? table : sweep) : 0) + (writes ? (p == 0 ? table : sweep) : 0);
            h->prof.state_bytes += alg;
            h->prof.moved_bytes += moved;
            h->prof.kernel_bytes[kind] += alg;
            h->prof.kernel_moved_bytes[kind] += moved;
            h->prof.kernel_states[kind] += 1;
            if (size_t(p) < st.pass_pairs.size()) h->prof.kernel_flops[kind] += 24.0 * st.pass_pairs[size_t(p)];
        }
    }
    if (any_split && !(mode & kModeSidesOnly) && factor_path(h)) {
      if (n_unfused > 0) {
        hipStream_t const lane_of_group = h->work;
        struct Restore {
            qsv_t* h;
            hipStream_t st;
            ~Restore() { h->work = st; }
        } restore{h, lane_of_group};
        if (two_chains) h->work = chain_st;
        // quadratic operator: the expectation value from the two side tables alone, written straight to the result buffer
        int most_keys = 0;
        for (size_t i = 0; i < n_unfused; ++i) most_keys = std::max(most_keys, circs[eval_of(first + i)]->split.n_keys);
        if (most_keys > 3) {
            int rc3 = ensure(h, h->d_factor_big, factor_big_slot_doubles() * sizeof(double) * size_t(h->side_slots));
            if (rc3) return rc3;
            if (!h->d_factor_big_count.ptr) {
                const size_t cbytes = factor_big_slot_counters() * sizeof(uint32_t) * size_t(h->side_slots);
                if ((rc3 = ensure(h, h->d_factor_big_count, cbytes))) return rc3;
                QSV_HIP(h, hipMemsetAsync(h->d_factor_big_count.ptr, 0, cbytes, ws(h)));
            }
        }
        a.evals = batch_evals(h) + first;
        a.result_out = h->out_target ? h->out_target : h->h_out;
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[2], true));
        QSV_HIP(h, launch_factor(h->dtype, unsigned(n_unfused), static_cast<double*>(h->d_factor.ptr),
                                 static_cast<const double*>(h->d_quad.ptr), h->n, ws(h), a, static_cast<double*>(h->d_factor_big.ptr),
                                 static_cast<uint32_t*>(h->d_factor_big_count.ptr), most_keys));
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[2], false));
        h->prof.kernel_launches[2] += 1;  // (the launches of the route, timed as one)
      }
        for (size_t i = 0; i < n_unfused; ++i) {
            const SplitInfo& sp = circs[eval_of(first + i)]->split;
            // what the two kernels read per state: the side tables and one value of D per table row
            uint64_t moved = 0;
            double flops = 0.0;
            for (int s = 0; s < 2; ++s) {
                const int side_bits = sp.n_virtual[s] - sp.n_keys;
                moved += (uint64_t(1) << sp.n_virtual[s]) * h->amp_bytes + (uint64_t(8) << side_bits);
                flops += double(uint64_t(1) << side_bits) * double(1u << (2 * sp.n_keys)) * (side_bits / 2.0 + 5.0) * 2.0;
                h->prof.kernel_states[0] += 1;
                if (!sp.stats[s].pass_pairs.empty()) h->prof.kernel_flops[0] += 24.0 * sp.stats[s].pass_pairs[0];
            }
            h->prof.state_bytes += moved;
            h->prof.kernel_bytes[2] += moved;
            h->prof.moved_bytes += moved;
            h->prof.kernel_moved_bytes[2] += moved;
            h->prof.kernel_states[2] += 1;
            h->prof.kernel_flops[2] += flops;
        }
    } else if (any_split && !(mode & kModeSidesOnly) && !h->diagonal) {
        // general operator: every term's two small matrices from the side tables, summed per evaluation into d_out
        a.evals = batch_evals(h) + first;
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[2], true));
        QSV_HIP(h, launch_factor_terms(h->dtype, unsigned(n_split), static_cast<const FactorTerm*>(h->d_fterms.ptr), h->n_fterms,
                                       static_cast<double*>(h->d_fpart.ptr), ws(h), a));
        QSV_HIP(h, launch_reduce_partials(static_cast<const double*>(h->d_fpart.ptr), kFactorTermWaves, int(n_split),
                                          static_cast<double*>(h->d_out.ptr), ws(h), batch_evals(h) + first));
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[2], false));
        h->prof.kernel_launches[2] += 1;
        for (size_t i = 0; i < n_split; ++i) {
            const SplitInfo& sp = circs[eval_of(first + i)]->split;
            for (int s = 0; s < 2; ++s) {
                const uint64_t bytes = (uint64_t(1) << sp.n_virtual[s]) * h->amp_bytes * uint64_t(h->n_fterms);
                h->prof.kernel_bytes[2] += bytes;
                h->prof.kernel_moved_bytes[2] += bytes;
                h->prof.kernel_flops[2] += 8.0 * double(uint64_t(1) << sp.n_virtual[s]) * double(1u << sp.n_keys) * double(h->n_fterms);
                h->prof.kernel_states[0] += 1;
            }
            h->prof.kernel_states[2] += 1;
        }
    } else if (any_split && !(mode & kModeSidesOnly)) {
        // the contraction of the split evaluations
        a.evals = batch_evals(h) + first;
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[2], true));
        const unsigned contract_chunks = unsigned((uint64_t(1) << h->n) / (uint64_t(h->geo.threads_launch) << kSplitLoopBits));
        QSV_HIP(h, launch_contract(h->dtype, contract_chunks, unsigned(n_split), h->geo.threads_launch, ws(h), a));
        if (h->stamping) QSV_HIP(h, stamp(h, h->batch.launch_events[2], false));
        h->prof.kernel_launches[2] += 1;
        for (size_t i = 0; i < n_split; ++i) {
            const SplitInfo& sp = circs[eval_of(first + i)]->split;
            // what the contraction reads per state: D once (8 * 2^n bytes) and the two side tables
            const uint64_t moved = (uint64_t(8) << h->n) + ((uint64_t(1) << sp.n_virtual[0]) + (uint64_t(1) << sp.n_virtual[1])) * h->amp_bytes;
            h->prof.state_bytes += moved;
            h->prof.kernel_bytes[2] += moved;
            h->prof.moved_bytes += moved;
            h->prof.kernel_moved_bytes[2] += moved;
            h->prof.kernel_states[2] += 1;
            h->prof.kernel_flops[2] += double(uint64_t(1) << h->n) * (8.0 * double(1u << sp.n_keys) + 5.0);
            for (int s = 0; s < 2; ++s) {
                h->prof.kernel_states[0] += 1;
                if (!sp.stats[s].pass_pairs.empty()) h->prof.kernel_flops[0] += 24.0 * sp.stats[s].pass_pairs[0];
            }
        }
    }
    return QSV_OK;
}

int eval_begin(qsv_t* h, const std::vector<Circuit*>& circs, const std::vector<int64_t>& n_params) {
    if (h->n_terms == 0) return fail(h, QSV_E_STATE, "no operator set (call qsv_set_operator first)");
    h->prof = qsv_profile{};
    h->prof.n_evals = uint64_t(circs.size());
    int rc;
    const size_t n_evals = circs.size();
    if ((rc = ensure(h, h->d_partials, std::max<size_t>(1, n_evals) * partials_per_state(h) * sizeof(double)))) return rc;
    if ((rc = ensure(h, h->d_out, std::max<size_t>(1, n_evals) * sizeof(double)))) return rc;
    if ((rc = ensure_host_out(h, std::max<size_t>(1, n_evals)))) return rc;
    if (h->profiling) {
        QSV_HIP(h, hipEventCreate(&h->batch.ev0));
        QSV_HIP(h, hipEventCreate(&h->batch.ev1));
        QSV_HIP(h, hipEventRecord(h->batch.ev0, h->stream));
    }
    if ((rc = batch_layout(h, circs, n_params, h->diagonal || factor_terms_path(h), factor_path(h) ? kMaxSplitKeys : 3))) return rc;
    if (factor_terms_path(h) && h->batch.split_any &&
        (rc = ensure(h, h->d_fpart, std::max<size_t>(1, n_evals) * kFactorTermWaves * sizeof(double))))
        return rc;
    qsv_handle::Batch& b = h->batch;
    // Two streams: consecutive pushes alternate between them, so that kernels of different pushes share the chip
    // (+15 % on the benchmark population).  Each stream owns one half of the state slots (eval_push assigns them),
    // so a slot is only ever reused on the stream that used it before and needs no cross-stream ordering.  Only when
    // the expectation is fused into the last pass (no scratch shared between pushes).
    // Measured alternative: pass 0 of every push on one stream and the later passes on the other (compute-bound
    // beside memory-bound by construction) is 10 % slower: two resident kernels mostly take workgroup slots from
    // each other.
    b.ways = 1;
    if (h->diagonal && n_evals >= 2)
        b.ways = std::max(1, std::min({h->n_streams, h->n_lane_streams + 1, h->group}));  // (lanes: not the auxiliary stream)
    // A batch that mixes split evaluations (three short launches, no state) with ordinary ones (passes over resident
    // states): the ordinary ones go to the auxiliary stream, every push's, so that the two kinds run side by side instead
    // of one after the other (n = 14, 64 four-layer circuits of which a third has no split form: 114 -> see DESIGN.md).
    // Only where the split ones leave no partial sums for the push's common reduction (quadratic operator).
    b.aux_plain = false;
    if (h->diagonal && b.split_any && factor_path(h) && h->geo.blocks_per_state > 1)
        for (size_t i = 0; i < n_evals && !b.aux_plain; ++i) b.aux_plain = b.split[i] == 0;
    if (b.aux_plain && h->aux_stream < 0) {
        hipStream_t st = nullptr;
        QSV_HIP(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        h->side_streams.push_back(st);
        h->aux_stream = int(h->side_streams.size()) - 1;
        // (the plans of this batch were uploaded on the handle's stream before this stream existed)
        QSV_HIP(h, hipEventRecord(h->ev_join, h->stream));
        QSV_HIP(h, hipStreamWaitEvent(st, h->ev_join, 0));
    }
    b.aux_count = 0;
    b.used_mask = 0;
    b.n_pushes = 0;
    b.chain_crossed = false;
    // Every evaluation's result is ONE 8-byte store to the pinned result buffer by the kernel that finishes it (diagonal
    // operators): marked with a value no kernel writes, the buffer itself says when the batch is done (eval_end).
    b.sentinels = h->poll_results && h->diagonal && !h->profiling && n_evals > 0;
    if (b.sentinels) {
        uint64_t* v = reinterpret_cast<uint64_t*>(h->h_out);
        for (size_t i = 0; i < n_evals; ++i) v[i] = kResultSentinel;
    }
    return QSV_OK;
}

// Descriptors [first, first + count) of the current batch: split evaluations first, the others behind them (results,
// partial sums and slots go by the descriptor's fields, not by its position), so that the side circuits, the ordinary
// passes and the contraction are each launched over the evaluations they concern -- a workgroup that only finds out that
// it has nothing to do still costs its dispatch, and a mixed launch was mostly such workgroups.  Among the split ones
// those with more keys first: their workgroups of the contraction take longest and should not be the tail of the
// launch.  Returns the number of split evaluations; batch.eval_at maps positions back to evaluations.
// Evaluations that continue a kept state (Circuit::prefix_id) come LAST, behind the ordinary ones: their first pass runs the
// later-pass instantiation of the kernel, so they are launch groups of their own; *n_cont receives their number.
size_t order_split_first(qsv_t* h, size_t first, size_t count, size_t* n_cont = nullptr) {
    qsv_handle::Batch& b = h->batch;
    if (n_cont) *n_cont = 0;
    if (!b.split_any && !b.cont_any) return 0;
    EvalDesc* hd = static_cast<EvalDesc*>(h->h_batch);
    const size_t P = b.circs.size();
    std::vector<EvalDesc> tmp(hd + first, hd + first + count), tmp2;
    if (b.split_any) tmp2.assign(hd + P + first, hd + P + first + count);
    size_t at = first, n_split = 0;
    if (!b.split_any) {
        for (int cont = 0; cont <= 1; ++cont)
            for (size_t j = 0; j < count; ++j) {
                if (int(b.circs[first + j]->prefix_id >= 0) != cont) continue;
                hd[at] = tmp[j];
                b.eval_at[at] = uint32_t(first + j);
                ++at;
                if (cont && n_cont) *n_cont += 1;
            }
        return 0;
    }
    // (among the split ones first those that need launches of their own after the virtual circuits -- kEvalFused ones are
    // finished by the launch that runs theirs --, so that each kind is one contiguous range of every launch group)
    // (... and among the kEvalFused ones first those whose sides have eight amplitudes per thread: a launch is ONE instantiation of
    // the kernel)
    // (... and of those first the ones with half sides: a group too large for one launch with the sides' states in LDS gives them
    // a launch of their own, run_group)
    for (int fused = 0; fused <= 1; ++fused)
      for (int r = 3; r <= 4; ++r)
       for (int halved = 1; halved >= 0; --halved)
        for (int cls = kMaxSplitKeys; cls >= 0; --cls)
            for (size_t j = 0; j < count; ++j) {
                if (!b.split[first + j]) continue;
                const SplitInfo& sp = b.circs[first + j]->split;
                if (sp.n_keys != cls || int(sp.fused) != fused || ((sp.side_r == 3) != (r == 3)) || int(sp.halves) != halved) continue;
                hd[at] = tmp[j];
                hd[P + at] = tmp2[j];
                b.eval_at[at] = uint32_t(first + j);
                ++at;
                ++n_split;
            }
    for (int cont = 0; cont <= 1; ++cont)
        for (size_t j = 0; j < count; ++j) {
            if (b.split[first + j] || int(b.circs[first + j]->prefix_id >= 0) != cont) continue;
            hd[at] = tmp[j];
            hd[P + at] = tmp2[j];
            b.eval_at[at] = uint32_t(first + j);
            ++at;
            if (cont && n_cont) *n_cont += 1;
        }
    return n_split;
}

int eval_push(qsv_t* h, size_t first, size_t count, const double* values, const double* device_values = nullptr) {
    qsv_handle::Batch& b = h->batch;
    if (first != b.pushed) return fail(h, QSV_E_STATE, "evaluations must be pushed in order");
    if (first + count > b.circs.size()) return fail(h, QSV_E_ARG, "push exceeds the batch");
    // Launch groups: G evaluations whose states are resident together.  On one stream the slot of an evaluation is its
    // index mod G (batch_layout) and the stream orders every reuse.  With two streams each push takes its stream's
    // half of the slots and is cut into launch groups of G / 2.
    size_t G = size_t(h->group);
    struct WorkGuard {
        qsv_t* h;
        ~WorkGuard() {
            h->work = nullptr;
            h->stamping = false;
            h->batch.dev_params = nullptr;  // (a property of the push: other paths lay batches out and ship them too)
        }
    } guard{h};
    h->stamping = h->profiling;
    EvalDesc* hd = static_cast<EvalDesc*>(h->h_batch);  // pinned; prepare_kernel reads it after this point
    if (b.repeat && !(first == 0 && count == b.circs.size())) {
        // (pushed differently than the batch whose layout this one kept: lay it out afresh)
        const std::vector<Circuit*> circs = b.circs;
        const std::vector<int64_t> np(b.cur_counts.begin(), b.cur_counts.end());
        const int rc0 = eval_begin(h, circs, np);
        if (rc0) return rc0;
        if (h->out_target) b.ways = 1;  // (qsv_eval_set_output's choice, made before this push)
    }
    if (!device_values && h->async_pending) {  // (the staging buffer is about to be written: see qsv_eval_begin)
        QSV_HIP(h, sync_streams(h));
        h->async_pending = false;
    }
    b.whole_push = first == 0 && count == b.circs.size();
    // (values in device memory: the descriptors' offsets are those of the staging buffer, so the base is where evaluation 0's
    // values would be)
    b.dev_params = device_values && count > 0 ? device_values - b.param_base[first] : nullptr;
    const size_t ways = size_t(std::max(1, b.ways)), lane = ways > 1 ? size_t(b.n_pushes) % ways : 0;
    const size_t P = b.circs.size();
    size_t n_cont = b.snap_n_cont;
    const size_t n_split = b.repeat ? b.snap_n_split : order_split_first(h, first, count, &n_cont);
    if (b.whole_push) {
        b.snap_n_split = n_split;
        b.snap_n_cont = n_cont;
    }
    // Slots.  Ordinary evaluations: G states are resident together; on one stream the slot of an evaluation is its
    // position mod G and the stream orders every reuse; with several streams each push takes its stream's share of the
    // slots.  (The expectation kernels of the general-operator path index states by position in the launch group.)
    // Split evaluations have no state, only two small tables: their own, much larger set of slots.
    if (ways > 1) {
        G = G / ways;
        if (lane) {
            h->work = h->side_streams[lane - 1];
            b.used_mask |= 1u << (lane - 1);
        }
    }
    const size_t SG = h->side_slots > 0 ? std::max<size_t>(1, size_t(h->side_slots) / ways) : 1;
    for (size_t j = 0; j < n_split; ++j) {
        const uint32_t slot = uint32_t(lane * SG + j % SG);
        hd[first + j].state_slot = slot;
        hd[P + first + j].state_slot = slot;
    }
    hipStream_t const lane_stream = h->work;  // (null: the handle's own)
    // Split evaluations of both kinds in this push -- finished by the launch that runs their virtual circuits (kEvalFused) /
    // with launches of their own: the second kind's chain (virtual circuits, Gram matrices, combination) goes to the chain
    // stream and runs beside the first kind's one launch.  Every split evaluation of the push has its own slot then (n_split
    // <= SG), so the two streams share nothing.
    // The chain stream is the OTHER lane's: idle when the push is the whole batch, and in a batch of two pushes (one per lane)
    // each stream then carries one push's single launch and the other push's chain -- 128 five-layer circuits: 230 -> see
    // DESIGN.md 4.2.  (A stream of its own would be the handle's fourth, and a process has four hardware queues by default: it
    // landed on the auxiliary stream's queue and the two chains ran one after the other; L = 6 with five keys: 397 -> 449 us
    // per step.)  A lane's slots are reused by the lane's later pushes in stream order -- which a chain on the other lane's
    // stream is outside of: a push that comes to a lane again after such a chain first joins the two lanes' streams.
    b.chain_now = false;
    h->chain_stream = -1;
    const size_t push_index = b.n_pushes;
    if (b.chain_crossed && push_index >= ways && h->n_lane_streams >= 1) {
        QSV_HIP(h, hipEventRecord(h->ev_join, h->stream));
        QSV_HIP(h, hipStreamWaitEvent(h->side_streams[0], h->ev_join, 0));
        QSV_HIP(h, hipEventRecord(h->ev_join, h->side_streams[0]));
        QSV_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
        b.chain_crossed = false;
    }
    if (h->chain_enabled && ways <= 2 && lane <= 1 && h->n_lane_streams >= 1 && n_split > 0 && n_split <= SG && factor_path(h) &&
        fused_route(h)) {
        size_t n_fused = 0;
        for (size_t j = 0; j < n_split; ++j) n_fused += b.circs[b.eval_at[first + j]]->split.fused ? 1 : 0;
        b.chain_now = n_fused > 0 && n_fused < n_split;
        // (... or every split evaluation is of the first kind, too many for one launch with the sides' states in LDS, and some have
        // half sides, which need them there: those few get a launch of their own beside the others', run_group)
        size_t n_halved = 0;
        for (size_t j = 0; j < n_split; ++j) n_halved += b.circs[b.eval_at[first + j]]->split.halves ? 1 : 0;
        size_t working = 0;
        for (size_t j = 0; j < n_split; ++j) working += working_workgroups(b.circs[b.eval_at[first + j]]->split);
        if (n_fused == n_split && n_halved > 0 && working > size_t(h->n_cus)) b.chain_now = true;
    }
    if (b.chain_now) {
        h->chain_stream = lane == 0 ? 0 : -2;  // (-2: the handle's own stream, lane 0's)
        b.chain_crossed = true;
        if (lane == 0) {
            b.used_mask |= 1u;
            if (h->out_target) {  // (the caller may have work queued on the handle's stream that the results must come after)
                QSV_HIP(h, hipEventRecord(h->ev_join, h->stream));
                QSV_HIP(h, hipStreamWaitEvent(h->side_streams[0], h->ev_join, 0));
            }
        }
    }
    hipStream_t const plain_stream = b.aux_plain ? h->side_streams[size_t(h->aux_stream)] : lane_stream;
    if (b.aux_plain && n_split < count) b.used_mask |= 1u << h->aux_stream;

    for (size_t j = n_split; j < count; ++j) {
        // (on the auxiliary stream every ordinary evaluation of the batch: one stream orders every reuse of a slot)
        const uint32_t slot = b.aux_plain ? uint32_t((b.aux_count + (j - n_split)) % size_t(h->group))
                                          : uint32_t(lane * G + (j - n_split) % G);
        hd[first + j].state_slot = slot;
        if (b.split_any) hd[P + first + j].state_slot = slot;
    }
    if (b.aux_plain) {
        b.aux_count += count - n_split;
        G = size_t(h->group);
    }
    b.n_pushes += 1;
    const uint32_t mode = kModeSynthFirst | (h->diagonal ? kModeFinalDiag : kModeFinalStore) |
                          (h->has_diag_part ? kModeFinalDiag : 0u);
    bool direct = single_workgroup_path(h, mode) && n_cont == 0;  // (an evaluation on a kept state does not synthesise its input)
    for (size_t j = 0; direct && j < count; ++j) direct = b.circs[first + j]->plan.stats.n_passes == 1;
    h->work = plain_stream;  // (the preparation launch concerns the ordinary evaluations only)
    int rc = batch_ship(h, first, count, values, direct ? count : n_split);
    h->work = lane_stream;
    if (rc) return rc;
    const uint32_t group_mode = mode | (direct ? uint32_t(kModeDirectResult) : 0u);
    // launch groups: the split evaluations of the push in groups of SG, then the ordinary ones in groups of G, then -- in
    // groups of their own -- those that continue a kept state
    const size_t cont0 = first + count - n_cont;
    for (size_t g0 = first; g0 < first + count;) {
        const bool in_split = g0 < first + n_split;
        const size_t gc = in_split ? std::min(SG, first + n_split - g0) : std::min(G, (g0 < cont0 ? cont0 : first + count) - g0);
        struct Advance {
            size_t& g0;
            size_t gc;
            ~Advance() { g0 += gc; }
        } advance{g0, gc};
        h->work = in_split ? lane_stream : plain_stream;
        QSV_HIP(h, stamp(h, b.pass_events, true));
        rc = run_group(h, b.circs, g0, gc, group_mode);
        if (!rc) QSV_HIP(h, stamp(h, b.pass_events, false));
        if (rc) return rc;
        if (!h->diagonal && !in_split) {
            QSV_HIP(h, stamp(h, b.exp_events, true));
            QSV_HIP(h, launch_pauli_groups(h->dtype, h->d_states.ptr, uint64_t(1) << h->n, h->n, int(gc), h->n_groups,
                                           static_cast<const PauliGroup*>(h->d_groups.ptr),
                                           static_cast<const uint64_t*>(h->d_z.ptr),
                                           static_cast<const double*>(h->d_cre.ptr),
                                           static_cast<const uint32_t*>(h->d_term_odd.ptr), h->pauli_nb,
                                           static_cast<double*>(h->d_term_partials.ptr), h->stream));
            QSV_HIP(h, launch_pauli_combine(static_cast<const double*>(h->d_term_partials.ptr),
                                            uint32_t(h->n_groups) * uint32_t(h->pauli_nb),
                                            h->has_diag_part ? static_cast<const double*>(h->d_partials.ptr) : nullptr,
                                            partials_per_state(h), int(gc), batch_evals(h) + g0,
                                            static_cast<double*>(h->d_out.ptr), h->stream));
            QSV_HIP(h, stamp(h, b.exp_events, false));
        }
    }
    if (h->diagonal && !direct) {
        // This push's evaluations are reduced on the push's own stream, straight into the pinned result buffer: no
        // cross-stream join in front of one final reduction (the join alone cost 15-30 us at the end of every call).
        QSV_HIP(h, stamp(h, b.exp_events, true));
        h->work = n_split > 0 && factor_path(h) ? plain_stream : lane_stream;
        if (n_split > 0 && factor_path(h))  // (the split evaluations' results are already there: the others, by descriptor)
            QSV_HIP(h, launch_reduce_partials(static_cast<const double*>(h->d_partials.ptr), partials_per_state(h),
                                              int(count - n_split), h->out_target ? h->out_target : h->h_out, ws(h),
                                              batch_evals(h) + first + n_split));
        else
            QSV_HIP(h, launch_reduce_partials(static_cast<const double*>(h->d_partials.ptr) + first * size_t(partials_per_state(h)),
                                              partials_per_state(h), int(count), (h->out_target ? h->out_target : h->h_out) + first,
                                              ws(h)));
        QSV_HIP(h, stamp(h, b.exp_events, false));
    }
    b.pushed = first + count;
    return QSV_OK;
}

int eval_end(qsv_t* h, double* out) {
    qsv_handle::Batch& b = h->batch;
    const size_t n_evals = b.circs.size();
    if (b.pushed != n_evals) return fail(h, QSV_E_STATE, "not every evaluation of the batch was pushed");
    if (n_evals == 0) return QSV_OK;
    const unsigned used_mask = b.used_mask;
    if (h->profiling)  // (only so that ev1 below marks the end of EVERY stream's work)
        for (size_t i = 0; i < h->side_streams.size(); ++i)
            if (used_mask >> i & 1u) {
                QSV_HIP(h, hipEventRecord(h->ev_join, h->side_streams[i]));
                QSV_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
            }
    b.used_mask = 0;
    if (h->profiling) QSV_HIP(h, hipEventRecord(b.ev1, h->stream));
    // a batch that went through in one push leaves its layout for the next call to find (qsv_eval_begin)
    if (b.have_ids && b.whole_push && h->repeat_enabled && !h->profiling) {
        if (!b.repeat) {
            b.snap_ids = b.cur_ids;
            b.snap_counts = b.cur_counts;
        }
        b.snap_epoch = h->epoch;
    } else {
        b.snap_epoch = 0;
    }
    if (h->out_target && !out && !h->profiling) {
        // Results go to the caller's device buffer: nothing to wait for here.  Whatever the caller enqueues on the
        // handle's stream next (a collective over the results) runs after every push, also those of the other streams.
        for (size_t i = 0; i < h->side_streams.size(); ++i)
            if (used_mask >> i & 1u) {
                QSV_HIP(h, hipEventRecord(h->ev_join, h->side_streams[i]));
                QSV_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
            }
        if (!h->diagonal)
            // (hipMemcpyDefault: the caller's buffer may be host memory the device can address -- a node's shared fitness table)
            QSV_HIP(h, hipMemcpyAsync(h->out_target, h->d_out.ptr, n_evals * sizeof(double), hipMemcpyDefault, h->stream));
        h->async_pending = true;
        return QSV_OK;
    }
    if (!h->diagonal)
        QSV_HIP(h, hipMemcpyAsync(h->out_target ? h->out_target : h->h_out, h->d_out.ptr, n_evals * sizeof(double),
                                  h->out_target ? hipMemcpyDefault : hipMemcpyDeviceToHost, h->stream));
    // The results themselves say when they are there: every evaluation's is one store to the pinned buffer, visible to the host
    // about 5 us before the stream's completion signal is (scripts/ubench/flag_vs_sync.hip).  Watched for at most kPollMicros
    // -- a step of a shallow population; longer batches wait on the streams as before.  Results that have all arrived ARE the
    // end of the batch's work (every kernel that reads a staging buffer comes before the one that writes its evaluation's
    // result): nothing is left for a later call to wait for, and the runtime retires its own bookkeeping with the next
    // launches (synchronising at the start of the next call instead was measured: 77 -> 84 us per step, worse than not polling).
    bool arrived = false;
    if (b.sentinels && !h->out_target && out) {
        const volatile uint64_t* v = reinterpret_cast<const volatile uint64_t*>(h->h_out);
        const auto t0 = std::chrono::steady_clock::now();
        size_t done = 0;
        for (uint32_t spin = 1;; ++spin) {
            while (done < n_evals && v[done] != kResultSentinel) ++done;
            if (done == n_evals) {
                arrived = true;
                break;
            }
            if ((spin & 31u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kPollMicros)) break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!arrived) {
        // (polling hipStreamQuery instead was measured: no faster, and it slowed concurrent callers down threefold)
        QSV_HIP(h, hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < h->side_streams.size(); ++i)
            if (used_mask >> i & 1u) QSV_HIP(h, hipStreamSynchronize(h->side_streams[i]));
    }
    if (h->out_target) {  // (a waiting end of a batch with a device output: the caller also gets a host copy)
        if (out) QSV_HIP(h, hipMemcpy(out, h->out_target, n_evals * sizeof(double), hipMemcpyDefault));
    } else {
        std::memcpy(out, h->h_out, n_evals * sizeof(double));
    }

    if (h->profiling) {
        float ms = 0.f;
        QSV_HIP(h, hipEventElapsedTime(&ms, b.ev0, b.ev1));
        h->prof.total_ms = ms;
        for (auto& p : b.pass_events) {
            QSV_HIP(h, hipEventElapsedTime(&ms, p.first, p.second));
            h->prof.pass_ms += ms;
        }
        for (int kind = 0; kind < 3; ++kind)
            for (auto& p : b.launch_events[kind]) {
                QSV_HIP(h, hipEventElapsedTime(&ms, p.first, p.second));
                h->prof.kernel_ms[kind] += ms;
            }
        // wall-clock window of the gate passes: with two streams the per-push intervals above overlap
        for (auto& p : b.pass_events) {
            QSV_HIP(h, hipEventElapsedTime(&ms, b.pass_events.front().first, p.second));
            h->prof.pass_window_ms = std::max(h->prof.pass_window_ms, double(ms));
        }
        for (auto& p : b.exp_events) {
            QSV_HIP(h, hipEventElapsedTime(&ms, p.first, p.second));
            h->prof.expect_ms += ms;
        }
    }
    return QSV_OK;
}

// Releases whatever a batch holds (events) and marks it closed; safe to call on error paths.
void eval_close(qsv_t* h) {
    qsv_handle::Batch& b = h->batch;
    for (auto* list : {&b.pass_events, &b.exp_events, &b.launch_events[0], &b.launch_events[1], &b.launch_events[2]}) {
        for (auto& p : *list) {
            if (p.first) (void)hipEventDestroy(p.first);
            if (p.second) (void)hipEventDestroy(p.second);
        }
        list->clear();
    }
    if (b.ev0) (void)hipEventDestroy(b.ev0);
    if (b.ev1) (void)hipEventDestroy(b.ev1);
    b.ev0 = b.ev1 = nullptr;
    if (b.snap_epoch != h->epoch) b.circs.clear();  // (kept with a layout the next batch may reuse)
    b.have_ids = false;
    b.repeat = false;
    b.open = false;
    h->out_target = nullptr;
    b.ways = 1;
    b.used_mask = 0;
    h->work = nullptr;
}

// One-shot evaluation: begin + one push + end.
int eval_all(qsv_t* h, const std::vector<Circuit*>& circs, const int64_t* param_offsets, const double* params,
             double* out) {
    if (circs.empty()) return QSV_OK;
    std::vector<int64_t> np(circs.size());
    for (size_t i = 0; i < circs.size(); ++i) np[i] = param_offsets[i + 1] - param_offsets[i];
    int rc = eval_begin(h, circs, np);
    // the caller's vectors may have gaps between them: pack what each evaluation declared
    std::vector<double> packed;
    if (!rc) {
        size_t total = 0;
        for (int64_t v : np) total += size_t(v);
        packed.resize(total + 1);
        size_t cur = 0;
        for (size_t i = 0; i < circs.size(); ++i) {
            if (np[i]) std::memcpy(packed.data() + cur, params + param_offsets[i], size_t(np[i]) * sizeof(double));
            cur += size_t(np[i]);
        }
        rc = eval_push(h, 0, circs.size(), packed.data());
    }
    if (!rc) rc = eval_end(h, out);
    eval_close(h);
    return rc;
}

// Prepare the final state of one circuit in slot 0 (used by statevector / probabilities / sample).
int run_single_to_state(qsv_t* h, int circuit_id, const double* params, int n_params) {
    auto it = h->circuits.find(circuit_id);
    if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id");
    std::vector<Circuit*> cc{&it->second};
    h->prof = qsv_profile{};
    int rc = batch_layout(h, cc, std::vector<int64_t>{int64_t(n_params)});
    if (!rc) rc = batch_ship(h, 0, 1, params);
    if (!rc) rc = ensure(h, h->d_partials, size_t(partials_per_state(h)) * sizeof(double));
    if (!rc) rc = run_group(h, cc, 0, 1, kModeSynthFirst | kModeFinalStore);
    h->batch.circs.clear();
    return rc;
}


}  // namespace

extern "C" {

const char* qsv_version(void) { return "libqsv 0.1.0 (gfx950)"; }

// Not part of include/qsv.h: read-out of the per-phase cycle stamps of a diagnostic (-DQSV_STAMPS) build.
int qsv_debug_stamps(unsigned long long* out, int reset) {
    return qsv::read_stamps(out, reset) == hipSuccess ? QSV_OK : QSV_E_UNSUPPORTED;
}

// Likewise not part of include/qsv.h: the phase timeline of a -DQSV_TIMELINE build (scripts/timeline.py).
int qsv_debug_timeline(unsigned long long* out, size_t max_words, unsigned int* n_records, int reset) {
    return qsv::read_timeline(out, max_words, n_records, reset) == hipSuccess ? QSV_OK : QSV_E_UNSUPPORTED;
}

int qsv_create(int n_qubits, int dtype, int device, const qsv_plan_config* cfg, qsv_t** out) {
    if (!out) return fail(nullptr, QSV_E_ARG, "out is null");
    *out = nullptr;
    if (dtype != QSV_F64 && dtype != QSV_F32) return fail(nullptr, QSV_E_ARG, "dtype must be QSV_F64 or QSV_F32");
    PlanConfig pc;
    Geometry geo;
    try {
        pc = resolve_config(cfg, dtype, n_qubits);
        geo = make_geometry(n_qubits, pc);
    } catch (const std::exception& e) {
        return fail(nullptr, QSV_E_ARG, e.what());
    }
    if (geo.threads_launch > 512) return fail(nullptr, QSV_E_ARG, "tile_bits - reg_bits must be <= 9");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(nullptr, QSV_E_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0 || device >= n_dev) return fail(nullptr, QSV_E_ARG, "device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, QSV_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));

    qsv_t* h = new qsv_t();
    h->n = n_qubits;
    h->dtype = dtype;
    h->device = device;
    h->cfg = pc;
    h->geo = geo;
    h->amp_bytes = size_t(pc.amp_bytes);
    const size_t state_bytes = (size_t(1) << n_qubits) * h->amp_bytes;
    int group = cfg && cfg->group > 0 ? cfg->group : 0;
    if (const char* env = getenv("QSV_GROUP")) group = atoi(env);
    if (group <= 0) {
        // Evaluations that run side by side in one launch.  Measured on MI355X (scripts/sweep.sh): the passes are
        // bound by fp64 issue and LDS traffic rather than by HBM, so keeping a group inside the 256 MiB Infinity
        // Cache buys nothing, while larger groups amortise launch ramp-up and tail: 2 GiB of resident states, at
        // most 256 evaluations (n = 16: 256, n = 20: 128, n = 24: 8, n >= 27: 1).
        const size_t budget = size_t(2) << 30;
        group = int(std::max<size_t>(1, std::min<size_t>(256, budget / state_bytes)));
    }
    h->group = group;
    // a workgroup sweeps two consecutive tiles once a launch has plenty of workgroups anyway (>= 4096 tiles)
    {
        const uint64_t tiles_per_launch = uint64_t(geo.blocks_per_state) * uint64_t(group);
        h->tiles_per_block = (tiles_per_launch >= 4096 && geo.blocks_per_state >= 2) ? 2 : 1;
    }
    h->tiles_per_block_later = h->tiles_per_block;
    if (const char* env = getenv("QSV_TILES_PER_BLOCK")) h->tiles_per_block = h->tiles_per_block_later = std::max(1, atoi(env));
    if (const char* env = getenv("QSV_TILES_PER_BLOCK_LATER")) h->tiles_per_block_later = std::max(1, atoi(env));
    auto bail = [&](hipError_t err, const char* what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(err);
        qsv_destroy(h);
        return fail(nullptr, QSV_E_DEVICE, msg);
    };
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    h->own_stream = true;
    if (const char* env = getenv("QSV_SPLIT")) h->split_enabled = atoi(env) != 0;
    if (const char* env = getenv("QSV_SPLIT_SAMPLE")) h->split_sampling = atoi(env) != 0;
    if (const char* env = getenv("QSV_FACTOR")) h->factor_enabled = atoi(env) != 0;
    if (getenv("QSV_NO_FUSED_FACTOR")) h->fused_factor = false;
    if (const char* env = getenv("QSV_CHAIN_STREAM")) h->chain_enabled = atoi(env) != 0;
    if (const char* env = getenv("QSV_POLL")) h->poll_results = atoi(env) != 0;
    if (const char* env = getenv("QSV_REPEAT")) h->repeat_enabled = atoi(env) != 0;
    if (const char* env = getenv("QSV_REPEAT_DESCS")) h->repeat_device_descs = atoi(env) != 0;
    if (const char* env = getenv("QSV_FUSED_LDS")) h->fused_lds_table = atoi(env) != 0;
    if (const char* env = getenv("QSV_SIDE_DIAG")) h->side_diag = atoi(env) != 0;
    if (const char* env = getenv("QSV_SIDES_R3")) h->sides_r3 = atoi(env) != 0;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cus = cus;
    }
    {
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) large_bar = 0;
        // (measured, n = 20, 64 evaluations: the launch is 1.4 us shorter with the descriptors local -- 54.4 -> 53.0 us --, the
        // step is not, 73.8 -> 73.6 us: off unless asked for)
        h->bar_ship = false;
        if (const char* env = getenv("QSV_BAR")) h->bar_ship = large_bar != 0 && atoi(env) != 0;
    }
    if (const char* env = getenv("QSV_SPLIT_MAX_KEYS")) h->split_max_keys = std::max(0, std::min(kMaxSplitKeys, atoi(env)));
    h->stream_mode = state_bytes > (size_t(256) << 20) ? uint32_t(kModeStreaming) : 0u;
    if (const char* env = getenv("QSV_STREAMING")) h->stream_mode = atoi(env) ? uint32_t(kModeStreaming) : 0u;
    if (const char* env = getenv("QSV_STREAMS")) h->n_streams = std::max(1, std::min(4, atoi(env)));
    for (int i = 1; i < std::max(2, h->n_streams); ++i) {  // (the lanes' streams; the auxiliary one on first need)
        hipStream_t st = nullptr;
        if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
        h->side_streams.push_back(st);
    }
    h->n_lane_streams = int(h->side_streams.size());
    if ((e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(&h->d_states.ptr, state_bytes * size_t(group))) != hipSuccess) return bail(e, "hipMalloc(states)");
    h->d_states.bytes = state_bytes * size_t(group);
    // compact tables: at most 2^kMaxCompactBits tiles per slot, never more than a state
    h->wtab_stride = std::min<uint64_t>(uint64_t(1) << n_qubits, uint64_t(1) << (geo.k + int(kMaxCompactBits)));
    if ((e = hipMalloc(&h->d_wtab.ptr, size_t(h->wtab_stride) * h->amp_bytes * size_t(group))) != hipSuccess)
        return bail(e, "hipMalloc(compact tables)");
    h->d_wtab.bytes = size_t(h->wtab_stride) * h->amp_bytes * size_t(group);
    if (h->split_enabled && n_qubits > geo.k && n_qubits <= 28) {
        // side tables of split evaluations: two virtual circuits of at most tile + kSideExtraBits qubits per slot
        h->side_stride = uint64_t(2) << (geo.k + kSideExtraBits);
        // (a lane's split evaluations run in launch groups of its share of the slots: with 128 slots config 3's step --
        // 256 evaluations at 24 qubits, two lanes -- was four groups of 64, two chains of launches per lane; with 256 it is
        // one per lane: 0.276 -> 0.211 ms.  4 MiB per slot at 13-qubit tiles.)
        h->side_slots = 256;
        if (const char* env = getenv("QSV_SIDE_SLOTS")) h->side_slots = std::max(8, std::min(1024, atoi(env)));
        if ((e = hipMalloc(&h->d_side.ptr, size_t(h->side_stride) * h->amp_bytes * size_t(h->side_slots))) != hipSuccess)
            return bail(e, "hipMalloc(side tables)");
        h->d_side.bytes = size_t(h->side_stride) * h->amp_bytes * size_t(h->side_slots);
        if (h->factor_enabled) {
            const size_t bytes = factor_slot_doubles() * sizeof(double) * size_t(h->side_slots);
            if ((e = hipMalloc(&h->d_factor.ptr, bytes)) != hipSuccess) return bail(e, "hipMalloc(partial Gram matrices)");
            h->d_factor.bytes = bytes;
            const size_t cbytes = sizeof(uint32_t) * size_t(kFactorCountersPerSlot) * size_t(h->side_slots);
            if ((e = hipMalloc(&h->d_factor_count.ptr, cbytes)) != hipSuccess) return bail(e, "hipMalloc(hand-off counters)");
            h->d_factor_count.bytes = cbytes;
            if ((e = hipMemset(h->d_factor_count.ptr, 0, cbytes)) != hipSuccess) return bail(e, "hipMemset(hand-off counters)");
        }
    }
    {
        // the most dynamic LDS a pass launch of this handle may ask for: the exchange plane of the largest tile (a side of a
        // split circuit may have a tile one qubit larger than the handle's), the fused preparation, the fused factor tail
        const int side_tile = std::max(geo.k, std::min(geo.r + 9, int(kMaxTileBits)));
        const size_t plane = (size_t(1) << side_tile) * h->amp_bytes / (pc.xmode == 2 ? 2 : 1);
        const size_t most = std::max({geo.lds_bytes, plane, kFusedPrepareLdsBytes, kFusedFactorLdsBytes, kFusedLdsTableEnd});
        if ((e = configure_pass_kernels(dtype, geo.r, pc.xmode, most)) != hipSuccess) return bail(e, "hipFuncSetAttribute");
        // (sides of split circuits may run with eight amplitudes per thread whatever the handle's own geometry: build_circuit)
        if (geo.r != 3 && (e = configure_pass_kernels(dtype, 3, pc.xmode, most)) != hipSuccess) return bail(e, "hipFuncSetAttribute");
    }
    *out = h;
    return QSV_OK;
}

void qsv_destroy(qsv_t* h) {
    if (!h) return;
    if (getenv("QSV_COALESCE_DEBUG") && h->cq_batches)
        fprintf(stderr, "qsv_eval_coalesced: %llu batches, last expectation %zu\n", (unsigned long long)h->cq_batches, h->cq_expected);
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipStream_t st : h->side_streams) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (DeviceBuffer* b : {&h->d_z, &h->d_cre, &h->d_diag, &h->d_order, &h->d_sorted, &h->d_term_partials, &h->d_groups, &h->d_term_odd, &h->d_arena,
                            &h->d_states, &h->d_wtab, &h->d_side, &h->d_factor, &h->d_factor_count, &h->d_factor_big, &h->d_factor_big_count, &h->d_quad, &h->d_fterms, &h->d_fpart, &h->d_batch, &h->d_mats, &h->d_partials, &h->d_out, &h->d_scratch, &h->d_prefix, &h->d_sdiag})
        if (b->ptr) (void)hipFree(b->ptr);
    if (h->h_batch) (void)hipHostFree(h->h_batch);
    if (h->d_ship) (void)hipFree(h->d_ship);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_out) (void)hipHostFree(h->h_out);
    if (h->h_samples) (void)hipHostFree(h->h_samples);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* qsv_last_error(const qsv_t* h) {
    if (!h) return g_create_error.c_str();
    return g_handle_error_owner == h ? g_handle_error.c_str() : "";
}

int qsv_set_stream(qsv_t* h, void* hip_stream) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    QSV_HIP(h, hipSetDevice(h->device));
    QSV_HIP(h, hipStreamSynchronize(h->stream));
    if (h->own_stream) {
        (void)hipStreamDestroy(h->stream);
        h->own_stream = false;
    }
    if (hip_stream) {
        h->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        QSV_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return QSV_OK;
}

int qsv_n_qubits(const qsv_t* h) { return h ? h->n : QSV_E_ARG; }
int qsv_group_size(const qsv_t* h) { return h ? h->group : QSV_E_ARG; }

int qsv_set_operator(qsv_t* h, int n_terms, const uint64_t* x_mask, const uint64_t* z_mask, const double* coeff_re,
                     const double* coeff_im) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    if (n_terms < 1 || !x_mask || !z_mask || !coeff_re) return fail(h, QSV_E_ARG, "operator needs at least one term");
    (void)coeff_im;  // <P_k> is real for every Pauli string, so real(<H>) only needs the real parts
    const uint64_t limit = (uint64_t(1) << h->n) - 1;
    for (int k = 0; k < n_terms; ++k)
        if ((x_mask[k] | z_mask[k]) & ~limit) return fail(h, QSV_E_ARG, "Pauli term acts on a qubit >= n_qubits");
    QSV_HIP(h, hipSetDevice(h->device));
    // ---- split into the diagonal part (x = 0) and x-mask groups of the rest -------------------------------
    std::vector<uint64_t> diag_z, off_z;
    std::vector<double> diag_c, off_c;
    std::vector<uint32_t> off_odd;
    std::vector<PauliGroup> groups;
    std::vector<int> order;
    for (int k = 0; k < n_terms; ++k) {
        if (x_mask[k] == 0) {
            diag_z.push_back(z_mask[k]);
            diag_c.push_back(coeff_re[k]);
        } else {
            order.push_back(k);
        }
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return x_mask[a] < x_mask[b]; });
    for (int k : order) {
        const uint64_t x = x_mask[k];
        if (groups.empty() || groups.back().x != x) {
            PauliGroup g{};
            g.x = x;
            g.first = uint32_t(off_z.size());
            g.count = 0;
            g.pivot = uint32_t(63 - __builtin_clzll(x));
            groups.push_back(g);
        }
        groups.back().count += 1;
        const int ny = __builtin_popcountll(x & z_mask[k]);
        off_z.push_back(z_mask[k]);
        off_odd.push_back(uint32_t(ny & 1));
        off_c.push_back(((ny >> 1) & 1) ? -coeff_re[k] : coeff_re[k]);  // (-1)^{floor(ny/2)}
    }
    int rc;
    const bool all_diag = groups.empty();
    if (!diag_z.empty()) {
        DeviceBuffer tmp_z, tmp_c;
        const size_t mb = diag_z.size() * 8;
        if ((rc = ensure(h, tmp_z, mb)) || (rc = ensure(h, tmp_c, mb))) return rc;
        QSV_HIP(h, hipMemcpyAsync(tmp_z.ptr, diag_z.data(), mb, hipMemcpyHostToDevice, h->stream));
        QSV_HIP(h, hipMemcpyAsync(tmp_c.ptr, diag_c.data(), mb, hipMemcpyHostToDevice, h->stream));
        if ((rc = ensure(h, h->d_diag, (size_t(1) << h->n) * sizeof(double)))) return rc;
        QSV_HIP(h, launch_diag_table(h->n, int(diag_z.size()), static_cast<const uint64_t*>(tmp_z.ptr),
                                     static_cast<const double*>(tmp_c.ptr), static_cast<double*>(h->d_diag.ptr),
                                     h->stream));
        QSV_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(tmp_z.ptr);
        (void)hipFree(tmp_c.ptr);
    }
    if (!all_diag) {
        const size_t nt = off_z.size();
        if ((rc = ensure(h, h->d_z, nt * 8)) || (rc = ensure(h, h->d_cre, nt * 8)) ||
            (rc = ensure(h, h->d_term_odd, nt * 4)) || (rc = ensure(h, h->d_groups, groups.size() * sizeof(PauliGroup))))
            return rc;
        QSV_HIP(h, hipMemcpyAsync(h->d_z.ptr, off_z.data(), nt * 8, hipMemcpyHostToDevice, h->stream));
        QSV_HIP(h, hipMemcpyAsync(h->d_cre.ptr, off_c.data(), nt * 8, hipMemcpyHostToDevice, h->stream));
        QSV_HIP(h, hipMemcpyAsync(h->d_term_odd.ptr, off_odd.data(), nt * 4, hipMemcpyHostToDevice, h->stream));
        QSV_HIP(h, hipMemcpyAsync(h->d_groups.ptr, groups.data(), groups.size() * sizeof(PauliGroup),
                                  hipMemcpyHostToDevice, h->stream));
        const uint64_t n_pairs = uint64_t(1) << (h->n - 1);
        // enough workgroups per group to fill the chip once the group count is accounted for
        const uint64_t want = std::max<uint64_t>(1, 2048 / std::max<size_t>(1, groups.size()));
        h->pauli_nb = int(std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(want, 1024), n_pairs / 256 + 1)));
        if ((rc = ensure(h, h->d_term_partials, size_t(h->group) * groups.size() * size_t(h->pauli_nb) * 8))) return rc;
        QSV_HIP(h, hipStreamSynchronize(h->stream));
    }
    // quadratic diagonal operators: the couplings as a dense matrix (split evaluations then factorise, launch_factor)
    h->quadratic = false;
    if (all_diag) {
        bool quadratic = true;
        std::vector<double> quad(size_t(h->n) * size_t(h->n), 0.0);
        for (size_t k = 0; k < diag_z.size() && quadratic; ++k) {
            const int weight = __builtin_popcountll(diag_z[k]);
            if (weight > 2) quadratic = false;
            if (weight == 2) {
                const int a = __builtin_ctzll(diag_z[k]), b = 63 - __builtin_clzll(diag_z[k]);
                quad[size_t(a) * size_t(h->n) + size_t(b)] += diag_c[k];
                quad[size_t(b) * size_t(h->n) + size_t(a)] += diag_c[k];
            }
        }
        if (quadratic) {
            if ((rc = ensure(h, h->d_quad, quad.size() * sizeof(double)))) return rc;
            QSV_HIP(h, hipMemcpyAsync(h->d_quad.ptr, quad.data(), quad.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            QSV_HIP(h, hipStreamSynchronize(h->stream));
            h->quadratic = true;
        }
    }
    // general operators: the terms as a plain list (split evaluations then need no state either, launch_factor_terms)
    h->n_fterms = 0;
    if (!all_diag && h->n <= 32) {
        std::vector<FactorTerm> list(size_t(n_terms), FactorTerm{0, 0, 0.0});
        for (int k = 0; k < n_terms; ++k) list[size_t(k)] = FactorTerm{uint32_t(x_mask[k]), uint32_t(z_mask[k]), coeff_re[k]};
        if ((rc = ensure(h, h->d_fterms, list.size() * sizeof(FactorTerm)))) return rc;
        QSV_HIP(h, hipMemcpyAsync(h->d_fterms.ptr, list.data(), list.size() * sizeof(FactorTerm), hipMemcpyHostToDevice, h->stream));
        QSV_HIP(h, hipStreamSynchronize(h->stream));
        h->n_fterms = uint32_t(n_terms);
    }
    // (plans in the arena name tables of the OLD operator's D: everything is uploaded again when it is next used)
    QSV_HIP(h, sync_streams(h));
    for (auto& kv : h->circuits) kv.second.uploaded = false;
    h->sdiag_used = 0;
    h->n_terms = n_terms;
    h->order_valid = false;
    h->diagonal = all_diag;
    h->has_diag_part = !diag_z.empty();
    h->n_groups = int(groups.size());
    return QSV_OK;
}

int qsv_circuit_create(qsv_t* h, int n_ops, const qsv_op* ops, int n_params, int* out_circuit_id) {
    if (!h) return QSV_E_ARG;
    if (!out_circuit_id || n_params < 0) return fail(h, QSV_E_ARG, "bad arguments");
    // the scheduler runs outside the handle lock: callers on several threads register structures side by side
    Circuit c;
    std::string err;
    int rc = build_circuit(h, n_ops, ops, n_params, true, &c, &err);
    if (rc) return fail(h, rc, err);
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    *out_circuit_id = insert_circuit(h, std::move(c));
    return QSV_OK;
}

int qsv_circuits_create(qsv_t* h, int n_circuits, const int64_t* op_offsets, const qsv_op* ops, const int* n_params,
                        int* out_circuit_ids) {
    if (!h) return QSV_E_ARG;
    if (n_circuits < 0 || (n_circuits > 0 && (!op_offsets || !n_params || !out_circuit_ids)))
        return fail(h, QSV_E_ARG, "bad arguments");
    for (int i = 0; i < n_circuits; ++i)
        if (op_offsets[i + 1] < op_offsets[i] || n_params[i] < 0) return fail(h, QSV_E_ARG, "bad offsets or parameter counts");
    std::vector<BuiltCircuit> built;
    build_many(h, size_t(n_circuits), [&](size_t i, BuiltCircuit& b) {
        b.rc = build_circuit(h, int(op_offsets[i + 1] - op_offsets[i]), ops + op_offsets[i], n_params[i], true, &b.circuit, &b.err);
    }, built);
    for (int i = 0; i < n_circuits; ++i)
        if (built[size_t(i)].rc) return fail(h, built[size_t(i)].rc, built[size_t(i)].err + " (circuit " + std::to_string(i) + ")");
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    for (int i = 0; i < n_circuits; ++i) out_circuit_ids[i] = insert_circuit(h, std::move(built[size_t(i)].circuit));
    return QSV_OK;
}

int qsv_circuit_destroy(qsv_t* h, int circuit_id) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    auto it = h->circuits.find(circuit_id);
    if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id");
    if (it->second.prefix_id >= 0) prefix_unref(h, it->second.prefix_id);
    h->circuits.erase(it);
    // the arena space is reclaimed when the arena is next rebuilt
    return QSV_OK;
}

// ---- kept states ---------------------------------------------------------------------------------------------------------
// (reference: mutation.py:57-59 -- optimize_layer_of_individual evaluates get_partially_parameterized_quantum_circuit({layer_id})
// over and over: everything in front of that layer is the same state in every evaluation of the search)

int qsv_prefix_create(qsv_t* h, int n_states, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                      int* out_prefix_ids) {
    if (!h) return QSV_E_ARG;
    if (h->batch_owner.load() == std::this_thread::get_id())
        return fail(h, QSV_E_STATE, "a batch is open on this handle (qsv_prefix_create goes between batches)");
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_states < 0 || (n_states > 0 && (!circuit_ids || !param_offsets || !out_prefix_ids))) return fail(h, QSV_E_ARG, "bad arguments");
    if (n_states == 0) return QSV_OK;
    QSV_HIP(h, hipSetDevice(h->device));
    const size_t n = size_t(n_states);
    std::vector<Circuit*> circs(n, nullptr);
    std::vector<int64_t> np(n);
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        auto it = h->circuits.find(circuit_ids[i]);
        if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id " + std::to_string(circuit_ids[i]));
        if (it->second.prefix_id >= 0) return fail(h, QSV_E_UNSUPPORTED, "a kept state of a circuit that itself continues a kept state");
        circs[i] = &it->second;
        np[i] = param_offsets[i + 1] - param_offsets[i];
        if (np[i] < 0) return fail(h, QSV_E_ARG, "param_offsets must be non-decreasing");
        total += size_t(np[i]);
    }
    std::vector<double> packed(total + 1, 0.0);
    for (size_t i = 0, cur = 0; i < n; cur += size_t(np[i]), ++i)
        if (np[i]) std::memcpy(packed.data() + cur, params + param_offsets[i], size_t(np[i]) * sizeof(double));
    // room for n more states
    const size_t state_bytes = (size_t(1) << h->n) * h->amp_bytes;
    const size_t fresh_needed = n > h->prefix_free.size() ? n - h->prefix_free.size() : 0;
    if (h->prefix_used + fresh_needed > h->prefix_slots) {
        const size_t want = h->prefix_used + fresh_needed;
        size_t cap = std::max(want, h->prefix_slots * 2);
        void* fresh = nullptr;
        hipError_t e = hipMalloc(&fresh, cap * state_bytes);
        if (e != hipSuccess && cap > want) {
            (void)hipGetLastError();
            cap = want;
            e = hipMalloc(&fresh, cap * state_bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, QSV_E_DEVICE, std::string("hipMalloc(kept states): ") + hipGetErrorString(e));
        }
        if (h->d_prefix.ptr) {
            hipError_t e2 = sync_streams(h);  // (nothing may still read the old buffer)
            if (e2 == hipSuccess && h->prefix_used)
                e2 = hipMemcpy(fresh, h->d_prefix.ptr, h->prefix_used * state_bytes, hipMemcpyDeviceToDevice);
            if (e2 != hipSuccess) {
                (void)hipFree(fresh);
                return fail(h, QSV_E_DEVICE, std::string("moving the kept states: ") + hipGetErrorString(e2));
            }
            (void)hipFree(h->d_prefix.ptr);
        }
        h->d_prefix.ptr = fresh;
        h->d_prefix.bytes = cap * state_bytes;
        h->prefix_slots = cap;
    }
    // the circuits run their ordinary plans (a state is wanted, not an expectation value), a launch group at a time, on the
    // handle's stream; each final state is copied from its slot of the group to its kept slot
    h->prof = qsv_profile{};
    int rc = batch_layout(h, circs, np);
    if (rc) return rc;
    const size_t G = size_t(h->group);
    {
        EvalDesc* hd = static_cast<EvalDesc*>(h->h_batch);
        for (size_t j = 0; j < n; ++j) hd[j].state_slot = uint32_t(j % G);
    }
    if (!rc) rc = ensure(h, h->d_partials, std::max<size_t>(1, n) * partials_per_state(h) * sizeof(double));
    if (!rc) rc = batch_ship(h, 0, n, packed.data());
    std::vector<uint32_t> slots(n);
    for (size_t i = 0; i < n; ++i) {
        if (!h->prefix_free.empty()) {
            slots[i] = h->prefix_free.back();
            h->prefix_free.pop_back();
        } else {
            slots[i] = uint32_t(h->prefix_used++);
        }
    }
    auto give_back = [&]() {
        for (uint32_t sl : slots) h->prefix_free.push_back(sl);
        h->batch.circs.clear();
        (void)sync_streams(h);
    };
    for (size_t g0 = 0; !rc && g0 < n; g0 += G) {
        const size_t gc = std::min(G, n - g0);
        rc = run_group(h, circs, g0, gc, kModeSynthFirst | kModeFinalStore);
        for (size_t i = g0; !rc && i < g0 + gc; ++i) {
            hipError_t e = hipMemcpyAsync(static_cast<char*>(h->d_prefix.ptr) + size_t(slots[i]) * state_bytes,
                                          static_cast<const char*>(h->d_states.ptr) + (i % G) * state_bytes, state_bytes,
                                          hipMemcpyDeviceToDevice, h->stream);
            if (e != hipSuccess) rc = fail(h, QSV_E_DEVICE, std::string("hipMemcpyAsync(kept state): ") + hipGetErrorString(e));
        }
    }
    if (rc) {
        give_back();
        return rc;
    }
    h->batch.circs.clear();
    // (the staging buffers this batch's preparation read are free again, and whatever stream continues a kept state finds it)
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        give_back();
        return fail(h, QSV_E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    for (size_t i = 0; i < n; ++i) {
        const int id = h->next_prefix_id++;
        h->prefixes.emplace(id, PrefixState{slots[i], 0, false});
        out_prefix_ids[i] = id;
    }
    return QSV_OK;
}

int qsv_prefix_destroy(qsv_t* h, int n_states, const int* prefix_ids) {
    if (!h) return QSV_E_ARG;
    if (n_states < 0 || (n_states > 0 && !prefix_ids)) return fail(h, QSV_E_ARG, "bad arguments");
    // (finalizers of host objects call this: it must not wait for a handle its own thread holds between begin and end)
    if (h->batch_owner.load() == std::this_thread::get_id()) return fail(h, QSV_E_STATE, "a batch is open on this handle");
    std::lock_guard<std::mutex> lock(h->mu);
    int rc = QSV_OK;
    for (int i = 0; i < n_states; ++i) {
        auto it = h->prefixes.find(prefix_ids[i]);
        if (it == h->prefixes.end() || it->second.released) {
            rc = fail(h, QSV_E_ARG, "unknown kept state " + std::to_string(prefix_ids[i]));
            continue;
        }
        it->second.released = true;
        if (it->second.refs == 0) {
            h->prefix_free.push_back(it->second.slot);
            h->prefixes.erase(it);
        }
    }
    return rc;
}

int qsv_prefix_count(const qsv_t* h) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    return int(h->prefixes.size());
}

int qsv_circuits_create_on_prefixes(qsv_t* h, int n_circuits, const int64_t* op_offsets, const qsv_op* ops, const int* n_params,
                                    const int* prefix_ids, int* out_circuit_ids) {
    if (!h) return QSV_E_ARG;
    if (n_circuits < 0 || (n_circuits > 0 && (!op_offsets || !n_params || !prefix_ids || !out_circuit_ids)))
        return fail(h, QSV_E_ARG, "bad arguments");
    for (int i = 0; i < n_circuits; ++i)
        if (op_offsets[i + 1] < op_offsets[i] || n_params[i] < 0) return fail(h, QSV_E_ARG, "bad offsets or parameter counts");
    // unfolded plans (the input is an arbitrary state, not a product state), never split
    std::vector<BuiltCircuit> built;
    build_many(h, size_t(n_circuits), [&](size_t i, BuiltCircuit& b) {
        b.rc = build_circuit(h, int(op_offsets[i + 1] - op_offsets[i]), ops + op_offsets[i], n_params[i], false, &b.circuit, &b.err);
    }, built);
    for (int i = 0; i < n_circuits; ++i)
        if (built[size_t(i)].rc) return fail(h, built[size_t(i)].rc, built[size_t(i)].err + " (circuit " + std::to_string(i) + ")");
    std::lock_guard<std::mutex> lock(h->mu);
    for (int i = 0; i < n_circuits; ++i) {
        auto it = h->prefixes.find(prefix_ids[i]);
        if (it == h->prefixes.end() || it->second.released) return fail(h, QSV_E_ARG, "unknown kept state " + std::to_string(prefix_ids[i]));
    }
    h->epoch += 1;
    for (int i = 0; i < n_circuits; ++i) {
        built[size_t(i)].circuit.prefix_id = prefix_ids[i];
        h->prefixes.find(prefix_ids[i])->second.refs += 1;
        out_circuit_ids[i] = insert_circuit(h, std::move(built[size_t(i)].circuit));
    }
    return QSV_OK;
}

int qsv_circuit_create_on_prefix(qsv_t* h, int prefix_id, int n_ops, const qsv_op* ops, int n_params, int* out_circuit_id) {
    const int64_t offsets[2] = {0, n_ops};
    return qsv_circuits_create_on_prefixes(h, 1, offsets, ops, &n_params, &prefix_id, out_circuit_id);
}

int qsv_circuit_cost(qsv_t* h, int circuit_id, qsv_circuit_cost_t* out) {
    if (!h) return QSV_E_ARG;
    if (!out) return fail(h, QSV_E_ARG, "out is null");
    std::lock_guard<std::mutex> lock(h->mu);
    auto it = h->circuits.find(circuit_id);
    if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id");
    Circuit& c = it->second;
    *out = qsv_circuit_cost_t{};
    // which way an expectation value of this circuit goes under the operator set now (eval_begin's rules)
    const bool allow_split = h->n_terms == 0 || h->diagonal || factor_terms_path(h);
    const int max_keys = (h->n_terms == 0 || factor_path(h)) ? kMaxSplitKeys : 3;
    const double scale = std::ldexp(1.0, h->n - 20) * (h->dtype == QSV_F64 ? 1.0 : 0.5);
    if (c.split.ok && allow_split && c.split.n_keys <= max_keys) {
        const SplitInfo& sp = c.split;
        const bool one_launch = sp.fused && h->fused_factor && (h->n_terms == 0 || factor_path(h));
        out->route = one_launch ? QSV_ROUTE_SPLIT_ONE_LAUNCH : QSV_ROUTE_SPLIT;
        out->n_keys = sp.n_keys;
        out->n_passes = std::max(sp.stats[0].n_passes, sp.stats[1].n_passes);
        // measured at 20 qubits (profiles/r03_split_by_depth.txt, r03_chain_stream.txt): one launch of 64 evaluations 49 us; own
        // launches 1.2 us per evaluation up to three keys, 16 / 32 product terms 5 / 9 us; sides larger than 2^10 cost in proportion
        const double sides = 0.5 * (std::ldexp(1.0, sp.n_virtual[0] - sp.n_keys - 10) + std::ldexp(1.0, sp.n_virtual[1] - sp.n_keys - 10));
        const double per = one_launch ? 0.8 : (sp.n_keys <= 3 ? 1.2 : (sp.n_keys == 4 ? 5.0 : 9.0));
        out->microseconds = per * std::max(1.0, sides) * double(out->n_passes);
        return QSV_OK;
    }
    int rc = ensure_plan(h, c);
    if (rc) return rc;
    out->n_passes = c.plan.stats.n_passes;
    out->on_kept_state = c.prefix_id >= 0 ? 1 : 0;
    if (h->geo.blocks_per_state == 1) {
        out->route = QSV_ROUTE_ONE_TILE;
        out->microseconds = 1.0;
        return QSV_OK;
    }
    out->route = QSV_ROUTE_PASSES;
    // a pass over a 2^20 state inside a full launch, measured (profiles/r04_prefix_reuse.txt): a deep individual's whole circuit
    // 17.1 us per evaluation at 3.13 passes = 5.5 us per pass (the synthesising, compact first pass included); on a kept state
    // 10.5 - 11 us at 1.89 passes, 18.4 at 2.8: 5.8 - 6.6 us per pass -- few gates, but every pass reads and writes the whole state
    // (HBM bound: 56 MiB per two-pass evaluation at 5.3 TB/s); n = 24: 16 times both
    // ... refitted at the end of round 4 (scripts/prefix_cache_experiment.py after the scheduler's tile search): a pass that
    // moves the state costs 4.2 us at 20 qubits, a scheduled gate entry 0.09 us on top, a synthesising first pass 1 us and no
    // state traffic: whole eight-layer circuit 2.98 passes, 78 entries -> 15 us (14.6 measured); seven layers 13.4 (13.5); a last
    // layer on a kept state 1.89 passes, 15 entries -> 9.3 (9.6); the upper four of eight layers 2.02 passes, 70 entries -> 14.8
    // (14.8: as dear as the whole circuit, which is why a search in the middle keeps it)
    {
        const bool synth = c.prefix_id < 0;
        const double moving = double(out->n_passes) - (synth ? 1.0 : 0.0);
        out->microseconds = (4.2 * std::max(0.0, moving) + 0.09 * double(c.plan.stats.n_real_gates) + (synth ? 1.0 : 0.0)) * scale;
    }
    return QSV_OK;
}

int qsv_eval_circuits(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                      double* out) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_evals < 0 || (n_evals > 0 && (!circuit_ids || !param_offsets || !out)))
        return fail(h, QSV_E_ARG, "bad arguments");
    QSV_HIP(h, hipSetDevice(h->device));
    std::vector<Circuit*> circs(size_t(n_evals), nullptr);
    for (int i = 0; i < n_evals; ++i) {
        auto it = h->circuits.find(circuit_ids[i]);
        if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id " + std::to_string(circuit_ids[i]));
        circs[size_t(i)] = &it->second;
        if (param_offsets[i + 1] < param_offsets[i]) return fail(h, QSV_E_ARG, "param_offsets must be non-decreasing");
    }
    static const double dummy = 0.0;
    return eval_all(h, circs, param_offsets, params ? params : &dummy, out);
}

// Collect the requests of concurrent callers for a moment, evaluate them as ONE batch, hand every caller its value.
// (caller holds h->cq_mu through `lock`; returns with it held)
static void coalesce_lead(qsv_t* h, std::unique_lock<std::mutex>& lock, double window_us) {
    using clock = std::chrono::steady_clock;
    const double window = window_us > 0 ? window_us : 300.0;
    const double quiet = getenv("QSV_COALESCE_QUIET_US") ? atof(getenv("QSV_COALESCE_QUIET_US")) : 60.0;
    // A batch is complete when as many callers as last time have arrived, or nobody new came for `quiet` us, or the
    // window is over.  The collector spins (no timed sleep is shorter than the kernel's timer slack of ~50 us).
    lock.unlock();
    const auto t0 = clock::now();
    auto last_arrival = t0;
    size_t seen = h->cq_count.load();
    for (;;) {
        const auto now = clock::now();
        const size_t count = h->cq_count.load();
        if (count != seen) {
            seen = count;
            last_arrival = now;
        }
        const double since_start = std::chrono::duration<double, std::micro>(now - t0).count();
        const double since_arrival = std::chrono::duration<double, std::micro>(now - last_arrival).count();
        if ((h->cq_expected > 1 && count >= h->cq_expected) || since_arrival >= quiet || since_start >= window) break;
        __builtin_ia32_pause();
    }
    lock.lock();
    std::vector<qsv_handle::CoalesceRequest*> batch;
    batch.swap(h->cq);
    h->cq_count.store(0);
    lock.unlock();
    {
        std::lock_guard<std::mutex> hl(h->mu);
        std::vector<Circuit*> circs;
        std::vector<int64_t> offsets{0};
        std::vector<double> params;
        std::vector<qsv_handle::CoalesceRequest*> valid;
        for (auto* r : batch) {
            auto it = h->circuits.find(r->circuit_id);
            if (it == h->circuits.end() || r->n_params < it->second.n_params) {
                r->rc = QSV_E_ARG;
                r->err = it == h->circuits.end() ? "unknown circuit id " + std::to_string(r->circuit_id)
                                                 : "circuit needs " + std::to_string(it->second.n_params) + " parameter values";
                continue;
            }
            circs.push_back(&it->second);
            params.insert(params.end(), r->params, r->params + r->n_params);
            offsets.push_back(int64_t(params.size()));
            valid.push_back(r);
        }
        if (!valid.empty()) {
            std::vector<double> values(valid.size(), 0.0);
            params.push_back(0.0);
            int rc = hipSetDevice(h->device) == hipSuccess ? QSV_OK : QSV_E_DEVICE;
            if (!rc) rc = eval_all(h, circs, offsets.data(), params.data(), values.data());
            const std::string err = rc ? g_handle_error : std::string();
            for (size_t i = 0; i < valid.size(); ++i) {
                valid[i]->rc = rc;
                valid[i]->err = err;
                valid[i]->value = values[i];
            }
        }
    }
    lock.lock();
    // how many callers to expect next time: a high-water mark that decays slowly (a batch cut short by a pause in the
    // arrivals must not teach the next collector to stop early as well)
    h->cq_expected = std::max(batch.size(), h->cq_expected * 7 / 8);
    h->cq_batches += 1;
    for (auto* r : batch) r->done = true;
    h->cq_collecting = false;
    h->cq_cv.notify_all();
}

int qsv_eval_coalesced(qsv_t* h, int circuit_id, const double* params, int n_params, double window_us, double* out) {
    if (!h) return QSV_E_ARG;
    if (!out || n_params < 0 || (n_params > 0 && !params)) return fail(h, QSV_E_ARG, "bad arguments");
    qsv_handle::CoalesceRequest req;
    req.circuit_id = circuit_id;
    req.params = params;
    req.n_params = n_params;
    std::unique_lock<std::mutex> lock(h->cq_mu);
    h->cq.push_back(&req);
    h->cq_count.fetch_add(1);
    while (!req.done) {
        if (!h->cq_collecting) {
            h->cq_collecting = true;  // nobody is collecting: this caller does, for everyone queued with it
            coalesce_lead(h, lock, window_us);
        } else {
            h->cq_cv.wait(lock);
        }
    }
    lock.unlock();
    if (req.rc) return fail(h, req.rc, req.err);
    *out = req.value;
    return QSV_OK;
}

int qsv_eval_begin(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_counts) {
    if (!h) return QSV_E_ARG;
    // a second begin from the thread that already holds the handle (between begin and end) would wait for itself
    if (h->batch_owner.load() == std::this_thread::get_id())
        return fail(h, QSV_E_STATE, "a batch is already open on this handle");
    std::unique_lock<std::mutex> lock(h->mu);
    if (h->batch.open) return fail(h, QSV_E_STATE, "a batch is already open on this handle");
    if (n_evals < 0 || (n_evals > 0 && (!circuit_ids || !param_counts))) return fail(h, QSV_E_ARG, "bad arguments");
    QSV_HIP(h, hipSetDevice(h->device));
    {
        // The previous batch again (same circuits, same counts, nothing changed in between -- an optimiser's next iteration):
        // its layout is still in the staging buffer, descriptors ordered and slotted as its one push left them.
        qsv_handle::Batch& b = h->batch;
        b.cur_ids.assign(circuit_ids, circuit_ids + n_evals);
        b.cur_counts.assign(param_counts, param_counts + n_evals);
        b.have_ids = true;
        if (h->repeat_enabled && n_evals > 0 && b.snap_epoch == h->epoch && !h->profiling &&
            b.snap_ids == b.cur_ids && b.snap_counts == b.cur_counts && b.circs.size() == size_t(n_evals)) {
            // (A batch that ended without waiting may still be running.  Its kernels read the staging buffers: whoever WRITES
            // them waits for it first -- qsv_eval_push with host values, qsv_eval_staging --, and a batch whose values come from
            // device memory, qsv_eval_push_device, writes nothing: such batches follow each other on the stream without the
            // host ever waiting, which is what an optimiser that lives on the device needs.)
            h->prof = qsv_profile{};
            h->prof.n_evals = uint64_t(n_evals);
            b.repeat = true;
            b.whole_push = false;
            b.pushed = 0;
            b.n_pushes = 0;
            b.aux_count = 0;
            b.used_mask = 0;
            b.chain_crossed = false;
            b.ways = 1;
            if (h->diagonal && n_evals >= 2) b.ways = std::max(1, std::min({h->n_streams, h->n_lane_streams + 1, h->group}));
            b.sentinels = h->poll_results && h->diagonal;
            if (b.sentinels) {
                uint64_t* v = reinterpret_cast<uint64_t*>(h->h_out);
                for (int i = 0; i < n_evals; ++i) v[i] = kResultSentinel;
            }
            b.open = true;
            h->batch_owner.store(std::this_thread::get_id());
            h->batch_lock = std::move(lock);
            return QSV_OK;
        }
    }
    std::vector<Circuit*> circs(size_t(n_evals), nullptr);
    std::vector<int64_t> np(size_t(n_evals), 0);
    for (int i = 0; i < n_evals; ++i) {
        auto it = h->circuits.find(circuit_ids[i]);
        if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id " + std::to_string(circuit_ids[i]));
        circs[size_t(i)] = &it->second;
        if (param_counts[i] < 0) return fail(h, QSV_E_ARG, "negative parameter count");
        np[size_t(i)] = param_counts[i];
    }
    int rc = eval_begin(h, circs, np);
    if (rc) {
        eval_close(h);
        return rc;
    }
    h->batch.open = true;
    h->batch_owner.store(std::this_thread::get_id());
    h->batch_lock = std::move(lock);  // other threads wait until qsv_eval_end
    return QSV_OK;
}

int qsv_eval_push(qsv_t* h, int first, int count, const double* values) {
    if (!h) return QSV_E_ARG;
    if (!h->batch.open) return fail(h, QSV_E_STATE, "no open batch (call qsv_eval_begin first)");
    if (first < 0 || count < 0) return fail(h, QSV_E_ARG, "bad arguments");
    static const double dummy = 0.0;
    return eval_push(h, size_t(first), size_t(count), values ? values : &dummy);
}

int qsv_eval_push_device(qsv_t* h, int first, int count, const double* device_values, void* ready_event) {
    if (!h) return QSV_E_ARG;
    if (!h->batch.open) return fail(h, QSV_E_STATE, "no open batch (call qsv_eval_begin first)");
    if (first < 0 || count < 0) return fail(h, QSV_E_ARG, "bad arguments");
    if (!device_values) return qsv_eval_push(h, first, count, nullptr);  // (only legal for evaluations without parameters)
    if (device_values != h->dev_params_checked) {  // (an optimiser hands over the same buffer call after call: asked once)
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, device_values) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != h->device) {
            (void)hipGetLastError();
            return fail(h, QSV_E_ARG, "device_values is not memory of this handle's device");
        }
        h->dev_params_checked = device_values;
    }
    if (ready_event) {
        // whichever of the handle's streams runs a part of this push reads the values: all of them wait
        QSV_HIP(h, hipStreamWaitEvent(h->stream, static_cast<hipEvent_t>(ready_event), 0));
        for (hipStream_t st : h->side_streams) QSV_HIP(h, hipStreamWaitEvent(st, static_cast<hipEvent_t>(ready_event), 0));
    }
    static const double dummy = 0.0;
    return eval_push(h, size_t(first), size_t(count), &dummy, device_values);
}

int qsv_eval_staging(qsv_t* h, int first, int count, double** values) {
    if (!h || !values) return QSV_E_ARG;
    if (!h->batch.open) return fail(h, QSV_E_STATE, "no open batch (call qsv_eval_begin first)");
    const qsv_handle::Batch& b = h->batch;
    if (first < 0 || count < 0 || size_t(first) + size_t(count) > b.circs.size()) return fail(h, QSV_E_ARG, "range exceeds the batch");
    if (h->async_pending) {  // (the caller is about to write where an unfinished batch's kernels may still read)
        QSV_HIP(h, sync_streams(h));
        h->async_pending = false;
    }
    double* hp = reinterpret_cast<double*>(static_cast<char*>(h->h_batch) + b.desc_bytes);
    *values = hp + (count > 0 ? size_t(b.param_base[size_t(first)]) : 0);
    return QSV_OK;
}

int qsv_spsa_step(qsv_t* h, const qsv_spsa_step_args* in) {
    if (!h || !in) return QSV_E_ARG;
    // (between qsv_eval_begin and qsv_eval_end the calling thread holds the handle: it would wait for itself)
    if (h->batch_owner.load() == std::this_thread::get_id())
        return fail(h, QSV_E_STATE, "a batch is open on this handle (qsv_spsa_step goes between batches)");
    std::lock_guard<std::mutex> lock(h->mu);
    if (in->n_runs < 0 || in->width < 0 || !in->x || !in->active || !in->iterations) return fail(h, QSV_E_ARG, "bad arguments");
    if (in->values && !in->delta_accept) return fail(h, QSV_E_ARG, "values without the signs they were measured with");
    if (in->delta_propose && !in->points) return fail(h, QSV_E_ARG, "a proposal needs somewhere to go");
    if (in->window < 0 || (in->window > 0 && (!in->previous || !in->n_values || !in->changes)))
        return fail(h, QSV_E_ARG, "the termination rule needs its state");
    if (!(in->eps > 0)) return fail(h, QSV_E_ARG, "eps must be positive");
    QSV_HIP(h, hipSetDevice(h->device));
    SpsaStepArgs a{};
    a.n_runs = in->n_runs;
    a.width = in->width;
    a.x = in->x;
    a.active = in->active;
    a.iterations = reinterpret_cast<long long*>(in->iterations);
    a.delta_accept = in->delta_accept;
    a.values = in->values;
    a.delta_propose = in->delta_propose;
    a.points = in->points;
    a.eps = in->eps;
    a.lr = in->lr;
    a.trust_region = in->trust_region;
    a.maxiter = in->maxiter;
    a.window = in->window;
    a.min_rel = in->min_rel;
    a.maxfev = in->maxfev;
    a.previous = in->previous;
    a.n_values = reinterpret_cast<long long*>(in->n_values);
    a.changes = in->changes;
    QSV_HIP(h, launch_spsa_step(a, h->stream));
    return QSV_OK;
}

int qsv_eval_suggested_pushes(const qsv_t* h) {
    if (!h || !h->batch.open) return QSV_E_ARG;
    // Two pushes overlap the packing of the second half with the GPU work on the first -- worth it when there is GPU
    // work to speak of.  A batch of split evaluations under a quadratic operator is a chain of three short launches:
    // one push (measured at 20 qubits, 64 evaluations: 87.6 us per call against 90.5).
    const qsv_handle::Batch& b = h->batch;
    bool all_split = b.split_any;
    for (size_t i = 0; all_split && i < b.split.size(); ++i) all_split = b.split[i] != 0;
    // (... up to about a launch group of the side circuits: 256 evaluations at 24 qubits take 0.29 ms in two pushes,
    // 0.42 ms in one)
    // (a mixed batch whose ordinary evaluations run beside the split ones on the auxiliary stream likewise)
    // (128 since a push's split evaluations of both kinds run side by side, eval_push: 128 five-layer circuits at 20 qubits
    // 678 k evals/s in two pushes, 827 k in one; six layers 268 k / 288 k; four layers 1.51 M / 1.48 M; 256: two pushes)
    return ((all_split && factor_path(h)) || b.aux_plain) && b.split.size() <= 128 ? 1 : 2;
}

int qsv_eval_set_output(qsv_t* h, double* device_out) {
    if (!h) return QSV_E_ARG;
    if (!h->batch.open) return fail(h, QSV_E_STATE, "no open batch (call qsv_eval_begin first)");
    if (h->batch.pushed != 0) return fail(h, QSV_E_STATE, "the output must be set before the first push");
    h->out_target = device_out;
    // every push on the handle's own stream: the caller continues on that stream (joining the other streams into it
    // cost 15 us per batch of 64 evaluations at 20 qubits; callers should push such a batch in one go)
    if (device_out) {
        h->batch.ways = 1;
        // The caller may already have queued writes to the output buffer on the handle's stream (a fill of the unused
        // tail of an uneven shard).  The batch's ordinary evaluations can run on the auxiliary stream (mixed batches,
        // eval_begin): whatever stream of ours writes results must come after that work.
        if (h->batch.aux_plain && h->aux_stream >= 0) {
            QSV_HIP(h, hipEventRecord(h->ev_join, h->stream));
            QSV_HIP(h, hipStreamWaitEvent(h->side_streams[size_t(h->aux_stream)], h->ev_join, 0));
        }
    }
    return QSV_OK;
}

int qsv_eval_results_seen(qsv_t* h) {
    if (!h) return QSV_E_ARG;
    if (h->batch_owner.load() == std::this_thread::get_id()) return fail(h, QSV_E_STATE, "a batch is open");
    std::lock_guard<std::mutex> lock(h->mu);
    if (h->batch.open) return fail(h, QSV_E_STATE, "a batch is open");
    h->async_pending = false;
    return QSV_OK;
}

int qsv_eval_end(qsv_t* h, double* out_expectations) {
    if (!h) return QSV_E_ARG;
    if (!h->batch.open) return fail(h, QSV_E_STATE, "no open batch (call qsv_eval_begin first)");
    int rc = out_expectations || h->batch.circs.empty() || h->out_target ? eval_end(h, out_expectations)
                                                                         : fail(h, QSV_E_ARG, "out is null");
    if (rc) {  // nothing of the failed batch may still be running
        (void)sync_streams(h);
    }
    eval_close(h);
    h->batch_owner.store(std::thread::id());
    std::unique_lock<std::mutex> lock = std::move(h->batch_lock);
    return rc;  // `lock` releases the handle here
}

int qsv_eval_batch(qsv_t* h, int n_evals, const int64_t* op_offsets, const qsv_op* ops, const int64_t* param_offsets,
                   const double* params, double* out) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_evals < 0 || (n_evals > 0 && (!op_offsets || !param_offsets || !out)))
        return fail(h, QSV_E_ARG, "bad arguments");
    QSV_HIP(h, hipSetDevice(h->device));
    // The cache of inline structures is bounded.  It is emptied only HERE, before any circuit of this call has been
    // looked up: evicting inside the loop below would free plans that earlier evaluations of the same call point at.
    if (h->inline_cache.size() > kInlineCacheLimit) {
        QSV_HIP(h, sync_streams(h));
        for (auto& kv : h->inline_cache) h->circuits.erase(kv.second);
        h->inline_cache.clear();
    }
    // pass 1: ids of known structures, list of the new ones (each distinct structure once)
    std::vector<int> ids(size_t(n_evals), 0);
    std::vector<std::string> keys;
    keys.resize(size_t(n_evals));
    struct Fresh {
        int first_eval;
        Circuit circuit;
        std::string err;
        int rc = QSV_OK;
    };
    std::vector<Fresh> fresh;
    std::unordered_map<std::string, int> fresh_index;  // key -> index in `fresh`
    std::vector<int> pending(size_t(n_evals), -1);
    for (int i = 0; i < n_evals; ++i) {
        const int64_t b = op_offsets[i], e = op_offsets[i + 1];
        if (e < b) return fail(h, QSV_E_ARG, "op_offsets must be non-decreasing");
        const int64_t np = param_offsets[i + 1] - param_offsets[i];
        if (np < 0) return fail(h, QSV_E_ARG, "param_offsets must be non-decreasing");
        // structure key: the ops (literal angles are baked into a registered circuit, so they are part of its
        // identity) and the length of the parameter vector
        std::string& key = keys[size_t(i)];
        key.assign(reinterpret_cast<const char*>(ops + b), size_t(e - b) * sizeof(qsv_op));
        key.append(reinterpret_cast<const char*>(&np), sizeof(np));
        auto it = h->inline_cache.find(key);
        if (it != h->inline_cache.end()) {
            ids[size_t(i)] = it->second;
            continue;
        }
        auto fi = fresh_index.find(key);
        if (fi == fresh_index.end()) {
            fi = fresh_index.emplace(key, int(fresh.size())).first;
            fresh.emplace_back();
            fresh.back().first_eval = i;
        }
        pending[size_t(i)] = fi->second;
    }
    // pass 2: schedule the new structures, on several host threads when there are many (a generation of EVQE brings
    // up to a population of new structures at once)
    if (!fresh.empty()) {
        std::vector<BuiltCircuit> built;
        build_many(h, fresh.size(), [&](size_t j, BuiltCircuit& b) {
            const int i = fresh[j].first_eval;
            const int64_t ob = op_offsets[i], oe = op_offsets[i + 1];
            b.rc = build_circuit(h, int(oe - ob), ops + ob, int(param_offsets[i + 1] - param_offsets[i]), true, &b.circuit, &b.err);
        }, built);
        for (size_t j = 0; j < fresh.size(); ++j) {
            fresh[j].rc = built[j].rc;
            fresh[j].err = std::move(built[j].err);
            fresh[j].circuit = std::move(built[j].circuit);
        }
        for (Fresh& f : fresh)
            if (f.rc) return fail(h, f.rc, f.err + " (evaluation " + std::to_string(f.first_eval) + ")");
        std::vector<int> fresh_id(fresh.size());
        for (size_t j = 0; j < fresh.size(); ++j) {
            fresh_id[j] = insert_circuit(h, std::move(fresh[j].circuit));
            h->inline_cache.emplace(keys[size_t(fresh[j].first_eval)], fresh_id[j]);
        }
        for (int i = 0; i < n_evals; ++i)
            if (pending[size_t(i)] >= 0) ids[size_t(i)] = fresh_id[size_t(pending[size_t(i)])];
    }
    // pass 3: only now, with every registration done, take pointers into the circuit table
    std::vector<Circuit*> circs(size_t(n_evals), nullptr);
    for (int i = 0; i < n_evals; ++i) circs[size_t(i)] = &h->circuits.find(ids[size_t(i)])->second;
    static const double dummy = 0.0;
    return eval_all(h, circs, param_offsets, params ? params : &dummy, out);
}

int qsv_statevector(qsv_t* h, int circuit_id, const double* params, int n_params, double* out_re_im) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!out_re_im) return fail(h, QSV_E_ARG, "out is null");
    QSV_HIP(h, hipSetDevice(h->device));
    static const double dummy = 0.0;
    int rc = run_single_to_state(h, circuit_id, params ? params : &dummy, n_params);
    if (rc) return rc;
    const uint64_t dim = uint64_t(1) << h->n;
    if (h->dtype == QSV_F64) {
        QSV_HIP(h, hipMemcpyAsync(out_re_im, h->d_states.ptr, dim * 16, hipMemcpyDeviceToHost, h->stream));
    } else {
        if ((rc = ensure(h, h->d_scratch, dim * 16))) return rc;
        QSV_HIP(h, launch_state_to_f64(h->dtype, h->d_states.ptr, dim, static_cast<double*>(h->d_scratch.ptr), h->stream));
        QSV_HIP(h, hipMemcpyAsync(out_re_im, h->d_scratch.ptr, dim * 16, hipMemcpyDeviceToHost, h->stream));
    }
    QSV_HIP(h, hipStreamSynchronize(h->stream));
    return QSV_OK;
}

int qsv_probabilities(qsv_t* h, int circuit_id, const double* params, int n_params, double* out_probs) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!out_probs) return fail(h, QSV_E_ARG, "out is null");
    QSV_HIP(h, hipSetDevice(h->device));
    static const double dummy = 0.0;
    int rc = run_single_to_state(h, circuit_id, params ? params : &dummy, n_params);
    if (rc) return rc;
    const uint64_t dim = uint64_t(1) << h->n;
    if ((rc = ensure(h, h->d_scratch, dim * 8))) return rc;
    QSV_HIP(h, launch_probabilities(h->dtype, h->d_states.ptr, dim, 1, static_cast<double*>(h->d_scratch.ptr), h->stream));
    QSV_HIP(h, hipMemcpyAsync(out_probs, h->d_scratch.ptr, dim * 8, hipMemcpyDeviceToHost, h->stream));
    QSV_HIP(h, hipStreamSynchronize(h->stream));
    return QSV_OK;
}

// Sampler branch for a whole batch: run the circuits group by group, turn each resident state into probabilities,
// draw `shots` samples per evaluation on the device and (for a diagonal operator) gather each sample's value D[state].
// out_cvar != null: the samples and their values stay on the device, only CVaR_alpha per evaluation comes back.
static int sample_batch_locked(qsv_t* h, const std::vector<Circuit*>& circs, const int64_t* param_offsets,
                               const double* params, int shots, uint64_t seed, uint64_t* out_states,
                               double* out_values, double alpha = 1.0, double* out_cvar = nullptr) {
    const size_t n_evals = circs.size();
    if (n_evals == 0 || shots == 0) return QSV_OK;
    for (const Circuit* c : circs)
        if (c->prefix_id >= 0) return fail(h, QSV_E_UNSUPPORTED, "circuits on kept states are not sampled (qsv_eval_* only)");
    if ((out_values || out_cvar) && !(h->has_diag_part && h->diagonal))
        return fail(h, QSV_E_STATE, "sample values need a diagonal operator (call qsv_set_operator with I/Z terms only)");
    std::vector<int64_t> np(n_evals);
    std::vector<double> packed;
    size_t total = 0;
    for (size_t i = 0; i < n_evals; ++i) {
        np[i] = param_offsets[i + 1] - param_offsets[i];
        if (np[i] < 0) return fail(h, QSV_E_ARG, "param_offsets must be non-decreasing");
        total += size_t(np[i]);
    }
    packed.resize(total + 1);
    for (size_t i = 0, cur = 0; i < n_evals; cur += size_t(np[i]), ++i)
        if (np[i]) std::memcpy(packed.data() + cur, params + param_offsets[i], size_t(np[i]) * sizeof(double));
    h->prof = qsv_profile{};
    // circuits that have a split form are sampled from their two side tables: no state, no 2^n probabilities
    int rc = batch_layout(h, circs, np, h->split_sampling);
    if (rc) return rc;
    const size_t n_split = order_split_first(h, 0, n_evals), n_plain = n_evals - n_split;
    const uint64_t dim = uint64_t(1) << h->n;
    const size_t G = size_t(h->group), SG = size_t(std::max(1, h->side_slots));
    {
        EvalDesc* hd = static_cast<EvalDesc*>(h->h_batch);
        for (size_t j = 0; j < n_evals; ++j) {
            const uint32_t slot = uint32_t(j < n_split ? j % SG : (j - n_split) % G);
            hd[j].state_slot = slot;
            if (h->batch.split_any) hd[n_evals + j].state_slot = slot;
        }
    }
    // device scratch: probabilities and chunk sums of a group of ordinary evaluations | tables of a group of split
    // ones | (device-side CVaR) the samples
    const size_t probs_bytes = n_plain ? G * dim * 8 : 0, sums_bytes = n_plain ? G * size_t(sample_chunk_count(dim)) * 8 : 0;
    const size_t split_off = ((probs_bytes + sums_bytes + 63) / 64) * 64;
    const size_t split_bytes = n_split ? std::min(SG, n_split) * split_sample_slot_doubles(h->geo.k + kSideExtraBits) * 8 : 0;
    const size_t out_bytes = n_evals * size_t(shots) * 8;
    const size_t dev_samples_off = ((split_off + split_bytes + 63) / 64) * 64;
    if ((rc = ensure(h, h->d_scratch, dev_samples_off + (out_cvar ? 2 * out_bytes : 0)))) return rc;
    if ((rc = ensure(h, h->d_partials, std::max<size_t>(1, n_evals) * partials_per_state(h) * sizeof(double)))) return rc;
    if (out_cvar && (rc = ensure_host_out(h, n_evals))) return rc;
    // samples (and their operator values) are written by the kernel straight into pinned host memory: no copy operations
    if (!out_cvar && h->h_samples_bytes < 2 * out_bytes) {
        if (h->h_samples) {
            QSV_HIP(h, sync_streams(h));
            QSV_HIP(h, hipHostFree(h->h_samples));
            h->h_samples = nullptr;
            h->h_samples_bytes = 0;
        }
        QSV_HIP(h, hipHostMalloc(&h->h_samples, 4 * out_bytes, hipHostMallocDefault));
        h->h_samples_bytes = 4 * out_bytes;
    }
    double* probs = static_cast<double*>(h->d_scratch.ptr);
    double* sums = probs + (n_plain ? G * dim : 0);
    double* split_scratch = reinterpret_cast<double*>(static_cast<char*>(h->d_scratch.ptr) + split_off);
    uint64_t* d_states = static_cast<uint64_t*>(h->h_samples);
    double* d_values = out_values ? reinterpret_cast<double*>(static_cast<char*>(h->h_samples) + out_bytes) : nullptr;
    if (out_cvar) {  // (device scratch behind the probabilities and chunk sums)
        d_states = reinterpret_cast<uint64_t*>(static_cast<char*>(h->d_scratch.ptr) + dev_samples_off);
        d_values = reinterpret_cast<double*>(static_cast<char*>(h->d_scratch.ptr) + dev_samples_off + out_bytes);
    }
    const double* diag = d_values ? static_cast<const double*>(h->d_diag.ptr) : nullptr;
    // one-tile registers: the pass kernel prepares its evaluation itself; n <= 28: its last pass writes the
    // probabilities, not the state
    const bool fuse = h->geo.blocks_per_state == 1;
    const bool probs_in_pass = h->n <= 28;
    rc = batch_ship(h, 0, n_evals, packed.data(), fuse ? n_evals : n_split);
    // the split evaluations (they lead the descriptors), a group of side-table slots at a time
    for (size_t g0 = 0; !rc && g0 < n_split; g0 += SG) {
        const size_t gc = std::min(SG, n_split - g0);
        if ((rc = run_group(h, circs, g0, gc, kModeSynthFirst | kModeFinalStore | kModeSidesOnly))) break;
        PassArgs a{};
        a.plan = static_cast<const uint32_t*>(h->d_arena.ptr);
        a.evals = batch_evals(h) + g0;
        a.wtab = h->d_side.ptr;
        a.wtab_stride = h->side_stride;
        QSV_HIP(h, launch_split_tables(h->dtype, h->geo.k + kSideExtraBits, unsigned(gc), split_scratch, h->stream, a));
        uint32_t table_doubles = 64;  // the largest Gram table of the group (whichever side the contraction calls Y)
        for (size_t i = 0; i < gc; ++i) {
            const SplitInfo& sp = circs[h->batch.eval_at[g0 + i]]->split;
            for (int side = 0; side < 2; ++side)
                table_doubles = std::max(table_doubles, uint32_t(1) << (2 * sp.n_keys + std::max(0, sp.n_virtual[side] - sp.n_keys - 6)));
        }
        QSV_HIP(h, launch_split_sample(h->dtype, h->geo.k + kSideExtraBits, unsigned(gc), split_scratch, shots, seed, diag, d_states, d_values,
                                       h->stream, a, table_doubles));
    }
    const uint32_t mode = kModeSynthFirst | (probs_in_pass ? kModeFinalProbs : kModeFinalStore) | (fuse ? kModeFusedPrepare : 0u);
    for (size_t g0 = n_split; !rc && g0 < n_evals; g0 += G) {
        const size_t gc = std::min(G, n_evals - g0);
        if ((rc = run_group(h, circs, g0, gc, mode))) break;
        if (!probs_in_pass) QSV_HIP(h, launch_probabilities(h->dtype, h->d_states.ptr, dim, int(gc), probs, h->stream));
        QSV_HIP(h, launch_sample(probs, dim, int(gc), sums, shots, seed, uint32_t(g0), diag, d_states, d_values, h->stream,
                                 n_split ? batch_evals(h) + g0 : nullptr));
    }
    if (!rc && out_cvar) QSV_HIP(h, launch_cvar(d_values, int(n_evals), shots, alpha, h->h_out, h->stream));
    h->batch.circs.clear();
    if (rc) return rc;
    QSV_HIP(h, hipStreamSynchronize(h->stream));
    if (out_cvar) {
        std::memcpy(out_cvar, h->h_out, n_evals * sizeof(double));
        return QSV_OK;
    }
    std::memcpy(out_states, d_states, out_bytes);
    if (out_values) std::memcpy(out_values, d_values, out_bytes);
    return QSV_OK;
}

// Exact-probability CVaR for a whole batch: the circuits group by group as in the sampler branch (split circuits as their
// two virtual circuits, the others with the probabilities written by their last gate pass), then launch_cvar_exact.
static int exact_cvar_locked(qsv_t* h, const std::vector<Circuit*>& circs, const int64_t* param_offsets, const double* params,
                             double alpha, double* out_cvar) {
    const size_t n_evals = circs.size();
    if (n_evals == 0) return QSV_OK;
    for (const Circuit* c : circs)
        if (c->prefix_id >= 0) return fail(h, QSV_E_UNSUPPORTED, "circuits on kept states are not sampled (qsv_eval_* only)");
    if (!(h->has_diag_part && h->diagonal))
        return fail(h, QSV_E_STATE, "the exact CVaR needs a diagonal operator (call qsv_set_operator with I/Z terms only)");
    if (h->n > 28) return fail(h, QSV_E_UNSUPPORTED, "the exact CVaR is available up to 28 qubits");
    const uint64_t dim = uint64_t(1) << h->n;
    int rc;
    if (!h->order_valid) {
        if ((rc = ensure(h, h->d_order, dim * sizeof(uint32_t))) || (rc = ensure(h, h->d_sorted, dim * sizeof(double)))) return rc;
        QSV_HIP(h, sort_states_by_value(static_cast<const double*>(h->d_diag.ptr), dim, static_cast<uint32_t*>(h->d_order.ptr),
                                        static_cast<double*>(h->d_sorted.ptr), h->stream));
        h->order_valid = true;
    }
    std::vector<int64_t> np(n_evals);
    std::vector<double> packed;
    size_t total = 0;
    for (size_t i = 0; i < n_evals; ++i) {
        np[i] = param_offsets[i + 1] - param_offsets[i];
        if (np[i] < 0) return fail(h, QSV_E_ARG, "param_offsets must be non-decreasing");
        total += size_t(np[i]);
    }
    packed.resize(total + 1);
    for (size_t i = 0, cur = 0; i < n_evals; cur += size_t(np[i]), ++i)
        if (np[i]) std::memcpy(packed.data() + cur, params + param_offsets[i], size_t(np[i]) * sizeof(double));
    h->prof = qsv_profile{};
    if ((rc = batch_layout(h, circs, np, h->split_sampling))) return rc;
    const size_t n_split = order_split_first(h, 0, n_evals), n_plain = n_evals - n_split;
    const size_t G = size_t(h->group), SG = size_t(std::max(1, h->side_slots));
    {
        EvalDesc* hd = static_cast<EvalDesc*>(h->h_batch);
        for (size_t j = 0; j < n_evals; ++j) {
            const uint32_t slot = uint32_t(j < n_split ? j % SG : (j - n_split) % G);
            hd[j].state_slot = slot;
            if (h->batch.split_any) hd[n_evals + j].state_slot = slot;
        }
    }
    const uint32_t n_chunks = cvar_exact_chunks(dim);
    const size_t probs_bytes = n_plain ? G * dim * 8 : 0;
    const size_t chunk_off = ((probs_bytes + 63) / 64) * 64;
    const size_t chunk_bytes = 2 * std::max(G, std::min(SG, std::max<size_t>(1, n_split))) * size_t(n_chunks) * 8;
    if ((rc = ensure(h, h->d_scratch, chunk_off + chunk_bytes))) return rc;
    if ((rc = ensure(h, h->d_partials, std::max<size_t>(1, n_evals) * partials_per_state(h) * sizeof(double)))) return rc;
    if ((rc = ensure_host_out(h, n_evals))) return rc;
    double* probs = static_cast<double*>(h->d_scratch.ptr);
    double* chunk_scratch = reinterpret_cast<double*>(static_cast<char*>(h->d_scratch.ptr) + chunk_off);
    const bool fuse = h->geo.blocks_per_state == 1;
    rc = batch_ship(h, 0, n_evals, packed.data(), fuse ? n_evals : n_split);
    PassArgs a{};
    a.plan = static_cast<const uint32_t*>(h->d_arena.ptr);
    a.wtab = h->d_side.ptr;
    a.wtab_stride = h->side_stride;
    const uint32_t* order = static_cast<const uint32_t*>(h->d_order.ptr);
    const double* sorted = static_cast<const double*>(h->d_sorted.ptr);
    for (size_t g0 = 0; !rc && g0 < n_split; g0 += SG) {
        const size_t gc = std::min(SG, n_split - g0);
        if ((rc = run_group(h, circs, g0, gc, kModeSynthFirst | kModeFinalStore | kModeSidesOnly))) break;
        a.evals = batch_evals(h) + g0;
        QSV_HIP(h, launch_cvar_exact(h->dtype, probs, dim, unsigned(gc), order, sorted, alpha, chunk_scratch, h->h_out, h->stream, a));
    }
    const uint32_t mode = kModeSynthFirst | kModeFinalProbs | (fuse ? kModeFusedPrepare : 0u);
    for (size_t g0 = n_split; !rc && g0 < n_evals; g0 += G) {
        const size_t gc = std::min(G, n_evals - g0);
        if ((rc = run_group(h, circs, g0, gc, mode))) break;
        a.evals = batch_evals(h) + g0;
        QSV_HIP(h, launch_cvar_exact(h->dtype, probs, dim, unsigned(gc), order, sorted, alpha, chunk_scratch, h->h_out, h->stream, a));
    }
    h->batch.circs.clear();
    if (rc) return rc;
    QSV_HIP(h, hipStreamSynchronize(h->stream));
    std::memcpy(out_cvar, h->h_out, n_evals * sizeof(double));
    return QSV_OK;
}

int qsv_sample_batch(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                     int shots, uint64_t seed, uint64_t* out_states, double* out_values) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_evals < 0 || shots < 0 || (n_evals > 0 && shots > 0 && (!circuit_ids || !param_offsets || !out_states)))
        return fail(h, QSV_E_ARG, "bad arguments");
    QSV_HIP(h, hipSetDevice(h->device));
    std::vector<Circuit*> circs(size_t(n_evals), nullptr);
    for (int i = 0; i < n_evals; ++i) {
        auto it = h->circuits.find(circuit_ids[i]);
        if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id " + std::to_string(circuit_ids[i]));
        circs[size_t(i)] = &it->second;
    }
    static const double dummy = 0.0;
    return sample_batch_locked(h, circs, param_offsets, params ? params : &dummy, shots, seed, out_states, out_values);
}

int qsv_sample_cvar_batch(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                          int shots, uint64_t seed, double alpha, double* out_cvar) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_evals < 0 || shots < 1 || (n_evals > 0 && (!circuit_ids || !param_offsets || !out_cvar)))
        return fail(h, QSV_E_ARG, "bad arguments");
    if (!(alpha > 0.0) || alpha > 1.0) return fail(h, QSV_E_ARG, "alpha must be in (0, 1]");
    if (shots > kCvarMaxShots) return fail(h, QSV_E_ARG, "the device-side CVaR sorts at most 4096 samples per evaluation");
    QSV_HIP(h, hipSetDevice(h->device));
    std::vector<Circuit*> circs(size_t(n_evals), nullptr);
    for (int i = 0; i < n_evals; ++i) {
        auto it = h->circuits.find(circuit_ids[i]);
        if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id " + std::to_string(circuit_ids[i]));
        circs[size_t(i)] = &it->second;
    }
    static const double dummy = 0.0;
    return sample_batch_locked(h, circs, param_offsets, params ? params : &dummy, shots, seed, nullptr, nullptr, alpha, out_cvar);
}

int qsv_exact_cvar_batch(qsv_t* h, int n_evals, const int* circuit_ids, const int64_t* param_offsets, const double* params,
                         double alpha, double* out_cvar) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (n_evals < 0 || (n_evals > 0 && (!circuit_ids || !param_offsets || !out_cvar))) return fail(h, QSV_E_ARG, "bad arguments");
    if (!(alpha > 0.0) || alpha > 1.0) return fail(h, QSV_E_ARG, "alpha must be in (0, 1]");
    QSV_HIP(h, hipSetDevice(h->device));
    std::vector<Circuit*> circs(size_t(n_evals), nullptr);
    for (int i = 0; i < n_evals; ++i) {
        auto it = h->circuits.find(circuit_ids[i]);
        if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id " + std::to_string(circuit_ids[i]));
        circs[size_t(i)] = &it->second;
    }
    static const double dummy = 0.0;
    return exact_cvar_locked(h, circs, param_offsets, params ? params : &dummy, alpha, out_cvar);
}

int qsv_sample(qsv_t* h, int circuit_id, const double* params, int n_params, int shots, uint64_t seed,
               uint64_t* out_states) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (shots < 0 || n_params < 0 || (shots > 0 && !out_states)) return fail(h, QSV_E_ARG, "bad arguments");
    QSV_HIP(h, hipSetDevice(h->device));
    auto it = h->circuits.find(circuit_id);
    if (it == h->circuits.end()) return fail(h, QSV_E_ARG, "unknown circuit id");
    std::vector<Circuit*> circs{&it->second};
    const int64_t offsets[2] = {0, n_params};
    static const double dummy = 0.0;
    return sample_batch_locked(h, circs, offsets, params ? params : &dummy, shots, seed, out_states, nullptr);
}

int qsv_fitness_table_wait(const volatile uint64_t* own, int count, volatile int64_t* done, int stride, int world, int rank,
                           int64_t step, int budget_us) {
    if (!own || !done || count < 0 || world < 1 || rank < 0 || rank >= world || stride < 1) return QSV_E_ARG;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(budget_us);
    auto expired = [&](unsigned& spins) {
        __builtin_ia32_pause();
        return (++spins & 255u) == 0 && std::chrono::steady_clock::now() > deadline;
    };
    unsigned spins = 0;
    for (int i = 0; i < count;) {  // (values arrive in any order: every word is looked at until it is no sentinel)
        if (own[i] != QSV_TABLE_SENTINEL)
            ++i;
        else if (expired(spins))
            return 1;
    }
    // (the values were seen by this core before the counter is published: whoever sees the counter sees them)
    __atomic_store_n(&done[size_t(rank) * size_t(stride)], step, __ATOMIC_RELEASE);
    for (int r = 0; r < world;) {
        if (__atomic_load_n(&done[size_t(r) * size_t(stride)], __ATOMIC_ACQUIRE) >= step)
            ++r;
        else if (expired(spins))
            return 2;
    }
    return 0;
}

int qsv_set_option(qsv_t* h, const char* name, int value) {
    if (!h) return QSV_E_ARG;
    if (!name) return fail(h, QSV_E_ARG, "option name is null");
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    const std::string key(name);
    if (key == "split") {
        if (value != 0 && !h->d_side.ptr && h->n > h->geo.k && h->n <= 28)
            return fail(h, QSV_E_ARG, "this handle was created without side tables (QSV_SPLIT=0): splitting cannot be switched on");
        h->split_enabled = value != 0;
    } else if (key == "factor") {
        if (value != 0 && h->split_enabled && h->d_side.ptr && !h->d_factor.ptr)
            return fail(h, QSV_E_ARG, "this handle was created without the factorised path (QSV_FACTOR=0)");
        h->factor_enabled = value != 0;
    } else if (key == "fused_factor") {
        h->fused_factor = value != 0;
    } else if (key == "chain_stream") {
        h->chain_enabled = value != 0;
    } else if (key == "poll_results") {
        h->poll_results = value != 0;
    } else if (key == "repeat_layout") {
        h->repeat_enabled = value != 0;
    } else if (key == "split_max_keys") {
        if (value < 0 || value > kMaxSplitKeys) return fail(h, QSV_E_ARG, "split_max_keys must be between 0 and 5");
        h->split_max_keys = value;
    } else if (key == "split_sampling") {
        h->split_sampling = value != 0;
    } else if (key == "fused_lds_table") {  // one-launch route: sides' states handed to their Gram matrices through LDS (same bits either way)
        h->fused_lds_table = value != 0;
    } else if (key == "sides_r3") {  // one-launch route: sides with eight amplitudes per thread, three-key sides on two workgroups.  Circuits registered afterwards.
        h->sides_r3 = value != 0;
    } else if (key == "side_diag") {  // one-launch route: a side's values of D from a table of its own (one run) instead of gathered from D
        // (same values either way; the plans name the tables: everything is uploaded again when it is next used)
        if (h->side_diag != (value != 0)) {
            QSV_HIP(h, sync_streams(h));
            for (auto& kv : h->circuits) kv.second.uploaded = false;
            h->sdiag_used = 0;
        }
        h->side_diag = value != 0;
    } else if (key == "streams") {
        if (value < 1 || value > h->n_lane_streams + 1) return fail(h, QSV_E_ARG, "streams must be between 1 and the number the handle was created with");
        h->n_streams = value;
    } else {
        return fail(h, QSV_E_ARG, "unknown option '" + key + "'");
    }
    return QSV_OK;
}

int qsv_set_profiling(qsv_t* h, int enabled) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    h->epoch += 1;
    h->profiling = enabled != 0;
    return QSV_OK;
}

int qsv_get_profile(const qsv_t* h, qsv_profile* out) {
    if (!h || !out) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    *out = h->prof;
    return QSV_OK;
}

static int bench_ops_locked(qsv_t* h, int n_ops, const qsv_op* ops, int reps, double* out_ms_per_rep,
                            int* out_n_passes) {
    if (!out_ms_per_rep || reps < 1) return fail(h, QSV_E_ARG, "bad arguments");
    QSV_HIP(h, hipSetDevice(h->device));
    int id = 0;
    // no folding: the gates must really be applied to the resident state
    int rc = register_circuit(h, n_ops, ops, 0, &id, /*fold=*/false);
    if (rc) return rc;
    Circuit& c = h->circuits.find(id)->second;
    auto cleanup = [&]() { h->circuits.erase(id); };
    std::vector<Circuit*> cc{&c};
    static const double dummy = 0.0;
    if ((rc = batch_layout(h, cc, std::vector<int64_t>{0})) || (rc = batch_ship(h, 0, 1, &dummy))) {
        h->batch.circs.clear();
        cleanup();
        return rc;
    }
    h->batch.circs.clear();
    PassArgs a{};
    a.plan = static_cast<const uint32_t*>(h->d_arena.ptr);
    a.mats = static_cast<const double*>(h->d_mats.ptr);
    a.evals = batch_evals(h);
    a.states = h->d_states.ptr;
    a.wtab = h->d_wtab.ptr;
    a.wtab_stride = h->wtab_stride;
    a.state_stride = uint64_t(1) << h->n;
    a.mode = kModeFinalStore | h->stream_mode;  // read-modify-write of the resident state, no synthesis
    const unsigned chunks = chunks_per_state(h);
    a.tiles_per_block = (h->geo.blocks_per_state + chunks - 1) / chunks;
    dim3 grid(chunks, 1);
    const int n_passes = c.plan.stats.n_passes;
    auto sweep = [&]() -> hipError_t {
        for (int p = 0; p < n_passes; ++p) {
            a.pass_index = uint32_t(p);
            hipError_t e = launch_pass(h->dtype, h->geo.r, h->cfg.xmode, grid, h->geo.threads_launch, h->geo.lds_bytes, h->stream, a);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    hipEvent_t e0, e1;
    QSV_HIP(h, hipEventCreate(&e0));
    QSV_HIP(h, hipEventCreate(&e1));
    QSV_HIP(h, sweep());  // untimed: code object load, plan in cache
    QSV_HIP(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) QSV_HIP(h, sweep());
    QSV_HIP(h, hipEventRecord(e1, h->stream));
    QSV_HIP(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    QSV_HIP(h, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *out_ms_per_rep = double(ms) / reps;
    if (out_n_passes) *out_n_passes = n_passes;
    cleanup();
    return QSV_OK;
}

int qsv_bench_ops(qsv_t* h, int n_ops, const qsv_op* ops, int reps, double* out_ms_per_rep, int* out_n_passes) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    for (int i = 0; i < n_ops; ++i)
        if (ops && ops[i].kind != QSV_OP_ID && (ops[i].p_theta >= 0 || ops[i].p_phi >= 0 || ops[i].p_lambda >= 0))
            return fail(h, QSV_E_ARG, "qsv_bench_ops needs bound (literal) angles");
    return bench_ops_locked(h, n_ops, ops, reps, out_ms_per_rep, out_n_passes);
}

int qsv_bench_gate(qsv_t* h, int target, int control, double theta, double phi, double lambda, int reps,
                   double* out_ms_per_sweep) {
    if (!h) return QSV_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (target < 0 || target >= h->n || control >= h->n) return fail(h, QSV_E_ARG, "qubit out of range");
    qsv_op op{};
    op.kind = control >= 0 ? QSV_OP_CU3 : QSV_OP_U;
    op.target = uint8_t(target);
    op.control = control >= 0 ? uint8_t(control) : uint8_t(QSV_NO_CONTROL);
    op.p_theta = op.p_phi = op.p_lambda = -1;
    op.theta = theta;
    op.phi = phi;
    op.lambda = lambda;
    return bench_ops_locked(h, 1, &op, reps, out_ms_per_sweep, nullptr);
}

int qsv_split_describe(int n_qubits, int n_ops, const qsv_op* ops, int max_side, uint64_t* mask_a, qsv_op* ops_a,
                       int capacity_a, int* n_ops_a, qsv_op* ops_b, int capacity_b, int* n_ops_b) {
    if (!mask_a || !n_ops_a || !n_ops_b) return fail(nullptr, QSV_E_ARG, "null output") - 100;
    if (n_qubits < 1 || n_qubits > 32) return fail(nullptr, QSV_E_ARG, "n_qubits must be in [1, 32]") - 100;
    if (validate_ops(nullptr, n_qubits, n_ops, ops, 1 << 30)) return QSV_E_ARG - 100;
    std::vector<AngleSource> angles;
    const std::vector<GateIn> gates = gates_of(ops, n_ops, &angles);
    const SplitCircuits sc = find_split(n_qubits, gates, angles, max_side);
    if (!sc.ok) return -1;
    *mask_a = sc.mask[0];
    qsv_op* outs[2] = {ops_a, ops_b};
    const int caps[2] = {capacity_a, capacity_b};
    int* counts[2] = {n_ops_a, n_ops_b};
    for (int s = 0; s < 2; ++s) {
        *counts[s] = int(sc.gates[s].size());
        if (!outs[s] || caps[s] < *counts[s]) continue;
        for (size_t i = 0; i < sc.gates[s].size(); ++i) {
            const GateIn& g = sc.gates[s][i];
            const AngleSource& a = sc.angles[s][size_t(g.op)];
            qsv_op o{};
            o.kind = g.control < 0 ? QSV_OP_U : QSV_OP_CU3;
            o.target = uint8_t(g.target);
            o.control = g.control < 0 ? uint8_t(QSV_NO_CONTROL) : uint8_t(g.control);
            o.p_theta = a.p_theta; o.p_phi = a.p_phi; o.p_lambda = a.p_lambda;
            o.theta = a.theta; o.phi = a.phi; o.lambda = a.lambda;
            outs[s][i] = o;
        }
    }
    return sc.n_keys;
}

int qsv_plan_build(int n_qubits, int dtype, int n_ops, const qsv_op* ops, const qsv_plan_config* cfg,
                   uint32_t* out_words, size_t capacity_words, size_t* n_words) {
    if (!n_words) return fail(nullptr, QSV_E_ARG, "n_words is null");
    if (dtype != QSV_F64 && dtype != QSV_F32) return fail(nullptr, QSV_E_ARG, "bad dtype");
    if (n_qubits < 1 || n_qubits > 32) return fail(nullptr, QSV_E_ARG, "n_qubits must be in [1, 32]");
    int rc = validate_ops(nullptr, n_qubits, n_ops, ops, 1 << 30);
    if (rc) return rc;
    try {
        PlanConfig pc = resolve_config(cfg, dtype, n_qubits);
        std::vector<AngleSource> angles;
        std::vector<GateIn> gates = gates_of(ops, n_ops, &angles);
        CircuitPlan plan = build_plan(n_qubits, gates, angles, pc);
        *n_words = plan.words.size();
        if (out_words && capacity_words >= plan.words.size())
            std::memcpy(out_words, plan.words.data(), plan.words.size() * 4);
    } catch (const std::exception& e) {
        return fail(nullptr, QSV_E_ARG, e.what());
    }
    return QSV_OK;
}

}  // extern "C"
