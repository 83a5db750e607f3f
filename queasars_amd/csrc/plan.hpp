// Host-side pass scheduler for the tiled statevector kernel.  Pure C++ (no HIP): it is exercised on the CPU by
// tests/test_plan.py through qsv_plan_build().
//
// Model.  A circuit is a list of single-target gates (u: no control, cu3: one control) on n qubits.  The state is
// swept in PASSES.  A pass picks k qubits (the TILE); every workgroup owns one tile of 2^k amplitudes (the other
// n-k index bits are fixed per workgroup) and keeps it on chip for the whole pass.  Inside a pass the tile lives
// in registers: each of 2^t threads holds 2^r amplitudes (k = t + r).  A ROUND chooses which r tile bits are
// "register bits"; a gate is applied in a round in which its TARGET is a register bit (a 2x2 butterfly between
// two registers of one thread).  The CONTROL of a cu3 never has to be a register bit, nor even in the tile:
// it is a predicate on the register index, on the thread index or on the workgroup's fixed bits.  Between rounds
// the tile is transposed through LDS (an EXCHANGE) so that a different set of tile bits becomes register bits.
//
// All index maps (thread/register -> global offset, thread/register -> LDS offset) are GF(2)-linear, so each is
// shipped to the device as one column per thread bit and per register bit; the kernel XORs columns together.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace qsv {

struct PlanConfig {
    int tile_bits = 12;  // k for n >= k
    int reg_bits = 4;    // r
    int low_bits = 4;    // c: tile always contains qubits 0..c-1 (coalescing)
    int elem_bytes = 16; // size of one complex amplitude in LDS (16 = fp64, 8 = fp32)
};

struct GateIn {
    int target;
    int control;  // -1 if none
    int mat;      // index of the gate's 2x2 matrix in the per-evaluation matrix buffer
};

// Resolved geometry for a given n (same for every circuit on the handle).
struct Geometry {
    int n = 0, k = 0, r = 0, t = 0, c = 0;
    int threads_active = 0;  // 2^t
    int threads_launch = 0;  // max(64, 2^t)
    uint32_t blocks_per_state = 0;  // 2^(n-k)
    size_t lds_bytes = 0;
};
Geometry make_geometry(int n_qubits, const PlanConfig& cfg);

struct PlanStats {
    int n_passes = 0;
    int n_rounds = 0;
    int n_exchanges = 0;
    int n_gates = 0;
    int lds_conflict_cycles = 0;  // extra LDS cycles per wave-instruction summed over exchanges (0 = conflict free)
};

struct CircuitPlan {
    std::vector<uint32_t> words;  // encoded plan, see plan.cpp for the layout
    PlanStats stats;
};

// Word layout constants shared with the kernel (kernels.hip includes this header).
// circuit: [0] n_passes  [1] n_mats  [2 .. 2+n_passes) pass offsets (words, relative to the circuit base)
// pass:    [0] k | r<<8 | t<<16 | n_rounds<<24      [1] reserved
//          [2 .. 2+k) tile bit j -> qubit position (ascending)
//          then (t+r) global columns for the load layout, (t+r) for the store layout (amplitude offsets)
//          then the rounds
// round:   [0] n_gates | has_exchange<<16
//          if has_exchange: (t+r) LDS write columns (previous layout), (t+r) LDS read columns (this layout),
//                           both in ELEMENT units under the same swizzle
//          then 4 words per gate: [0] target register bit | mat<<8   [1] ctrl mask over register index
//                                 [2] ctrl mask over thread index     [3] ctrl mask over the global index
constexpr uint32_t kPassHeaderWords = 2;
constexpr uint32_t kGateWords = 4;

CircuitPlan build_plan(int n_qubits, const std::vector<GateIn>& gates, const PlanConfig& cfg);

}  // namespace qsv
