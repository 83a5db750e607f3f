// Register splitting: evaluate a weakly entangled circuit as two small circuits and a combination of their final states.
//
// EVQE individuals are shallow (a few layers of one gate per qubit), so their controlled rotations often leave the
// register in two halves A and B that interact through very few cu3 gates.  Every such CROSS gate (control c on one
// side, target on the other) is  P0(c) (x) I  +  P1(c) (x) U : applied to a sum of product terms it doubles the number
// of terms, unless the terms already have a definite value of c (the same control was used before and not rotated
// since: one KEY = one control qubit between two gates that target it).  With K keys cut by the partition
//
//     psi[i] = sum over kappa in {0,1}^K of  a_kappa[i restricted to A] * b_kappa[i restricted to B]
//
// where a_kappa / b_kappa are the final states of two VIRTUAL circuits on |A| + K and |B| + K qubits: the side's own
// gates, plus one extra qubit per key that starts as (1, 1) and is never targeted.  On the target side the cross gate
// becomes cu3(key -> target); on the control side the key's first use becomes the projector P_kappa(c), written as
// [X(c) if key] P0(c) [X(c) if key].  Both virtual circuits are small (at most a few qubits more than a tile), so the
// ordinary pass kernel runs each in one to sixteen workgroups.  What is left is the expectation value of
// psi = sum_kappa a_kappa (x) b_kappa:
//   * under a diagonal operator whose terms have at most two Z factors (Ising / QUBO), from weighted Gram matrices of the
//     two small states alone (kernels.hpp: launch_factor) -- nothing of size 2^n is touched;
//   * under any other diagonal operator, one streaming kernel (contract_kernel) forms psi on the fly and reduces
//     <psi|D|psi> -- the only sweep over 2^n indices, with 4 * 2^K + 3 fp64 operations per amplitude instead of a pass of
//     gates;
//   * under a general Pauli operator, two small matrices per term (launch_factor_terms);
//   * and samples of |psi|^2 are drawn from the two states directly (launch_split_sample).
//
// This generalises the compact first pass (plan.hpp): there only pass 0 worked on a table of tiles, here every gate
// does.  A circuit that has no such partition (deeper, well entangled circuits) keeps the ordinary multi-pass plan.
#pragma once

#include <cstdint>
#include <vector>

#include "plan.hpp"

namespace qsv {

// Angle-table entries with p_theta below -1 are FIXED matrices (prepare_kernel): the virtual circuits need a projector,
// an unnormalised |0> + |1> and an exact X.  Public op lists cannot contain them (validate_ops).
constexpr int32_t kFixedProj0 = -2;  // [[1, 0], [0, 0]]
constexpr int32_t kFixedOnes = -3;   // [[1, 0], [1, 0]]: applied to |0> it gives (1, 1)
constexpr int32_t kFixedX = -4;      // [[0, 1], [1, 0]]

constexpr int kMaxSplitKeys = 5;  // (beyond three: 16 / 32 product terms, kernels.hpp launch_factor_big; quadratic operators only)

struct SplitCircuits {
    bool ok = false;
    int n_keys = 0;
    int n_side[2] = {0, 0};          // real qubits of side A / B (virtual circuits have n_side + n_keys qubits)
    uint64_t mask[2] = {0, 0};       // qubits of side A / B
    std::vector<GateIn> gates[2];    // the virtual circuits: side qubits renumbered 0.. in ascending order, then the keys
    std::vector<AngleSource> angles[2];
};

// max_side: most qubits a virtual circuit may have.  Deterministic.
SplitCircuits find_split(int n_qubits, const std::vector<GateIn>& gates, const std::vector<AngleSource>& op_angles,
                         int max_side, int max_keys = kMaxSplitKeys);
// Several size limits in order of preference (smaller virtual circuits first): the result of the first limit that has a
// partition, found in ONE enumeration of key sets.
SplitCircuits find_split(int n_qubits, const std::vector<GateIn>& gates, const std::vector<AngleSource>& op_angles,
                         const std::vector<int>& max_sides, int max_keys = kMaxSplitKeys);

}  // namespace qsv
