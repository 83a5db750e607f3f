/*
 * Plain-C restatement of the circuit-evaluation hot path (TEST INFRASTRUCTURE, parity unpinned).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * library (queasars_amd/csrc) never links or calls it.  See oracle/__init__.py.
 *
 * It follows the same published semantics as oracle/statevector_oracle.py and is cross-checked
 * against it in tests/test_oracle.py:
 *   - gates id / u / cu3 as emitted by the EVQE genome
 *     (reference: queasars/minimum_eigensolvers/evqe/quantum_circuit/quantum_gate.py:78-79, :96-102, :157-165);
 *   - little-endian qubit order (reference: queasars/utility/pauli_strings.py:38-40);
 *   - result = real(<psi|H|psi>) from |0..0> (reference: queasars/circuit_evaluation/circuit_evaluation.py:200-215).
 *
 * One gate = one sweep over the 2^n amplitudes, the way a CPU statevector simulator without gate fusion
 * works.  OpenMP is optional (-fopenmp); without it the code is single threaded.
 *
 * State layout: interleaved (re, im) doubles, amplitude i at state[2*i], state[2*i+1].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { QSVO_ID = 0, QSVO_U = 1, QSVO_CU3 = 2 };

typedef struct {
    double re, im;
} cplx;

static inline cplx cmul(cplx a, cplx b) { return (cplx){a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static inline cplx cadd(cplx a, cplx b) { return (cplx){a.re + b.re, a.im + b.im}; }

void qsvo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int qsvo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Qiskit UGate matrix, row major m[0..3] = m00 m01 m10 m11 */
void qsvo_u_matrix(double theta, double phi, double lam, double* out8) {
    double c = cos(theta / 2.0), s = sin(theta / 2.0);
    cplx m00 = {c, 0.0};
    cplx m01 = {-cos(lam) * s, -sin(lam) * s};
    cplx m10 = {cos(phi) * s, sin(phi) * s};
    cplx m11 = {cos(phi + lam) * c, sin(phi + lam) * c};
    out8[0] = m00.re; out8[1] = m00.im; out8[2] = m01.re; out8[3] = m01.im;
    out8[4] = m10.re; out8[5] = m10.im; out8[6] = m11.re; out8[7] = m11.im;
}

static void apply_gate(cplx* st, int n, int target, int control, const double* m8) {
    const cplx m00 = {m8[0], m8[1]}, m01 = {m8[2], m8[3]}, m10 = {m8[4], m8[5]}, m11 = {m8[6], m8[7]};
    const int64_t half = (int64_t)1 << (n - 1);
    const int64_t tbit = (int64_t)1 << target;
    const int64_t cbit = control >= 0 ? ((int64_t)1 << control) : 0;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < half; ++p) {
        /* insert a zero at bit `target` */
        int64_t i0 = ((p >> target) << (target + 1)) | (p & (tbit - 1));
        if ((i0 & cbit) != cbit) continue;
        int64_t i1 = i0 | tbit;
        cplx a0 = st[i0], a1 = st[i1];
        st[i0] = cadd(cmul(m00, a0), cmul(m01, a1));
        st[i1] = cadd(cmul(m10, a0), cmul(m11, a1));
    }
}

/*
 * kinds[i] in {0,1,2}; targets[i]; controls[i] (-1 if none); angles[3*i..3*i+2] = theta, phi, lam.
 * state: 2 * 2^n doubles, overwritten with the final state (starts from |0..0> when init != 0).
 */
int qsvo_simulate(int n, int n_ops, const int32_t* kinds, const int32_t* targets, const int32_t* controls,
                  const double* angles, double* state, int init) {
    if (n < 1 || n > 40) return -1;
    cplx* st = (cplx*)state;
    const int64_t dim = (int64_t)1 << n;
    if (init) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < dim; ++i) st[i] = (cplx){0.0, 0.0};
        st[0].re = 1.0;
    }
    for (int g = 0; g < n_ops; ++g) {
        if (kinds[g] == QSVO_ID) continue;
        if (targets[g] < 0 || targets[g] >= n) return -2;
        double m8[8];
        qsvo_u_matrix(angles[3 * g], angles[3 * g + 1], angles[3 * g + 2], m8);
        if (kinds[g] == QSVO_U) {
            apply_gate(st, n, targets[g], -1, m8);
        } else if (kinds[g] == QSVO_CU3) {
            if (controls[g] < 0 || controls[g] >= n || controls[g] == targets[g]) return -3;
            apply_gate(st, n, targets[g], controls[g], m8);
        } else {
            return -4;
        }
    }
    return 0;
}

/* <psi| sum_k c_k P_k |psi>,  P = i^{|x&z|} X^x Z^z; out[0] = re, out[1] = im */
int qsvo_expectation(int n, const double* state, int n_terms, const uint64_t* x_mask, const uint64_t* z_mask,
                     const double* coeff_re, const double* coeff_im, double* out) {
    const cplx* st = (const cplx*)state;
    const int64_t dim = (int64_t)1 << n;
    double tot_re = 0.0, tot_im = 0.0;
    for (int k = 0; k < n_terms; ++k) {
        const uint64_t x = x_mask[k], z = z_mask[k];
        double acc_re = 0.0, acc_im = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc_re, acc_im)
        for (int64_t i = 0; i < dim; ++i) {
            uint64_t j = (uint64_t)i ^ x;
            double sgn = (__builtin_popcountll(j & z) & 1) ? -1.0 : 1.0;
            /* conj(a_i) * a_j */
            double re = st[i].re * st[j].re + st[i].im * st[j].im;
            double im = st[i].re * st[j].im - st[i].im * st[j].re;
            acc_re += sgn * re;
            acc_im += sgn * im;
        }
        /* multiply by i^{ny} then by the coefficient */
        int ny = __builtin_popcountll(x & z) & 3;
        double pr = acc_re, pi = acc_im, t;
        for (int r = 0; r < ny; ++r) { t = pr; pr = -pi; pi = t; }
        tot_re += coeff_re[k] * pr - coeff_im[k] * pi;
        tot_im += coeff_re[k] * pi + coeff_im[k] * pr;
    }
    out[0] = tot_re;
    out[1] = tot_im;
    return 0;
}

/* Diagonal fast path: sum_i |a_i|^2 * sum_k c_k (-1)^{popcount(i & z_k)} */
double qsvo_diagonal_expectation(int n, const double* state, int n_terms, const uint64_t* z_mask,
                                 const double* coeff_re) {
    const cplx* st = (const cplx*)state;
    const int64_t dim = (int64_t)1 << n;
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int64_t i = 0; i < dim; ++i) {
        double p = st[i].re * st[i].re + st[i].im * st[i].im;
        double d = 0.0;
        for (int k = 0; k < n_terms; ++k) d += (__builtin_popcountll((uint64_t)i & z_mask[k]) & 1) ? -coeff_re[k] : coeff_re[k];
        acc += p * d;
    }
    return acc;
}

/* D[i] = sum_k c_k (-1)^{popcount(i & z_k)}: built once per operator, reused by every evaluation */
void qsvo_diagonal_table(int n, int n_terms, const uint64_t* z_mask, const double* coeff_re, double* table) {
    const int64_t dim = (int64_t)1 << n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < dim; ++i) {
        double d = 0.0;
        for (int k = 0; k < n_terms; ++k) d += (__builtin_popcountll((uint64_t)i & z_mask[k]) & 1) ? -coeff_re[k] : coeff_re[k];
        table[i] = d;
    }
}

double qsvo_diagonal_expectation_table(int n, const double* state, const double* table) {
    const cplx* st = (const cplx*)state;
    const int64_t dim = (int64_t)1 << n;
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int64_t i = 0; i < dim; ++i) acc += (st[i].re * st[i].re + st[i].im * st[i].im) * table[i];
    return acc;
}

/*
 * One whole circuit evaluation (what the reference counts as one circuit-eval):
 * simulate from |0..0> in a caller-provided scratch state, then <H>.  Returns real(<H>).
 */
double qsvo_evaluate(int n, int n_ops, const int32_t* kinds, const int32_t* targets, const int32_t* controls,
                     const double* angles, int n_terms, const uint64_t* x_mask, const uint64_t* z_mask,
                     const double* coeff_re, const double* coeff_im, const double* diag_table /* may be NULL */,
                     double* scratch_state) {
    if (qsvo_simulate(n, n_ops, kinds, targets, controls, angles, scratch_state, 1) != 0) return NAN;
    if (diag_table) return qsvo_diagonal_expectation_table(n, scratch_state, diag_table);
    int diagonal = 1;
    for (int k = 0; k < n_terms; ++k)
        if (x_mask[k] != 0 || coeff_im[k] != 0.0) diagonal = 0;
    if (diagonal) return qsvo_diagonal_expectation(n, scratch_state, n_terms, z_mask, coeff_re);
    double out[2];
    qsvo_expectation(n, scratch_state, n_terms, x_mask, z_mask, coeff_re, coeff_im, out);
    return out[0];
}
