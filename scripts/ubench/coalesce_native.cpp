// qsv_eval_coalesced from native threads: what the library's merging of concurrent one-circuit callers delivers when
// the callers are not Python threads (bench.py's threaded_b1_evals_per_s is bounded by CPython's thread hand-over).
//   g++ -O2 -std=c++17 -pthread scripts/ubench/coalesce_native.cpp -Iinclude -Lqueasars_amd -lqsv -Wl,-rpath,$PWD/queasars_amd -o scripts/ubench/coalesce_native
//   scripts/ubench/coalesce_native [n_qubits] [threads] [rounds]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "qsv.h"

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20, P = argc > 2 ? atoi(argv[2]) : 64, rounds = argc > 3 ? atoi(argv[3]) : 50;
    qsv_t* h = nullptr;
    if (qsv_create(n, QSV_F64, 0, nullptr, &h)) { fprintf(stderr, "create: %s\n", qsv_last_error(nullptr)); return 1; }
    // Ising operator: all ZZ pairs + Z fields
    std::mt19937_64 rng(2020);
    std::normal_distribution<double> gauss;
    std::vector<uint64_t> x, z; std::vector<double> re, im;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { x.push_back(0); z.push_back((1ull << i) | (1ull << j)); re.push_back(gauss(rng)); im.push_back(0); }
    for (int i = 0; i < n; ++i) { x.push_back(0); z.push_back(1ull << i); re.push_back(gauss(rng)); im.push_back(0); }
    if (qsv_set_operator(h, int(x.size()), x.data(), z.data(), re.data(), im.data())) { fprintf(stderr, "operator: %s\n", qsv_last_error(h)); return 1; }
    // P random four-layer circuits in the EVQE style (every qubit one role per layer), every angle a parameter
    std::uniform_real_distribution<double> angle(0.0, 6.283185307179586);
    std::vector<int> ids(P); std::vector<std::vector<double>> params(P);
    for (int c = 0; c < P; ++c) {
        std::vector<qsv_op> ops;
        for (int layer = 0; layer < 4; ++layer) {
            std::vector<int> q(n); for (int i = 0; i < n; ++i) q[i] = i;
            std::shuffle(q.begin(), q.end(), rng);
            for (int i = 0; i < n;) {
                qsv_op op{};
                const int base = int(params[c].size());
                op.p_theta = base; op.p_phi = base + 1; op.p_lambda = base + 2;
                for (int k = 0; k < 3; ++k) params[c].push_back(angle(rng));
                if ((rng() & 1) || i + 1 >= n) { op.kind = QSV_OP_U; op.target = uint8_t(q[i]); op.control = QSV_NO_CONTROL; i += 1; }
                else { op.kind = QSV_OP_CU3; op.control = uint8_t(q[i]); op.target = uint8_t(q[i + 1]); i += 2; }
                ops.push_back(op);
            }
        }
        if (qsv_circuit_create(h, int(ops.size()), ops.data(), int(params[c].size()), &ids[c])) { fprintf(stderr, "circuit: %s\n", qsv_last_error(h)); return 1; }
    }
    std::vector<double> batch(P);
    {   // reference values from one batched call
        std::vector<int64_t> off(P + 1, 0); std::vector<double> flat;
        for (int c = 0; c < P; ++c) { flat.insert(flat.end(), params[c].begin(), params[c].end()); off[c + 1] = int64_t(flat.size()); }
        if (qsv_eval_circuits(h, P, ids.data(), off.data(), flat.data(), batch.data())) { fprintf(stderr, "eval: %s\n", qsv_last_error(h)); return 1; }
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < rounds; ++r) qsv_eval_circuits(h, P, ids.data(), off.data(), flat.data(), batch.data());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("one caller, batches of %d:            %9.0f evals/s\n", P, P * rounds / s);
    }
    std::vector<double> got(P);
    std::atomic<int> arrived{0}, generation{0}, failures{0};
    auto worker = [&](int c) {
        for (int r = 0; r < rounds + 3; ++r) {
            // all threads start a round together, like the futures of one selection step
            const int gen = generation.load();
            if (arrived.fetch_add(1) + 1 == P) { arrived.store(0); generation.fetch_add(1); }
            else while (generation.load() == gen) std::this_thread::yield();
            if (qsv_eval_coalesced(h, ids[c], params[c].data(), int(params[c].size()), 0.0, &got[c])) failures.fetch_add(1);
        }
    };
    std::vector<std::thread> pool;
    const auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < P; ++c) pool.emplace_back(worker, c);
    for (auto& t : pool) t.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int wrong = 0;
    for (int c = 0; c < P; ++c) wrong += got[c] != batch[c];
    printf("%d native threads, one circuit per call: %9.0f evals/s   (failures %d, values differing from the batched call %d)\n", P,
           P * (rounds + 3) / s, failures.load(), wrong);
    qsv_destroy(h);
    return failures.load() || wrong;
}
