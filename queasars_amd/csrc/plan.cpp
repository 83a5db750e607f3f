// Pass scheduler: see plan.hpp for the model and the encoded layout.
#include "plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>

namespace qsv {

Geometry make_geometry(int n, const PlanConfig& cfg) {
    if (n < 1 || n > 32) throw std::invalid_argument("n_qubits must be in [1, 32]");
    if (cfg.reg_bits < 1 || cfg.reg_bits > 4) throw std::invalid_argument("reg_bits must be in [1, 4]");
    if (cfg.tile_bits < cfg.reg_bits + 1 || cfg.tile_bits > cfg.reg_bits + 9)
        throw std::invalid_argument("tile_bits must be in [reg_bits+1, reg_bits+9]");
    if (cfg.xmode < 0 || cfg.xmode > 2) throw std::invalid_argument("xmode must be 0, 1 or 2");
    if (cfg.low_bits < 0 || cfg.low_bits > 8) throw std::invalid_argument("low_bits must be in [0, 8]");
    Geometry g;
    g.n = n;
    if (n >= cfg.tile_bits) {
        g.k = cfg.tile_bits;
        g.r = cfg.reg_bits;
    } else {
        g.k = n;
        g.r = std::min(cfg.reg_bits, std::max(1, n - 6));
    }
    g.t = g.k - g.r;
    g.c = std::min(cfg.low_bits, g.t);
    g.cl = std::max(0, std::min(cfg.lane_bits, g.c));
    g.threads_active = 1 << g.t;
    g.threads_launch = std::max(64, g.threads_active);
    g.blocks_per_state = 1u << (n - g.k);
    g.lds_bytes = (size_t(1) << g.k) * size_t(cfg.amp_bytes) / (cfg.xmode == 2 ? 2 : 1);
    return g;
}

namespace {

// A layout assigns every tile bit either to a thread bit or to a register bit.
struct Layout {
    std::vector<int> thr;  // thr[u] = tile bit under thread bit u (ascending)
    std::vector<int> reg;  // reg[v] = tile bit under register bit v (ascending)
};

// `last` lists tile bits that should become the HIGHEST thread bits (wave-index bits): controls of the round's
// gates, so that the control predicate is uniform per wave and a wave whose control bit is 0 skips the gate.
Layout make_layout(int k, const std::vector<int>& regbits_sorted, const std::vector<int>& last = {}) {
    Layout l;
    l.reg = regbits_sorted;
    std::vector<char> skip(k, 0);
    for (int b : regbits_sorted) skip[b] = 1;
    for (int b : last) skip[b] = 1;
    for (int b = 0; b < k; ++b)
        if (!skip[b]) l.thr.push_back(b);
    for (int b : last) l.thr.push_back(b);
    return l;
}

// ---- LDS swizzle -------------------------------------------------------------------------------------
// LDS element index of tile index x:  y = XOR_{bit p of x set} (2^pos[p] ^ M[p]),  M[p] < 2^min(pos[p], sb).
// pos is a permutation of the address bits (the identity, except for exchanges that stay inside each wave, see
// choose_swizzle); every M[p] only touches address bits below pos[p] (and below sb), so the map is unit lower
// triangular over GF(2) in address order: a bijection for any M.  sb = 4 for 16-byte elements (ds_read_b128 wants 16 distinct 16-B slots per lane
// group), 5 for 8-byte ones.
struct Swizzle {
    std::array<uint8_t, 16> m{};    // m[p] for tile bit p
    std::array<uint8_t, 16> pos{};  // address bit that carries tile bit p (a permutation of 0..k-1)
    Swizzle() {
        for (int p = 0; p < 16; ++p) pos[size_t(p)] = uint8_t(p);
    }
};

inline uint32_t lds_col(int tile_bit, const Swizzle& s) {
    return (1u << s.pos[size_t(tile_bit)]) ^ uint32_t(s.m[size_t(tile_bit)]);
}

// Extra LDS cycles of one wave-instruction (beyond the conflict-free count) for a given layout's lane map.
// Banking per MI355X_MICROARCH.md section LDS:
//   ds_write_b128: 8 groups of 8 contiguous lanes, bank = (addr/4) % 32  -> 16-B slot mod 8
//   ds_read_b128 : 4 groups {0-3,12-15,20-27} {4-11,16-19,28-31} and the same +32, bank = (addr/4) % 64 -> slot mod 16
//   ds_write_b64 : 4 groups of 16 contiguous lanes, bank = (addr/4) % 32 -> 8-B slot mod 16
//   ds_read_b64  : 2 groups of 32 lanes, bank = (addr/4) % 64 -> 8-B slot mod 32
// Lanes with identical addresses broadcast (cannot happen here: the map is injective).
int conflict_cycles(const Layout& l, const Swizzle& s, int elem_bytes, bool is_write) {
    // lane -> LDS element offset: XOR of the columns of the lane's set bits, filled in lowest-set-bit order
    uint32_t lane_off[64];
    const int nlane_bits = std::min<int>(6, int(l.thr.size()));
    const int active = 1 << nlane_bits;
    lane_off[0] = 0;
    for (int u = 0; u < nlane_bits; ++u) {
        const uint32_t col = lds_col(l.thr[size_t(u)], s);
        for (int lane = 0; lane < (1 << u); ++lane) lane_off[lane | (1 << u)] = lane_off[lane] ^ col;
    }
    // lane groups of the four access kinds (see the comment above), as flat tables
    static const uint8_t kReadB128[4][16] = {
        {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
        {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
        {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
        {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int group_size, slot_mod;
    bool contiguous = true;
    if (elem_bytes == 16) {
        group_size = is_write ? 8 : 16;
        slot_mod = is_write ? 8 : 16;
        contiguous = is_write;
    } else {
        group_size = is_write ? 16 : 32;
        slot_mod = is_write ? 16 : 32;
    }
    int extra = 0;
    for (int g = 0; g < 64 / group_size; ++g) {
        uint8_t count[32] = {0};
        int worst = 0;
        for (int i = 0; i < group_size; ++i) {
            const int lane = contiguous ? g * group_size + i : int(kReadB128[g][i]);
            if (lane >= active) continue;
            const int c = ++count[lane_off[lane] & uint32_t(slot_mod - 1)];
            worst = c > worst ? c : worst;
        }
        extra += worst > 0 ? worst - 1 : 0;
    }
    return extra;
}

struct SwizzleChoice {
    Swizzle s;
    int cost = 0;
};

// Pick M so that writing in layout `a` and reading in layout `b` are both bank-conflict free (or as close as a
// short coordinate descent gets).  Results are memoised: layouts live in tile-bit space, so few distinct
// pairs ever occur.
// `wave_bits` (exchanges that stay inside each wave): the tile bits held by the wave-index thread bits, the same
// before and after.  They are moved to the TOP address bits, so that a wave owns the contiguous LDS region numbered
// by its own index under any swizzle (the terms M[p] stay below bit sb): consecutive exchanges of this kind cannot
// touch each other's data, which is what lets the kernel drop their barriers.
SwizzleChoice choose_swizzle(const Layout& a, const Layout& b, int k, int elem_bytes,
                             const std::vector<int>& wave_bits = {}) {
    static std::mutex mu;
    static std::map<std::vector<int>, SwizzleChoice> memo;
    std::vector<int> key;
    key.push_back(k);
    key.push_back(elem_bytes);
    key.insert(key.end(), wave_bits.begin(), wave_bits.end());
    key.push_back(-2);
    key.insert(key.end(), a.thr.begin(), a.thr.end());
    key.push_back(-1);
    key.insert(key.end(), b.thr.begin(), b.thr.end());
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
    }
    Swizzle base;
    if (!wave_bits.empty()) {
        int next = 0;
        for (int p = 0; p < k; ++p)
            if (std::find(wave_bits.begin(), wave_bits.end(), p) == wave_bits.end()) base.pos[size_t(p)] = uint8_t(next++);
        for (int p : wave_bits) base.pos[size_t(p)] = uint8_t(next++);
    }
    const int sb = elem_bytes == 16 ? 4 : 5;
    auto cost_of = [&](const Swizzle& s) {
        return conflict_cycles(a, s, elem_bytes, true) + conflict_cycles(b, s, elem_bytes, false);
    };
    // tile bits that sit under a lane bit in either layout are the only ones whose M matters
    std::vector<int> relevant;
    for (const Layout* l : {&a, &b})
        for (size_t u = 0; u < l->thr.size() && u < 6; ++u)
            if (base.pos[size_t(l->thr[u])] >= 1 && std::find(relevant.begin(), relevant.end(), l->thr[u]) == relevant.end())
                relevant.push_back(l->thr[u]);
    auto span_of = [&](int p) { return 1 << std::min(int(base.pos[size_t(p)]), sb); };
    SwizzleChoice best;
    best.s = base;
    best.cost = cost_of(best.s);
    uint32_t rng = 0x9E3779B9u;
    // Cheap first: random draws.  A conflict-free swizzle only asks that the images of the tile bits on the lowest
    // lane bits be linearly independent modulo the bank count, which a random assignment satisfies about one time
    // in ten; a draw costs two conflict counts, a descent sweep several hundred.
    for (int trial = 0; trial < 96 && best.cost > 0; ++trial) {
        Swizzle s = base;
        for (int p : relevant) {
            rng = rng * 1664525u + 1013904223u;
            s.m[size_t(p)] = uint8_t((rng >> 24) & uint32_t(span_of(p) - 1));
        }
        const int cst = cost_of(s);
        if (cst < best.cost) {
            best.s = s;
            best.cost = cst;
        }
    }
    for (int restart = 0; restart < 24 && best.cost > 0; ++restart) {
        Swizzle s = base;
        if (restart > 0)
            for (int p : relevant) {
                rng = rng * 1664525u + 1013904223u;
                s.m[p] = uint8_t((rng >> 24) & uint32_t(span_of(p) - 1));
            }
        int cur = cost_of(s);
        bool improved = true;
        while (improved && cur > 0) {
            improved = false;
            for (int p : relevant) {
                uint8_t keep = s.m[p];
                uint8_t arg = keep;
                for (int v = 0; v < span_of(p); ++v) {
                    s.m[p] = uint8_t(v);
                    int cst = cost_of(s);
                    if (cst < cur) {
                        cur = cst;
                        arg = uint8_t(v);
                        improved = true;
                    }
                }
                s.m[p] = arg;
            }
        }
        if (cur < best.cost) {
            best.s = s;
            best.cost = cur;
        }
    }
    std::lock_guard<std::mutex> lock(mu);
    memo[key] = best;
    return best;
}

// ---- dependency-aware first-come selection -----------------------------------------------------------
// Gates are visited in program order.  A visited gate is either taken or deferred; a deferred gate blocks
// later gates that do not commute with it:
//   - anything touching its target (as target or control)
//   - anything targeting its control (gates that merely share the control commute: both are diagonal there)
struct Blocker {
    std::vector<char> full;    // qubit may be neither targeted nor used as a control
    std::vector<char> target;  // qubit may not be targeted (it is the control of a deferred gate)
    explicit Blocker(int n) : full(n, 0), target(n, 0) {}
    bool allows(const GateIn& g) const {
        if (full[g.target] || target[g.target]) return false;
        if (g.control >= 0 && full[g.control]) return false;
        return true;
    }
    void defer(const GateIn& g) {
        full[g.target] = 1;
        if (g.control >= 0) target[g.control] = 1;
    }
};

struct RoundPlan {
    std::vector<int> regbits;  // tile bits, ascending
    std::vector<int> gates;    // indices into the real-gate list, program order
};

struct PassPlan {
    std::vector<int> pos;  // tile bit -> qubit, ascending
    std::vector<RoundPlan> rounds;
};

void push_angle_entry(std::vector<uint32_t>& w, const AngleSource& a) {
    w.push_back(uint32_t(a.p_theta));
    w.push_back(uint32_t(a.p_phi));
    w.push_back(uint32_t(a.p_lambda));
    for (double v : {a.theta, a.phi, a.lambda}) {
        uint64_t bits;
        std::memcpy(&bits, &v, 8);
        w.push_back(uint32_t(bits & 0xffffffffu));
        w.push_back(uint32_t(bits >> 32));
    }
}

}  // namespace

CircuitPlan build_plan(int n, const std::vector<GateIn>& all_gates, const std::vector<AngleSource>& op_angles,
                       const PlanConfig& cfg) {
    const Geometry geo = make_geometry(n, cfg);
    const int k = geo.k, r = geo.r, t = geo.t, c = geo.c, cl = geo.cl;
    for (const GateIn& g : all_gates) {
        if (g.target < 0 || g.target >= n || g.control >= n || g.control == g.target)
            throw std::invalid_argument("gate qubit index out of range");
        if (g.op < 0 || size_t(g.op) >= op_angles.size()) throw std::invalid_argument("gate op index out of range");
    }
    CircuitPlan out;

    // ---- 1. fold leading gates into the initial product state --------------------------------------------
    enum : char { kZero = 0, kProduct = 1, kEntangled = 2 };
    std::vector<char> qstate(n, kZero);
    std::vector<std::vector<int>> folds(n);  // per qubit: op indices of folded u gates, program order
    std::vector<GateIn> gates;               // the real gates
    for (const GateIn& g : all_gates) {
        if (cfg.fold && g.control < 0 && qstate[g.target] != kEntangled) {
            folds[g.target].push_back(g.op);
            qstate[g.target] = kProduct;
            out.stats.n_folded_gates += 1;
        } else if (cfg.fold && g.control >= 0 && qstate[g.control] == kZero) {
            out.stats.n_dropped_gates += 1;  // control is exactly |0>: identity
        } else {
            gates.push_back(g);
            qstate[g.target] = kEntangled;
            if (g.control >= 0) qstate[g.control] = kEntangled;
        }
    }

    // ---- 1b. fusion (plan.hpp): u gates next to a gate on the same target are multiplied into it -------------
    // chain1[i]: the ops whose matrices, multiplied in the order listed, give gate i's matrix (where its control is 1);
    // chain0[i]: the u gates among them (the matrix where the control is 0), empty for a gate that is not multiplexed
    std::vector<std::vector<int>> chain1, chain0;
    {
        std::vector<GateIn> fused;
        std::vector<char> dead;
        std::vector<int> last(size_t(n), -1);       // the latest gate that touches the qubit ..
        std::vector<char> as_target(size_t(n), 0);  // .. and whether it targets it
        for (const GateIn& g : gates) {
            const int j = last[size_t(g.target)];
            const bool open = cfg.fuse && j >= 0 && as_target[size_t(g.target)] && chain1[size_t(j)].size() < kMaxChain;
            int at = -1;
            if (open && g.control < 0) {
                // a u joins the gate before it on this qubit (nothing touched the qubit since)
                chain1[size_t(j)].push_back(g.op);
                if (fused[size_t(j)].control >= 0) chain0[size_t(j)].push_back(g.op);
                at = j;
                out.stats.n_fused_gates += 1;
            } else if (open && fused[size_t(j)].control < 0) {
                // a cu3 takes in the u gates before it: they move here, to the cu3's place in the program
                at = int(fused.size());
                fused.push_back(g);
                chain1.push_back(chain1[size_t(j)]);
                chain1.back().push_back(g.op);
                chain0.push_back(chain1[size_t(j)]);
                dead.push_back(0);
                dead[size_t(j)] = 1;
                out.stats.n_fused_gates += int(chain1[size_t(j)].size());
            } else {
                at = int(fused.size());
                fused.push_back(g);
                chain1.push_back({g.op});
                chain0.emplace_back();
                dead.push_back(0);
            }
            last[size_t(g.target)] = at;
            as_target[size_t(g.target)] = 1;
            if (g.control >= 0) {
                last[size_t(g.control)] = at;
                as_target[size_t(g.control)] = 0;
            }
        }
        size_t keep = 0;
        for (size_t i = 0; i < fused.size(); ++i)
            if (!dead[i]) {
                fused[keep] = fused[i];
                chain1[keep].swap(chain1[i]);
                chain0[keep].swap(chain0[i]);
                ++keep;
            }
        fused.resize(keep);
        chain1.resize(keep);
        chain0.resize(keep);
        gates.swap(fused);
    }

    // ---- 2. passes and rounds ----------------------------------------------------------------------------
    std::vector<int> default_regs;
    for (int b = k - r; b < k; ++b) default_regs.push_back(b);

    // One scheduling attempt.  try_no = 0 is the plain first-come rule; later attempts decline, with probability 1/7, to
    // give a new qubit a place in the tile when its first gate comes by (a deterministic pseudo-random sequence per
    // attempt), which reaches tile sets the greedy rule cannot.  build_plan keeps the attempt with the fewest passes:
    // a pass is a full sweep of the state, and at n = 24 one circuit in six needs three passes under the greedy rule
    // where two suffice.
    // tiles_only: just the choice of tiles and the gates each pass takes (what decides the NUMBER of passes): the rounds of the
    // best attempt are laid out afterwards, once -- an attempt then costs a tenth, and build_plan can afford many more of them
    // (round 4: n = 24, L = 8 3.94 -> 3.5 passes per circuit with 256 attempts, n = 20, L = 6 2.44 -> 2.16).
    // forced: the tiles are given (one qubit mask per pass, the local search's: below) instead of found first come.
    auto attempt_schedule = [&](int try_no, bool tiles_only, const std::vector<uint64_t>* forced = nullptr) {
    uint32_t lcg = 0x2545F491u * uint32_t(try_no + 1);
    auto decline = [&]() {
        if (try_no == 0) return false;
        lcg = lcg * 1664525u + 1013904223u;
        return (lcg >> 16) % 7u == 0u;
    };
    std::vector<char> done(gates.size(), 0);
    size_t n_done = 0;
    std::vector<PassPlan> passes;

    while (n_done < gates.size() || passes.empty()) {
        PassPlan pass;
        std::vector<char> in_tile(n, 0);
        int tile_count = 0;
        if (k == n) {
            std::fill(in_tile.begin(), in_tile.end(), 1);
            tile_count = n;
        } else {
            for (int q = 0; q < c; ++q) in_tile[q] = 1;
            tile_count = c;
        }
        std::vector<int> selected;
        if (forced && passes.size() < forced->size() && k < n) {
            const uint64_t mask = (*forced)[passes.size()];
            tile_count = 0;
            for (int q = 0; q < n; ++q) {
                in_tile[q] = char(mask >> q & 1u);
                tile_count += in_tile[q];
            }
            Blocker blk(n);
            for (size_t i = 0; i < gates.size(); ++i) {
                if (done[i]) continue;
                const GateIn& g = gates[i];
                if (blk.allows(g) && in_tile[g.target])
                    selected.push_back(int(i));
                else
                    blk.defer(g);
            }
        } else {
            Blocker blk(n);
            for (size_t i = 0; i < gates.size(); ++i) {
                if (done[i]) continue;
                const GateIn& g = gates[i];
                bool ok = blk.allows(g);
                if (ok && !in_tile[g.target]) {
                    if (tile_count < k && !decline()) {
                        in_tile[g.target] = 1;
                        ++tile_count;
                    } else {
                        ok = false;
                    }
                }
                if (ok)
                    selected.push_back(int(i));
                else
                    blk.defer(g);
            }
        }
        if (tiles_only) {
            for (int s2 : selected) {
                done[size_t(s2)] = 1;
                ++n_done;
            }
            if (selected.empty() && n_done < gates.size()) return std::vector<PassPlan>();
            passes.emplace_back();
            continue;
        }
        for (int q = 0; q < n && tile_count < k; ++q)
            if (!in_tile[q]) {
                in_tile[q] = 1;
                ++tile_count;
            }
        std::vector<int> tile_bit_of(n, -1);
        for (int q = 0; q < n; ++q)
            if (in_tile[q]) {
                tile_bit_of[q] = int(pass.pos.size());
                pass.pos.push_back(q);
            }

        std::vector<char> placed(selected.size(), 0);
        size_t n_placed = 0;
        bool first = true;
        while (n_placed < selected.size() || pass.rounds.empty()) {
            RoundPlan round;
            std::vector<char> is_reg(k, 0);
            int reg_count = 0;
            Blocker blk(n);
            for (size_t s = 0; s < selected.size(); ++s) {
                if (placed[s]) continue;
                const GateIn& g = gates[selected[s]];
                const int tb = tile_bit_of[g.target];
                bool ok = blk.allows(g);
                // the first layout doubles as the global load layout: its lanes must sit on the low tile bits
                if (ok && first && tb < cl) ok = false;
                if (ok && !is_reg[tb]) {
                    if (reg_count < r) {
                        is_reg[tb] = 1;
                        ++reg_count;
                    } else {
                        ok = false;
                    }
                }
                if (ok) {
                    round.gates.push_back(selected[s]);
                    placed[s] = 1;
                    ++n_placed;
                } else {
                    blk.defer(g);
                }
            }
            for (int b = k - 1; b >= 0 && reg_count < r; --b)
                if (!is_reg[b] && !(first && b < cl)) {
                    is_reg[b] = 1;
                    ++reg_count;
                }
            for (int b = 0; b < k; ++b)
                if (is_reg[b]) round.regbits.push_back(b);
            pass.rounds.push_back(std::move(round));
            first = false;
        }
        // (with lane swaps the encoder appends a restoring swap round itself where the last layout needs one)
        bool low_in_regs = false;
        for (int b : pass.rounds.back().regbits) low_in_regs |= (b < cl);
        if (low_in_regs && !cfg.swaps) {
            RoundPlan tail;
            tail.regbits = default_regs;
            pass.rounds.push_back(std::move(tail));
        }
        for (int s : selected) {
            done[s] = 1;
            ++n_done;
        }
        if (selected.empty() && n_done < gates.size()) {
            if (try_no == 0) throw std::logic_error("scheduler made no progress");
            return std::vector<PassPlan>();  // (an attempt that declined everything: discarded)
        }
        passes.push_back(std::move(pass));
    }
    return passes;
    };
    std::vector<PassPlan> passes = attempt_schedule(0, false);
    // LOCAL SEARCH over the tiles (round 4).  Pass by pass: start from the first-come tile, then trade one tile qubit for one
    // outside it as long as that leaves FEWER QUBITS WITH PENDING TARGETS behind (what the last pass must hold at once), or as
    // many and more gates done.  Deterministic, about a quarter of a millisecond for an eight-layer circuit at 24 qubits.
    std::vector<uint64_t> searched_tiles;
    if (cfg.retries > 0 && n > k && n <= 62 && passes.size() > 2) {
        std::vector<char> done(gates.size(), 0);
        size_t n_done = 0;
        const uint64_t low_mask = (uint64_t(1) << c) - 1;
        // what a pass with tile `mask` takes: (gates selected, qubits that still have a pending target afterwards)
        auto simulate = [&](uint64_t mask, std::vector<int>* taken) {
            Blocker blk(n);
            int count = 0;
            uint64_t pending = 0;
            for (size_t i = 0; i < gates.size(); ++i) {
                if (done[i]) continue;
                const GateIn& g = gates[i];
                if (blk.allows(g) && (mask >> g.target & 1u)) {
                    ++count;
                    if (taken) taken->push_back(int(i));
                } else {
                    blk.defer(g);
                    pending |= uint64_t(1) << g.target;
                }
            }
            return std::make_pair(count, __builtin_popcountll(pending));
        };
        while (n_done < gates.size() && searched_tiles.size() < 16) {
            // the first-come tile
            uint64_t mask = low_mask;
            int tile_count = c;
            {
                Blocker blk(n);
                for (size_t i = 0; i < gates.size(); ++i) {
                    if (done[i]) continue;
                    const GateIn& g = gates[i];
                    bool ok = blk.allows(g);
                    if (ok && !(mask >> g.target & 1u)) {
                        if (tile_count < k) {
                            mask |= uint64_t(1) << g.target;
                            ++tile_count;
                        } else {
                            ok = false;
                        }
                    }
                    if (!ok) blk.defer(g);
                }
                for (int q = 0; q < n && tile_count < k; ++q)
                    if (!(mask >> q & 1u)) {
                        mask |= uint64_t(1) << q;
                        ++tile_count;
                    }
            }
            auto score = simulate(mask, nullptr);
            for (int sweep = 0; sweep < 12 && score.second > 0; ++sweep) {
                uint64_t best_mask = mask;
                auto best_score = score;
                for (int qi = c; qi < n; ++qi) {
                    if (!(mask >> qi & 1u)) continue;
                    for (int qo = c; qo < n; ++qo) {
                        if (mask >> qo & 1u) continue;
                        const uint64_t trial = (mask & ~(uint64_t(1) << qi)) | uint64_t(1) << qo;
                        const auto sc = simulate(trial, nullptr);
                        if (sc.second < best_score.second || (sc.second == best_score.second && sc.first > best_score.first)) {
                            best_score = sc;
                            best_mask = trial;
                        }
                    }
                }
                if (best_mask == mask) break;
                mask = best_mask;
                score = best_score;
            }
            std::vector<int> taken;
            simulate(mask, &taken);
            if (taken.empty()) {  // (no progress: give the search up, the attempts below stand)
                searched_tiles.clear();
                break;
            }
            for (int i : taken) {
                done[size_t(i)] = 1;
                ++n_done;
            }
            searched_tiles.push_back(mask);
        }
        if (n_done < gates.size()) searched_tiles.clear();
    }
    if (cfg.retries > 0 && n > k && passes.size() > 2) {
        size_t best = passes.size();
        int best_try = 0, last_gain = 0;
        if (!searched_tiles.empty() && searched_tiles.size() < best) best = searched_tiles.size();
        // (patience: a circuit whose attempts stop improving is not tried to the end -- most eight-layer circuits at 20 qubits
        // stay at three passes whatever is tried, and their plans are built while a generation waits)
        const int patience = std::max(16, cfg.retries / 4);
        for (int attempt = 1; attempt <= cfg.retries && best > 2 && attempt - last_gain <= patience; ++attempt) {
            const size_t count = attempt_schedule(attempt, true).size();
            if (count > 0 && count < best) {
                best = count;
                best_try = attempt;
                last_gain = attempt;
            }
        }
        if (best_try != 0)
            passes = attempt_schedule(best_try, false);
        else if (!searched_tiles.empty() && searched_tiles.size() < passes.size())
            passes = attempt_schedule(0, false, &searched_tiles);
    }

    // ---- 3. encode ---------------------------------------------------------------------------------------
    std::vector<uint32_t>& w = out.words;
    w.assign(kCircuitHeaderWords, 0);
    w[0] = uint32_t(passes.size());
    w[2] = uint32_t(n);  // ([1], the number of scheduled entries, is known after the rounds are encoded)
    const size_t off_table = w.size();
    w.resize(w.size() + passes.size(), 0);
    std::vector<std::pair<int, int>> schedule;  // (real gate, 1: its matrix where the control is 1 / 0: where it is 0)

    // ---- COMPACT first pass (plan.hpp): worthwhile when there is a second pass to read the table and the outer
    // control patterns are far fewer than the tiles
    int compact_bits = -1;
    std::vector<int> compact_ctrl;  // outer qubits pass 0 uses as controls, ascending
    if (cfg.fold && cfg.compact && passes.size() >= 2 && n > k) {
        const PassPlan& p0 = passes[0];
        for (const RoundPlan& rd : p0.rounds)
            for (int gi : rd.gates) {
                const int c = gates[size_t(gi)].control;
                if (c >= 0 && !std::binary_search(p0.pos.begin(), p0.pos.end(), c) &&
                    std::find(compact_ctrl.begin(), compact_ctrl.end(), c) == compact_ctrl.end())
                    compact_ctrl.push_back(c);
            }
        std::sort(compact_ctrl.begin(), compact_ctrl.end());
        if (int(compact_ctrl.size()) <= int(kMaxCompactBits) && int(compact_ctrl.size()) + 1 <= n - k &&
            n - k <= int(kMaxOuterBits))
            compact_bits = int(compact_ctrl.size());
    }
    out.stats.compact_bits = compact_bits;

    for (size_t pi = 0; pi < passes.size(); ++pi) {
        PassPlan& pass = passes[pi];
        w[off_table + pi] = uint32_t(w.size());
        const size_t pass_header_at = w.size();
        w.push_back(uint32_t(k) | uint32_t(r) << 8 | uint32_t(t) << 16 | uint32_t(pass.rounds.size()) << 24);
        w.push_back(uint32_t(schedule.size()));
        // COMPACT (plan.hpp): pass 0 over the patterns of its outer control qubits, pass 1 reading W x F
        uint32_t pass_flags = 0;
        if (compact_bits >= 0 && pi == 0) pass_flags = kPassCompactStore | uint32_t(compact_bits) << 8;
        if (compact_bits >= 0 && pi == 1) pass_flags = kPassCompactLoad | uint32_t(compact_bits) << 8;
        w.push_back(pass_flags);
        w.push_back(0);
        for (uint32_t j = 0; j < kMaxTileBits; ++j) w.push_back(j < pass.pos.size() ? uint32_t(pass.pos[j]) : kPosPad);
        auto tile_bit = [&](int q) {
            auto it = std::lower_bound(pass.pos.begin(), pass.pos.end(), q);
            return (it != pass.pos.end() && *it == q) ? int(it - pass.pos.begin()) : -1;
        };

        // Layouts.  Every layout keeps `nw` tile bits on the wave-index thread bits (the wave set W).  As long as
        // consecutive rounds can keep the same W (none of its bits is needed in registers) the exchange between
        // them moves data only inside each wave and needs no barrier, so W is chosen like a cache victim: the bits
        // whose next turn in registers is farthest away; ties go to controls of the round's gates (a wave whose
        // control bit is 0 skips the gate), then to high bits.  The first and the last layout also face global
        // memory: the lowest `cl` tile bits stay on the lowest lanes there, so they never enter W or the registers.
        const size_t n_rounds_pass = pass.rounds.size();
        const int nw = t > 6 ? t - 6 : 0;
        std::vector<std::vector<int>> needed(n_rounds_pass);  // tile bits targeted by the round's gates
        for (size_t m = 0; m < n_rounds_pass; ++m)
            for (int gi : pass.rounds[m].gates) {
                const int tb = tile_bit(gates[gi].target);
                if (std::find(needed[m].begin(), needed[m].end(), tb) == needed[m].end()) needed[m].push_back(tb);
            }
        auto next_use = [&](int b, size_t m) {
            for (size_t m2 = m + 1; m2 < n_rounds_pass; ++m2)
                if (std::find(needed[m2].begin(), needed[m2].end(), b) != needed[m2].end()) return int(m2);
            return int(n_rounds_pass) + 1;
        };
        auto contains = [](const std::vector<int>& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
        std::vector<Layout> layouts;
        std::vector<char> intra(n_rounds_pass, 0);
        std::vector<std::vector<int>> wave_sets(n_rounds_pass);
        std::vector<std::vector<std::pair<int, int>>> swaps(n_rounds_pass);  // (register bit, lane bit) transpositions
        std::vector<char> by_swap(n_rounds_pass, 0);                          // the round's relayout needs no LDS
        std::vector<int> prev_w;
        // does any later round of this pass target tile bit b?
        auto targeted_later = [&](int b, size_t m) { return next_use(b, m) <= int(n_rounds_pass); };
        for (size_t m = 0; m < n_rounds_pass; ++m) {
            RoundPlan& rd = pass.rounds[m];
            const bool edge = m == 0 || m + 1 == n_rounds_pass;
            std::vector<int> must = needed[m];
            if (must.empty() && rd.gates.empty()) must = {};  // relayout-only round: registers are all filler
            // controls of the round's gates that are tile bits but not targets in this round
            std::vector<int> ctrl;
            for (int gi : rd.gates) {
                const int cb = gates[gi].control >= 0 ? tile_bit(gates[gi].control) : -1;
                if (cb >= 0 && !contains(must, cb) && !contains(ctrl, cb)) ctrl.push_back(cb);
            }
            // SWAP relayout (plan.hpp): every target of the round already sits in a register or on one of the lane
            // bits a swap can reach; the last layout of a pass must also leave the low tile bits on the low lanes
            // (swaps never move those, so it is enough that the previous layout has them there).
            if (m > 0 && cfg.swaps) {
                const Layout& cur = layouts.back();
                bool ok = true;
                std::vector<std::pair<int, int>> want;  // (tile bit, lane bit it sits on)
                for (int b : must) {
                    if (contains(cur.reg, b)) continue;
                    int u = -1;
                    for (size_t i = 0; i < cur.thr.size(); ++i)
                        if (cur.thr[i] == b) u = int(i);
                    if (u < kSwapLaneLo || u >= kSwapLaneHi || u >= t) ok = false;
                    want.push_back({b, u});
                }
                if (ok && want.size() <= kMaxSwaps) {
                    Layout nl = cur;
                    std::vector<char> taken(nl.reg.size(), 0);
                    for (const auto& wb : want) {
                        // victim: a register whose bit this round does not target; rather not a control of the
                        // round's gates (a register-held control halves a gate's work), then the bit needed latest
                        int best = -1;
                        for (size_t v = 0; v < nl.reg.size(); ++v) {
                            if (taken[v] || contains(must, nl.reg[v])) continue;
                            if (best < 0) {
                                best = int(v);
                                continue;
                            }
                            const bool cv = contains(ctrl, nl.reg[v]), cb = contains(ctrl, nl.reg[size_t(best)]);
                            if (cv != cb) {
                                if (!cv) best = int(v);
                                continue;
                            }
                            if (next_use(nl.reg[v], m) > next_use(nl.reg[size_t(best)], m)) best = int(v);
                        }
                        if (best < 0) {
                            ok = false;
                            break;
                        }
                        taken[size_t(best)] = 1;
                        swaps[m].push_back({best, wb.second});
                        std::swap(nl.reg[size_t(best)], nl.thr[size_t(wb.second)]);
                    }
                    if (ok) {
                        rd.regbits = nl.reg;
                        layouts.push_back(nl);
                        by_swap[m] = 1;
                        wave_sets[m] = prev_w;
                        continue;
                    }
                    swaps[m].clear();
                }
            }
            // Spare registers go to controls first: a register-held control halves the gate's work, whereas a
            // control on a lane bit only masks lanes (the full butterfly still issues).
            std::vector<int> regs = must;
            for (int cb : ctrl)
                if (int(regs.size()) < r && !(edge && cb < cl)) regs.push_back(cb);
            // Wave set: controls not in registers come first (whole waves skip the gate), then the previous wave
            // bits (an unchanged set makes the exchange barrier-free), then the bits needed in registers latest.
            std::vector<int> wset;
            if (nw > 0) {
                std::vector<int> cand;
                for (int b = cl; b < k; ++b)
                    if (!contains(regs, b)) cand.push_back(b);
                std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) {
                    if (cfg.swaps) {
                        // wave-index bits are the only ones a swap cannot reach: bits no later round targets go first
                        const bool lx = targeted_later(x, m), ly = targeted_later(y, m);
                        if (lx != ly) return !lx;
                    }
                    const bool cx = contains(ctrl, x), cy = contains(ctrl, y);
                    if (cx != cy) return cx;
                    const bool px = contains(prev_w, x), py = contains(prev_w, y);
                    if (px != py) return px;
                    const int ux = next_use(x, m), uy = next_use(y, m);
                    if (ux != uy) return ux > uy;
                    return x > y;
                });
                // leave enough bits outside W for the registers
                const int spare = int(cand.size()) - (r - int(regs.size()));
                const int take = std::max(0, std::min(nw, spare));
                wset.assign(cand.begin(), cand.begin() + take);
                std::sort(wset.begin(), wset.end());
            }
            // remaining registers: filled up from the top with bits that are neither in W nor pinned to lanes
            for (int b = k - 1; b >= 0 && int(regs.size()) < r; --b)
                if (!contains(regs, b) && !contains(wset, b) && !(edge && b < cl)) regs.push_back(b);
            for (int b = k - 1; b >= 0 && int(regs.size()) < r; --b)  // tiny tiles: give up wave bits, then low bits
                if (!contains(regs, b) && !(edge && b < cl)) {
                    regs.push_back(b);
                    wset.erase(std::remove(wset.begin(), wset.end(), b), wset.end());
                }
            for (int b = k - 1; b >= 0 && int(regs.size()) < r; --b)
                if (!contains(regs, b)) regs.push_back(b);
            std::sort(regs.begin(), regs.end());
            rd.regbits = regs;
            layouts.push_back(make_layout(k, regs, wset));
            // A pass whose first layout faces nothing (cl = 0: a synthesised one-tile side) may seat its lane bits anywhere: a
            // transposition with lane bit 4 or 5 is 32 vector instructions at 16 amplitudes per thread, with bit 2 or 3 64, with
            // bit 0 or 1 128 -- the qubits the rounds will want soonest go to the cheap end, those nobody targets to lanes 0, 1.
            if (m == 0 && cl == 0 && cfg.swaps) {
                Layout& l0 = layouts.back();
                const size_t n_lanes = std::min<size_t>(6, l0.thr.size() - std::min(l0.thr.size(), wset.size()));
                std::stable_sort(l0.thr.begin(), l0.thr.begin() + long(n_lanes),
                                 [&](int x, int y) { return next_use(x, 0) > next_use(y, 0); });
            }
            // a workgroup of one wave (t <= 6) never needs a barrier
            intra[m] = m > 0 && (nw == 0 || (int(wset.size()) == nw && wset == prev_w));
            wave_sets[m] = wset;
            prev_w = wset;
        }
        // The last layout faces global memory again: the low `cl` tile bits must be back on the low lanes.  Swaps may
        // have taken them away (a gate targeted one of them): a final gate-less round of swaps brings them home --
        // from a register in one transposition, from another lane in two (through any register).
        if (cfg.swaps) {
            Layout fin = layouts.back();
            std::vector<std::pair<int, int>> fix;
            for (int u = 0; u < cl && u < int(fin.thr.size()); ++u) {
                if (fin.thr[size_t(u)] == u) continue;
                auto in_reg = std::find(fin.reg.begin(), fin.reg.end(), u);
                if (in_reg == fin.reg.end()) {
                    size_t u2 = 0;
                    while (fin.thr[u2] != u) ++u2;
                    if (int(u2) >= 6) throw std::logic_error("a low tile bit ended on a wave-index bit");
                    size_t v = 0;  // any register that does not hold another low bit waiting to go home
                    while (v + 1 < fin.reg.size() && fin.reg[v] < cl) ++v;
                    fix.push_back({int(v), int(u2)});
                    std::swap(fin.reg[v], fin.thr[u2]);
                    in_reg = fin.reg.begin() + long(v);
                }
                fix.push_back({int(in_reg - fin.reg.begin()), u});
                std::swap(*in_reg, fin.thr[size_t(u)]);
            }
            if (!fix.empty()) {
                // (more than a round's worth of transpositions -- three low lane bits of single-precision plans can need six --
                // take several gate-less rounds, each with the layout its own swaps leave)
                Layout cur = layouts.back();
                for (size_t at = 0; at < fix.size(); at += kMaxSwaps) {
                    const size_t end = std::min(fix.size(), at + size_t(kMaxSwaps));
                    std::vector<std::pair<int, int>> part(fix.begin() + long(at), fix.begin() + long(end));
                    for (const auto& vu : part) std::swap(cur.reg[size_t(vu.first)], cur.thr[size_t(vu.second)]);
                    pass.rounds.emplace_back();
                    pass.rounds.back().regbits = cur.reg;
                    layouts.push_back(cur);
                    swaps.push_back(part);
                    by_swap.push_back(1);
                    intra.push_back(0);
                    wave_sets.push_back(prev_w);
                    needed.emplace_back();
                }
                w[pass_header_at] = uint32_t(k) | uint32_t(r) << 8 | uint32_t(t) << 16 | uint32_t(pass.rounds.size()) << 24;
            }
        }
        // one layout's columns in the fixed shape: kMaxThreadBits thread columns, kMaxRegBits register columns
        auto push_cols = [&](const Layout& l, auto&& col) {
            for (uint32_t u = 0; u < kMaxThreadBits; ++u) w.push_back(u < l.thr.size() ? col(l.thr[u]) : 0u);
            for (uint32_t v = 0; v < kMaxRegBits; ++v) w.push_back(v < l.reg.size() ? col(l.reg[v]) : 0u);
        };
        auto push_global_cols = [&](const Layout& l) { push_cols(l, [&](int b) { return 1u << pass.pos[b]; }); };
        push_global_cols(layouts.front());
        const bool cstore = compact_bits >= 0 && pi == 0, cload = compact_bits >= 0 && pi == 1;
        if (cstore)
            push_cols(layouts.back(), [&](int b) { return 1u << b; });  // offsets inside the pattern's own tile
        else
            push_global_cols(layouts.back());
        {
            // compact block (fixed size, zero unless used)
            const PassPlan& p0 = passes[0];
            auto rank_in = [](const std::vector<int>& v, int q) {
                auto it = std::lower_bound(v.begin(), v.end(), q);
                return (it != v.end() && *it == q) ? int(it - v.begin()) : -1;
            };
            std::vector<int> outer0;  // pass 0's outer qubits, ascending: bit j of a pass-0 tile number
            for (int q = 0; q < n; ++q)
                if (rank_in(p0.pos, q) < 0) outer0.push_back(q);
            auto wcol = [&](int q) {  // W index = control pattern * 2^k + index inside pass 0's tile
                uint32_t c = 0;
                if (rank_in(p0.pos, q) >= 0) c |= 1u << rank_in(p0.pos, q);
                if (rank_in(compact_ctrl, q) >= 0) c |= 1u << (k + rank_in(compact_ctrl, q));
                return c;
            };
            auto fcol = [&](int q) { return rank_in(outer0, q) >= 0 ? 1u << rank_in(outer0, q) : 0u; };
            for (uint32_t j = 0; j < kMaxCompactBits; ++j)
                w.push_back(cstore && j < compact_ctrl.size() ? uint32_t(compact_ctrl[j]) : 63u);
            if (cload) {
                push_cols(layouts.front(), [&](int b) { return wcol(pass.pos[b]); });
                push_cols(layouts.front(), [&](int b) { return fcol(pass.pos[b]); });
                std::vector<int> outer_here;  // this pass's outer qubits: bit j of ITS tile number
                for (int q = 0; q < n; ++q)
                    if (rank_in(pass.pos, q) < 0) outer_here.push_back(q);
                for (uint32_t j = 0; j < kMaxOuterBits; ++j) w.push_back(j < outer_here.size() ? wcol(outer_here[j]) : 0u);
                for (uint32_t j = 0; j < kMaxOuterBits; ++j) w.push_back(j < outer_here.size() ? fcol(outer_here[j]) : 0u);
            } else {
                for (uint32_t j = 0; j < 2 * kColumnWords + 2 * kMaxOuterBits; ++j) w.push_back(0u);
            }
        }

        const double pass_tiles = cstore ? double(uint64_t(1) << compact_bits) : double(uint64_t(1) << (n - k));
        double pairs_this_pass = 0.0;
        for (size_t m = 0; m < pass.rounds.size(); ++m) {
            const RoundPlan& rd = pass.rounds[m];
            const Layout& lay = layouts[m];
            const bool swap_round = by_swap[m] && !swaps[m].empty();
            const bool exch = m > 0 && !by_swap[m];
            size_t n_entries = 0;
            for (int gi : rd.gates) n_entries += chain0[size_t(gi)].empty() ? 1 : 2;
            w.push_back(uint32_t(n_entries) | (exch ? 1u << 16 : 0u) | (exch && intra[m] ? 1u << 17 : 0u) |
                        (swap_round ? 1u << 18 : 0u));
            if (swap_round) {
                for (uint32_t i = 0; i < kMaxSwaps; ++i)
                    w.push_back(i < swaps[m].size() ? uint32_t(swaps[m][i].first) | uint32_t(swaps[m][i].second) << 8 : kSwapPad);
                out.stats.n_swap_rounds += 1;
                out.stats.n_swaps += int(swaps[m].size());
            }
            if (exch) {
                const SwizzleChoice sw =
                    choose_swizzle(layouts[m - 1], lay, k, cfg.elem_bytes, intra[m] ? wave_sets[m] : std::vector<int>{});
                out.stats.n_intra_wave_exchanges += intra[m] ? 1 : 0;
                out.stats.lds_conflict_cycles += sw.cost;
                out.stats.n_exchanges += 1;
                push_cols(layouts[m - 1], [&](int b) { return lds_col(b, sw.s); });
                push_cols(lay, [&](int b) { return lds_col(b, sw.s); });
            }
            std::vector<int> reg_index_of(k, -1), thr_index_of(k, -1);
            for (size_t v = 0; v < lay.reg.size(); ++v) reg_index_of[lay.reg[v]] = int(v);
            for (size_t u = 0; u < lay.thr.size(); ++u) thr_index_of[lay.thr[u]] = int(u);
            for (int gi : rd.gates) {
                const GateIn& g = gates[gi];
                const int tb = tile_bit(g.target);
                uint32_t creg = 0xFF, ct = 0, cg = 0;
                if (g.control >= 0) {
                    const int cb = tile_bit(g.control);
                    if (cb < 0)
                        cg = 1u << g.control;
                    else if (reg_index_of[cb] >= 0)
                        creg = uint32_t(reg_index_of[cb]);
                    else
                        ct = 1u << thr_index_of[cb];
                }
                // bit 16 + p: the p-th amplitude pair (register indices with the target bit clear, ascending) takes
                // part; a register-held control switches off the pairs whose control bit is 0
                const int jbit = reg_index_of[tb];
                // a multiplexed gate (plan.hpp FUSION) takes two entries: where the control is 1, then where it is 0
                for (int which = 1; which >= (chain0[size_t(gi)].empty() ? 1 : 0); --which) {
                    uint32_t pair_mask = 0;
                    for (int e0 = 0, pr = 0; e0 < (1 << r); ++e0) {
                        if ((e0 >> jbit) & 1) continue;
                        if (creg == 0xFF || int((e0 >> creg) & 1) == which) pair_mask |= 1u << pr;
                        ++pr;
                    }
                    const size_t factors = which ? chain1[size_t(gi)].size() : chain0[size_t(gi)].size();
                    // the control-is-0 entry lists its control among the COMPLEMENTED bits: thread bits 9 .. 17 of the
                    // extended thread index (tid, ~tid), word [3] for the global index
                    w.push_back(uint32_t(jbit) | creg << 8 | pair_mask << 16 | (factors > 1 ? kGateGeneral : 0u) |
                                (which ? 0u : kGateNegated));
                    w.push_back(which ? ct : ct << kMaxThreadBits);
                    w.push_back(which ? cg : 0u);
                    w.push_back(which ? 0u : cg);
                    schedule.push_back({gi, which});
                    // (a product's butterfly is 16 operations where a plain one is 14)
                    pairs_this_pass += pass_tiles * double(uint64_t(1) << (k - 1)) * (g.control >= 0 ? 0.5 : 1.0) *
                                       (factors > 1 ? 16.0 / 14.0 : 1.0);
                }
            }
            out.stats.n_rounds += 1;
        }
        out.stats.pass_pairs.push_back(pairs_this_pass);
    }
    // angle table: scheduled gates first, then the fold entries; fold index per qubit
    w[1] = uint32_t(schedule.size());
    w[3] = uint32_t(w.size());
    std::vector<uint32_t> chain_index;
    uint32_t entry = 0;
    for (const auto& se : schedule) {
        const std::vector<int>& chain = se.second ? chain1[size_t(se.first)] : chain0[size_t(se.first)];
        chain_index.push_back(entry | uint32_t(chain.size()) << 24);
        for (int op : chain) push_angle_entry(w, op_angles[size_t(op)]);
        entry += uint32_t(chain.size());
    }
    if (entry >= (1u << 24)) throw std::invalid_argument("too many gates");
    const uint32_t n_factors = entry;
    std::vector<std::pair<uint32_t, uint32_t>> fold_index(size_t(n), {0u, 0u});
    for (int q = 0; q < n; ++q) {
        fold_index[size_t(q)] = {entry, uint32_t(folds[size_t(q)].size())};
        for (int op : folds[size_t(q)]) {
            push_angle_entry(w, op_angles[size_t(op)]);
            ++entry;
        }
    }
    w[4] = uint32_t(w.size());
    w[5] = entry - n_factors;
    for (const auto& fi : fold_index) {
        w.push_back(fi.first);
        w.push_back(fi.second);
    }
    w[6] = uint32_t(w.size());
    w[7] = n_factors;
    w.insert(w.end(), chain_index.begin(), chain_index.end());
    w.resize(w.size() + kPlanPadWords, 0);
    out.stats.n_passes = int(passes.size());
    out.stats.n_real_gates = int(schedule.size());
    return out;
}

}  // namespace qsv
