// Register splitting (split.hpp): partition search and construction of the two virtual circuits.  Pure C++.
#include "split.hpp"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <functional>

namespace qsv {

namespace {

struct Key {
    int control;
    std::vector<int> targets;  // distinct targets of the key's gates
};

struct UnionFind {
    std::array<int8_t, 64> parent;
    explicit UnionFind(int n) {
        for (int i = 0; i < n; ++i) parent[size_t(i)] = int8_t(i);
    }
    int find(int x) {
        while (parent[size_t(x)] != x) {
            parent[size_t(x)] = parent[size_t(parent[size_t(x)])];
            x = parent[size_t(x)];
        }
        return x;
    }
    void join(int a, int b) { parent[size_t(find(a))] = int8_t(find(b)); }
};

}  // namespace

SplitCircuits find_split(int n, const std::vector<GateIn>& all_gates, const std::vector<AngleSource>& op_angles,
                         int max_side, int max_keys) {
    return find_split(n, all_gates, op_angles, std::vector<int>{max_side}, max_keys);
}

SplitCircuits find_split(int n, const std::vector<GateIn>& all_gates, const std::vector<AngleSource>& op_angles,
                         const std::vector<int>& max_sides, int max_keys) {
    // which of the size limits can apply at all
    std::vector<char> open_stage(max_sides.size(), 0);
    bool any_open = false;
    for (size_t s = 0; s < max_sides.size(); ++s) {
        open_stage[s] = !(n > 60 || n <= max_sides[s] || 2 * max_sides[s] < n);
        any_open |= open_stage[s] != 0;
    }
    if (!any_open) return SplitCircuits{};

    // ---- which gates act at all, and the keys -------------------------------------------------------------------
    // (the drop rule of build_plan: a cu3 whose control nobody has targeted yet acts on |0> and is the identity)
    std::vector<char> touched(size_t(n), 0);
    std::vector<int> epoch(size_t(n), 0);           // gates that have targeted the qubit so far
    std::vector<int> key_of(all_gates.size(), -1);  // per gate: its key (cu3 that act), -1 otherwise
    std::vector<char> dropped(all_gates.size(), 0);
    std::vector<Key> keys;
    std::vector<std::pair<int, int>> key_id;  // (control, epoch) of keys[i]
    for (size_t i = 0; i < all_gates.size(); ++i) {
        const GateIn& g = all_gates[i];
        if (g.control >= 0) {
            if (!touched[size_t(g.control)]) {
                dropped[i] = 1;
                continue;
            }
            const std::pair<int, int> id{g.control, epoch[size_t(g.control)]};
            int ki = -1;
            for (size_t j = keys.size(); j-- > 0;)
                if (key_id[j] == id) {
                    ki = int(j);
                    break;
                }
            if (ki < 0) {
                ki = int(keys.size());
                keys.push_back(Key{g.control, {}});
                key_id.push_back(id);
            }
            key_of[i] = ki;
            std::vector<int>& ts = keys[size_t(ki)].targets;
            if (std::find(ts.begin(), ts.end(), g.target) == ts.end()) ts.push_back(g.target);
        }
        touched[size_t(g.target)] = 1;
        epoch[size_t(g.target)] += 1;
    }
    const int nk = int(keys.size());

    // ---- partition: the fewest cut keys such that both virtual circuits fit a tile --------------------------------
    // Remove a set R of keys (|R| = 0, 1, 2, ..), take the connected components of what is left and pack them into
    // two bins (subset sum over the component sizes, as balanced as the size limit allows).
    // One enumeration serves every size limit: a trial's components are packed under each limit that has no partition
    // yet, and the enumeration stops at the first partition for the FIRST (smallest) limit -- the same partition per
    // limit as one search per limit would find, at a third of the cost for a circuit that has none.
    std::vector<uint64_t> side_of_stage(max_sides.size(), 0);
    std::vector<char> hit(max_sides.size(), 0);
    size_t first_open = 0;
    while (!open_stage[first_open]) ++first_open;
    bool found = false;
    std::vector<int> without;  // the keys removed in this trial, ascending
    auto try_without = [&]() {
        UnionFind uf(n);
        for (int j = 0; j < nk; ++j) {
            if (std::find(without.begin(), without.end(), j) != without.end()) continue;
            for (int t : keys[size_t(j)].targets) uf.join(keys[size_t(j)].control, t);
        }
        // components in order of their lowest qubit
        std::array<int8_t, 64> comp_of{};
        std::vector<int> size;
        std::vector<uint64_t> members;
        for (int q = 0; q < n; ++q) {
            const int root = uf.find(q);
            if (root == q) {
                comp_of[size_t(q)] = int8_t(size.size());
                size.push_back(0);
                members.push_back(0);
            }
        }
        // (roots are not necessarily the lowest member: number them in a second sweep)
        for (int q = 0; q < n; ++q) {
            const int c = comp_of[size_t(uf.find(q))];
            size[size_t(c)] += 1;
            members[size_t(c)] |= uint64_t(1) << q;
        }
        const int removed = int(without.size());
        // subset sums: choice[s] = set of components (bit mask) with total size s, component 0 always in A
        std::vector<uint64_t> choice(size_t(n) + 1, 0);
        std::vector<char> reach(size_t(n) + 1, 0);
        reach[size_t(size[0])] = 1;
        choice[size_t(size[0])] = 1;
        for (size_t c = 1; c < size.size(); ++c)
            for (int s = n - size[c]; s >= 0; --s)
                if (reach[size_t(s)] && !reach[size_t(s + size[c])]) {
                    reach[size_t(s + size[c])] = 1;
                    choice[size_t(s + size[c])] = choice[size_t(s)] | uint64_t(1) << c;
                }
        for (size_t stage = 0; stage < max_sides.size(); ++stage) {
            if (!open_stage[stage] || hit[stage]) continue;
            const int lo = n - max_sides[stage] + removed, hi = max_sides[stage] - removed;  // admissible |A|
            int best = -1;
            for (int s = lo; s <= hi; ++s)
                if (s >= 0 && s <= n && reach[size_t(s)] && (best < 0 || std::abs(2 * s - n) < std::abs(2 * best - n))) best = s;
            if (best < 0) continue;
            uint64_t side_a = 0;
            for (size_t c = 0; c < size.size(); ++c)
                if (choice[size_t(best)] >> c & 1u) side_a |= members[c];
            side_of_stage[stage] = side_a;
            hit[stage] = 1;
        }
        return hit[first_open] != 0;
    };
    // Sets of 0, 1, 2, .. keys, each size in lexicographic order: the first hit has the fewest keys.  The LAST key of a
    // set is only worth trying if its removal splits a component of what the others leave (an articulation point of the
    // qubit / key incidence graph): otherwise the components are those of the smaller set, which has already failed under
    // looser size bounds.  That filter is exact and turns the C(keys, 3) trials of a deep circuit (10 - 40 ms per
    // registration at eight to twelve layers, all in vain) into C(keys, 2) linear-time sweeps with hardly a trial.
    std::vector<std::vector<int>> keys_of_qubit(static_cast<size_t>(n));
    for (int j = 0; j < nk; ++j) {
        keys_of_qubit[size_t(keys[size_t(j)].control)].push_back(j);
        for (int t : keys[size_t(j)].targets) keys_of_qubit[size_t(t)].push_back(j);
    }
    const int n_nodes = n + nk;  // qubits, then keys
    std::vector<int> disc(static_cast<size_t>(n_nodes)), low(static_cast<size_t>(n_nodes)), parent(static_cast<size_t>(n_nodes));
    std::vector<int> next_edge(static_cast<size_t>(n_nodes)), stack;
    std::vector<char> removed(static_cast<size_t>(nk), 0), articulation(static_cast<size_t>(nk), 0);
    auto neighbour = [&](int u, int i) -> int {  // i-th neighbour of node u, -1 past the end
        if (u < n) return i < int(keys_of_qubit[size_t(u)].size()) ? n + keys_of_qubit[size_t(u)][size_t(i)] : -1;
        const Key& k = keys[size_t(u - n)];
        if (i == 0) return k.control;
        return i <= int(k.targets.size()) ? k.targets[size_t(i - 1)] : -1;
    };
    // keys (not removed) whose removal disconnects their component, ascending
    auto splitting_keys = [&](std::vector<int>& result) {
        result.clear();
        std::fill(disc.begin(), disc.end(), -1);
        std::fill(articulation.begin(), articulation.end(), 0);
        int clock = 0;
        for (int root = 0; root < n; ++root) {
            if (disc[size_t(root)] >= 0) continue;
            int root_children = 0;
            disc[size_t(root)] = low[size_t(root)] = clock++;
            parent[size_t(root)] = -1;
            next_edge[size_t(root)] = 0;
            stack.assign(1, root);
            while (!stack.empty()) {
                const int u = stack.back();
                const int v = neighbour(u, next_edge[size_t(u)]++);
                if (v < 0) {
                    stack.pop_back();
                    const int p = parent[size_t(u)];
                    if (p >= 0) {
                        low[size_t(p)] = std::min(low[size_t(p)], low[size_t(u)]);
                        if (p != root && p >= n && low[size_t(u)] >= disc[size_t(p)]) articulation[size_t(p - n)] = 1;
                        if (p == root) ++root_children;
                    }
                    continue;
                }
                if (v >= n && removed[size_t(v - n)]) continue;
                if (disc[size_t(v)] < 0) {
                    disc[size_t(v)] = low[size_t(v)] = clock++;
                    parent[size_t(v)] = u;
                    next_edge[size_t(v)] = 0;
                    stack.push_back(v);
                } else if (v != parent[size_t(u)]) {
                    low[size_t(u)] = std::min(low[size_t(u)], disc[size_t(v)]);
                }
            }
            (void)root_children;  // (roots are qubits: only key nodes are asked about)
        }
        for (int j = 0; j < nk; ++j)
            if (articulation[size_t(j)]) result.push_back(j);
    };
    std::vector<int> last_candidates;
    // (a bound on the sweeps, for circuits with very many keys: C(70, 2) sweeps are 8 ms, and such circuits never split)
    constexpr int kMaxSweeps = 2000;
    int sweeps = 0;
    std::function<bool(int, int)> choose = [&](int start, int left) {
        if (left == 0) return try_without();
        if (left == 1) {
            if (++sweeps > kMaxSweeps) return false;
            for (int r : without) removed[size_t(r)] = 1;
            splitting_keys(last_candidates);
            for (int r : without) removed[size_t(r)] = 0;
            const std::vector<int> candidates = last_candidates;
            for (int r : candidates) {
                if (r < start) continue;
                without.push_back(r);
                if (try_without()) return true;
                without.pop_back();
            }
            return false;
        }
        for (int r = start; r + left <= nk && sweeps <= kMaxSweeps; ++r) {
            without.push_back(r);
            if (choose(r + 1, left - 1)) return true;
            without.pop_back();
        }
        return false;
    };
    // Up to three keys by enumeration (above); four and five by branch and bound over the QUBITS (below): C(keys, 4) key
    // sets are tens of thousands of sweeps for the circuits that need them, whereas few partial assignments of a well
    // connected register cut at most five keys.
    for (int size = 0; !found && size <= std::min(max_keys, 3) && size <= nk; ++size) {
        without.clear();
        found = choose(0, size);
    }
    bool any_hit = false;
    for (size_t stage = 0; stage < max_sides.size(); ++stage) any_hit |= hit[stage] != 0;
    if (!any_hit && max_keys > 3 && nk > 3) {
        // Assign the qubits that take part in keys one by one (breadth first from the busiest one, so that keys are
        // decided early); a key is cut once it has members on both sides; a branch dies when it cuts more keys than the
        // bound or a side no longer fits the largest limit.  Bounds 4, then 5: the first leaf found has the fewest keys
        // beyond three.  Qubits without keys go wherever the sizes need them.  Exact up to the node budget.
        std::vector<int> order;
        {
            std::vector<char> seen(size_t(n), 0);
            std::vector<int> degree(size_t(n), 0);
            for (int q = 0; q < n; ++q) degree[size_t(q)] = int(keys_of_qubit[size_t(q)].size());
            for (;;) {
                int start = -1;
                for (int q = 0; q < n; ++q)
                    if (!seen[size_t(q)] && degree[size_t(q)] > 0 && (start < 0 || degree[size_t(q)] > degree[size_t(start)])) start = q;
                if (start < 0) break;
                size_t head = order.size();
                order.push_back(start);
                seen[size_t(start)] = 1;
                while (head < order.size()) {
                    const int u = order[head++];
                    for (int j : keys_of_qubit[size_t(u)]) {
                        const Key& k = keys[size_t(j)];
                        auto visit = [&](int v) {
                            if (!seen[size_t(v)]) {
                                seen[size_t(v)] = 1;
                                order.push_back(v);
                            }
                        };
                        visit(k.control);
                        for (int t : k.targets) visit(t);
                    }
                }
            }
        }
        const int n_active = int(order.size()), n_free = n - n_active;
        int largest = 0;
        for (size_t stage = 0; stage < max_sides.size(); ++stage)
            if (open_stage[stage]) largest = std::max(largest, max_sides[stage]);
        std::vector<int> side(size_t(n), -1), on_a(size_t(nk), 0), on_b(size_t(nk), 0);
        int count[2] = {0, 0}, cut = 0;
        long nodes = 0;
        constexpr long kNodeBudget = 200000;
        int bound = 0;
        bool done = false;
        std::function<void(int)> descend = [&](int depth) {
            if (done || ++nodes > kNodeBudget) return;
            if (depth == n_active) {
                // the free qubits fill the sides up: most balanced split the sizes allow, per stage
                for (size_t stage = 0; stage < max_sides.size(); ++stage) {
                    if (!open_stage[stage] || hit[stage]) continue;
                    const int room = max_sides[stage] - cut;  // own qubits a side may have
                    if (count[0] > room || count[1] > room || n > 2 * room) continue;
                    int to_a = std::max(0, std::min(n_free, n / 2 - count[0]));
                    to_a = std::max(to_a, n_free - (room - count[1]));
                    to_a = std::min(to_a, room - count[0]);
                    uint64_t side_a = 0;
                    int given = 0;
                    for (int q = 0; q < n; ++q) {
                        if (side[size_t(q)] == 0 || (side[size_t(q)] < 0 && given < to_a)) {
                            side_a |= uint64_t(1) << q;
                            given += side[size_t(q)] < 0;
                        }
                    }
                    side_of_stage[stage] = side_a;
                    hit[stage] = 1;
                }
                // (a smaller limit cannot have a partition with this many keys or more: n / 2 + keys is what a side needs)
                bool more = false;
                for (size_t stage = 0; stage < max_sides.size(); ++stage)
                    more |= open_stage[stage] && !hit[stage] && 2 * (max_sides[stage] - bound) >= n;
                done = !more;
                return;
            }
            const int v = order[size_t(depth)];
            const int first_side = (depth == 0 || count[0] <= count[1]) ? 0 : 1;
            for (int turn = 0; turn < 2 && !done; ++turn) {
                const int s2 = turn == 0 ? first_side : 1 - first_side;
                if (depth == 0 && s2 == 1) break;  // (the mirror image)
                int newly_cut = 0;
                for (int j : keys_of_qubit[size_t(v)]) {
                    int& mine = s2 == 0 ? on_a[size_t(j)] : on_b[size_t(j)];
                    const int other = s2 == 0 ? on_b[size_t(j)] : on_a[size_t(j)];
                    if (mine == 0 && other > 0) ++newly_cut;
                    ++mine;
                }
                cut += newly_cut;
                count[s2] += 1;
                side[size_t(v)] = s2;
                if (cut <= bound && count[0] + cut <= largest && count[1] + cut <= largest) descend(depth + 1);
                side[size_t(v)] = -1;
                count[s2] -= 1;
                cut -= newly_cut;
                for (int j : keys_of_qubit[size_t(v)]) --(s2 == 0 ? on_a[size_t(j)] : on_b[size_t(j)]);
            }
        };
        // (a qubit is listed once per key it belongs to: a key that names it as control AND target cannot exist)
        for (bound = 4; bound <= max_keys && !done; ++bound) {
            bool possible = false;
            for (size_t stage = 0; stage < max_sides.size(); ++stage) possible |= open_stage[stage] && 2 * (max_sides[stage] - bound) >= n;
            if (!possible) break;
            nodes = 0;
            descend(0);
            bool got = false;
            for (size_t stage = 0; stage < max_sides.size(); ++stage) got |= hit[stage] != 0;
            if (got) break;
        }
    }
    (void)found;
    for (size_t stage = 0; stage < max_sides.size(); ++stage) {
    if (!hit[stage]) continue;
    const uint64_t side_a = side_of_stage[stage];
    const int max_side = max_sides[stage];
    SplitCircuits out;

    // ---- the keys this partition really cuts ----------------------------------------------------------------------
    auto side_of = [&](int q) { return int(!(side_a >> q & 1u)); };  // 0 = A, 1 = B
    std::vector<int> cut_index(size_t(nk), -1);
    int n_cut = 0;
    for (int j = 0; j < nk; ++j) {
        bool cut = false;
        for (int t : keys[size_t(j)].targets) cut |= side_of(t) != side_of(keys[size_t(j)].control);
        if (cut) cut_index[size_t(j)] = n_cut++;
    }
    out.mask[0] = side_a;
    out.mask[1] = ~side_a & ((uint64_t(1) << n) - 1);
    for (int s = 0; s < 2; ++s) out.n_side[s] = __builtin_popcountll(out.mask[s]);
    if (n_cut > max_keys || out.n_side[0] + n_cut > max_side || out.n_side[1] + n_cut > max_side) continue;
    out.n_keys = n_cut;

    // ---- the two virtual circuits -------------------------------------------------------------------------------------
    std::vector<int> local(size_t(n), 0);
    {
        int ca = 0, cb = 0;
        for (int q = 0; q < n; ++q) local[size_t(q)] = side_of(q) == 0 ? ca++ : cb++;
    }
    auto fixed = [](int32_t code) { return AngleSource{code, -1, -1, 0.0, 0.0, 0.0}; };
    for (int s = 0; s < 2; ++s) {
        std::vector<GateIn>& vg = out.gates[s];
        std::vector<AngleSource>& va = out.angles[s];
        for (int j = 0; j < n_cut; ++j) {  // every key qubit starts as (1, 1)
            va.push_back(fixed(kFixedOnes));
            vg.push_back(GateIn{out.n_side[s] + j, -1, int(va.size()) - 1});
        }
    }
    std::vector<char> projected(size_t(nk), 0);
    for (size_t i = 0; i < all_gates.size(); ++i) {
        if (dropped[i]) continue;
        const GateIn& g = all_gates[i];
        const int st = side_of(g.target);
        const int key = key_of[i] >= 0 ? cut_index[size_t(key_of[i])] : -1;
        if (g.control < 0 || side_of(g.control) == st) {
            // a gate of one side -- but the first use of a cut key projects its control first, also when this
            // particular gate of the key stays inside the side
            if (key >= 0 && !projected[size_t(key_of[i])]) {
                projected[size_t(key_of[i])] = 1;
                std::vector<GateIn>& vg = out.gates[st];
                std::vector<AngleSource>& va = out.angles[st];
                const int c = local[size_t(g.control)], kq = out.n_side[st] + key;
                va.push_back(fixed(kFixedX));
                vg.push_back(GateIn{c, kq, int(va.size()) - 1});
                va.push_back(fixed(kFixedProj0));
                vg.push_back(GateIn{c, -1, int(va.size()) - 1});
                va.push_back(fixed(kFixedX));
                vg.push_back(GateIn{c, kq, int(va.size()) - 1});
            }
            out.angles[st].push_back(op_angles[size_t(g.op)]);
            out.gates[st].push_back(GateIn{local[size_t(g.target)], g.control < 0 ? -1 : local[size_t(g.control)],
                                           int(out.angles[st].size()) - 1});
            continue;
        }
        // cross gate: control on the other side
        const int sc = 1 - st;
        if (!projected[size_t(key_of[i])]) {
            projected[size_t(key_of[i])] = 1;
            std::vector<GateIn>& vg = out.gates[sc];
            std::vector<AngleSource>& va = out.angles[sc];
            const int c = local[size_t(g.control)], kq = out.n_side[sc] + key;
            va.push_back(fixed(kFixedX));
            vg.push_back(GateIn{c, kq, int(va.size()) - 1});
            va.push_back(fixed(kFixedProj0));
            vg.push_back(GateIn{c, -1, int(va.size()) - 1});
            va.push_back(fixed(kFixedX));
            vg.push_back(GateIn{c, kq, int(va.size()) - 1});
        }
        out.angles[st].push_back(op_angles[size_t(g.op)]);
        out.gates[st].push_back(GateIn{local[size_t(g.target)], out.n_side[st] + key, int(out.angles[st].size()) - 1});
    }
    out.ok = true;
    return out;
    }
    return SplitCircuits{};
}

}  // namespace qsv
