// CPU experiment (not part of the product): how much surface-area cost does parallel reinsertion (Meister & Bittner 2018) take out of
// a PLOC tree?  Emulates the GPU algorithm pass by pass: every node searches the best place for itself in the CURRENT tree, the
// moves are applied in order of decreasing gain while their topology nodes are free, the boxes are refitted, repeat.
// Prints sum of inner-node areas / root area after every pass (the figure tools/bvh_quality.py uses).
//   g++ -O2 -o /tmp/bq/probe tools/bvh_reinsert_probe.cpp && /tmp/bq/probe /tmp/bq/sponza.tri [passes] [fraction-mod]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Box {
    float lo[3], hi[3];
    void grow(const Box& b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    float area() const { float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2]; return x * y + y * z + z * x; }
};
static Box join(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }
static Box empty_box() { Box b; for (int k = 0; k < 3; k++) { b.lo[k] = 1e30f; b.hi[k] = -1e30f; } return b; }

struct Tree {
    int n = 0, root = 0;
    std::vector<int> left, right, parent;
    std::vector<Box> box;
    bool leaf(int i) const { return i < n; }
    double cost() const {
        double c = 0;
        for (int i = n; i < 2 * n - 1; i++) c += box[i].area();
        return c / box[root].area();
    }
    int sibling(int i) const { int p = parent[i]; return left[p] == i ? right[p] : left[p]; }
    void refit() {                                               // post-order from the root (topology is arbitrary)
        std::vector<int> order; order.reserve(2 * n); std::vector<int> st{root};
        while (!st.empty()) { int v = st.back(); st.pop_back(); order.push_back(v); if (!leaf(v)) { st.push_back(left[v]); st.push_back(right[v]); } }
        for (size_t k = order.size(); k-- > 0;) { int v = order[k]; if (!leaf(v)) box[v] = join(box[left[v]], box[right[v]]); }
    }
    int depth() const {
        int best = 0; std::vector<std::pair<int, int>> st{{root, 1}};
        while (!st.empty()) { auto [v, d] = st.back(); st.pop_back(); best = std::max(best, d); if (!leaf(v)) { st.push_back({left[v], d + 1}); st.push_back({right[v], d + 1}); } }
        return best;
    }
};

static uint64_t expand21(uint64_t v) {
    v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull; v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v;
}

static Tree ploc(const std::vector<Box>& tri, int radius) {
    const int n = (int)tri.size();
    Box all = empty_box(), cen = empty_box();
    for (auto& b : tri) { all.grow(b); Box c; for (int k = 0; k < 3; k++) c.lo[k] = c.hi[k] = 0.5f * (b.lo[k] + b.hi[k]); cen.grow(c); }
    std::vector<std::pair<uint64_t, int>> keys(n);
    for (int i = 0; i < n; i++) {
        uint64_t q[3];
        for (int k = 0; k < 3; k++) { float e = cen.hi[k] - cen.lo[k]; float t = e > 0 ? (0.5f * (tri[i].lo[k] + tri[i].hi[k]) - cen.lo[k]) / e : 0; q[k] = (uint64_t)std::min(2097151.0f, t * 2097152.0f); }
        keys[i] = {expand21(q[0]) << 2 | expand21(q[1]) << 1 | expand21(q[2]), i};
    }
    std::sort(keys.begin(), keys.end());
    Tree t; t.n = n; t.left.assign(2 * n - 1, -1); t.right.assign(2 * n - 1, -1); t.parent.assign(2 * n - 1, -1); t.box.resize(2 * n - 1);
    std::vector<int> cl(n);
    for (int i = 0; i < n; i++) { t.box[i] = tri[keys[i].second]; cl[i] = i; }
    int next = n;
    while (cl.size() > 1) {
        const int m = (int)cl.size();
        std::vector<int> nn(m);
        for (int i = 0; i < m; i++) {
            float best = 1e38f; int bj = -1;
            for (int j = std::max(0, i - radius); j <= std::min(m - 1, i + radius); j++) if (j != i) { float a = join(t.box[cl[i]], t.box[cl[j]]).area(); if (a < best) { best = a; bj = j; } }
            nn[i] = bj;
        }
        std::vector<int> out; out.reserve(m);
        for (int i = 0; i < m; i++) {
            int j = nn[i];
            if (nn[j] == i) { if (i < j) { int v = next++; t.left[v] = cl[i]; t.right[v] = cl[j]; t.parent[cl[i]] = t.parent[cl[j]] = v; t.box[v] = join(t.box[cl[i]], t.box[cl[j]]); out.push_back(v); } }
            else out.push_back(cl[i]);
        }
        cl.swap(out);
    }
    t.root = cl[0];
    return t;
}

// binned SAH, single-triangle leaves: only its cost is wanted
static double sah_cost(std::vector<Box> tri) {
    const int n = (int)tri.size(); double total = 0; Box all = empty_box(); for (auto& b : tri) all.grow(b);
    struct R { int a, b; }; std::vector<R> st{{0, n}};
    while (!st.empty()) {
        R r = st.back(); st.pop_back();
        if (r.b - r.a < 2) continue;
        Box bb = empty_box(), cb = empty_box();
        for (int i = r.a; i < r.b; i++) { bb.grow(tri[i]); Box c; for (int k = 0; k < 3; k++) c.lo[k] = c.hi[k] = 0.5f * (tri[i].lo[k] + tri[i].hi[k]); cb.grow(c); }
        total += bb.area();
        const int B = 32; double bestc = 1e300; int bax = -1, bsplit = 0;
        for (int ax = 0; ax < 3; ax++) {
            float e = cb.hi[ax] - cb.lo[ax]; if (!(e > 0)) continue;
            Box bin[B]; int cnt[B] = {0}; for (auto& x : bin) x = empty_box();
            for (int i = r.a; i < r.b; i++) { int k = std::min(B - 1, (int)((0.5f * (tri[i].lo[ax] + tri[i].hi[ax]) - cb.lo[ax]) / e * B)); bin[k].grow(tri[i]); cnt[k]++; }
            Box acc = empty_box(); float la[B]; int lc[B]; int c = 0;
            for (int k = 0; k < B; k++) { acc.grow(bin[k]); c += cnt[k]; la[k] = c ? acc.area() : 0; lc[k] = c; }
            acc = empty_box(); c = 0;
            for (int k = B - 1; k > 0; k--) { acc.grow(bin[k]); c += cnt[k]; if (c && lc[k - 1]) { double cost = (double)la[k - 1] * lc[k - 1] + (double)acc.area() * c; if (cost < bestc) { bestc = cost; bax = ax; bsplit = k; } } }
        }
        int mid;
        if (bax < 0) mid = (r.a + r.b) / 2;
        else {
            float e = cb.hi[bax] - cb.lo[bax];
            mid = (int)(std::partition(tri.begin() + r.a, tri.begin() + r.b, [&](const Box& t) { return std::min(B - 1, (int)((0.5f * (t.lo[bax] + t.hi[bax]) - cb.lo[bax]) / e * B)) < bsplit; }) - tri.begin());
            if (mid == r.a || mid == r.b) mid = (r.a + r.b) / 2;
        }
        st.push_back({r.a, mid}); st.push_back({mid, r.b});
    }
    return total / all.area();
}

struct Move { float gain; int in, out, top; };
#ifdef PROBE_CHECK
int check_main(const std::vector<Box>& tri_all);
#endif

// best new place for node `in` in the current tree (no stack: the walk goes down into the sibling subtrees hanging off the path to
// the root and back up through parent links)
static Move find_best(const Tree& t, int in) {
    Move mv{0.0f, in, -1, -1};
    const int p = t.parent[in];
    if (p < 0 || t.parent[p] < 0) return mv;
    const Box bin = t.box[in];
    const float area_in = bin.area();
    float d_path = t.box[p].area();          // gain so far: the parent disappears
    int pivot = p;                           // ancestor whose other subtree is being searched
    Box pivot_box = empty_box();             // box of `pivot` without `in` (valid once the search left pivot's subtree)
    int out = t.sibling(in);
    bool down = true;
    float d = d_path;                        // gain along the current descent
    std::vector<float> undo;                 // growth charged on the way down (a GPU version recomputes it going up)
    while (true) {
        if (down) {
            const Box& bo = t.box[out];
            const float merged = join(bo, bin).area();
            const float here = d - merged;
            if (here > mv.gain) { mv.gain = here; mv.out = out; mv.top = pivot == p ? t.parent[p] : pivot; }
            const float grow = merged - bo.area();
            if (t.leaf(out) || d - grow - area_in <= mv.gain) down = false;
            else { undo.push_back(grow); d -= grow; out = t.left[out]; }
        } else {
            const int po = t.parent[out];
            if (po == pivot) {
                // the sibling subtree of `pivot`'s path child is done: move the pivot one level up
                pivot_box = (pivot == p) ? t.box[out] : join(pivot_box, t.box[out]);
                // now pivot_box = box of pivot without `in`
                const int up = t.parent[pivot];
                if (up < 0) break;
                if (pivot != p) {
                    d_path += t.box[pivot].area() - pivot_box.area();     // an ancestor shrinks (p itself is removed: counted already)
                    const float here = d_path - t.box[pivot].area();     // `in` as the sibling of the shrunk ancestor itself
                    if (here > mv.gain && t.parent[pivot] >= 0) { mv.gain = here; mv.out = pivot; mv.top = t.parent[pivot]; }
                }
                out = t.sibling(pivot);
                pivot = up;
                d = d_path; down = true; undo.clear();
            } else if (out == t.left[po]) { out = t.right[po]; down = true; }         // second child of the same parent: same d
            else { d += undo.back(); undo.pop_back(); out = po; }
        }
    }
    return mv;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    const int n = (int)(bytes / 36); std::vector<float> v(n * 9); if (fread(v.data(), 36, n, f) != (size_t)n) return 1; fclose(f);
    const int passes = argc > 2 ? atoi(argv[2]) : 8;
    const int mod = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<Box> tri(n);
    for (int i = 0; i < n; i++) { tri[i] = empty_box(); for (int k = 0; k < 3; k++) { Box p; for (int a = 0; a < 3; a++) p.lo[a] = p.hi[a] = v[i * 9 + k * 3 + a]; tri[i].grow(p); } }
#ifdef PROBE_CHECK
    return check_main(tri);
#endif
    printf("%d triangles; binned SAH (32 bins, 1-triangle leaves): %.2f\n", n, sah_cost(tri));
    Tree t = ploc(tri, 16);
    printf("PLOC r16: %.2f  depth %d\n", t.cost(), t.depth());
    if (getenv("SEQUENTIAL")) {                                   // Bittner 2013 style: every move applied at once, paths refitted
        auto refit_up = [&](int v) { for (; v >= 0; v = t.parent[v]) t.box[v] = join(t.box[t.left[v]], t.box[t.right[v]]); };
        for (int pass = 0; pass < passes; pass++) {
            std::vector<int> order; for (int i = 0; i < 2 * n - 1; i++) order.push_back(i);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return t.box[a].area() > t.box[b].area(); });
            int applied = 0;
            for (int in : order) {
                if (in == t.root) continue;
                Move m = find_best(t, in);
                if (m.out < 0 || !(m.gain > 0)) continue;
                const int out = m.out, p = t.parent[in], s = t.sibling(in), g = t.parent[p];
                (t.left[g] == p ? t.left[g] : t.right[g]) = s; t.parent[s] = g;
                const int po2 = t.parent[out];
                (t.left[po2] == out ? t.left[po2] : t.right[po2]) = p; t.parent[p] = po2;
                t.left[p] = out; t.right[p] = in; t.parent[out] = p; t.parent[in] = p;
                refit_up(g); refit_up(p);
                applied++;
            }
            printf("sequential pass %d: %d applied -> %.2f  depth %d\n", pass + 1, applied, t.cost(), t.depth());
        }
        return 0;
    }
    for (int pass = 0; pass < passes; pass++) {
        std::vector<Move> mv;
        for (int i = 0; i < 2 * n - 1; i++) { if (mod > 1 && (i + pass) % mod) continue; if (i == t.root) continue; Move m = find_best(t, i); if (m.out >= 0 && m.gain > 0) mv.push_back(m); }
        std::sort(mv.begin(), mv.end(), [](const Move& a, const Move& b) { return a.gain > b.gain; });
        std::vector<char> locked(2 * n - 1, 0);
        // phase 1 (on the unmodified tree): the path in -> top <- out is locked (Meister & Bittner): boxes along it change, and
        // disjoint paths cannot form a cycle.  Highest gain first = what atomicMax on (gain, index) keys decides on the GPU.
        std::vector<Move> win;
        if (getenv("ATOMIC_LOCKS")) {                            // what the GPU does: EVERY candidate stamps max(key) on its path, then keeps the move if it holds all of it
            std::vector<long long> key(2 * n - 1, -1);
            auto path = [&](const Move& m, auto&& f) {
                for (int v = m.in; v != m.top; v = t.parent[v]) f(v);
                for (int v = m.out; v != m.top; v = t.parent[v]) f(v);
                f(m.top);
            };
            const int rounds = atoi(getenv("ATOMIC_LOCKS"));
            std::vector<char> owned(2 * n - 1, 0), state(mv.size(), 0);      // state: 0 candidate, 1 winner, 2 dead (touches a winner's path)
            for (int r = 0; r < rounds; r++) {
                std::fill(key.begin(), key.end(), -1);
                for (size_t k = 0; k < mv.size(); k++) if (state[k] == 0) { bool dead = false; path(mv[k], [&](int v) { if (owned[v]) dead = true; }); if (dead) state[k] = 2; }
                for (size_t k = 0; k < mv.size(); k++) if (state[k] == 0) path(mv[k], [&](int v) { key[v] = std::max(key[v], (long long)(mv.size() - k)); });
                for (size_t k = 0; k < mv.size(); k++) if (state[k] == 0) { bool held = true; path(mv[k], [&](int v) { if (key[v] != (long long)(mv.size() - k)) held = false; }); if (held) state[k] = 1; }
                for (size_t k = 0; k < mv.size(); k++) if (state[k] == 1) path(mv[k], [&](int v) { owned[v] = 1; });
            }
            for (size_t k = 0; k < mv.size(); k++) if (state[k] == 1) win.push_back(mv[k]);
        } else
        for (auto& m : mv) {
            std::vector<int> lock;
            for (int v = m.in; v != m.top; v = t.parent[v]) lock.push_back(v);
            for (int v = m.out; v != m.top; v = t.parent[v]) lock.push_back(v);
            lock.push_back(m.top);
            bool free_ = true; for (int k : lock) if (locked[k]) free_ = false;
            if (!free_) continue;
            for (int k : lock) locked[k] = 1;
            win.push_back(m);
        }
        int applied = 0;
        for (auto& m : win) {
            const int in = m.in, out = m.out, p = t.parent[in], s = t.sibling(in), g = t.parent[p];
            (t.left[g] == p ? t.left[g] : t.right[g]) = s; t.parent[s] = g;
            const int po2 = t.parent[out];
            (t.left[po2] == out ? t.left[po2] : t.right[po2]) = p; t.parent[p] = po2;
            t.left[p] = out; t.right[p] = in; t.parent[out] = p; t.parent[in] = p;
            applied++;
        }
        t.refit();
        printf("pass %d: %zu candidates, %d applied -> %.2f  depth %d\n", pass + 1, mv.size(), applied, t.cost(), t.depth());
    }
    return 0;
}

// validation (argv[2] == "check"): on the first 3000 triangles, the predicted gain of find_best against brute force over every position
#ifdef PROBE_CHECK
static bool in_subtree(const Tree& t, int root, int v) { for (; v >= 0; v = t.parent[v]) if (v == root) return true; return false; }
static double total_area(const Tree& t) { double c = 0; for (int i = t.n; i < 2 * t.n - 1; i++) c += t.box[i].area(); return c; }
int check_main(const std::vector<Box>& tri_all) {
    std::vector<Box> tri; for (size_t i = 0; i < tri_all.size(); i += 101) tri.push_back(tri_all[i]);
    Tree t = ploc(tri, 16);
    const double base = total_area(t);
    srand(1);
    for (int trial = 0; trial < 400; trial++) {
        int in = rand() % (2 * t.n - 1);
        if (in == t.root || t.parent[in] == t.root) continue;
        Move m = find_best(t, in);
        double best = 0; int bx = -1;
        for (int x = 0; x < 2 * t.n - 1; x++) {
            if (x == t.root || in_subtree(t, in, x) || x == t.parent[in]) continue;
            Tree c = t;
            const int p = c.parent[in], s = c.sibling(in), g = c.parent[p];
            (c.left[g] == p ? c.left[g] : c.right[g]) = s; c.parent[s] = g;
            const int po = c.parent[x];
            (c.left[po] == x ? c.left[po] : c.right[po]) = p; c.parent[p] = po;
            c.left[p] = x; c.right[p] = in; c.parent[x] = p; c.parent[in] = p;
            c.refit();
            const double gain = base - total_area(c);
            if (gain > best) { best = gain; bx = x; }
        }
        if (best > 0 || m.gain > 0) printf("node %d: search gain %.6g at %d | brute force %.6g at %d\n", in, m.gain, m.out, best, bx);
    }
    return 0;
}
#endif
