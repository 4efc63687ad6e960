// Host-side sector matching of a block-sparse contraction, in C++ like the reference's abelian_compose_worker
// (src/backends/abelian.cpp:1239-1469): key packing of the contracted sector indices (:1265-1283), lexsort and grouping of
// both block tables (:1286-1345), coupled charges of the kept legs (:1384-1418), charge lookup and the merge walk over the
// contracted keys that yields, per result block, the list of (a-block, b-block) pairs whose products are summed
// (:1420-1460).  int64 bookkeeping only -- no device work, no block data.  cyten_amd/abelian.py::compose_plan holds the
// same algorithm in numpy (and stays as the specification the tests compare against); this version takes the ~0.8 ms of
// Python per contraction off the hot path (3 ms of a 12 ms step at chi = 1024).
#include "common.h"

#include <algorithm>
#include <map>
#include <numeric>
#include <vector>

struct cyb_compose_plan_s {
    int64_t n_cols = 0; // kept legs of a + kept legs of b
    std::vector<int64_t> res_block_inds, res_shapes; // n_res x n_cols
    std::vector<int64_t> group_off;                  // n_res + 1
    std::vector<int64_t> pair_a, pair_b;             // indices into the ORIGINAL block lists
    std::vector<int64_t> pair_k;                     // contracted extent K of every pair
    std::vector<int64_t> res_m, res_n;               // M = prod of a's kept extents, N = prod of b's kept extents per result
    int64_t na_blocks = 0, nb_blocks = 0;
    int32_t num_contr = 0;
    double flops = 0.0;
};

namespace {

// row order of np.lexsort(rows.T): the LAST column is the primary key, the first the least significant
struct RowLess {
    const int64_t* data;
    int64_t ncol;
    bool operator()(int64_t x, int64_t y) const
    {
        for (int64_t c = ncol - 1; c >= 0; --c) {
            const int64_t a = data[x * ncol + c], b = data[y * ncol + c];
            if (a != b) return a < b;
        }
        return false;
    }
};

int64_t mod_reduce(int64_t q, int64_t m)
{
    if (m == 0) return q;
    const int64_t r = q % m;
    return r < 0 ? r + m : r;
}

} // namespace

extern "C" {

int cyb_compose_plan_create(const int64_t* moduli, int32_t n_sym, const cyb_leg* a_legs, int32_t na_legs,
                            const int64_t* a_block_inds, int64_t na_blocks, const cyb_leg* b_legs, int32_t nb_legs,
                            const int64_t* b_block_inds, int64_t nb_blocks, int32_t num_contr, cyb_compose_plan_t* out)
{
    CYB_REQUIRE(out, "cyb_compose_plan_create: out is NULL");
    CYB_REQUIRE(n_sym >= 0 && (n_sym == 0 || moduli), "cyb_compose_plan_create: bad symmetry");
    CYB_REQUIRE(na_legs >= 0 && nb_legs >= 0 && num_contr >= 0 && num_contr <= na_legs && num_contr <= nb_legs,
                "cyb_compose_plan_create: bad leg counts (%d, %d legs, %d contracted)", na_legs, nb_legs, num_contr);
    CYB_REQUIRE((na_legs == 0 || a_legs) && (nb_legs == 0 || b_legs), "cyb_compose_plan_create: legs are NULL");
    CYB_REQUIRE(na_blocks >= 0 && nb_blocks >= 0 && (na_blocks == 0 || na_legs == 0 || a_block_inds) &&
                    (nb_blocks == 0 || nb_legs == 0 || b_block_inds),
                "cyb_compose_plan_create: bad block tables");
    const int64_t na_keep = na_legs - num_contr, nb_keep = nb_legs - num_contr;
    // the contracted legs must match: a.legs[na_legs - 1 - i] pairs with b.legs[i]
    for (int i = 0; i < num_contr; ++i) {
        const cyb_leg& la = a_legs[na_legs - 1 - i];
        const cyb_leg& lb = b_legs[i];
        bool ok = la.sign == -lb.sign && la.n_sectors == lb.n_sectors;
        for (int64_t s = 0; ok && s < la.n_sectors; ++s) {
            ok = la.mults[s] == lb.mults[s];
            for (int k = 0; ok && k < n_sym; ++k) ok = la.sectors[s * n_sym + k] == lb.sectors[s * n_sym + k];
        }
        CYB_REQUIRE(ok, "legs a[%d] and b[%d] are not contractible", na_legs - 1 - i, i);
    }
    for (int64_t r = 0; r < na_blocks; ++r)
        for (int c = 0; c < na_legs; ++c)
            CYB_REQUIRE(a_block_inds[r * na_legs + c] >= 0 && a_block_inds[r * na_legs + c] < a_legs[c].n_sectors,
                        "cyb_compose_plan_create: a.block_inds[%lld, %d] out of range", (long long)r, c);
    for (int64_t r = 0; r < nb_blocks; ++r)
        for (int c = 0; c < nb_legs; ++c)
            CYB_REQUIRE(b_block_inds[r * nb_legs + c] >= 0 && b_block_inds[r * nb_legs + c] < b_legs[c].n_sectors,
                        "cyb_compose_plan_create: b.block_inds[%lld, %d] out of range", (long long)r, c);
    auto* pl = new cyb_compose_plan_s();
    pl->n_cols = na_keep + nb_keep;
    pl->na_blocks = na_blocks;
    pl->nb_blocks = nb_blocks;
    pl->num_contr = num_contr;
    pl->group_off.push_back(0);
    *out = pl;
    if (na_blocks == 0 || nb_blocks == 0) return CYB_OK;

    // ---- keys of the contracted columns, F-style strides over b's leg order (:1265-1283)
    std::vector<int64_t> stride((size_t)std::max(num_contr, 1), 1);
    for (int i = 1; i < num_contr; ++i) stride[(size_t)i] = stride[(size_t)i - 1] * b_legs[i - 1].n_sectors;
    // sort tables: rows [key, keep...] ordered like np.lexsort(hstack([key, keep]).T)
    auto sorted_table = [&](const int64_t* inds, int64_t nblocks, int nlegs, bool is_a, std::vector<int64_t>& rows,
                            std::vector<int64_t>& order) {
        const int64_t nkeep = is_a ? na_keep : nb_keep, ncol = nkeep + 1;
        rows.assign((size_t)(nblocks * ncol), 0);
        for (int64_t r = 0; r < nblocks; ++r) {
            int64_t key = 0;
            for (int i = 0; i < num_contr; ++i)
                key += (is_a ? inds[r * nlegs + (nlegs - 1 - i)] : inds[r * nlegs + i]) * stride[(size_t)i];
            rows[(size_t)(r * ncol)] = key;
            for (int64_t c = 0; c < nkeep; ++c) rows[(size_t)(r * ncol + 1 + c)] = inds[r * nlegs + (is_a ? c : num_contr + c)];
        }
        order.resize((size_t)nblocks);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), RowLess{rows.data(), ncol});
    };
    std::vector<int64_t> a_rows_t, b_rows_t, a_sort, b_sort;
    sorted_table(a_block_inds, na_blocks, na_legs, true, a_rows_t, a_sort);
    sorted_table(b_block_inds, nb_blocks, nb_legs, false, b_rows_t, b_sort);
    // groups of equal keep columns (in sorted order): slice boundaries
    auto groups = [&](const std::vector<int64_t>& rows, const std::vector<int64_t>& order, int64_t nkeep) {
        std::vector<int64_t> sl{0};
        const int64_t ncol = nkeep + 1;
        for (size_t k = 1; k < order.size(); ++k) {
            bool diff = false;
            for (int64_t c = 0; c < nkeep && !diff; ++c)
                diff = rows[(size_t)(order[k] * ncol + 1 + c)] != rows[(size_t)(order[k - 1] * ncol + 1 + c)];
            if (diff) sl.push_back((int64_t)k);
        }
        sl.push_back((int64_t)order.size());
        return sl;
    };
    const std::vector<int64_t> a_sl = groups(a_rows_t, a_sort, na_keep), b_sl = groups(b_rows_t, b_sort, nb_keep);
    const int64_t n_ga = (int64_t)a_sl.size() - 1, n_gb = (int64_t)b_sl.size() - 1;
    // ---- coupled charge of the kept legs of every group (:1384-1418); b's legs enter with the opposite sign
    auto charge = [&](const int64_t* inds, const cyb_leg* legs, int nlegs, int64_t block, int64_t nkeep, int first, int sgn,
                      std::vector<int64_t>& q) {
        q.assign((size_t)n_sym, 0);
        for (int64_t c = 0; c < nkeep; ++c) {
            const cyb_leg& lg = legs[first + c];
            const int64_t sec = inds[block * nlegs + first + c];
            for (int k = 0; k < n_sym; ++k) q[(size_t)k] += (int64_t)sgn * lg.sign * lg.sectors[sec * n_sym + k];
        }
        for (int k = 0; k < n_sym; ++k) q[(size_t)k] = mod_reduce(q[(size_t)k], moduli[k]);
    };
    std::map<std::vector<int64_t>, std::vector<int64_t>> lookup; // charge -> groups of a, ascending
    std::vector<int64_t> q;
    for (int64_t g = 0; g < n_ga; ++g) {
        charge(a_block_inds, a_legs, na_legs, a_sort[(size_t)a_sl[(size_t)g]], na_keep, 0, +1, q);
        lookup[q].push_back(g);
    }
    // ---- merge walk over the contracted keys (:1424-1460)
    struct Res {
        std::vector<int64_t> row, shape;
        std::vector<int64_t> pa, pb, pk;
        int64_t M = 1, N = 1;
    };
    std::vector<Res> res;
    double flops = 0.0;
    for (int64_t gb = 0; gb < n_gb; ++gb) {
        charge(b_block_inds, b_legs, nb_legs, b_sort[(size_t)b_sl[(size_t)gb]], nb_keep, num_contr, -1, q);
        auto it = lookup.find(q);
        if (it == lookup.end()) continue;
        for (int64_t ga : it->second) {
            Res r;
            int64_t i = a_sl[(size_t)ga], j = b_sl[(size_t)gb];
            const int64_t i1 = a_sl[(size_t)ga + 1], j1 = b_sl[(size_t)gb + 1];
            while (i < i1 && j < j1) { // both key lists ascend and hold no duplicates
                const int64_t ka = a_rows_t[(size_t)(a_sort[(size_t)i] * (na_keep + 1))];
                const int64_t kb = b_rows_t[(size_t)(b_sort[(size_t)j] * (nb_keep + 1))];
                if (ka == kb) {
                    r.pa.push_back(a_sort[(size_t)i]);
                    r.pb.push_back(b_sort[(size_t)j]);
                    ++i, ++j;
                } else if (ka < kb) {
                    ++i;
                } else {
                    ++j;
                }
            }
            if (r.pa.empty()) continue;
            const int64_t ra = a_sort[(size_t)a_sl[(size_t)ga]], rb = b_sort[(size_t)b_sl[(size_t)gb]];
            double M = 1.0, N = 1.0;
            for (int64_t c = 0; c < na_keep; ++c) {
                const int64_t sec = a_block_inds[ra * na_legs + c];
                r.row.push_back(sec);
                r.shape.push_back(a_legs[c].mults[sec]);
                M *= (double)a_legs[c].mults[sec];
            }
            for (int64_t c = 0; c < nb_keep; ++c) {
                const int64_t sec = b_block_inds[rb * nb_legs + num_contr + c];
                r.row.push_back(sec);
                r.shape.push_back(b_legs[num_contr + c].mults[sec]);
                N *= (double)b_legs[num_contr + c].mults[sec];
            }
            for (int64_t ai : r.pa) {
                double K = 1.0;
                for (int c = 0; c < num_contr; ++c) K *= (double)a_legs[na_keep + c].mults[a_block_inds[ai * na_legs + na_keep + c]];
                flops += 2.0 * M * N * K;
                r.pk.push_back((int64_t)K);
            }
            r.M = (int64_t)M;
            r.N = (int64_t)N;
            res.push_back(std::move(r));
        }
    }
    // ---- result rows in lexsorted order (last column primary)
    const int64_t ncol = pl->n_cols;
    std::vector<int64_t> flat((size_t)((int64_t)res.size() * std::max<int64_t>(ncol, 1)), 0);
    for (size_t g = 0; g < res.size(); ++g)
        for (int64_t c = 0; c < ncol; ++c) flat[g * (size_t)ncol + (size_t)c] = res[g].row[(size_t)c];
    std::vector<int64_t> order(res.size());
    std::iota(order.begin(), order.end(), 0);
    if (ncol > 0) std::stable_sort(order.begin(), order.end(), RowLess{flat.data(), ncol});
    for (int64_t g : order) {
        const Res& r = res[(size_t)g];
        pl->res_block_inds.insert(pl->res_block_inds.end(), r.row.begin(), r.row.end());
        pl->res_shapes.insert(pl->res_shapes.end(), r.shape.begin(), r.shape.end());
        pl->pair_a.insert(pl->pair_a.end(), r.pa.begin(), r.pa.end());
        pl->pair_b.insert(pl->pair_b.end(), r.pb.begin(), r.pb.end());
        pl->pair_k.insert(pl->pair_k.end(), r.pk.begin(), r.pk.end());
        pl->res_m.push_back(r.M);
        pl->res_n.push_back(r.N);
        pl->group_off.push_back((int64_t)pl->pair_a.size());
    }
    pl->flops = flops;
    return CYB_OK;
}

int cyb_compose_plan_sizes(cyb_compose_plan_t pl, int64_t* n_res, int64_t* n_pairs, int64_t* n_cols)
{
    CYB_REQUIRE(pl, "cyb_compose_plan_sizes: plan is NULL");
    if (n_res) *n_res = (int64_t)pl->group_off.size() - 1;
    if (n_pairs) *n_pairs = (int64_t)pl->pair_a.size();
    if (n_cols) *n_cols = pl->n_cols;
    return CYB_OK;
}

int cyb_compose_plan_get(cyb_compose_plan_t pl, int64_t* res_block_inds, int64_t* res_shapes, int64_t* group_offsets,
                         int64_t* pair_a, int64_t* pair_b, double* flops)
{
    CYB_REQUIRE(pl, "cyb_compose_plan_get: plan is NULL");
    if (res_block_inds) std::copy(pl->res_block_inds.begin(), pl->res_block_inds.end(), res_block_inds);
    if (res_shapes) std::copy(pl->res_shapes.begin(), pl->res_shapes.end(), res_shapes);
    if (group_offsets) std::copy(pl->group_off.begin(), pl->group_off.end(), group_offsets);
    if (pair_a) std::copy(pl->pair_a.begin(), pl->pair_a.end(), pair_a);
    if (pair_b) std::copy(pl->pair_b.begin(), pl->pair_b.end(), pair_b);
    if (flops) *flops = pl->flops;
    return CYB_OK;
}

// The hot loop of abelian_compose_worker (abelian.cpp:1424-1460) for a plan whose operand blocks are C-contiguous float64
// arrays: descriptor arrays built HERE (no per-block host objects on the caller's side) and ONE asynchronous grouped launch.
// An a-block (kept extents..., contracted extents...) is the row-major M x K matrix, a b-block (contracted..., kept...) the
// row-major K x N matrix; with more than one contracted leg the caller passes b-blocks whose contracted axes are already in
// a's (reversed) order (abelian.cpp:1349-1382 reshapes after the same permutation).
int cyb_compose_plan_enqueue_f64(cyb_ctx_t ctx, cyb_compose_plan_t pl, const int64_t* a_ptrs, const int64_t* b_ptrs,
                                 const int64_t* which, int64_t n_which, const int64_t* out_ptrs, double* flops, double* bytes)
{
    CYB_REQUIRE(ctx && pl, "cyb_compose_plan_enqueue_f64: NULL argument");
    const int64_t n_res = (int64_t)pl->group_off.size() - 1;
    const int64_t n = which ? n_which : n_res;
    CYB_REQUIRE(n >= 0 && (n == 0 || (a_ptrs && b_ptrs && out_ptrs)), "cyb_compose_plan_enqueue_f64: NULL pointer table");
    std::vector<cyb_gemm_prob> probs;
    std::vector<cyb_gemm_seg> segs;
    probs.reserve((size_t)n);
    double fl = 0.0, by = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t g = which ? which[k] : k;
        CYB_REQUIRE(g >= 0 && g < n_res, "cyb_compose_plan_enqueue_f64: result block %lld out of range", (long long)g);
        const int64_t M = pl->res_m[(size_t)g], N = pl->res_n[(size_t)g];
        CYB_REQUIRE(out_ptrs[k] != 0 || M * N == 0, "cyb_compose_plan_enqueue_f64: NULL output for result block %lld", (long long)g);
        cyb_gemm_prob p;
        p.C = reinterpret_cast<double*>(out_ptrs[k]);
        p.M = M;
        p.N = N;
        p.ldc = std::max<int64_t>(N, 1);
        p.seg_begin = (int32_t)segs.size();
        for (int64_t t = pl->group_off[(size_t)g]; t < pl->group_off[(size_t)g + 1]; ++t) {
            const int64_t ia = pl->pair_a[(size_t)t], ib = pl->pair_b[(size_t)t], K = pl->pair_k[(size_t)t];
            CYB_REQUIRE((a_ptrs[ia] && b_ptrs[ib]) || K * M * N == 0, "cyb_compose_plan_enqueue_f64: NULL operand block");
            cyb_gemm_seg sg;
            sg.A = reinterpret_cast<const double*>(a_ptrs[ia]);
            sg.B = reinterpret_cast<const double*>(b_ptrs[ib]);
            sg.K = K;
            sg.a_rs = std::max<int64_t>(K, 1);
            sg.a_cs = 1;
            sg.b_rs = std::max<int64_t>(N, 1);
            sg.b_cs = 1;
            segs.push_back(sg);
            fl += 2.0 * (double)M * (double)N * (double)K;
            by += 8.0 * ((double)M * (double)K + (double)K * (double)N);
        }
        p.seg_end = (int32_t)segs.size();
        p.alpha = 1.0;
        p.beta = 0.0;
        by += 8.0 * (double)M * (double)N;
        probs.push_back(p);
    }
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    if (probs.empty()) return CYB_OK;
    return cyb::gemm_launch_async(ctx, probs.data(), (int64_t)probs.size(), segs.data(), (int64_t)segs.size());
}

int cyb_compose_plan_destroy(cyb_compose_plan_t pl)
{
    delete pl;
    return CYB_OK;
}

} // extern "C"
