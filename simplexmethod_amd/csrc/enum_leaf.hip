// enum_leaf.hip — leaf kernels of the shared-prefix enumeration: ONE LANE PER SUBSET.
//
// Input: the depth m-7 records produced breadth-first by k_enum_expand (enum_prefix.hip): the
// tableau [W[:, c > last] | rhs] after the first m-7 Gauss-Jordan steps, shared by every subset
// with that prefix.
//
// k_enum_leaves<1> (a third of the subsets).  A wave owns one work item (record, child column,
// chunk of subsets) at a time: it copies the record to its private LDS slice, performs the pivot
// on the child column cooperatively (16 rows x 4 columns per step; the depth m-6 tableau never
// travels through HBM; its rows are written permuted, unused rows first), and each of its 64
// lanes then finishes one subset on its own: it picks the remaining 6 columns (table of all
// 6-subsets in lexicographic order), replays the last four Gauss-Jordan steps and the 2x2 block
// exactly as oracle/lp_oracle.c: orc_enum_subset orders them, and tests feasibility.  No
// cross-lane traffic, no barriers, no divergence beyond the leaf loop's tail: the ~800
// instructions per subset are ordinary lane-parallel fp64 work, which is what this chip has in
// abundance (the cooperative sweep of enum_prefix.hip shares two more levels but is bound by the
// latency of its broadcasts, pivot searches and workgroup barriers).
//
// k_enum_leaves<2> / <3> (two thirds of the subsets; <3> adds a third shared pivot for the large sub-groups, see MODE 3).  The subsets below a child are grouped by their
// first remaining column; a group that leaves >= kGrandMin selectable columns gets a second
// in-LDS pivot and its lanes take 5 columns each (items of table 1).  k_enum_leaves<1> finishes
// what is left of each child (items of table 0).
//
// k_enum_thin (the tails).  A depth m-6 node with fewer than 8 selectable columns holds at most
// C(7,6) = 7 subsets: as a work item it would cost a pivot and a wave pass for a handful of
// lanes, and such nodes are two thirds of their level.  Their subsets are exactly the ones that
// lie inside the LAST 8 columns of a depth m-7 node — at most C(8,7) = 8 per record — and the
// thin kernel takes them straight from the records in HBM: 8 lanes per record, 7 columns per
// lane, every lane reading its own record (no LDS staging, nothing wave-uniform).
//
//   phase 1  the KD rows not used by the prefix x the KD chosen columns (+ rhs) sit in
//            registers; KD-2 Gauss-Jordan steps with partial pivoting among the unused rows
//            (pivot row = select chain), then the 2x2 block -> x_a, x_b.
//   phase 2  each of the m-KD rows used by the prefix streams through: KD+1 reads, the same
//            eliminations against the stored pivot rows, back-substitution, x >= -1e-9.
#include <type_traits>

#include "enum_tree.hpp"

using namespace lptree;

namespace {

constexpr int LEAF_THREADS = 256;
constexpr int LEAF_WAVES = LEAF_THREADS / 64;
#ifndef LP_GRAND_MIN
#define LP_GRAND_MIN 10
#endif
constexpr int kGrandMin = LP_GRAND_MIN;         // a depth m-5 node with >= 10 selectable columns (>= 252 subsets) is
                                      // finished by the two-level kernel (second in-LDS pivot, 5 columns per lane)
#ifndef LP_GREAT_MIN
#define LP_GREAT_MIN 9
#endif
constexpr int kGreatMin = LP_GREAT_MIN;         // k_enum_leaves<3>: a sub-group (child, j2, j3) that leaves >= 9 selectable
                                      // columns (>= 126 subsets) gets a third in-LDS pivot, 4 columns per lane
constexpr int THIN_TAIL = 8;          // the thin kernel takes the subsets inside the last 8 columns
constexpr int TS = PG + 1;            // LDS column stride (doubles): odd, so that lanes reading the
                                      // same row of different columns hit different banks

// One subset, one lane.  tab: the node's tableau, column q (stride STRIDE doubles) = the q-th
// selectable column, column R = the rhs; c[]: the KD chosen columns; U[]: the KD rows not used by
// the prefix, ascending; used: mask of the rows the prefix did use.  PERM: the tableau's rows were
// permuted when it was written — the KD unused rows (ascending) sit at positions 0..KD-1 and the
// used ones at KD..m-1 — so every row index below is a compile-time constant (LDS reads with
// immediate offsets, pairs of rows per instruction) and U / used are ignored; the wave is then
// uniform (one tableau per wave).  Returns 0 feasible, 1 infeasible, 2 singular.
//
// OBJ (list evaluation, records in HBM): every row is finished — no early exit — and obj->z receives
// the objective, summed as the direct kernel sums it (enum_direct.hip: subset_objective): one fma per
// basis column in ascending column order, i.e. the prefix's columns (depth order), then c[0..KD-1].
struct LeafObj {
    const double* cost;           // objective coefficients by original column
    const unsigned char* prow;    // pivot row / column of each prefix depth (NodeMeta)
    const unsigned char* pcol;
    int D;                        // depth of the record
    int col0;                     // original column of tableau column 0
    unsigned used;                // PERM: rows used by the prefix (row i sits at KD + #used rows below i)
    int stride;                   // STRIDE == 0: column stride of the tableau (32-row records: EnumDev::rs)
    double z;
};

// leaf_verdict<PERM>'s row order by (step 0's pivot row p0, step 1's q1): entry p0 * KD + q1 holds the byte
// offsets (8 x row) of the KD-2 other rows, ascending, one per byte.  KD <= 6.
template <int KD>
__device__ __forceinline__ void leaf_row_table(unsigned* tab, int tid) {
    static_assert(KD <= 6, "four rows per 32-bit entry");
    if (tid < KD * KD) {
        const int p0 = tid / KD, q1 = tid - p0 * KD;
        unsigned pk = 0;
        int k = 0;
        for (int r = 0; r < KD; ++r)
            if (r != p0 && r != q1 && k < KD - 2) pk |= (unsigned)(8 * r) << (8 * k++);
        tab[tid] = pk;
    }
}

// 1.0 / x for 2^-500 <= |x| <= 2^500: the instructions the compiler emits for an f64 division without its
// range scaling (v_div_scale_f64 x 2) and special-case fix-up (v_div_fixup_f64), which leave an operand
// in that range untouched (v_div_fmas_f64 is then a plain fma, and the numerator 1.0 makes the quotient
// multiply an identity) — the same bits as `1.0 / x`, 7 instructions instead of 11
// (tests/test_gpu_enum.py::test_leaf_reciprocal_matches_division).  Outside the range: garbage, no trap.
__device__ __forceinline__ double recip_midrange(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double r1 = fma(r0, fma(-x, r0, 1.0), r0);
    const double r2 = fma(r1, fma(-x, r1, 1.0), r1);
    return fma(fma(-x, r2, 1.0), r2, r2);
}
constexpr double kRecipLo = 0x1p-500, kRecipHi = 0x1p500;

// FAST: every division is recip_midrange(pivot); *redo is set if a pivot of THIS function that is a number other
// than zero lay outside its range — whatever the verdict: every later pivot, and with it `sing`, then rests on
// quotients that may differ from the division's — and the caller repeats the pass with FAST = false
// (leaf_verdict below).  Tracked as maxp > 2^500 or max |1/pivot| > 2^500: a zero / NaN pivot has a NaN
// reciprocal, which v_max_f64 drops, and everything behind it is NaN as well (every later `big` stays at its
// initial -1), so the garbage behind a zero pivot — whose verdict, singular, is final and exact — never raises
// the flag; the pivots in front of the first out-of-range one are the division's bits.  (The prefix's pivots
// were divided plainly by the level kernels; a huge one, through maxp0, raises the flag all the same.)
template <int KD, int STRIDE, bool PERM, bool OBJ, bool FAST, bool TAB>
__device__ __forceinline__ int leaf_verdict_impl(const double* tab, const int (&c)[KD], int R, const int (&U)[KD],
                                                 unsigned used, double minp0, double maxp0, int m,
                                                 LeafObj* obj, bool* redo, const unsigned* rowtab) {
    auto recip = [](double x) { return FAST ? recip_midrange(x) : 1.0 / x; };
    // column stride: a template constant, or (STRIDE == 0, list evaluation from 32-row records) obj->stride
    const int S = STRIDE > 0 ? STRIDE : obj->stride;
    // ---- phase 1: unused rows x chosen columns
    double E[KD][KD], H[KD];
    // PERM (tableau in LDS): step 0's pivot row is chosen BEFORE the rows are loaded — the entries
    // are still the tableau's own, so instead of loading rows 0..KD-1 and rotating the chosen one
    // to the front (2 x (KD-1) selects per element: the largest single cost of a subset), columns
    // c[0] and c[1] are read first, and the rows then come from LDS already in their rotated
    // order: position 0 = step 0's row, position 1 = step 1's, then the other rows ascending.
    // (Not for the thin kernel's records in HBM: a second dependent round trip costs more there
    // than the selects.)
    // The same for step 1: its pivot row is the first largest entry of column c[1] AFTER step 0's
    // elimination, among the rows other than p0 — five fmas on the two columns already read (the
    // same operations, operand for operand, that step 0 performs on the loaded matrix below).
    double big0 = -1.0, bigs1 = -1.0, inv0 = 0.0;
    if constexpr (PERM) {
        double col0[KD], col1[KD];
        const double* cz = tab + c[0] * S;
        const double* cy = tab + c[1] * S;
#pragma unroll
        for (int r = 0; r < KD; ++r) {
            col0[r] = cz[r];
            col1[r] = cy[r];
        }
        int p0 = 0;
#pragma unroll
        for (int r = 0; r < KD; ++r) big0 = fmax(big0, fabs(col0[r]));
#pragma unroll
        for (int r = KD - 1; r >= 0; --r) p0 = (fabs(col0[r]) == big0) ? r : p0;   // descending: the first wins
        // the pivot element and the pivot row's entry in column c[1]: read again at the chosen row (two LDS
        // reads instead of 4 x (KD-1) selects)
        const double piv0 = cz[p0], pr01 = cy[p0];
        inv0 = recip(piv0);
        double a1[KD];
#pragma unroll
        for (int r = 0; r < KD; ++r) {
            const double u = fma(-(col0[r] * inv0), pr01, col1[r]);
            // |u|, or for row p0 some value in (-2, -1] (the high word of -1.0 over u's low word: one select
            // instead of two; all that matters is that it is negative, i.e. below every |u| and caught by the
            // `bigs1 < 0` rule below if nothing else is a number)
            a1[r] = __hiloint2double((r == p0) ? (int)0xBFF00000u : (__double2hiint(u) & 0x7FFFFFFF), __double2loint(u));
        }
        // (v_max_f64 itself: the values are |arithmetic results| or the negative marker, never signalling NaNs,
        // which the compiler cannot see behind the word-wise construction and would quiet one by one first)
#pragma unroll
        for (int r = 0; r < KD; ++r) asm("v_max_f64 %0, %1, %2" : "=v"(bigs1) : "v"(bigs1), "v"(a1[r]));
        int q1 = 0;
#pragma unroll
        for (int r = KD - 1; r >= 0; --r) q1 = (a1[r] == bigs1) ? r : q1;
        q1 = (bigs1 < 0.0) ? (p0 == 0 ? 1 : 0) : q1;   // (all NaN: any row other than p0; the subset is singular)
        // byte offsets of the rows in their rotated order: p0, q1, then the others ascending — from the
        // caller's table of all (p0, q1) pairs (leaf_row_table: one LDS read and a bit-field extract per row
        // against ~8 integer instructions per row), or computed
        unsigned off[KD];
        off[0] = (unsigned)p0 * 8u;
        off[1] = (unsigned)q1 * 8u;
        if constexpr (TAB) {
            static_assert(KD <= 6, "leaf_row_table");
            const unsigned pk = rowtab[p0 * KD + q1];
#pragma unroll
            for (int k = 0; k < KD - 2; ++k) off[k + 2] = (pk >> (8 * k)) & 0xFFu;
        } else {
            const int lo = p0 < q1 ? p0 : q1, hi = p0 < q1 ? q1 : p0;
#pragma unroll
            for (int k = 0; k < KD - 2; ++k) {
                int idx = k + (k >= lo ? 1 : 0);
                idx += (idx >= hi) ? 1 : 0;
                off[k + 2] = (unsigned)idx * 8u;
            }
        }
        auto at = [](const double* col, unsigned o) {
            return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(col) + o);
        };
#pragma unroll
        for (int t = 0; t < KD; ++t) {
            const double* col = tab + c[t] * S;
#pragma unroll
            for (int r = 0; r < KD; ++r) E[r][t] = at(col, off[r]);
        }
#pragma unroll
        for (int r = 0; r < KD; ++r) H[r] = at(tab + R * S, off[r]);
    } else {
#pragma unroll
        for (int t = 0; t < KD; ++t) {
            const double* col = tab + c[t] * S;
#pragma unroll
            for (int r = 0; r < KD; ++r) E[r][t] = col[U[r]];
        }
#pragma unroll
        for (int r = 0; r < KD; ++r) H[r] = tab[R * S + U[r]];
    }
    // Invariant: before step t, rows t..KD-1 of E are the rows not yet used, in ascending
    // original order (so "first row of largest |entry|" keeps its meaning), and rows 0..t-1
    // are the pivot rows of steps 0..t-1.  The chosen row p is ROTATED into position t
    // (rows t..p-1 move down by one): afterwards every access below has a static index —
    // no per-element "is this the pivot row" selects, and the 2x2 block is the last two rows.
    double PR[KD - 2][KD], PRH[KD - 2], INV[KD - 2];
    double minp = minp0, maxp = maxp0;
    double maxinv = 0.0;   // FAST: largest |reciprocal of a pivot| of this function
    bool sing = false;
#pragma unroll
    for (int t = 0; t < KD - 2; ++t) {
        // first row of largest |entry| (a NaN entry is never the maximum: fmax drops it)
        double big = -1.0;
        int p = t;
        if (PERM && t <= 1) {
            big = t == 0 ? big0 : bigs1;   // chosen while loading; the pivot rows already sit at positions 0, 1
        } else {
#pragma unroll
            for (int r = t; r < KD; ++r) big = fmax(big, fabs(E[r][t]));
#pragma unroll
            for (int r = KD - 1; r >= t; --r) p = (fabs(E[r][t]) == big) ? r : p;   // descending: the first wins
        }
        // (FAST: no test here — a zero column leaves big = 0, a NaN column -1, so minp <= 0 and the threshold
        // test behind the 2x2 block says singular: one compare per step less in the leaf kernels' hot loop)
        if constexpr (!FAST) {
            if (!(big > 0.0)) sing = true;
        }
        minp = fmin(minp, big);
        maxp = fmax(maxp, big);
        // rotate row p to position t (columns t..KD-1 and the rhs)
        if (!(PERM && t <= 1))
#pragma unroll
        for (int cc = t; cc <= KD; ++cc) {
            double pr = (cc < KD) ? E[t][cc < KD ? cc : 0] : H[t];
#pragma unroll
            for (int r = t + 1; r < KD; ++r) {
                const double v = (cc < KD) ? E[r][cc < KD ? cc : 0] : H[r];
                pr = (r == p) ? v : pr;
            }
#pragma unroll
            for (int r = KD - 1; r > t; --r) {
                if (cc < KD)
                    E[r][cc < KD ? cc : 0] = (r <= p) ? E[r - 1][cc < KD ? cc : 0] : E[r][cc < KD ? cc : 0];
                else
                    H[r] = (r <= p) ? H[r - 1] : H[r];
            }
            if (cc < KD) E[t][cc < KD ? cc : 0] = pr; else H[t] = pr;
        }
        // the pivot element, now in its static position (PERM, step 0: its reciprocal is already there)
        const double inv = (PERM && t == 0) ? inv0 : recip(E[t][t]);
        INV[t] = inv;
        if constexpr (FAST) maxinv = fmax(maxinv, fabs(inv));
#pragma unroll
        for (int cc = t + 1; cc < KD; ++cc) PR[t][cc] = E[t][cc];
        PRH[t] = H[t];
#pragma unroll
        for (int r = 0; r < KD; ++r) {
            if (r == t) continue;
            const double lx = -(E[r][t] * inv);
#pragma unroll
            for (int cc = t + 1; cc < KD; ++cc) E[r][cc] = fma(lx, PR[t][cc], E[r][cc]);
            H[r] = fma(lx, PRH[t], H[r]);
        }
#pragma unroll
        for (int cc = t + 1; cc < KD; ++cc) E[t][cc] = PR[t][cc] * inv;
        H[t] = PRH[t] * inv;
    }
    // ---- 2x2 block on the two rows still unused and the last two chosen columns
    const double a1 = E[KD - 2][KD - 2], a2 = E[KD - 1][KD - 2], b1 = E[KD - 2][KD - 1],
                 b2 = E[KD - 1][KD - 1], h1 = H[KD - 2], h2 = H[KD - 1];
    const bool second = fabs(a2) > fabs(a1);
    const double pa = second ? a2 : a1, pb = second ? b2 : b1, ph = second ? h2 : h1;
    const double qa = second ? a1 : a2, qb = second ? b1 : b2, qh = second ? h1 : h2;
    const double big1 = fabs(pa);
    const double inv1 = recip(pa);
    const double l = -(qa * inv1);
    const double wqb = fma(l, pb, qb);
    const double rq = fma(l, ph, qh);
    const double big2 = fabs(wqb);
    const double inv2 = recip(wqb);
    const double xb = rq * inv2;
    const double xa = fma(-pb, xb, ph) * inv1;
    if (!(big1 > 0.0) || !(big2 > 0.0)) sing = true;
    minp = fmin(minp, fmin(big1, big2));
    maxp = fmax(maxp, fmax(big1, big2));
    if (minp <= DBL_EPSILON * (double)m * maxp) sing = true;
    if constexpr (FAST) {
        maxinv = fmax(maxinv, fmax(fabs(inv1), fabs(inv2)));
        *redo = maxp > kRecipHi || maxinv > 1.0 / kRecipLo;
    }
    bool feas = (xa >= -1e-9) && (xb >= -1e-9);
    // the rows pivoted in phase 1: back-substitution
    double xr[OBJ ? KD - 2 : 1];
#pragma unroll
    for (int r = 0; r < KD - 2; ++r) {
        const double x = fma(-E[r][KD - 1], xb, fma(-E[r][KD - 2], xa, H[r]));
        feas = feas && (x >= -1e-9);
        if constexpr (OBJ) xr[r] = x;
    }
    // ---- phase 2: rows already used by the prefix, one at a time
    // (a subset survives phase 1 with probability ~2^-8, so the loop below usually ends
    // after a row or two: it stops as soon as no lane of the wave is still feasible)
    bool alive = !sing && feas;
    unsigned rows = used & (m >= 32 ? ~0u : ((1u << m) - 1u));
    auto one_row = [&](int i, bool has) {
        double v[KD];
#pragma unroll
        for (int t = 0; t < KD; ++t) v[t] = tab[c[t] * S + i];
        double h = tab[R * S + i];
#pragma unroll
        for (int t = 0; t < KD - 2; ++t) {
            const double lx = -(v[t] * INV[t]);
#pragma unroll
            for (int cc = t + 1; cc < KD; ++cc) v[cc] = fma(lx, PR[t][cc], v[cc]);
            h = fma(lx, PRH[t], h);
        }
        const double x = fma(-v[KD - 1], xb, fma(-v[KD - 2], xa, h));
        feas = feas && (!has || x >= -1e-9);
        alive = alive && feas;
        return x;
    };
    if constexpr (OBJ) {
        double z = 0.0;
        // (unrolled: the rows' loads of four prefix depths are in flight together — the list evaluation is one
        // lane per entry and nothing but dependent round trips to the records in HBM)
#pragma unroll 4
        for (int k = 0; k < obj->D; ++k) {
            int i = obj->prow[k];
            if constexpr (PERM) i = KD + __builtin_popcount(obj->used & ((1u << i) - 1u));
            z = fma(obj->cost[obj->pcol[k]], one_row(i, true), z);
        }
#pragma unroll
        for (int r = 0; r < KD - 2; ++r) z = fma(obj->cost[obj->col0 + c[r]], xr[r], z);
        z = fma(obj->cost[obj->col0 + c[KD - 2]], xa, z);
        z = fma(obj->cost[obj->col0 + c[KD - 1]], xb, z);
        obj->z = z;
    } else if constexpr (PERM) {
        for (int i = KD; i < m && __any(alive); ++i) one_row(i, true);
    } else {
        while (__any(alive && rows != 0u)) {
            const bool has = rows != 0u;
            const int i = has ? __builtin_ctz(rows) : 0;
            rows &= rows - 1u;
            one_row(i, has);
        }
    }
    return sing ? 2 : (feas ? 0 : 1);
}

// FAST: the leaf kernels' hot form.  A subset whose pivots leave recip_midrange's range (and that is not
// singular anyway) raises *viol; the kernel then sets EnumResult::range_flag, and the host repeats the pass
// on the EXACT instantiations of the kernels (plain divisions; enum_prefix.hip: lp_enum_prefix_range).  Only
// a problem scaled to ~1e-150 or ~1e150 as a whole gets there: a non-singular subset's pivots lie within
// 1/(eps m) = 2^48 of each other.  (Repeating the one subset in place — the exact form inlined behind the
// fast one, or called — cost the leaf loops 50-200 bytes of spills and the thin kernel a wave per SIMD.)
template <int KD, int STRIDE, bool PERM, bool OBJ = false, bool FAST = false, bool TAB = false>
__device__ __forceinline__ int leaf_verdict(const double* tab, const int (&c)[KD], int R, const int (&U)[KD],
                                            unsigned used, double minp0, double maxp0, int m,
                                            LeafObj* obj = nullptr, unsigned* viol = nullptr,
                                            const unsigned* rowtab = nullptr) {
    static_assert(!(OBJ && FAST), "the list evaluation divides plainly");
    bool redo = false;
    const int v = leaf_verdict_impl<KD, STRIDE, PERM, OBJ, FAST, TAB>(tab, c, R, U, used, minp0, maxp0, m, obj, &redo, rowtab);
    if constexpr (FAST) *viol |= redo ? 1u : 0u;
    return v;
}

#ifdef LP_LEAF_WAVELOG
// Diagnostic build (scripts/leaf_wavelog.py): every wave of the leaf kernels leaves {kernel, hardware id, start,
// first item, end (100 MHz clock), items, steals}; read back through lp_debug_leaf_wavelog.
constexpr int kWaveLogCap = 1 << 16;
__device__ unsigned long long g_wavelog[kWaveLogCap][6];
__device__ unsigned int g_wavelog_n;
__device__ __forceinline__ void wavelog_put(int kernel, unsigned long long t0, unsigned long long t1, unsigned int items,
                                            unsigned int steals) {
    if ((threadIdx.x & 63) == 0) {
        const unsigned int k = atomicAdd(&g_wavelog_n, 1u);
        if (k < (unsigned)kWaveLogCap) {
            unsigned int hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_wavelog[k][0] = (unsigned long long)kernel | ((unsigned long long)blockIdx.x << 8);
            g_wavelog[k][1] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            g_wavelog[k][2] = t0;
            g_wavelog[k][3] = t1;
            g_wavelog[k][4] = __builtin_amdgcn_s_memrealtime();
            g_wavelog[k][5] = (unsigned long long)items | ((unsigned long long)steals << 32);
        }
    }
}
#endif

constexpr int kChunk = 1024;  // subsets per work item (16 wave passes)
// lanes of k_enum_make_items that share one record: 4 on wide levels, 8 on the narrow ones of a small
// rank range (there the kernel's run time is the length of one lane's loop)
constexpr int kItemLanesWide = 4, kItemLanesNarrow = 8;

// Work items of the leaf kernels, so that no item is longer than 16 wave passes — a depth m-6 node
// can hold up to C(22,6) = 74,613 subsets, and a rank-range shard of an 8-GPU run is only a few
// milliseconds of work in total.  One lane per record; slots in the item tables are allocated with
// one atomic per wave and table.  FUSED: records are depth m-7 nodes; for each child a (with at
// least min_child_R selectable columns; the thin kernel takes the rest) whose subsets meet
// [begin, end):
//   table 1 (k_enum_leaves<3>): the child's subsets are grouped by their first remaining column
//     j2 (lexicographic order); every group that leaves at least kGrandMin selectable columns is
//     cut into items (record, a | j2 << 8 | groups << 16, first subset of the chunk inside the
//     group, rank offset of the group inside the record) — a large group in chunks of kChunk, or
//     several consecutive small groups packed into one item;
//   table 0 (k_enum_leaves<1>): the child's remaining subsets — (record, a, first subset of the
//     chunk inside the child, rank offset of the child inside the record).
// !FUSED (m = 6): the one record is the depth m-6 root itself, table 0 only.
template <bool FUSED>
__global__ __launch_bounds__(1024) void k_enum_make_items(EnumDev d, PrefixDev pd,
                                                         const double* __restrict__ roots,
                                                         int root_level, int root_cap, int min_child_R,
                                                         int kItemLanes, unsigned long long begin,
                                                         unsigned long long end) {
    constexpr int KD = 6;
    // C(r, 5) and C(r, 6) for r <= NMX + KD + 1, from LDS (the loops below are chains of dependent
    // lookups; out of L2 they were half of this kernel's time)
    __shared__ unsigned int s_b5[NMX + KD + 2], s_b6[NMX + KD + 2];
    if (threadIdx.x < NMX + KD + 2) {
        s_b5[threadIdx.x] = (unsigned int)d.binom[threadIdx.x * kBinomK + KD - 1];
        s_b6[threadIdx.x] = (unsigned int)d.binom[threadIdx.x * kBinomK + KD];
    }
    __syncthreads();
    // s_n1[R]: table-1 items of a child with R selectable columns whose whole rank interval lies inside
    // [begin, end) — the count pass below then costs such a child one lookup instead of the group loop
    // (nearly every child of a pass is one; the count pass was ~40 % of this kernel, which is 10 % of
    // an 8-way shard's run time)
    __shared__ int s_n1[NMX + KD + 2];
    if (threadIdx.x < NMX + KD + 2) {
        const int R = threadIdx.x;
        int n1 = 0, pack_ng = 0;
        unsigned long long pack_n = 0;
        for (int j2 = 0; R - 1 - j2 >= kGrandMin; ++j2) {
            const unsigned long long cnt2 = s_b5[R - 1 - j2];
            if (cnt2 >= (unsigned long long)kChunk) {
                n1 += pack_ng ? 1 : 0;
                pack_ng = 0;
                pack_n = 0;
                n1 += (int)((cnt2 + kChunk - 1) / kChunk);
            } else {
                if (pack_n + cnt2 > (unsigned long long)kChunk) {
                    n1 += pack_ng ? 1 : 0;
                    pack_ng = 0;
                    pack_n = 0;
                }
                ++pack_ng;
                pack_n += cnt2;
            }
        }
        s_n1[R] = n1 + (pack_ng ? 1 : 0);
    }
    __syncthreads();
    const int n = d.n, m = d.m, D = m - KD - (FUSED ? 1 : 0);
    const int nroots = min(pd.level_counts[root_level], root_cap);
    // kItemLanes lanes share a record: lane `sub` takes the children sub, sub + kItemLanes, ... (the
    // order of the items in the tables does not matter)
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int rec = FUSED ? gid / kItemLanes : gid, sub = FUSED ? gid % kItemLanes : 0;
    const int lane = threadIdx.x & 63;
    int last = kHole;
    unsigned long long rb0 = 0ULL;
    if (rec < nroots) {
        const NodeMeta* pm = reinterpret_cast<const NodeMeta*>(roots + (size_t)rec * rec_doubles(n, D) +
                                                               (size_t)PG * (n - D + 1));
        last = pm->last_col;
        rb0 = pm->rank_base;
    }
    // chunks of [lo0, hi0) (subset indices relative to `base`) whose rank interval meets [begin, end)
    auto chunks = [&](unsigned long long base, unsigned long long lo0, unsigned long long hi0, auto&& f) {
        for (unsigned long long lo = lo0; lo < hi0; lo += kChunk)
            if (overlap(base + lo, (hi0 - lo < kChunk) ? hi0 - lo : (unsigned long long)kChunk, begin, end)) f((int)lo);
    };
    // one child a (R selectable columns, L subsets from rank rb) of this lane's share
    auto child = [&](int a, int R, unsigned long long L, unsigned long long rb, int lim, auto&& emit) {
        // groups j2 = 0, 1, ... with at least kGrandMin columns left; consecutive small
        // groups are packed into one item (the child pivot is paid once per item)
        unsigned long long off2 = 0, pack_off = 0, pack_n = 0;
        int pack_j2 = 0, pack_ng = 0;
        auto flush = [&]() {
            if (pack_ng && overlap(rb + pack_off, pack_n, begin, end))
                emit(1, a | (pack_j2 << 8) | (pack_ng << 16), 0, (int)(rb + pack_off - rb0));
            pack_ng = 0;
            pack_n = 0;
        };
        for (int j2 = 0; R - 1 - j2 >= kGrandMin; ++j2) {
            const unsigned long long cnt2 = s_b5[R - 1 - j2];
            if (cnt2 >= (unsigned long long)kChunk) {
                flush();
                chunks(rb + off2, 0, cnt2,
                       [&](int lo) { emit(1, a | (j2 << 8) | (1 << 16), lo, (int)(rb + off2 - rb0)); });
            } else {
                if (pack_n + cnt2 > (unsigned long long)kChunk) flush();
                if (pack_ng == 0) {
                    pack_j2 = j2;
                    pack_off = off2;
                }
                ++pack_ng;
                pack_n += cnt2;
            }
            off2 += cnt2;
        }
        flush();
        // table 0: what is left of the child, [off2, L) — at most C(kGrandMin, 6) subsets.
        // Children 2k and 2k+1 of a record share one item (k_enum_leaves<1> pivots both
        // from the one record and fills its passes from both tails), emitted by the even one.
        const bool odd = (a - last - 1) & 1;
        const bool mine = overlap(rb + off2, L - off2, begin, end) != 0ULL;
        bool partner = false;   // does the other child of the pair exist and overlap?
        if (!odd && a + 1 <= lim && n - 2 - a >= min_child_R) {
            const int Rn = R - 1;
            unsigned long long offn = 0;
            for (int j2 = 0; Rn - 1 - j2 >= kGrandMin; ++j2) offn += s_b5[Rn - 1 - j2];
            partner = overlap(rb + L + offn, s_b6[Rn] - offn, begin, end) != 0ULL;
        }
        if (odd) {
            // (its tail rides in the even child's item: that one is emitted whenever either
            // tail meets the range)
        } else if (mine || partner) {
            emit(0, a | (partner ? 1 << 16 : 0), (int)off2, (int)(rb - rb0));
        }
    };
    // COUNT: only the number of items per table is wanted — a child whose whole interval lies inside
    // the range takes its table-1 count from s_n1 and (even children) one table-0 item
    auto visit = [&](auto&& emit, auto count_only) {   // emit(table, a_field, first subset, rank offset)
        if (last == kHole) return;
        if (FUSED) {
            unsigned long long rb = rb0;
            const int lim = n - m + D;  // largest column selectable at depth D
            for (int a = last + 1; a <= lim; ++a) {
                const int R = n - 1 - a;
                const unsigned long long L = s_b6[R];
                if (R >= min_child_R && ((a - last - 1) >> 1) % kItemLanes == sub) {
                    if (decltype(count_only)::value && rb >= begin && rb + L <= end) {
                        for (int k = 0; k < s_n1[R]; ++k) emit(1, 0, 0, 0);
                        if (!((a - last - 1) & 1)) emit(0, 0, 0, 0);
                    } else {
                        child(a, R, L, rb, lim, emit);
                    }
                }
                rb += L;
            }
        } else {
            chunks(rb0, 0, s_b6[n - 1 - last], [&](int lo) { emit(0, last, lo, 0); });
        }
    };
    int nch[2] = {0, 0};
    visit([&](int tab, int, int, int) { ++nch[tab]; }, std::true_type{});
    // slots: wave scan, then ONE atomic per block and table (a returning atomic on one word costs
    // ~11 ns chip-wide; at one per wave they were most of this kernel's time)
    __shared__ int s_wave_total[2][16], s_block_base[2];
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    int at[2];
#pragma unroll
    for (int tab = 0; tab < 2; ++tab) {
        int incl = nch[tab];   // inclusive wave scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wave_total[tab][wave] = incl;
        at[tab] = incl - nch[tab];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int tab = threadIdx.x;
        int total = 0;
        for (int w = 0; w < nwaves; ++w) {
            const int t = s_wave_total[tab][w];
            s_wave_total[tab][w] = total;   // exclusive prefix over the waves
            total += t;
        }
        s_block_base[tab] = total ? atomicAdd(&pd.item_count[tab], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int tab = 0; tab < 2; ++tab) at[tab] += s_block_base[tab] + s_wave_total[tab][wave];
    visit([&](int tab, int a, int lo, int roff) {
        int4* items = tab ? pd.items2 : pd.items;
        const int cap = tab ? pd.item_cap2 : pd.item_cap;
        const int k = tab ? at[1]++ : at[0]++;
        if (k < cap) items[k] = make_int4(rec, a, lo, roff);
    }, std::false_type{});
}

// MODE 1: work items of table 0 over the depth m-7 records; the wave pivots on the child column
//         itself (depth m-6 tableau in LDS), the lanes take the 6 remaining columns.
// MODE 2: work items of table 1; the wave pivots twice (child, then the group's first column j2:
//         depth m-5 tableau in LDS) and the lanes take the 5 remaining columns — consecutive
//         subsets share their first remaining column and with it the whole first elimination
//         step, which costs a lane a third of its instructions.  A kernel of its own so that the
//         register allocation of MODE 1's loop is not disturbed (the two loops in one kernel cost
//         that loop 12 %).
// MODE 3: MODE 2's items; inside a group the subsets are in lexicographic order again, so those that share their
//         first remaining column j3 are consecutive — C(R2 - 1 - j3, 4) of them.  Every such sub-group that leaves
//         >= kGreatMin selectable columns (half of all subsets of C(32,16)) gets a THIRD in-LDS pivot and its lanes
//         take 4 columns each (two Gauss-Jordan steps whose rows are chosen before they are loaded, no register
//         rotation at all, then the 2x2 block: ~200 instructions per subset against ~320 with 5 columns); what is
//         left of the group is finished with 5 columns as in MODE 2.  The great-grandchild tableau lives in the
//         half of the record's double buffer that the NEXT item's record will be written to (free until then).
// MODE 0 (m = 6): the root record is the depth m-6 node.
template <int MODE, bool EXACT>
__global__ __launch_bounds__(LEAF_THREADS) __attribute__((amdgpu_waves_per_eu(3)))
void k_enum_leaves(EnumDev d, PrefixDev pd, const double* __restrict__ roots, unsigned long long begin,
                   unsigned long long end) {
    constexpr bool FUSED = MODE != 0;
    constexpr int KD = 6;
    constexpr int MAXCOLS = NMX + KD + 1 + (FUSED ? 1 : 0);  // columns of a record (+ rhs)
    constexpr int CHILDCOLS = NMX + KD + 1;                  // columns of a depth m-6 tableau (+ rhs)
    __shared__ __attribute__((aligned(16))) double s_tab[LEAF_WAVES * 2][MAXCOLS * TS];  // double-buffered
    __shared__ __attribute__((aligned(16))) double s_child[FUSED ? LEAF_WAVES : 1][FUSED ? CHILDCOLS * TS : 1];
    __shared__ __attribute__((aligned(16))) double s_child2[MODE == 1 ? LEAF_WAVES : 1][MODE == 1 ? CHILDCOLS * TS : 1];
    __shared__ __attribute__((aligned(16))) double s_grand[MODE >= 2 ? LEAF_WAVES : 1][MODE >= 2 ? (NMX + KD) * TS : 1];
    __shared__ unsigned int s_binom[(NMX + KD + 2) * (KD + 1)];  // C(r, k), r <= NMX+KD+1, k <= KD
    __shared__ unsigned long long s_cnt[3];
    __shared__ unsigned int s_off[32];  // offsets of the per-R subset tables inside pd.comb6
    __shared__ unsigned int s_rows[36];  // leaf_row_table of this kernel's lanes (5 columns each in MODE 2 / 3, else 6)
    __shared__ unsigned int s_off4[MODE == 3 ? 32 : 1], s_rows4[MODE == 3 ? 16 : 1];   // MODE 3: the same for 4 columns
    __shared__ unsigned long long s_run[LEAF_WAVES];   // every wave's run of work items {next, end}: see draw() below
    __shared__ int s_dry;                              // a wave of this workgroup has seen the end of the item table

    const int m = d.m, n = d.n, D = m - KD - (FUSED ? 1 : 0);   // depth of the records
    const int4* const items = MODE >= 2 ? pd.items2 : pd.items;   // built by k_enum_make_items
    const int nitems = MODE >= 2 ? min(pd.item_count[1], pd.item_cap2) : min(pd.item_count[0], pd.item_cap);
    int* const cursor = pd.root_cursor + (MODE >= 2 ? 1 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef LP_LEAF_WAVELOG
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wl_t1 = 0;
    unsigned int wl_items = 0, wl_steals = 0;
#endif
    for (int k = tid; k < (NMX + KD + 2) * (KD + 1); k += LEAF_THREADS) {
        const int r = k / (KD + 1), kk = k - r * (KD + 1);
        s_binom[k] = (unsigned int)d.binom[r * kBinomK + kk];
    }
    if (tid < 3) s_cnt[tid] = 0ULL;
    if (tid < 32) s_off[tid] = MODE >= 2 ? pd.comb5[tid] : pd.comb6[tid];
    leaf_row_table<(MODE >= 2 ? 5 : 6)>(s_rows, tid);
    if constexpr (MODE == 3) {
        if (tid < 32) s_off4[tid] = pd.comb4[tid];
        leaf_row_table<4>(s_rows4, tid);
    }
    __syncthreads();
    unsigned int cntF = 0, cntI = 0, cntS = 0;
    unsigned int viol = 0;   // !EXACT: a pivot left the fast reciprocal's range (leaf_verdict)

    // Software pipeline over work items: while item A is computed from LDS slice `buf`, the whole
    // record of item B (its index was drawn one iteration earlier) is in flight into registers,
    // and the index of item C is being drawn — neither the atomic nor the HBM round trip of a
    // record sits between two items.
    constexpr int NLOAD = (MAXCOLS * PG + 63) / 64;   // doubles per lane to hold one record
    const int rec_cols = n - D + 1;                   // columns of a record incl. rhs
    // Items are dealt in runs of kRun: one returning atomic on a single word costs ~11 ns chip-wide, which at one
    // draw per item bounds the kernel.  The first deal is static — wave w owns items [w*k0, (w+1)*k0) — so that
    // the launch does not open with two atomics per wave on that word (~0.2 ms for a full grid).
    // A wave's run lives in LDS as one 64-bit word {next item, end of the run}, and items leave it by compare-and-
    // swap: when the table is dealt out, a wave that has nothing left takes items out of the runs of the other
    // waves of its workgroup (a workgroup's wave slots and LDS are free for the next kernel's blocks only when its
    // LAST wave ends).  That is what lets the runs stay at kRun to the end: rounds 1-3 shrank them with what was
    // left (guided self-scheduling), and the last ~4 items per wave were then drawn one by one — 12 k single draws
    // and 3 k failing ones queued on the one word at the end of each leaf kernel, ~0.1 ms of every pass whatever
    // its size (scripts/leaf_wavelog.py: on an 8-way shard of C(32,16) the waves of k_enum_leaves<1> lived 9 %
    // longer per item than on the whole range).  A wave that sees the end of the table tells its siblings.
    constexpr int kRun = 4;
    const int nwaves = (int)gridDim.x * LEAF_WAVES;
    const int k0 = max(1, min(kRun, nitems / (nwaves * 4)));
    const int dyn_base = nwaves * k0;
    auto pack_run = [](int next, int end) { return ((unsigned long long)(unsigned)next << 32) | (unsigned)end; };
    {
        const int first = ((int)blockIdx.x * LEAF_WAVES + wave) * k0;
        if (lane == 0) s_run[wave] = pack_run(min(first, nitems), min(first + k0, nitems));
        if (tid == 0) s_dry = 0;
    }
    __syncthreads();
    auto take = [&](int w) {   // the next item of wave w's run, -1 if it is empty
        int item = -1;
        if (lane == 0) {
            unsigned long long cur = __hip_atomic_load(&s_run[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((unsigned)(cur >> 32) < (unsigned)cur) {
                if (__hip_atomic_compare_exchange_strong(&s_run[w], &cur, cur + (1ULL << 32), __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    item = (int)(cur >> 32);
                    break;
                }
            }
        }
        return __builtin_amdgcn_readfirstlane(item);
    };
    bool dry = dyn_base >= nitems;   // the table is dealt out
    bool none = false;               // this wave has been told "nothing left" (final: the loop below ends on the first
                                     // such answer, so a later, luckier draw — a sibling's last run arriving — would be lost)
    auto draw = [&]() {
        if (none) return nitems;
        int item = take(wave);
        if (item >= 0) return item;
        if (!dry && __hip_atomic_load(&s_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) dry = true;
        if (!dry) {
            constexpr int k = kRun;
            int v = 0, over = 0;
            if (lane == 0) {
                v = atomicAdd(cursor, k);
                // a list far beyond its capacity (a degenerate LP): the caller switches to the dense form
                // whatever else this pass finds, so the waves stop drawing (the load rides with the atomic)
                over = __hip_atomic_load(pd.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pd.list_abort;
            }
            v = __builtin_amdgcn_readfirstlane(v) + dyn_base;
            if (__builtin_amdgcn_readfirstlane(over)) v = nitems;
            if (v < nitems) {
                if (k > 1 && lane == 0)
                    __hip_atomic_store(&s_run[wave], pack_run(v + 1, min(v + k, nitems)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                return v;
            }
            dry = true;
            if (lane == 0) __hip_atomic_store(&s_dry, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        for (int q = 1; q < LEAF_WAVES; ++q) {
            item = take((wave + q) & (LEAF_WAVES - 1));
#ifdef LP_LEAF_WAVELOG
            if (item >= 0) ++wl_steals;
#endif
            if (item >= 0) return item;
        }
        none = true;
        return nitems;
    };
    double pre[NLOAD];
    NodeMeta pmB;
    int chunkB = 0, childB = 0, roffB = 0, recB = 0;
    auto fetch = [&](int item) {   // issue the loads of item's record (no use of the data here)
        const int4 it = items[item];
        recB = it.x;
        childB = it.y;
        chunkB = it.z;
        roffB = it.w;
        const double* Q = roots + (size_t)it.x * rec_doubles(n, D);
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int k = lane + 64 * q;
            pre[q] = (k < rec_cols * PG) ? Q[k] : 0.0;
        }
        pmB = *reinterpret_cast<const NodeMeta*>(Q + (size_t)PG * rec_cols);
    };
    int itemB = draw();
    int itemC = draw();
    if (itemB < nitems) fetch(itemB);
    int buf = 0;
    for (;;) {
        if (itemB >= nitems) break;
#ifdef LP_LEAF_WAVELOG
        if (!wl_items) wl_t1 = __builtin_amdgcn_s_memrealtime();
        ++wl_items;
#endif
        // ---- item B becomes the current item: registers -> LDS slice (odd column stride)
        const int chunk = chunkB, child = childB, roff = roffB, rec = recB;
        const NodeMeta pm = pmB;
        double* tab = s_tab[wave * 2 + buf];
#pragma unroll
        for (int q = 0; q < NLOAD; ++q) {
            const int k = lane + 64 * q;
            if (k < rec_cols * PG) tab[(k >> 4) * TS + (k & 15)] = pre[q];
        }
        buf ^= 1;
        itemB = itemC;
        itemC = draw();
        if (itemB < nitems) fetch(itemB);
        if (pm.last_col == kHole) continue;
        const int child_col = FUSED ? (child & 0xFF) : pm.last_col;   // last chosen column of the depth m-6 node
        const int j2_first = (child >> 8) & 0xFF;      // MODE 2: first remaining column of the (first) group
        const int ngroups = child >> 16;               // MODE 2: groups packed into this item
        const int last = child_col;
        const int R = n - 1 - last;                    // selectable columns of the depth m-6 node
        if (R < KD) continue;
        // subsets of this item: [leaf_lo, leaf_hi) inside the node (MODE 0/1) or inside the group (MODE 2)
        const unsigned int L = s_binom[R * (KD + 1) + KD];   // (MODE 0/1)
        // subset table: all 6-subsets of R columns in lexicographic order, 5 bits per index
        // (one L2-resident load; a dependent unranking loop over binomials costs ~2k cycles)
        const unsigned* comb = pd.comb6 + s_off[MODE >= 2 ? 0 : R];
        unsigned long long rb = pm.rank_base + (unsigned long long)(FUSED ? roff : 0);
        const unsigned int leaf_lo = (unsigned int)chunk;
        const unsigned int leaf_hi = (leaf_lo + kChunk < L) ? leaf_lo + kChunk : L;
        double minp0 = pm.minp, maxp0 = pm.maxp;
        unsigned umask = __builtin_amdgcn_readfirstlane(pm.used_mask);
        bool sing0 = false;   // MODE 1: the first child of a paired item turned out singular
        if (FUSED) {
            // ---- the wave pivots on column `child`: lane = (row r, column group g), exactly the
            // arithmetic of k_enum_expand (first unused row of largest |w|; l = -(w/piv) as w * (1/piv))
            const int r = lane & (PG - 1), g = lane >> 4;
            const bool row_used = (r >= m) || ((umask >> r) & 1u);
            const double* pcol = tab + (child_col - D) * TS;
            const double w = pcol[r];
            double big;
            const int p = __builtin_amdgcn_readfirstlane(pick_pivot_row(w, row_used, lane & ~(PG - 1), big));
            minp0 = fmin(minp0, big);
            maxp0 = fmax(maxp0, big);
            if (!(big > 0.0) || minp0 <= DBL_EPSILON * (double)m * maxp0) {
                // every subset below this node is singular
                unsigned int span = leaf_hi - leaf_lo;
                if (MODE >= 2) {
                    span = 0;
                    for (int gi = 0; gi < ngroups; ++gi) span += s_binom[(R - 1 - j2_first - gi) * (KD + 1) + KD - 1];
                    if (ngroups == 1) span = min(span - leaf_lo, (unsigned int)kChunk);
                }
                if (lane == 0) cntS += (unsigned int)overlap(rb + leaf_lo, span, begin, end);
                if (MODE == 1 && ((child >> 16) & 1))
                    sing0 = true;   // the item's second child is still to do
                else
                    continue;
            }
            const double inv = 1.0 / pcol[p];
            const bool isp = (r == p);
            const double lx = isp ? inv : -(w * inv);
            umask |= 1u << p;
            // The child's rows are written PERMUTED: the 6 rows still unused (ascending) at positions
            // 0..5, the used ones (ascending) behind them — leaf_verdict<PERM> then reads every row at
            // a compile-time offset.
            // (`below` is recomputed per item — the opaque copy of r keeps it from being hoisted out of the item
            // loop, where it was spilled to scratch and reloaded in front of the row permutation)
            unsigned rr = (unsigned)r;
            asm volatile("" : "+v"(rr));
            const unsigned all = (1u << m) - 1u, below = (1u << rr) - 1u, freem = ~umask & all;
            const int pos = (r >= m) ? r
                            : ((freem >> r) & 1u) ? __builtin_popcount(freem & below)
                                                  : __builtin_popcount(freem) + __builtin_popcount(umask & all & below);
            double* ctab = s_child[wave];
#pragma unroll
            for (int q = 0; q < (CHILDCOLS + 3) / 4; ++q) {
                const int j = g + 4 * q;            // child column j = column child+1+j (j = R: rhs)
                if (j <= R && !sing0) {
                    const double* pc = tab + (child_col + 1 + j - D) * TS;
                    ctab[j * TS + pos] = fma(lx, pc[p], isp ? -0.0 : pc[r]);
                }
            }
            tab = ctab;
        } else {
            tab += (last + 1 - D) * TS;              // column q below = column last+1+q
            // (m = 6: the root record, no row used yet — the identity is the permuted order)
        }
        if constexpr (MODE >= 2) {
            constexpr int K5 = KD - 1;
            const double minp1 = minp0, maxp1 = maxp0;
            for (int gi = 0; gi < ngroups; ++gi) {
            // ---- second pivot, on column j2 of the (row-permuted) child tableau: positions 0..5 are
            // the unused rows in ascending original order, so "first row of largest |w|" keeps its meaning
            const int j2 = j2_first + gi;
            const int R2 = R - 1 - j2;                 // selectable columns of the depth m-5 node
            const unsigned int L2 = s_binom[R2 * (KD + 1) + K5];
            const unsigned int lo2 = leaf_lo;          // (0 unless the item is a chunk of one large group)
            const unsigned int hi2 = (lo2 + kChunk < L2) ? lo2 + kChunk : L2;
            const unsigned long long rb2 = rb;
            rb += L2;                                  // next group of a packed item
            if (overlap(rb2 + lo2, hi2 - lo2, begin, end) == 0ULL) continue;
            const int r = lane & (PG - 1), g = lane >> 4;
            const double* pcol = tab + j2 * TS;
            const double w = pcol[r];
            double big;
            const int p2 = __builtin_amdgcn_readfirstlane(pick_pivot_row(w, r >= KD, lane & ~(PG - 1), big));
            const double minp2 = fmin(minp1, big), maxp2 = fmax(maxp1, big);
            if (!(big > 0.0) || minp2 <= DBL_EPSILON * (double)m * maxp2) {
                if (lane == 0) cntS += (unsigned int)overlap(rb2 + lo2, hi2 - lo2, begin, end);
                continue;
            }
            const double inv = 1.0 / pcol[p2];
            const bool isp = (r == p2);
            const double lx = isp ? inv : -(w * inv);
            // rows of the grandchild: the 5 unused ones at 0..4, the new pivot row at 5, the rest stay
            const int pos = (r < p2) ? r : (r == p2) ? K5 : (r < KD) ? r - 1 : r;
            double* gtab = s_grand[wave];
#pragma unroll
            for (int q = 0; q < (NMX + KD + 3) / 4; ++q) {
                const int j = g + 4 * q;        // grandchild column j = child column j2+1+j (j = R2: rhs)
                if (j <= R2) {
                    const double* pc = tab + (j2 + 1 + j) * TS;
                    gtab[j * TS + pos] = fma(lx, pc[p2], isp ? -0.0 : pc[r]);
                }
            }
            const unsigned* comb5 = pd.comb5 + s_off[R2];
            const int U5[K5] = {0, 1, 2, 3, 4};        // unused by leaf_verdict<PERM>
            unsigned int tail_lo = lo2;                // MODE 3: the leaves in front of it went to sub-groups
            if constexpr (MODE == 3) {
                constexpr int K4 = KD - 2;
                double* ggtab = s_tab[wave * 2 + buf];  // (the next item's slice: written at the top of the next iteration)
                unsigned int off3 = 0;                  // first leaf of sub-group j3 inside the group
                for (int j3 = 0; R2 - 1 - j3 >= kGreatMin && off3 < hi2; ++j3) {
                    const int R3 = R2 - 1 - j3;         // selectable columns of the depth m-4 node
                    const unsigned int L3 = s_binom[R3 * (KD + 1) + K4];
                    const unsigned int a3 = off3 > lo2 ? off3 : lo2, b3 = off3 + L3 < hi2 ? off3 + L3 : hi2;
                    const unsigned int base3 = off3;
                    off3 += L3;
                    tail_lo = off3 > lo2 ? off3 : lo2;
                    if (a3 >= b3 || overlap(rb2 + a3, b3 - a3, begin, end) == 0ULL) continue;
                    // ---- third pivot, on column j3 of the grandchild tableau (unused rows at positions 0..4, ascending)
                    const double* pcol3 = gtab + j3 * TS;
                    const double w3 = pcol3[r];
                    double big3;
                    const int p3 = __builtin_amdgcn_readfirstlane(pick_pivot_row(w3, r >= K5, lane & ~(PG - 1), big3));
                    const double minp3 = fmin(minp2, big3), maxp3 = fmax(maxp2, big3);
                    if (!(big3 > 0.0) || minp3 <= DBL_EPSILON * (double)m * maxp3) {
                        if (lane == 0) cntS += (unsigned int)overlap(rb2 + a3, b3 - a3, begin, end);
                        continue;
                    }
                    const double inv3 = 1.0 / pcol3[p3];
                    const bool isp3 = (r == p3);
                    const double lx3 = isp3 ? inv3 : -(w3 * inv3);
                    // rows: the 4 unused ones at 0..3, the new pivot row at 4, the rest stay
                    const int pos3 = (r < p3) ? r : (r == p3) ? K4 : (r < K5) ? r - 1 : r;
#pragma unroll
                    for (int q = 0; q < (NMX + KD + 2) / 4; ++q) {
                        const int j = g + 4 * q;        // column j = grandchild column j3+1+j (j = R3: rhs)
                        if (j <= R3) {
                            const double* pc = gtab + (j3 + 1 + j) * TS;
                            ggtab[j * TS + pos3] = fma(lx3, pc[p3], isp3 ? -0.0 : pc[r]);
                        }
                    }
                    const unsigned* comb4 = pd.comb4 + s_off4[R3];
                    const int U4[K4] = {0, 1, 2, 3};    // unused by leaf_verdict<PERM>
                    for (unsigned int leaf = a3 + lane; leaf < b3; leaf += 64) {
                        const unsigned long long rank = rb2 + leaf;
                        if (rank < begin || rank >= end) continue;
                        int c4[K4];
                        const unsigned pk = comb4[leaf - base3];
#pragma unroll
                        for (int t = 0; t < K4; ++t) c4[t] = (int)((pk >> (5 * t)) & 31u);
                        const int verdict = leaf_verdict<K4, TS, true, false, !EXACT, true>(ggtab, c4, R3, U4, 0u, minp3, maxp3, m, nullptr, &viol, s_rows4);
                        if (verdict == 2) {
                            ++cntS;
                        } else if (verdict == 1) {
                            ++cntI;
                        } else {
                            ++cntF;
                            const unsigned long long at = atomicAdd(pd.list_count, 1ULL);
                            if (at < pd.list_cap) {
                                pd.list[at] = rank;
                                pd.list_rec[at] = rec;
                            }
                        }
                    }
                }
            }
            for (unsigned int leaf = tail_lo + lane; leaf < hi2; leaf += 64) {
                const unsigned long long rank = rb2 + leaf;
                if (rank < begin || rank >= end) continue;
                int c[K5];
                const unsigned pk = comb5[leaf];
#pragma unroll
                for (int t = 0; t < K5; ++t) c[t] = (int)((pk >> (5 * t)) & 31u);
                const int verdict = leaf_verdict<K5, TS, true, false, !EXACT, true>(gtab, c, R2, U5, 0u, minp2, maxp2, m, nullptr, &viol, s_rows);
                if (verdict == 2) {
                    ++cntS;
                } else if (verdict == 1) {
                    ++cntI;
                } else {
                    ++cntF;
                    const unsigned long long at = atomicAdd(pd.list_count, 1ULL);
                    if (at < pd.list_cap) {
                        pd.list[at] = rank;
                        pd.list_rec[at] = rec;
                    }
                }
            }
            }
        } else if constexpr (MODE == 1) {
            // ---- table 0 item: the tail [leaf_lo, L) of child `child_col` and, if flagged, of the next
            // child too (pivoted from the same record): the lanes are filled from both tails
            const bool pair = (child >> 16) & 1;
            const double* par = s_tab[wave * 2 + (buf ^ 1)];   // the record's slice (tab now points at the child)
            unsigned int T0 = sing0 ? 0u : L - leaf_lo, T1 = 0, lo1 = 0;
            unsigned long long rb1 = rb + L;
            double minq = pm.minp, maxq = pm.maxp;
            const double* tab1 = tab;
            int R1 = R - 1;
            if (pair) {
                for (int j2 = 0; R1 - 1 - j2 >= kGrandMin; ++j2) lo1 += s_binom[(R1 - 1 - j2) * (KD + 1) + KD - 1];
                const unsigned int L1 = s_binom[R1 * (KD + 1) + KD];
                T1 = L1 - lo1;
                // second pivot, on column child_col + 1 of the record
                const int r = lane & (PG - 1), g = lane >> 4;
                unsigned um1 = __builtin_amdgcn_readfirstlane(pm.used_mask);
                const bool row_used = (r >= m) || ((um1 >> r) & 1u);
                const double* pcol = par + (child_col + 1 - D) * TS;
                const double w = pcol[r];
                double big;
                const int p = __builtin_amdgcn_readfirstlane(pick_pivot_row(w, row_used, lane & ~(PG - 1), big));
                minq = fmin(minq, big);
                maxq = fmax(maxq, big);
                if (!(big > 0.0) || minq <= DBL_EPSILON * (double)m * maxq) {
                    if (lane == 0) cntS += (unsigned int)overlap(rb1 + lo1, T1, begin, end);
                    T1 = 0;   // every subset below the second child is singular
                } else {
                    const double inv = 1.0 / pcol[p];
                    const bool isp = (r == p);
                    const double lx = isp ? inv : -(w * inv);
                    um1 |= 1u << p;
                    unsigned rr = (unsigned)r;
                    asm volatile("" : "+v"(rr));
                    const unsigned all = (1u << m) - 1u, below = (1u << rr) - 1u, freem = ~um1 & all;
                    const int pos = (r >= m) ? r
                                    : ((freem >> r) & 1u) ? __builtin_popcount(freem & below)
                                                          : __builtin_popcount(freem) + __builtin_popcount(um1 & all & below);
                    int wv = wave;   // (opaque copy: the slice address is recomputed here, not kept live — and spilled — across items)
                    asm volatile("" : "+v"(wv));
                    double* ctab1 = s_child2[wv];
#pragma unroll
                    for (int q = 0; q < (CHILDCOLS + 3) / 4; ++q) {
                        const int j = g + 4 * q;        // column j of the second child = record column child_col+2+j
                        if (j <= R1) {
                            const double* pc = par + (child_col + 2 + j - D) * TS;
                            ctab1[j * TS + pos] = fma(lx, pc[p], isp ? -0.0 : pc[r]);
                        }
                    }
                    tab1 = ctab1;
                }
            }
            const unsigned* comb1 = pd.comb6 + s_off[R1 >= KD ? R1 : KD];
            const int U[KD] = {0, 1, 2, 3, 4, 5};        // unused by leaf_verdict<PERM>
            for (unsigned int gidx = lane; gidx < T0 + T1; gidx += 64) {
                const bool second = gidx >= T0;
                const unsigned int leaf = second ? lo1 + (gidx - T0) : leaf_lo + gidx;
                const unsigned long long rank = (second ? rb1 : rb) + leaf;
                if (rank < begin || rank >= end) continue;
                int c[KD];
                const unsigned pk = (second ? comb1 : comb)[leaf];
#pragma unroll
                for (int t = 0; t < KD; ++t) c[t] = (int)((pk >> (5 * t)) & 31u);
                const int verdict = leaf_verdict<KD, TS, true, false, !EXACT, true>(second ? tab1 : tab, c, second ? R1 : R, U, 0u,
                                                                              second ? minq : minp0, second ? maxq : maxp0, m,
                                                                              nullptr, &viol, s_rows);
                if (verdict == 2) {
                    ++cntS;
                } else if (verdict == 1) {
                    ++cntI;
                } else {
                    ++cntF;
                    const unsigned long long at = atomicAdd(pd.list_count, 1ULL);
                    if (at < pd.list_cap) {
                        pd.list[at] = rank;
                        pd.list_rec[at] = rec;
                    }
                }
            }
        } else {
        const int U[KD] = {0, 1, 2, 3, 4, 5};        // unused by leaf_verdict<PERM>
        for (unsigned int leaf = leaf_lo + lane; leaf < leaf_hi; leaf += 64) {
            const unsigned long long rank = rb + leaf;
            if (rank < begin || rank >= end) continue;
            int c[KD];
            const unsigned pk = comb[leaf];
#pragma unroll
            for (int t = 0; t < KD; ++t) c[t] = (int)((pk >> (5 * t)) & 31u);
            const int verdict = leaf_verdict<KD, TS, true, false, !EXACT, true>(tab, c, R, U, umask, minp0, maxp0, m, nullptr, &viol, s_rows);
            if (verdict == 2) {
                ++cntS;
            } else if (verdict == 1) {
                ++cntI;
            } else {
                ++cntF;
                const unsigned long long at = atomicAdd(pd.list_count, 1ULL);
                if (at < pd.list_cap) {
            pd.list[at] = rank;
            pd.list_rec[at] = rec;
        }
            }
        }
        }
    }
#ifdef LP_LEAF_WAVELOG
    wavelog_put(MODE, wl_t0, wl_t1, wl_items, wl_steals);
#endif
    if (cntF) atomicAdd(&s_cnt[0], (unsigned long long)cntF);
    if (cntI) atomicAdd(&s_cnt[1], (unsigned long long)cntI);
    if (cntS) atomicAdd(&s_cnt[2], (unsigned long long)cntS);
    if (!EXACT && viol) atomicOr(&d.result->range_flag, 1ULL);
    __syncthreads();
    if (tid < 3 && s_cnt[tid]) atomicAdd(&d.result->counts[tid], s_cnt[tid]);
}

// The tails: 8 lanes per depth m-7 record, lane j takes the j-th 7-subset (lexicographic) of the
// record's last min(R, 8) selectable columns — C(8,7) = 8, C(7,7) = 1 — reading the record where
// it lies in HBM (column stride PG).
template <bool EXACT>
__global__ __launch_bounds__(LEAF_THREADS) void k_enum_thin(EnumDev d, PrefixDev pd,
                                                             const double* __restrict__ roots,
                                                             int root_level, int root_cap,
                                                             unsigned long long begin,
                                                             unsigned long long end) {
    constexpr int KD = 7;
    __shared__ unsigned long long s_cnt[3];
    const int m = d.m, n = d.n, D = m - KD;
    const int nrec = min(pd.level_counts[root_level], root_cap);
    const int tid = threadIdx.x;
#ifdef LP_LEAF_WAVELOG
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < 3) s_cnt[tid] = 0ULL;
    __syncthreads();
    // grid-stride over (record, tail subset) pairs: the grid is sized to the chip, and the three
    // counters leave a block with three atomics in total (one block per 32 records would put
    // 190 k atomics on three words for C(32,16): 0.7 ms at ~11 ns each)
    unsigned int cnt[3] = {0u, 0u, 0u};
    unsigned int viol = 0;
    for (long long gid = (long long)blockIdx.x * LEAF_THREADS + tid; gid < (long long)nrec * 8;
         gid += (long long)gridDim.x * LEAF_THREADS) {
    const int rec = (int)(gid >> 3), j = (int)(gid & 7);
    int verdict = -1;
    unsigned long long rank = 0ULL;
    if (rec < nrec) {
        const double* Q = roots + (size_t)rec * rec_doubles(n, D);
        const NodeMeta* pm = reinterpret_cast<const NodeMeta*>(Q + (size_t)PG * (n - D + 1));
        const int last = pm->last_col;
        const int R = n - 1 - last;
        const int R8 = R < THIN_TAIL ? R : THIN_TAIL;
        // the 7-subsets of 8 columns in lexicographic order: the j-th leaves out column 7 - j
        const int ntail = (R8 == THIN_TAIL) ? THIN_TAIL : 1;
        if (last != kHole && R >= KD && j < ntail) {
            // L - ntail + j: the tail subsets are the last ones of the node
            rank = pm->rank_base + (binom(d, R, KD) - (unsigned long long)ntail + (unsigned long long)j);
            if (rank >= begin && rank < end) {
                int c[KD];
#pragma unroll
                for (int t = 0; t < KD; ++t) c[t] = (R - R8) + t + ((R8 == THIN_TAIL && t >= 7 - j) ? 1 : 0);
                const unsigned umask = pm->used_mask;
                int U[KD];
                unsigned free_rows = ~umask & (m >= 32 ? ~0u : ((1u << m) - 1u));
#pragma unroll
                for (int r = 0; r < KD; ++r) {
                    U[r] = free_rows ? __builtin_ctz(free_rows) : 0;
                    free_rows &= free_rows - 1u;
                }
                const double* tab = Q + (size_t)(last + 1 - D) * PG;   // column q = column last+1+q
                verdict = leaf_verdict<KD, PG, false, false, !EXACT>(tab, c, R, U, umask, pm->minp, pm->maxp, m, nullptr, &viol);
            }
        }
    }
    if (verdict == 0) {
        const unsigned long long at = atomicAdd(pd.list_count, 1ULL);
        if (at < pd.list_cap) {
            pd.list[at] = rank;
            pd.list_rec[at] = rec;
        }
    }
    cnt[0] += verdict == 0;
    cnt[1] += verdict == 1;
    cnt[2] += verdict == 2;
    }
#ifdef LP_LEAF_WAVELOG
    wavelog_put(7, wl_t0, wl_t0, 0, 0);
#endif
    // counts: wave reduction, one LDS atomic per wave and verdict
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        unsigned int c = cnt[v];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
        if ((tid & 63) == 0 && c) atomicAdd(&s_cnt[v], (unsigned long long)c);
    }
    if (!EXACT && viol) atomicOr(&d.result->range_flag, 1ULL);
    __syncthreads();
    if (tid < 3 && s_cnt[tid]) atomicAdd(&d.result->counts[tid], s_cnt[tid]);
}

// ---------------------------------------------------------------------------------------------
// The general leaf kernel: ANY record height (PGT = 16 or 32 rows) and any number of selectable
// columns.  The three kernels above are tuned for m <= 16, n - m <= 16 (subset tables, LDS slices
// sized for 24 columns, 16-lane cooperative pivots); shapes outside that box used to fall to the
// from-scratch solver of enum_direct.hip at ~2 ns per subset.  Here a wave takes an item = (depth
// m-7 record, run of consecutive subsets of its C(R,7)); every lane unranks the first subset of its
// own contiguous share once (binomials from LDS), then walks lexicographic successors, reading the
// record where it lies (L1/L2: a record is at most 40 x 32 doubles) — leaf_verdict<7, PGT, false>,
// the arithmetic of the thin kernel.
constexpr int kGenChunk = 2048;   // subsets per item (32 per lane)
// max n - m of the general path: 32-row records (m > 16) with n <= 64 leave n - m <= 47 — the host
// admits 32 there; 16-row records (7 <= m <= 16) go up to n - m = 57
constexpr int nmxw(int pgt) { return pgt == 16 ? 57 : 32; }

template <int PGT>
__global__ __launch_bounds__(256) void k_enum_generic_items(EnumDev d, PrefixDev pd,
                                                            const double* __restrict__ roots, int root_level,
                                                            int root_cap, unsigned long long begin,
                                                            unsigned long long end) {
    constexpr int KD = 7;
    const int n = d.n, m = d.m, D = m - KD;
    const int nrec = min(pd.level_counts[root_level], root_cap);
    const int rec = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long lo0 = 0, hi0 = 0;
    if (rec < nrec) {
        const NodeMetaT<PGT>* pm = reinterpret_cast<const NodeMetaT<PGT>*>(roots + (size_t)rec * rec_doubles_g<PGT>(n, D, d.rs) +
                                                               (size_t)rec_rs<PGT>(d.rs) * (n - D + 1));
        const int last = pm->last_col;
        const int R = n - 1 - last;
        if (last != kHole && R >= KD) {
            const unsigned long long rb = pm->rank_base, L = binom(d, R, KD);
            const unsigned long long lo = rb > begin ? rb : begin, hi = rb + L < end ? rb + L : end;
            if (hi > lo) {
                lo0 = lo - rb;
                hi0 = hi - rb;
            }
        }
    }
    const int nitem = (int)((hi0 - lo0 + kGenChunk - 1) / kGenChunk);
    int at = nitem ? atomicAdd(&pd.item_count[0], nitem) : 0;
    for (int k = 0; k < nitem; ++k, ++at) {
        const unsigned long long lo = lo0 + (unsigned long long)k * kGenChunk;
        const unsigned long long cnt = hi0 - lo < (unsigned long long)kGenChunk ? hi0 - lo : (unsigned long long)kGenChunk;
        if (at < pd.item_cap) pd.items[at] = make_int4(rec, (int)lo, (int)cnt, 0);
    }
}

// DENSE (a degenerate LP: a large part of the range is feasible): no list — every subset is finished
// without the early exit, its objective summed as the direct solver sums it, and its score (-inf if
// not feasible) written to dense_scores[rank - begin]; the tie rule then runs over that array.
template <int PGT, bool DENSE, bool EXACT>
__global__ __launch_bounds__(LEAF_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_enum_generic_leaves(EnumDev d, PrefixDev pd,
                                                                       const double* __restrict__ roots,
                                                                       unsigned long long range_subsets,
                                                                       unsigned long long begin) {
    constexpr int KD = 7;
    constexpr int TSG = PGT + 1;                 // LDS column stride (odd: rows of different columns, different banks)
    constexpr int NMXW = nmxw(PGT);
    constexpr int GCOLS = NMXW + KD + 1;         // <= n - m + 7 selectable columns + rhs
    __shared__ unsigned int s_bin[(NMXW + KD + 2) * (KD + 1)];   // C(r, k), r <= NMXW + KD + 1, k <= KD
    __shared__ unsigned long long s_cnt[3];
    // the item's record, one private slice per wave, rows PERMUTED: the 7 rows the prefix has not used
    // first (ascending), then the used ones — leaf_verdict<PERM> then reads every row at a
    // compile-time offset (no per-lane gathers from HBM/L1: those, ~100 per subset at a quarter of
    // the LDS rate, bounded the first version of this kernel at 14 G subsets/s on C(32,16))
    __shared__ double s_rec[LEAF_WAVES][GCOLS * TSG];
    __shared__ double s_cost[DENSE ? kEnumMaxN : 1];
    __shared__ unsigned long long s_best;
    __shared__ unsigned long long s_run[LEAF_WAVES];   // every wave's run of work items {next, end}
    __shared__ int s_dry;                              // a wave of this workgroup has seen the end of the item table
    double best = -INFINITY;
    const int m = d.m, n = d.n, D = m - KD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < (NMXW + KD + 2) * (KD + 1); k += LEAF_THREADS) {
        const int r = k / (KD + 1), kk = k - r * (KD + 1);
        s_bin[k] = (unsigned int)d.binom[r * kBinomK + kk];
    }
    if (tid < 3) s_cnt[tid] = 0ULL;
    if (DENSE && tid < d.n) s_cost[tid] = d.c[tid];
    if (tid == 0) s_best = lp_f64_key(-INFINITY);
    __syncthreads();
    const int nitems = min(pd.item_count[0], pd.item_cap);
    unsigned int viol = 0;
    unsigned int cnt[3] = {0u, 0u, 0u};
    double* tab = s_rec[wave];
    int staged = -1;   // record in this wave's slice
    const int U[KD] = {0, 1, 2, 3, 4, 5, 6};   // unused by leaf_verdict<PERM>
    // Items are dealt in runs (a returning atomic on one word costs ~11 ns chip-wide: at one draw per
    // item, the 1.4 M mostly tiny items of C(30,18) took 15.6 ms whatever the lanes did): a static first
    // deal, then runs of a fixed length — ~1024 subsets on average at most (16 items of tiny records, one full
    // item) — held in LDS and emptied by compare-and-swap, so that a wave that finds the table dealt out takes
    // what the other waves of its workgroup still hold (as k_enum_leaves: runs that shrank towards the end of
    // the table made the last thousands of draws queue on the one word)
    const int kMaxRun = (int)max(1ULL, min(16ULL, 1024ULL * (unsigned long long)max(nitems, 1) / max(range_subsets, 1ULL)));
    const int nwaves = (int)gridDim.x * LEAF_WAVES;
    const int k0 = max(1, min(kMaxRun, nitems / (nwaves * 4)));
    const int dyn_base = nwaves * k0;
    auto pack_run = [](int next, int end) { return ((unsigned long long)(unsigned)next << 32) | (unsigned)end; };
    {
        const int first = ((int)blockIdx.x * LEAF_WAVES + wave) * k0;
        if (lane == 0) s_run[wave] = pack_run(min(first, nitems), min(first + k0, nitems));
        if (tid == 0) s_dry = 0;
    }
    __syncthreads();
    auto take = [&](int w) {   // the next item of wave w's run, -1 if it is empty
        int item = -1;
        if (lane == 0) {
            unsigned long long cur = __hip_atomic_load(&s_run[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((unsigned)(cur >> 32) < (unsigned)cur) {
                if (__hip_atomic_compare_exchange_strong(&s_run[w], &cur, cur + (1ULL << 32), __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    item = (int)(cur >> 32);
                    break;
                }
            }
        }
        return __builtin_amdgcn_readfirstlane(item);
    };
    bool dry = dyn_base >= nitems;   // the table is dealt out
    auto draw = [&]() {
        int item = take(wave);
        if (item >= 0) return item;
        if (!dry && __hip_atomic_load(&s_dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) dry = true;
        if (!dry) {
            const int k = kMaxRun;
            int v = 0, over = 0;
            if (lane == 0) {
                v = atomicAdd(&pd.root_cursor[0], k);
                if (!DENSE)   // (a list far beyond its capacity: the caller switches to the dense form)
                    over = __hip_atomic_load(pd.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pd.list_abort;
            }
            v = __builtin_amdgcn_readfirstlane(v) + dyn_base;
            if (__builtin_amdgcn_readfirstlane(over)) v = nitems;
            if (v < nitems) {
                if (k > 1 && lane == 0)
                    __hip_atomic_store(&s_run[wave], pack_run(v + 1, min(v + k, nitems)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                return v;
            }
            dry = true;
            if (lane == 0) __hip_atomic_store(&s_dry, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        for (int q = 1; q < LEAF_WAVES; ++q) {
            item = take((wave + q) & (LEAF_WAVES - 1));
            if (item >= 0) return item;
        }
        return nitems;
    };
    for (;;) {
        const int item = draw();
        if (item >= nitems) break;
        const int4 it = pd.items[item];
        const int rec = __builtin_amdgcn_readfirstlane(it.x);
        const double* Q = roots + (size_t)rec * rec_doubles_g<PGT>(n, D, d.rs);
        const int RS = rec_rs<PGT>(d.rs);   // row stride of the record's columns in HBM
        const NodeMetaT<PGT>* pm = reinterpret_cast<const NodeMetaT<PGT>*>(Q + (size_t)RS * (n - D + 1));
        const int last = __builtin_amdgcn_readfirstlane(pm->last_col);
        const int R = n - 1 - last;
        const unsigned umask = __builtin_amdgcn_readfirstlane(pm->used_mask);
        const unsigned long long rb = pm->rank_base;
        const double minp0 = pm->minp, maxp0 = pm->maxp;
        if (rec != staged) {
            // columns last+1 .. n-1 and the rhs, RS rows each: 64 / PGT columns per round, a lane's row is
            // the same in every round
            const int row = lane & (PGT - 1), cl = lane / PGT;
            const unsigned all = m >= 32 ? ~0u : ((1u << m) - 1u), below = (1u << row) - 1u, freem = ~umask & all;
            const int pos = (row >= m) ? row
                            : ((freem >> row) & 1u) ? __builtin_popcount(freem & below)
                                                    : __builtin_popcount(freem) + __builtin_popcount(umask & all & below);
            const double* src = Q + (size_t)(last + 1 - D) * RS;
            __builtin_amdgcn_wave_barrier();   // (the previous item's reads of the slice are done)
            if (row < RS)
                for (int col = cl; col <= R; col += 64 / PGT) tab[col * TSG + pos] = src[(size_t)col * RS + row];
            __builtin_amdgcn_wave_barrier();
            staged = rec;
        }
        // this lane's share: K consecutive subsets from `mine`
        const unsigned int total = (unsigned int)it.z;
        const unsigned int K = (total + 63u) / 64u;
        const unsigned int mine = (unsigned int)it.y + (unsigned int)lane * K;
        const unsigned int have = (unsigned int)lane * K < total ? min(K, total - (unsigned int)lane * K) : 0u;
        int c[KD];
        {
            unsigned int left = have ? mine : 0u;
            int j = 0;
#pragma unroll
            for (int t = 0; t < KD; ++t) {
                for (;; ++j) {
                    const unsigned int cn = s_bin[(R - 1 - j) * (KD + 1) + (KD - 1 - t)];
                    if (left < cn) break;
                    left -= cn;
                }
                c[t] = j++;
            }
        }
        for (unsigned int k = 0; k < K; ++k) {
            if (k < have) {
                int verdict;
                if constexpr (DENSE) {
                    LeafObj obj;
                    obj.cost = s_cost;
                    obj.prow = pm->prow;
                    obj.pcol = pm->pcol;
                    obj.D = D;
                    obj.col0 = last + 1;
                    obj.used = umask;
                    obj.stride = TSG;
                    obj.z = 0.0;
                    verdict = leaf_verdict<KD, TSG, true, true>(tab, c, R, U, 0u, minp0, maxp0, m, &obj);
                    const double score = verdict == 0 ? (d.maximize ? obj.z : -obj.z) : -INFINITY;
                    pd.dense_scores[rb + mine + k - begin] = score;
                    best = fmax(best, score);
                } else {
                    verdict = leaf_verdict<KD, TSG, true, false, !EXACT>(tab, c, R, U, 0u, minp0, maxp0, m, nullptr, &viol);
                    if (verdict == 0) {
                        const unsigned long long at = atomicAdd(pd.list_count, 1ULL);
                        if (at < pd.list_cap) {
                            pd.list[at] = rb + mine + k;
                            pd.list_rec[at] = rec;
                        }
                    }
                }
                cnt[0] += verdict == 0;
                cnt[1] += verdict == 1;
                cnt[2] += verdict == 2;
                // lexicographic successor inside {0 .. R-1}
                int tpos = -1;
#pragma unroll
                for (int t = 0; t < KD; ++t)
                    if (c[t] < R - KD + t) tpos = t;
#pragma unroll
                for (int t = 0; t < KD; ++t) {
                    if (t == tpos)
                        c[t] += 1;
                    else if (t > tpos && t > 0)
                        c[t] = c[t - 1] + 1;
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        unsigned int x = cnt[v];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        if (lane == 0 && x) atomicAdd(&s_cnt[v], (unsigned long long)x);
    }
    if (DENSE && best > -INFINITY) atomicMax(&s_best, lp_f64_key(best));
    if (!DENSE && !EXACT && viol) atomicOr(&d.result->range_flag, 1ULL);
    __syncthreads();
    if (tid < 3 && s_cnt[tid]) atomicAdd(&d.result->counts[tid], s_cnt[tid]);
    if (DENSE && tid == 0 && s_best != lp_f64_key(-INFINITY)) atomicMax(&d.result->best_key, s_best);
}

// Objectives of the listed (feasible) subsets from the depth m-7 records they were found under: one
// lane per list entry.  The lane unranks its 7 remaining columns inside the record, repeats the
// leaf's arithmetic (leaf_verdict<7, PG, false, OBJ>: the operations, operand for operand, of the
// direct solver, so the score is the one k_enum_eval_list would produce) and keeps every row's
// value for the objective — ~1/15 of the price of a from-scratch m x m solve per entry, which is
// what a degenerate LP (every non-singular basis feasible: hundreds of millions of entries) pays.
template <int PGT>
__global__ __launch_bounds__(LEAF_THREADS) void k_enum_eval_records(EnumDev d, PrefixDev pd,
                                                                     const double* __restrict__ roots,
                                                                     double* __restrict__ scores) {
    constexpr int KD = 7;
    constexpr int NMXW = nmxw(PGT);
    __shared__ unsigned int s_bin[(NMXW + KD + 2) * (KD + 1)];   // C(r, k), r <= NMXW + KD + 1, k <= KD
    __shared__ double s_cost[kEnumMaxN];
    __shared__ unsigned long long s_best;
    const int m = d.m, n = d.n, D = m - KD;
    const int tid = threadIdx.x;
    for (int k = tid; k < (NMXW + KD + 2) * (KD + 1); k += LEAF_THREADS) {
        const int r = k / (KD + 1), kk = k - r * (KD + 1);
        s_bin[k] = (unsigned int)d.binom[r * kBinomK + kk];
    }
    if (tid < n) s_cost[tid] = d.c[tid];
    if (tid == 0) s_best = lp_f64_key(-INFINITY);
    __syncthreads();
    const unsigned long long count = *pd.list_count < pd.list_cap ? *pd.list_count : pd.list_cap;
    double best = -INFINITY;
    for (unsigned long long e = (unsigned long long)blockIdx.x * LEAF_THREADS + tid; e < count;
         e += (unsigned long long)gridDim.x * LEAF_THREADS) {
        const unsigned long long rank = pd.list[e];
        const double* Q = roots + (size_t)pd.list_rec[e] * rec_doubles_g<PGT>(n, D, d.rs);
        const int RS = rec_rs<PGT>(d.rs);
        const NodeMetaT<PGT>* pm = reinterpret_cast<const NodeMetaT<PGT>*>(Q + (size_t)RS * (n - D + 1));
        const int last = pm->last_col;
        const int R = n - 1 - last;
        // the 7 remaining columns: lexicographic unranking inside the R selectable ones
        unsigned int left = (unsigned int)(rank - pm->rank_base);
        int c[KD];
        int j = 0;
#pragma unroll
        for (int t = 0; t < KD; ++t) {
            for (;; ++j) {
                const unsigned int cnt = s_bin[(R - 1 - j) * (KD + 1) + (KD - 1 - t)];
                if (left < cnt) break;
                left -= cnt;
            }
            c[t] = j++;
        }
        const unsigned umask = pm->used_mask;
        int U[KD];
        unsigned free_rows = ~umask & (m >= 32 ? ~0u : ((1u << m) - 1u));
#pragma unroll
        for (int r = 0; r < KD; ++r) {
            U[r] = free_rows ? __builtin_ctz(free_rows) : 0;
            free_rows &= free_rows - 1u;
        }
        LeafObj obj;
        obj.cost = s_cost;
        obj.prow = pm->prow;
        obj.pcol = pm->pcol;
        obj.D = D;
        obj.col0 = last + 1;
        obj.used = umask;
        obj.stride = RS;
        obj.z = 0.0;
        const double* tab = Q + (size_t)(last + 1 - D) * RS;   // column q = column last+1+q
        const int verdict = leaf_verdict<KD, (PGT == 32 ? 0 : PGT), false, true>(tab, c, R, U, umask, pm->minp, pm->maxp, m, &obj);
        // (listed subsets are feasible by construction; a verdict mismatch would be a bug and shows
        // up as -inf here)
        const double score = verdict == 0 ? (d.maximize ? obj.z : -obj.z) : -INFINITY;
        scores[e] = score;
        best = fmax(best, score);
    }
    if (best > -INFINITY) atomicMax(&s_best, lp_f64_key(best));
    __syncthreads();
    if (tid == 0 && s_best != lp_f64_key(-INFINITY)) atomicMax(&d.result->best_key, s_best);
}

}  // namespace

void lp_enum_queue_record_eval(lp_enum_problem* p, const double* roots) {
    lp_context* ctx = p->ctx;
    if (p->dev.m > PG)
        hipLaunchKernelGGL(k_enum_eval_records<32>, (unsigned)ctx->num_cus * 8, LEAF_THREADS, 0, ctx->stream, p->dev,
                           p->prefix, roots, p->prefix.scores);
    else
        hipLaunchKernelGGL(k_enum_eval_records<PG>, (unsigned)ctx->num_cus * 8, LEAF_THREADS, 0, ctx->stream, p->dev,
                           p->prefix, roots, p->prefix.scores);
}

// Second phase of the shared-prefix pass over the records of the last breadth-first level
// (`roots`, count in level_counts[level], at most `bound`): depth m-7 records (fused = true: the
// regular kernel pivots once more itself, the thin kernel takes the small tails) or, for m = 6, the
// root record.  `bound6` bounds the number of depth m-6 nodes (sizes the item table).
int lp_enum_launch_leaves(lp_enum_problem* p, const double* roots, int bound, int level, bool fused,
                          int shape, bool dense, uint64_t begin, uint64_t end) {
    lp_context* ctx = p->ctx;
    PrefixDev& pd = p->prefix;
    const int n = p->dev.n, m = p->dev.m;
    auto ensure = [&](int4*& items, int& cap, uint64_t want) -> int {
        if ((uint64_t)cap >= want) return LP_OPTIMAL;
        lp_pool_release(ctx, items, sizeof(int4) * (size_t)cap);
        items = nullptr;
        cap = 0;
        size_t got = 0;
        LP_HIP(ctx, lp_pool_alloc(ctx, (void**)&items, sizeof(int4) * want, &got));
        cap = (int)std::min<uint64_t>(got / sizeof(int4), 0x7FFFFFFFULL);
        return LP_OPTIMAL;
    };
    const bool exact = p->exact_div;   // plain divisions (a pass of the fast kernels reported pivots out of range)
    const bool general = shape != 1 || dense;
    if (general) {
        // one item per started run of kGenChunk subsets of a record
        int rc = ensure(pd.items, pd.item_cap, (end - begin) / kGenChunk + (uint64_t)bound + 1024);
        if (rc) return rc;
        const unsigned long long b = begin, e = end;
        const unsigned grid_items = (unsigned)lp_ceil_div(bound, 256), grid = (unsigned)ctx->num_cus * 8;
#define LP_GEN(PGT)                                                                                                      \
    do {                                                                                                                 \
        hipLaunchKernelGGL(k_enum_generic_items<PGT>, grid_items, 256, 0, ctx->stream, p->dev, pd, roots, level, bound, \
                           b, e);                                                                                        \
        if (dense)                                                                                                       \
            hipLaunchKernelGGL((k_enum_generic_leaves<PGT, true, true>), grid, LEAF_THREADS, 0, ctx->stream, p->dev,    \
                               pd, roots, e - b, b);                                                                     \
        else if (exact)                                                                                                  \
            hipLaunchKernelGGL((k_enum_generic_leaves<PGT, false, true>), grid, LEAF_THREADS, 0, ctx->stream, p->dev,   \
                               pd, roots, e - b, b);                                                                     \
        else                                                                                                             \
            hipLaunchKernelGGL((k_enum_generic_leaves<PGT, false, false>), grid, LEAF_THREADS, 0, ctx->stream, p->dev,  \
                               pd, roots, e - b, b);                                                                     \
    } while (0)
        if (shape == 3) LP_GEN(32); else LP_GEN(PG);
#undef LP_GEN
        return LP_OPTIMAL;
    }
    const uint64_t total = lp_host_binom(n, m);
    const uint64_t bound6 = lp_host_binom(n - m + level + 1, level + 1);   // depth m-6 nodes (sizes the item table)
    int rc = ensure(pd.items, pd.item_cap, bound6 + total / kChunk + 1024);
    if (rc) return rc;
    if (fused) {
        // table 1: one item per chunk of a depth m-5 node with >= kGrandMin selectable columns
        // (their last chosen column is < n - kGrandMin: at most C(n - kGrandMin, m - 5) nodes)
        rc = ensure(pd.items2, pd.item_cap2, lp_host_binom(n - kGrandMin, m - 5) + total / kChunk + 1024);
        if (rc) return rc;
    }
    // persistent waves (items are dealt dynamically): as many blocks as are resident
    const int grid6 = ctx->num_cus * 3;
    const unsigned long long b = begin, e = end;
    if (fused) {
        // The three leaf kernels are independent of each other (own item table / work cursor, results
        // through atomics), and the thin kernel does not even need the item tables: they run on three
        // streams, so that the thin kernel runs beside the leaf kernels and every kernel's tail (persistent
        // waves running out of items) is filled by the next kernel's blocks.  The library's stream
        // waits for the other two before anything that follows (list evaluation, result copy).
        if (!ctx->aux_stream[0]) {
            for (hipStream_t& a : ctx->aux_stream) LP_HIP(ctx, hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
            for (hipEvent_t& ev : ctx->aux_event) LP_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        hipStream_t s = ctx->stream, sT = ctx->aux_stream[0], s1 = ctx->aux_stream[1];
        const int lanes = bound < 300000 ? kItemLanesNarrow : kItemLanesWide;
        LP_HIP(ctx, hipEventRecord(ctx->aux_event[0], s));          // the level records are complete
        LP_HIP(ctx, hipStreamWaitEvent(sT, ctx->aux_event[0], 0));
        // The thin kernel is latency-bound (8 lanes per record, gathers from HBM: half of its VALU slots idle) and
        // nothing waits for it: a small grid of it (ONE workgroup per CU, grid-stride) is queued BEHIND the item
        // builder, which is on the leaf kernels' critical path, and runs beside the leaf kernels.  (Launched first
        // with 12 workgroups per CU it held the chip while the item builder ran: 14.72 -> 14.50 ms on the whole
        // range, 2.26 -> 2.21 ms on the slowest 8-way shard.  Round 4, with the leaf kernels' new dealing: 1 / 2 / 3 / 4
        // workgroups per CU 13.16 / 13.26 / 13.89 / 13.67 ms, slowest shard 1.92 / 1.95 / 2.01 / 1.99 ms — its waves
        // hold slots that table 0's kernel, which ends the pass, would use better; scripts/time_enum_variant.py.)
        const int thin_per_cu = 1;
        const unsigned grid_thin = (unsigned)std::min<uint64_t>(lp_ceil_div<uint64_t>((uint64_t)bound * 8, LEAF_THREADS), (uint64_t)ctx->num_cus * thin_per_cu);
        auto launch_thin = [&]() {
            if (exact)
                hipLaunchKernelGGL(k_enum_thin<true>, grid_thin, LEAF_THREADS, 0, sT, p->dev, pd, roots, level, bound, b, e);
            else
                hipLaunchKernelGGL(k_enum_thin<false>, grid_thin, LEAF_THREADS, 0, sT, p->dev, pd, roots, level, bound, b, e);
        };
        hipLaunchKernelGGL(k_enum_make_items<true>, lp_ceil_div(bound * lanes, 1024), 1024, 0, s, p->dev, pd,
                           roots, level, bound, THIN_TAIL, lanes, b, e);
        LP_HIP(ctx, hipEventRecord(ctx->aux_event[1], s));          // both item tables are built
        LP_HIP(ctx, hipStreamWaitEvent(s1, ctx->aux_event[1], 0));
        LP_HIP(ctx, hipStreamWaitEvent(sT, ctx->aux_event[1], 0));
        launch_thin();
        // both leaf kernels with resident-sized grids (3 workgroups per CU).  (2 for table 1's kernel gives the best
        // single passes — C(28,14) 1.17 against 1.19 ms, C(32,16) 13.2 against 13.3 — but half of the passes then take
        // 14.0-14.1 ms: the average is worse.)
        const int grid2 = grid6, grid1 = grid6;
#ifndef LP_LEAF_LEVELS   // diagnostic builds: 2 = table 1's kernel without the third in-LDS pivot
#define LP_LEAF_LEVELS 3
#endif
        if (exact) {
            hipLaunchKernelGGL((k_enum_leaves<LP_LEAF_LEVELS, true>), grid2, LEAF_THREADS, 0, s, p->dev, pd, roots, b, e);
            hipLaunchKernelGGL((k_enum_leaves<1, true>), grid1, LEAF_THREADS, 0, s1, p->dev, pd, roots, b, e);
        } else {
            hipLaunchKernelGGL((k_enum_leaves<LP_LEAF_LEVELS, false>), grid2, LEAF_THREADS, 0, s, p->dev, pd, roots, b, e);
            hipLaunchKernelGGL((k_enum_leaves<1, false>), grid1, LEAF_THREADS, 0, s1, p->dev, pd, roots, b, e);
        }
        LP_HIP(ctx, hipEventRecord(ctx->aux_event[0], sT));
        LP_HIP(ctx, hipEventRecord(ctx->aux_event[2], s1));
        LP_HIP(ctx, hipStreamWaitEvent(s, ctx->aux_event[0], 0));
        LP_HIP(ctx, hipStreamWaitEvent(s, ctx->aux_event[2], 0));
    } else {
        hipLaunchKernelGGL(k_enum_make_items<false>, lp_ceil_div(bound, 1024), 1024, 0, ctx->stream, p->dev, pd,
                           roots, level, bound, 0, 1, b, e);
        if (exact)
            hipLaunchKernelGGL((k_enum_leaves<0, true>), grid6, LEAF_THREADS, 0, ctx->stream, p->dev, pd, roots, b, e);
        else
            hipLaunchKernelGGL((k_enum_leaves<0, false>), grid6, LEAF_THREADS, 0, ctx->stream, p->dev, pd, roots, b, e);
    }
    return LP_OPTIMAL;
}

// ---- self-test of recip_midrange against the compiler's division (lp_debug_reciprocal) ----
namespace {
__global__ void k_debug_reciprocal(const double* __restrict__ x, int n, double* __restrict__ fast,
                                   double* __restrict__ plain) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    fast[i] = recip_midrange(v);
    asm volatile("" : "+v"(v));   // (two separate computations, whatever the optimiser thinks of them)
    plain[i] = 1.0 / v;
}
}  // namespace

int lp_enum_debug_reciprocal(lp_context* ctx, const double* x, int n, double* fast_out, double* plain_out) {
    double *dx = nullptr, *df = nullptr, *dp = nullptr;
    const size_t bytes = sizeof(double) * (size_t)n;
    LP_HIP(ctx, hipMalloc(&dx, bytes));
    hipError_t e = hipMalloc(&df, bytes);
    if (e == hipSuccess) e = hipMalloc(&dp, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_reciprocal, (unsigned)lp_ceil_div(n, 256), 256, 0, ctx->stream, dx, n, df, dp);
        e = hipMemcpyAsync(fast_out, df, bytes, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(plain_out, dp, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dx);
    (void)hipFree(df);
    (void)hipFree(dp);
    LP_HIP(ctx, e);
    return LP_OPTIMAL;
}

#ifdef LP_LEAF_WAVELOG
// Diagnostic builds only (-DLP_LEAF_WAVELOG, scripts/leaf_wavelog.py): copies the wave log out (6 words per wave),
// returns the number of entries and empties the log.
extern "C" int lp_debug_leaf_wavelog(unsigned long long* out, int cap) {
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wavelog_n), sizeof(n)) != hipSuccess) return -1;
    if (n > (unsigned)kWaveLogCap) n = kWaveLogCap;
    if ((int)n > cap) n = (unsigned)cap;
    if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wavelog), sizeof(unsigned long long) * 6 * n) != hipSuccess) return -1;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wavelog_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif
