// enum_prefix.hip — vertex enumeration with SHARED-PREFIX elimination over the combination
// tree (LP_ENUM_ALGO_PREFIX).  Results are bit-identical to enum_direct.hip / the oracle's
// orc_enum_subset: the per-subset solve (Gauss-Jordan with partial pivoting on the first m-2
// columns in ascending order, then a 2x2 block + back-substitution) depends, up to step t,
// only on the first t+1 columns of the subset, so all subsets sharing a prefix share that
// part of the elimination — ~O(m) flops per subset instead of O(m^3).
//
// The sorted m-subsets form a tree (depth t = t columns chosen).  A node carries the
// partially eliminated tableau restricted to the columns still selectable, [W[:, c > last] | rhs]
// (16 or 32 rows x <= n-t+1 columns), the used-row mask, min/max |pivot| and the pivot row and
// column of every depth so far.
//
//   phase 1  k_enum_expand / k_enum_expand_narrow (this file): levels 0 .. D0 breadth-first
//            through HBM (D0 = m-7 for the default second phase).  One wave per parent, its
//            four 16-lane groups (lane = row) pivot four children at a time and write the child
//            records; a block of 64 parents takes all its children's slots with one atomic.
//            Narrow levels: one wave per (parent, child).  Wide levels of 16-row records:
//            k_enum_expand_staged (parent record and the subset-count table in LDS).
//   phase 2  the leaf kernels of enum_leaf.hip (one lane per subset from the depth
//            m-7 records, with one or two more pivots done by the wave in LDS).
//   Feasible subsets are rare; each is appended to a list as (rank, record index) and its objective
//   is evaluated afterwards from that record (enum_leaf.hip: k_enum_eval_records; m = 6:
//   enum_direct.hip: k_enum_eval_list); the scored list also serves pass 2 (tie rule) without a
//   second enumeration.
//   Records have 16 rows (m <= 16) or 32 (m <= 32): the kernels below are templates over the group
//   width PGT and the column bound NMXT; <16, 16> is the tuned shape of C(32,16).
//
// Singular prefixes prune their whole subtree (min|piv| only falls, max|piv| only grows).
#include <cfloat>
#include <cstdlib>

#include "enum_problem.hpp"
#include "enum_tree.hpp"

namespace {

using namespace lptree;

constexpr int kExpandParents = 64;  // parents per block of k_enum_expand (one slot allocation per block)
constexpr int kStagedParents = 256; // ... of k_enum_expand_staged (16-row records): a lane per parent in all four waves

// ---------------------------------------------------------------------------
// phase 1: expand level t -> t+1 (records in HBM)
// ---------------------------------------------------------------------------
template <int PGT, int NMXT>
__global__ __launch_bounds__(256) void k_enum_expand(EnumDev d, PrefixDev pd, int t,
                                                     const double* __restrict__ src, int src_cap,
                                                     double* __restrict__ dst, int dst_cap, int ppw,
                                                     unsigned long long begin,
                                                     unsigned long long end) {
    // A block owns 4 * ppw consecutive parents (ppw = parents per wave, 1..16: 16 on wide levels,
    // fewer on levels that could not fill the chip otherwise).  Phase A: one lane per parent counts the
    // children whose rank interval meets [begin, end) and the block takes all their slots with ONE
    // atomic — a returning atomic on one word costs ~11 ns chip-wide, and at one per wave of four
    // parents (184 k of them for the 735 k parents of C(32,16)'s last level) that alone was 2 ms.
    // Phase B: one WAVE per parent, 16 parents per wave in turn: its four 16-lane groups
    // (lane = row) pivot four different children at a time, so a parent's up to n-m+1 children
    // take 5 rounds of dependent HBM round trips instead of 17, with four children's stores in
    // flight together.
    __shared__ int s_base[kExpandParents];
    // subsets below a child with last column a, C(n-1-a, m-t-1): the level's column of the binomial table, in LDS
    // (round 3: looked up in L2 child by child these dependent little loads were what a parent cost — see
    // k_enum_expand_staged below)
    __shared__ unsigned long long s_cnt[kEnumMaxN + 1];
    const int m = d.m, n = d.n;
    const int tid = threadIdx.x;
    if (tid <= kEnumMaxN) s_cnt[tid] = (tid < n) ? binom(d, n - 1 - tid, m - t - 1) : 0ULL;
    constexpr int GW = 64 / PGT;   // groups (children in flight) per wave
    using Meta = NodeMetaT<PGT>;
    const int RS = rec_rs<PGT>(d.rs);   // row stride of the records' columns
    const int lane = tid & 63, gl = lane & (PGT - 1), g = lane / PGT, gbase = lane & ~(PGT - 1);
    // the level's record count lives on the device (level_counts[t]): the host queues all levels
    // without synchronising, with grids sized for an upper bound
    const int nsrc = min(pd.level_counts[t], src_cap);  // (an overflowed level is reported by the host)
    const int first = blockIdx.x * 4 * ppw;
    if (first >= nsrc) return;
    const int lim = n - m + t;  // largest column selectable at depth t
    __syncthreads();
    if (tid < 64) {
        const int node = first + tid;
        int nch = 0;
        if (tid < 4 * ppw && node < nsrc) {
            const Meta* q = reinterpret_cast<const Meta*>(src + (size_t)node * rec_doubles_g<PGT>(n, t, d.rs) +
                                                                  (size_t)RS * (n - t + 1));
            const int last = q->last_col;
            if (last != kHole) {
                unsigned long long rb = q->rank_base;
                for (int a = last + 1; a <= lim; ++a) {
                    const unsigned long long cnt = s_cnt[a];
                    if (overlap(rb, cnt, begin, end) != 0ULL) ++nch;
                    rb += cnt;
                }
            }
        }
        int incl = nch;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        int base = 0;
        if (lane == 63 && total > 0) base = atomicAdd(&pd.level_counts[t + 1], total);
        base = __shfl(base, 63, 64);
        s_base[tid] = base + incl - nch;
    }
    __syncthreads();
    for (int it = 0; it < ppw; ++it) {
    const int local = (tid >> 6) * ppw + it;
    const int node = first + local;
    if (node >= nsrc) break;
    const double* P = src + (size_t)node * rec_doubles_g<PGT>(n, t, d.rs);
    const Meta pm = *reinterpret_cast<const Meta*>(P + (size_t)RS * (n - t + 1));
    if (pm.last_col == kHole) continue;
    // ---- the children, one per lane: subset counts, rank bases (exclusive scan), range overlap
    const int a_l = pm.last_col + 1 + lane;
    const unsigned long long cnt_l = (a_l <= lim) ? s_cnt[a_l] : 0ULL;
    unsigned long long incl = cnt_l;
#pragma unroll
    for (int off = 1; off < (NMXT > 16 ? 64 : 32); off <<= 1) {   // at most NMXT + 1 children
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    const unsigned long long rb_l = pm.rank_base + (incl - cnt_l);
    const unsigned long long ov_l = overlap(rb_l, cnt_l, begin, end);
    const unsigned long long vmask = __ballot(ov_l != 0ULL);
    const int nchild = __popcll(vmask);
    if (nchild == 0) continue;
    const int wbase = s_base[local];
    const bool prow_used = (gl >= m) || ((pm.used_mask >> gl) & 1u);
    const double prhs = gl < RS ? P[(size_t)(n - t) * RS + gl] : 0.0;
    unsigned long long sing = 0ULL;
    for (int k0 = 0; k0 < nchild; k0 += GW) {
        const int k = k0 + g;                  // this group's child (k-th valid one)
        const bool active = k < nchild;        // whole groups are active or not
        unsigned long long vm = vmask;
        for (int i = 0; i < (active ? k : 0); ++i) vm &= vm - 1ULL;
        const int src_lane = active ? (int)__builtin_ctzll(vm) : 0;
        const int a = pm.last_col + 1 + src_lane;
        const unsigned long long rb_child = __shfl(rb_l, src_lane, 64);
        const unsigned long long ov = __shfl(ov_l, src_lane, 64);
        const int myslot = wbase + k;
        // (no wave-level operation below: groups proceed independently)
        if (!active) continue;
        if (myslot >= dst_cap) {
            if (gl == 0) atomicExch(pd.overflow, 1);
            continue;
        }
        double* C = dst + (size_t)myslot * rec_doubles_g<PGT>(n, t + 1, d.rs);
        Meta* cmeta = reinterpret_cast<Meta*>(C + (size_t)RS * (n - t));
        const double w = gl < RS ? P[(size_t)(a - t) * RS + gl] : 0.0;
        double big;
        const int p = pick_pivot_row_g<PGT>(w, prow_used, gbase, big);
        const double minp = fmin(pm.minp, big), maxp = fmax(pm.maxp, big);
        if (!(big > 0.0) || minp <= DBL_EPSILON * (double)m * maxp) {
            sing += ov;  // the whole subtree is singular: leave a hole
            if (gl == 0) cmeta->last_col = kHole;
            continue;
        }
        const int addr = (gbase + p) << 2;
        const double piv = bcast16(w, addr);
        const double inv = 1.0 / piv;
        const bool isp = (gl == p);
        const double lx = isp ? inv : -(w * inv);
        // columns a+1 .. n-1, six at a time with all their loads issued before the first use (a
        // one-column loop waits out a memory round trip per column)
        for (int c0 = a + 1; c0 < n; c0 += 6) {
            double own[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) own[u] = (c0 + u < n && gl < RS) ? P[(size_t)(c0 + u - t) * RS + gl] : 0.0;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const double pc = bcast16(own[u], addr);
                if (c0 + u < n && gl < RS) C[(size_t)(c0 + u - t - 1) * RS + gl] = fma(lx, pc, isp ? -0.0 : own[u]);
            }
        }
        const double pr = bcast16(prhs, addr);
        if (gl < RS) C[(size_t)(n - t - 1) * RS + gl] = fma(lx, pr, isp ? -0.0 : prhs);
        if (gl == 0) {
            Meta cm;
            cm.rank_base = rb_child;
            cm.minp = minp;
            cm.maxp = maxp;
            cm.last_col = a;
            cm.used_mask = pm.used_mask | (1u << p);
            for (int q = 0; q < PGT; ++q) {
                cm.prow[q] = q == t ? (unsigned char)p : pm.prow[q];
                cm.pcol[q] = q == t ? (unsigned char)a : pm.pcol[q];
            }
            *cmeta = cm;
        }
    }
    if (gl == 0 && sing) atomicAdd(&d.result->counts[2], sing);
    }
}

// ---------------------------------------------------------------------------
// phase 1, wide levels of 16-row records (round 3): the PARENT RECORD STAGED IN LDS.
// k_enum_expand above walks a parent's children in rounds of four and fetches every operand from the
// record where it lies (HBM the first time, L2 afterwards): the pivot column, then the remaining
// columns six at a time — ~4 dependent memory round trips per round, ~20 per parent, one parent after the
// other in each wave.  PMC (scripts/pmc_kernel_bytes.py): the last level of C(32,16) moves 3.66 GB (1.14 GB
// of parents read, 2.58 GB of children written — only the columns behind a node's last chosen one are
// stored) in 1.7 ms = 2.2 TB/s: a latency chain, not a bandwidth limit.
// Here a wave copies the meaningful part of its parent (columns > last_col and the rhs: contiguous, <= 4.2 KB)
// into its private LDS slice with ONE coalesced load per lane-slot, issued one parent AHEAD (the registers
// of parent i+1 and its metadata are in flight while parent i is expanded; which part to fetch is known
// from the block's first phase, which reads every parent's last_col anyway).  Every operand of every child
// — its pivot column, the pivot row's entries (a plain LDS read at the pivot row instead of a
// ds_bpermute of the loaded column), its own entries — then comes from LDS, and the only global traffic
// inside the loop is the children's stores.  Same arithmetic, operand for operand; same slots, same
// metadata as k_enum_expand.  The per-level table of subset counts C(n-1-a, m-t-1) lives in LDS as well: read
// out of L2 by every parent's lane in phase A and again per lane in the expansion, those dependent little loads
// — not the records — were what a parent cost (staging the records alone changed nothing: 1.82 against 1.66 ms
// for the last level; with the table in LDS 1.04 ms = 3.5 TB/s of the 3.66 GB, the five wide levels of C(32,16)
// 2.70 -> 1.66 ms).  The child's metadata is assembled word-wise (the byte-wise form was a third of the loop's
// instructions).
// ---------------------------------------------------------------------------
// (4 waves per SIMD: 6 changed nothing, 8 spills and is 70 % slower)
#ifndef LP_STAGED_COUNTED
#define LP_STAGED_COUNTED 0
#endif
template <int NMXT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_enum_expand_staged(EnumDev d, PrefixDev pd, int t,
                                                            const double* __restrict__ src, int src_cap,
                                                            double* __restrict__ dst, int dst_cap, int ppw,
                                                            unsigned long long begin,
                                                            unsigned long long end) {
    constexpr int PGT = 16, RS = 16, GW = 4;
    constexpr int MAXC = NMXT + PGT + 1;            // columns of a record incl. the rhs: n - t + 1 <= m + NMXT + 1
    constexpr int PER = (MAXC * RS + META + 63) / 64;   // doubles per lane that hold one record's meaningful part + metadata
    using Meta = NodeMetaT<PGT>;
    __shared__ int s_base[kStagedParents];
    __shared__ int s_last[kStagedParents];
    __shared__ int s_wtot[5];
    __shared__ int s_nst[4];   // per wave: store instructions issued behind the prefetch in flight
    __shared__ __attribute__((aligned(16))) double s_rec[4][MAXC * RS + META];
    // subsets below a child with last column a: C(n-1-a, m-t-1) — one table per level, in LDS (looked up per
    // child by every parent's lane in phase A and again per lane in the expansion: out of L2 these dependent
    // little loads, not the records, were what a parent cost)
    __shared__ unsigned long long s_cnt[kEnumMaxN + 1];
    const int m = d.m, n = d.n;
    const int tid = threadIdx.x;
    if (tid <= kEnumMaxN) s_cnt[tid] = (tid < n) ? binom(d, n - 1 - tid, m - t - 1) : 0ULL;
    const int lane = tid & 63, gl = lane & (PGT - 1), g = lane / PGT, gbase = lane & ~(PGT - 1);
    const int wave = tid >> 6;
    const int nsrc = min(pd.level_counts[t], src_cap);  // (an overflowed level is reported by the host)
    const int first = blockIdx.x * 4 * ppw;
    if (first >= nsrc) return;
    const int lim = n - m + t;  // largest column selectable at depth t
    __syncthreads();
    const size_t rdP = rec_doubles_g<PGT>(n, t, d.rs), rdC = rec_doubles_g<PGT>(n, t + 1, d.rs);
    {   // phase A: children per parent and ONE slot allocation per block — here over up to 256 parents, a lane each in
        // all four waves (k_enum_expand: 64 parents by one wave while three wait).  Its three dependent round trips
        // (the table above, the parents' metadata, the returning atomic: ~7 us) were a third of a 64-parent block's life.
        const int node = first + tid;
        int nch = 0, last = kHole;
        if (tid < 4 * ppw && node < nsrc) {
            const Meta* q = reinterpret_cast<const Meta*>(src + (size_t)node * rdP + (size_t)RS * (n - t + 1));
            last = q->last_col;
            if (last != kHole) {
                unsigned long long rb = q->rank_base;
                for (int a = last + 1; a <= lim; ++a) {
                    const unsigned long long cnt = s_cnt[a];
                    if (overlap(rb, cnt, begin, end) != 0ULL) ++nch;
                    rb += cnt;
                }
            }
        }
        int incl = nch;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        if (tid == 0) {
            int total = 0;
            for (int w = 0; w < 4; ++w) {
                const int x = s_wtot[w];
                s_wtot[w] = total;   // exclusive prefix over the waves
                total += x;
            }
            s_wtot[4] = total > 0 ? atomicAdd(&pd.level_counts[t + 1], total) : 0;
        }
        __syncthreads();
        s_base[tid] = s_wtot[4] + s_wtot[wave] + incl - nch;
        s_last[tid] = nch > 0 ? last : kHole;   // (a parent without a child in the range is skipped like a hole)
    }
    __syncthreads();
    double* L = s_rec[wave];
    double pre[PER];
    // issue the loads of this wave's parent `it` (nothing for a hole or past the end: wave-uniform)
    auto fetch = [&](int it) {
        const int local = wave * ppw + it, node = first + local;
        const int last = (it < ppw && node < nsrc) ? s_last[local] : kHole;
        if (last == kHole) return;
        const double* P = src + (size_t)node * rdP;
        const double* Q = P + (size_t)(last + 1 - t) * RS;
        const int cnt = (n - last) * RS + META;   // columns last+1 .. n-1, the rhs, and the metadata right behind it
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = u * 64 + lane;
#if LP_STAGED_COUNTED
            // (asm: a load the compiler does not track — the wait for it is the counted one at the staging below)
            if (idx < cnt) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre[u]) : "v"(Q + idx) : "memory");
#else
            pre[u] = idx < cnt ? Q[idx] : 0.0;
#endif
        }
    };
#if LP_STAGED_COUNTED
#pragma unroll
    for (int u = 0; u < PER; ++u) pre[u] = 0.0;
    if (lane == 0) s_nst[wave] = 0;
#endif
    fetch(0);
    for (int it = 0; it < ppw; ++it) {
        const int local = wave * ppw + it;
        const int node = first + local;
        if (node >= nsrc) break;
        const int last = s_last[local];
        if (last == kHole) {
            fetch(it + 1);
            continue;
        }
        const int cnt = (n - last) * RS + META;
#if LP_STAGED_COUNTED
        {
            // The wave's memory counter retires loads and stores in issue order, and the compiler, not knowing how many
            // stores the previous parent issued behind this parent's loads, would wait for ALL of them (vmcnt(0)): every
            // parent then pays its predecessor's store acknowledgements, ~2-3 us.  The previous iteration counted the
            // store instructions it issued at wave level (a lower bound: the column loops' trips and the rhs stores) —
            // waiting until at most that many operations are outstanding retires exactly the loads and whatever is older.
            const int nst = __builtin_amdgcn_readfirstlane(s_nst[wave]);
            if (nst >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (nst >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (nst >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (nst >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) s_nst[wave] = 0;
        }
#endif
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = u * 64 + lane;
            if (idx < cnt) L[idx] = pre[u];
        }
        fetch(it + 1);   // the next parent travels while this one is expanded
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nrhs = (n - 1 - last) * RS;   // offset of the rhs column inside L
        const Meta& pm = *reinterpret_cast<const Meta*>(L + nrhs + RS);   // (read from LDS where needed)
        // ---- the children, one per lane: subset counts, rank bases (exclusive scan), range overlap
        const int a_l = last + 1 + lane;
        const unsigned long long cnt_l = (a_l <= lim) ? s_cnt[a_l] : 0ULL;
        unsigned long long incl = cnt_l;
#pragma unroll
        for (int off = 1; off < (NMXT > 16 ? 64 : 32); off <<= 1) {   // at most NMXT + 1 children
            const unsigned long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const unsigned long long rb_l = pm.rank_base + (incl - cnt_l);
        const unsigned long long ov_l = overlap(rb_l, cnt_l, begin, end);
        const unsigned long long vmask = __ballot(ov_l != 0ULL);
        const int nchild = __popcll(vmask);
        const int wbase = s_base[local];
        const bool prow_used = (gl >= m) || ((pm.used_mask >> gl) & 1u);
        const double prhs = L[nrhs + gl];
        unsigned long long sing = 0ULL;
        for (int k0 = 0; k0 < nchild; k0 += GW) {
            const int k = k0 + g;                  // this group's child (k-th valid one)
            const bool active = k < nchild;        // whole groups are active or not
            unsigned long long vm = vmask;
            for (int i = 0; i < (active ? k : 0); ++i) vm &= vm - 1ULL;
            const int src_lane = active ? (int)__builtin_ctzll(vm) : 0;
            const int a = last + 1 + src_lane;
            const unsigned long long rb_child = __shfl(rb_l, src_lane, 64);
            const unsigned long long ov = __shfl(ov_l, src_lane, 64);
            const int myslot = wbase + k;
            // (no wave-level operation below: groups proceed independently)
            if (!active) continue;
            if (myslot >= dst_cap) {
                if (gl == 0) atomicExch(pd.overflow, 1);
                continue;
            }
            double* C = dst + (size_t)myslot * rdC;
            Meta* cmeta = reinterpret_cast<Meta*>(C + (size_t)RS * (n - t));
            const double* La = L + (a - last - 1) * RS;   // the child's pivot column
            const double w = La[gl];
            double big;
            const int p = pick_pivot_row_g<PGT>(w, prow_used, gbase, big);
            const double minp = fmin(pm.minp, big), maxp = fmax(pm.maxp, big);
            if (!(big > 0.0) || minp <= DBL_EPSILON * (double)m * maxp) {
                sing += ov;  // the whole subtree is singular: leave a hole
                if (gl == 0) cmeta->last_col = kHole;
                continue;
            }
            const double piv = La[p];
            const double inv = 1.0 / piv;
            const bool isp = (gl == p);
            const double lx = isp ? inv : -(w * inv);
#if LP_STAGED_COUNTED
            {   // the groups that write a child this round: the first one has the smallest a, i.e. the longest column loop
                const unsigned long long run = __ballot(1);
                if (lane == (int)__builtin_ctzll(run)) s_nst[wave] += n - a;   // n - 1 - a columns + the rhs
            }
#endif
            double* Cc = C + (size_t)(a - t) * RS + gl;   // column c of the child at (c - t - 1) * RS
            const double* Lc = La + RS;                   // column a + 1 of the parent
#pragma unroll 2
            for (int c = a + 1; c < n; ++c, Lc += RS, Cc += RS) *Cc = fma(lx, Lc[p], isp ? -0.0 : Lc[gl]);
            C[(size_t)(n - t - 1) * RS + gl] = fma(lx, L[nrhs + p], isp ? -0.0 : prhs);
            if (gl == 0) {
                // the child's metadata, word-wise (64 bytes: four 16-byte stores; the pivot row / column lists
                // are the parent's with byte t patched — byte by byte this was a third of the loop's instructions)
                const uint4 pr4 = *reinterpret_cast<const uint4*>(pm.prow), pc4 = *reinterpret_cast<const uint4*>(pm.pcol);
                unsigned prw[4] = {pr4.x, pr4.y, pr4.z, pr4.w}, pcw[4] = {pc4.x, pc4.y, pc4.z, pc4.w};
                const unsigned sh = 8u * (unsigned)(t & 3), keep = ~(0xFFu << sh);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q == (t >> 2)) {
                        prw[q] = (prw[q] & keep) | ((unsigned)p << sh);
                        pcw[q] = (pcw[q] & keep) | ((unsigned)a << sh);
                    }
                uint4* out = reinterpret_cast<uint4*>(cmeta);
                const unsigned long long mn = (unsigned long long)__double_as_longlong(minp),
                                         mx = (unsigned long long)__double_as_longlong(maxp);
                out[0] = make_uint4((unsigned)rb_child, (unsigned)(rb_child >> 32), (unsigned)mn, (unsigned)(mn >> 32));
                out[1] = make_uint4((unsigned)mx, (unsigned)(mx >> 32), (unsigned)a, pm.used_mask | (1u << p));
                out[2] = make_uint4(prw[0], prw[1], prw[2], prw[3]);
                out[3] = make_uint4(pcw[0], pcw[1], pcw[2], pcw[3]);
            }
        }
        if (gl == 0 && sing) atomicAdd(&d.result->counts[2], sing);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the slice is rewritten by the next parent)
    }
}

// The staged expansion for 32-ROW records (m > 16; RS = m rounded up to even rows per column, two 32-lane groups =
// two children per wave at a time).  A record's meaningful part is up to 65 columns x 32 rows = 16.6 KB: no
// register prefetch of the next parent here (it would take 66 VGPRs) — the wave copies its parent to LDS at the
// top of each iteration, eight doubles per lane in flight at a time, and pays that one round trip per parent
// instead of the ~4 per round of k_enum_expand.
template <int NMXT>
__global__ __launch_bounds__(256) void k_enum_expand_staged32(EnumDev d, PrefixDev pd, int t,
                                                              const double* __restrict__ src, int src_cap,
                                                              double* __restrict__ dst, int dst_cap, int ppw,
                                                              unsigned long long begin,
                                                              unsigned long long end) {
    constexpr int PGT = 32, GW = 2;
    constexpr int MAXC = NMXT + PGT + 1;   // columns of a record incl. the rhs: n - t + 1 <= m + NMXT + 1
    using Meta = NodeMetaT<PGT>;
    constexpr int METAD = sizeof(Meta) / 8;
    __shared__ int s_base[kExpandParents];
    __shared__ int s_last[kExpandParents];
    __shared__ __attribute__((aligned(16))) double s_rec[4][MAXC * PGT + METAD];
    __shared__ unsigned long long s_cnt[kEnumMaxN + 1];   // C(n-1-a, m-t-1) by child column a
    const int m = d.m, n = d.n;
    const int RS = rec_rs<PGT>(d.rs);
    const int tid = threadIdx.x;
    if (tid <= kEnumMaxN) s_cnt[tid] = (tid < n) ? binom(d, n - 1 - tid, m - t - 1) : 0ULL;
    const int lane = tid & 63, gl = lane & (PGT - 1), g = lane / PGT, gbase = lane & ~(PGT - 1);
    const int wave = tid >> 6;
    const int nsrc = min(pd.level_counts[t], src_cap);
    const int first = blockIdx.x * 4 * ppw;
    if (first >= nsrc) return;
    const int lim = n - m + t;
    __syncthreads();
    const size_t rdP = rec_doubles_g<PGT>(n, t, d.rs), rdC = rec_doubles_g<PGT>(n, t + 1, d.rs);
    if (tid < 64) {   // phase A (as in k_enum_expand)
        const int node = first + tid;
        int nch = 0, last = kHole;
        if (tid < 4 * ppw && node < nsrc) {
            const Meta* q = reinterpret_cast<const Meta*>(src + (size_t)node * rdP + (size_t)RS * (n - t + 1));
            last = q->last_col;
            if (last != kHole) {
                unsigned long long rb = q->rank_base;
                for (int a = last + 1; a <= lim; ++a) {
                    const unsigned long long cnt = s_cnt[a];
                    if (overlap(rb, cnt, begin, end) != 0ULL) ++nch;
                    rb += cnt;
                }
            }
        }
        int incl = nch;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int total = __shfl(incl, 63, 64);
        int base = 0;
        if (lane == 63 && total > 0) base = atomicAdd(&pd.level_counts[t + 1], total);
        base = __shfl(base, 63, 64);
        s_base[tid] = base + incl - nch;
        s_last[tid] = nch > 0 ? last : kHole;
    }
    __syncthreads();
    double* L = s_rec[wave];
    const bool row = gl < RS;   // this lane holds a row of the record
    for (int it = 0; it < ppw; ++it) {
        const int local = wave * ppw + it;
        const int node = first + local;
        if (node >= nsrc) break;
        const int last = s_last[local];
        if (last == kHole) continue;
        {   // the parent's columns last+1 .. n-1, rhs and metadata (contiguous) -> LDS, eight doubles per lane in flight
            const double* Q = src + (size_t)node * rdP + (size_t)(last + 1 - t) * RS;
            const int cnt = (n - last) * RS + METAD;
            for (int base = 0; base < cnt; base += 8 * 64) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 64 + lane;
                    v[u] = idx < cnt ? Q[idx] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 64 + lane;
                    if (idx < cnt) L[idx] = v[u];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nrhs = (n - 1 - last) * RS;
        const Meta& pm = *reinterpret_cast<const Meta*>(L + nrhs + RS);
        const int a_l = last + 1 + lane;
        const unsigned long long cnt_l = (a_l <= lim) ? s_cnt[a_l] : 0ULL;
        unsigned long long incl = cnt_l;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const unsigned long long rb_l = pm.rank_base + (incl - cnt_l);
        const unsigned long long ov_l = overlap(rb_l, cnt_l, begin, end);
        const unsigned long long vmask = __ballot(ov_l != 0ULL);
        const int nchild = __popcll(vmask);
        const int wbase = s_base[local];
        const bool prow_used = (gl >= m) || ((pm.used_mask >> gl) & 1u);
        const double prhs = row ? L[nrhs + gl] : 0.0;
        unsigned long long sing = 0ULL;
        for (int k0 = 0; k0 < nchild; k0 += GW) {
            const int k = k0 + g;
            const bool active = k < nchild;
            unsigned long long vm = vmask;
            for (int i = 0; i < (active ? k : 0); ++i) vm &= vm - 1ULL;
            const int src_lane = active ? (int)__builtin_ctzll(vm) : 0;
            const int a = last + 1 + src_lane;
            const unsigned long long rb_child = __shfl(rb_l, src_lane, 64);
            const unsigned long long ov = __shfl(ov_l, src_lane, 64);
            const int myslot = wbase + k;
            // pick_pivot_row_g<32> exchanges between the two halves of a group: called by whole groups only, but a
            // group that is not active must still reach it together with the other one (wave-level shuffle)
            const double* La = L + (size_t)(active ? a - last - 1 : 0) * RS;
            const double w = row ? La[gl] : 0.0;
            double big;
            const int p = pick_pivot_row_g<PGT>(w, prow_used, gbase, big);
            if (!active) continue;
            if (myslot >= dst_cap) {
                if (gl == 0) atomicExch(pd.overflow, 1);
                continue;
            }
            double* C = dst + (size_t)myslot * rdC;
            Meta* cmeta = reinterpret_cast<Meta*>(C + (size_t)RS * (n - t));
            const double minp = fmin(pm.minp, big), maxp = fmax(pm.maxp, big);
            if (!(big > 0.0) || minp <= DBL_EPSILON * (double)m * maxp) {
                sing += ov;
                if (gl == 0) cmeta->last_col = kHole;
                continue;
            }
            const double piv = La[p];
            const double inv = 1.0 / piv;
            const bool isp = (gl == p);
            const double lx = isp ? inv : -(w * inv);
            double* Cc = C + (size_t)(a - t) * RS + gl;
            const double* Lc = La + RS;
            if (row) {
#pragma unroll 2
                for (int c = a + 1; c < n; ++c, Lc += RS, Cc += RS) *Cc = fma(lx, Lc[p], isp ? -0.0 : Lc[gl]);
                C[(size_t)(n - t - 1) * RS + gl] = fma(lx, L[nrhs + p], isp ? -0.0 : prhs);
            }
            if (gl == 0) {
                Meta cm = pm;   // (32-row records: 96 bytes, copied as a struct)
                cm.rank_base = rb_child;
                cm.minp = minp;
                cm.maxp = maxp;
                cm.last_col = a;
                cm.used_mask = pm.used_mask | (1u << p);
                cm.prow[t] = (unsigned char)p;
                cm.pcol[t] = (unsigned char)a;
                *cmeta = cm;
            }
        }
        if (gl == 0 && sing) atomicAdd(&d.result->counts[2], sing);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// The same expansion for NARROW levels (the first few, and every level of a small rank range):
// one WAVE per (parent, child) — lane = (row, column quarter) — with all of the child's columns
// in flight at once.  A lone 16-lane group pivoting up to n-m+1 children one after the other, four
// dependent HBM round trips per child, takes ~60 us per level: most of a small shard's run time.
template <int PGT, int NMXT>
__global__ __launch_bounds__(256) void k_enum_expand_narrow(EnumDev d, PrefixDev pd, int t,
                                                            const double* __restrict__ src, int src_cap,
                                                            double* __restrict__ dst, int dst_cap,
                                                            unsigned long long begin,
                                                            unsigned long long end) {
    const int m = d.m, n = d.n, S = n - m + 1;   // S = most children a node can have
    constexpr int GW = 64 / PGT;
    using Meta = NodeMetaT<PGT>;
    const int RS = rec_rs<PGT>(d.rs);   // row stride of the records' columns
    const int lane = threadIdx.x & 63, gl = lane & (PGT - 1), g = lane / PGT, gbase = lane & ~(PGT - 1);
    const int wid = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int node = wid / S, j = wid - node * S;
    const int nsrc = min(pd.level_counts[t], src_cap);
    if (node >= nsrc) return;
    const double* P = src + (size_t)node * rec_doubles_g<PGT>(n, t, d.rs);
    const Meta pm = *reinterpret_cast<const Meta*>(P + (size_t)RS * (n - t + 1));
    const int a = pm.last_col + 1 + j;
    if (pm.last_col == kHole || a > n - m + t) return;
    // subsets below the siblings in front of this child: one lane per sibling and a wave sum (a loop of up to
    // n - m dependent little loads from the table in L2 was most of a narrow level's 8-60 us)
    const int a2 = pm.last_col + 1 + lane;
    unsigned long long part = (a2 < a) ? binom(d, n - 1 - a2, m - t - 1) : 0ULL;
    const unsigned long long mine = binom(d, n - 1 - a, m - t - 1);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    const unsigned long long rb = pm.rank_base + part;
    const unsigned long long ov = overlap(rb, mine, begin, end);
    if (ov == 0ULL) return;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&pd.level_counts[t + 1], 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= dst_cap) {
        if (lane == 0) atomicExch(pd.overflow, 1);
        return;
    }
    double* C = dst + (size_t)slot * rec_doubles_g<PGT>(n, t + 1, d.rs);
    Meta* cmeta = reinterpret_cast<Meta*>(C + (size_t)RS * (n - t));
    const bool prow_used = (gl >= m) || ((pm.used_mask >> gl) & 1u);
    const double w = gl < RS ? P[(size_t)(a - t) * RS + gl] : 0.0;
    // this group's columns (a+1+g, a+5+g, ...; column n = the rhs): all loads issued before use
    constexpr int NC = (PGT + NMXT + 1 + GW - 1) / GW;   // n <= PGT + NMXT: at most that many columns + rhs
    double own[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = a + 1 + g + GW * q;
        own[q] = (c <= n && gl < RS) ? P[(size_t)(c - t) * RS + gl] : 0.0;
    }
    double big;
    const int p = pick_pivot_row_g<PGT>(w, prow_used, gbase, big);
    const double minp = fmin(pm.minp, big), maxp = fmax(pm.maxp, big);
    if (!(big > 0.0) || minp <= DBL_EPSILON * (double)m * maxp) {
        if (lane == 0) {
            cmeta->last_col = kHole;  // the whole subtree is singular
            atomicAdd(&d.result->counts[2], ov);
        }
        return;
    }
    const int addr = (gbase + p) << 2;
    const double inv = 1.0 / bcast16(w, addr);
    const bool isp = (gl == p);
    const double lx = isp ? inv : -(w * inv);
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int c = a + 1 + g + GW * q;
        const double pc = bcast16(own[q], addr);
        if (c <= n && gl < RS) C[(size_t)(c - t - 1) * RS + gl] = fma(lx, pc, isp ? -0.0 : own[q]);
    }
    if (lane == 0) {
        Meta cm;
        cm.rank_base = rb;
        cm.minp = minp;
        cm.maxp = maxp;
        cm.last_col = a;
        cm.used_mask = pm.used_mask | (1u << p);
        for (int k = 0; k < PGT; ++k) {
            cm.prow[k] = k == t ? (unsigned char)p : pm.prow[k];
            cm.pcol[k] = k == t ? (unsigned char)a : pm.pcol[k];
        }
        *cmeta = cm;
    }
}

// root record (depth 0): the original [A | b]
// (also resets every counter of the pass: one launch instead of two copies and four memsets,
// each of which costs a 10-30 us enqueue gap at the start of a small rank range)
template <int PGT>
__global__ void k_enum_root(EnumDev d, PrefixDev pd, double* dst) {
    const int gl = threadIdx.x;
    if (gl < 32) pd.level_counts[gl] = gl == 0 ? 1 : 0;
    if (gl == 32) {
        EnumResult r;
        r.best_key = lp_f64_key(-INFINITY);
        r.counts[0] = r.counts[1] = r.counts[2] = 0ULL;
        r.first_rank = ~0ULL;
        r.range_flag = r.pad[0] = r.pad[1] = 0ULL;
        *d.result = r;
        *pd.list_count = 0ULL;
        *pd.overflow = 0;
        pd.root_cursor[0] = pd.root_cursor[1] = 0;
        pd.item_count[0] = pd.item_count[1] = 0;
    }
    const int RS = rec_rs<PGT>(d.rs);
    if (gl >= RS) return;
    for (int c = 0; c < d.n; ++c) dst[(size_t)c * RS + gl] = gl < d.m ? d.A[gl * d.lda + c] : 0.0;
    dst[(size_t)d.n * RS + gl] = gl < d.m ? d.b[gl] : 0.0;
    if (gl == 0) {
        NodeMetaT<PGT> cm;
        cm.rank_base = 0ULL;
        cm.minp = INFINITY;
        cm.maxp = 0.0;
        cm.last_col = -1;
        cm.used_mask = 0u;
        for (int k = 0; k < PGT; ++k) cm.prow[k] = cm.pcol[k] = 0;
        *reinterpret_cast<NodeMetaT<PGT>*>(dst + (size_t)RS * (d.n + 1)) = cm;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------

static size_t host_rec_doubles(int n, int t, int pg, int rs) {
    return (size_t)(pg == 32 ? rs : pg) * (n - t + 1) + (pg == 32 ? 12 : 8);
}

// Number of depth-t tree nodes whose subtree meets the rank range [begin, end): the length-t
// prefixes of the subsets begin .. end-1 are consecutive in the lexicographic order of the
// t-subsets of {0 .. n-m+t-1}, so the count is the difference of two prefix ranks, plus one.
uint64_t lp_host_prefix_rank(int n, int m, uint64_t rank, int t) {
    // unrank the first t elements of the rank-th m-subset, accumulating their rank among t-subsets
    uint64_t pr = 0;
    int a = 0;
    for (int k = 0; k < t; ++k) {
        int j = a;
        for (;; ++j) {
            const uint64_t cnt = lp_host_binom(n - 1 - j, m - 1 - k);
            if (rank < cnt) break;
            rank -= cnt;
            pr += lp_host_binom(n - m + t - 1 - j, t - 1 - k);  // t-prefixes starting ..j.. lie before
        }
        a = j + 1;
    }
    return pr;
}
static uint64_t host_level_nodes(int n, int m, uint64_t begin, uint64_t end, int t) {
    if (end <= begin) return 0;
    return lp_host_prefix_rank(n, m, end - 1, t) - lp_host_prefix_rank(n, m, begin, t) + 1;
}

// Shapes of the shared-prefix path:
//   1  m in 6..16, n-m in 2..16: 16-row records, the tuned leaf kernels (subset tables, LDS slices)
//   2  m in 7..16, n-m in 17..57 (n <= 64): 16-row records, the general leaf kernel
//   3  m in 17..32, n-m in 2..32: 32-row records, the general leaf kernel
//   0  everything else (direct kernel)
int lp_enum_prefix_shape(const lp_enum_problem* p) {
    const int m = p->dev.m, nm = p->dev.n - p->dev.m;
    if (nm < 2) return 0;
    if (m >= 6 && m <= PG && nm <= NMX) return 1;
    if (m >= 7 && m <= PG && nm <= 57) return 2;
    if (m > PG && m <= 32 && nm <= 32) return 3;
    return 0;
}
bool lp_enum_prefix_supported(const lp_enum_problem* p) { return lp_enum_prefix_shape(p) != 0; }

static int prefix_range_once(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score_best,
                             uint64_t counts[3], lp_enum_stats* stats, bool dense);

// The leaf kernels run with the fast reciprocal (enum_leaf.hip: recip_midrange) until a pass reports a
// pivot outside its exponent range on a subset that is not singular anyway; that pass is repeated with
// plain divisions, and so is every later pass of the problem.
int lp_enum_prefix_range(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score_best,
                         uint64_t counts[3], lp_enum_stats* stats, bool dense) {
    if (const char* ev = getenv("LP_ENUM_EXACT_DIV")) p->exact_div = p->exact_div || atoi(ev) != 0;   // (A/B, tests)
    p->h_result->range_flag = 0ULL;
    int rc = prefix_range_once(p, begin, end, score_best, counts, stats, dense);
    if (!p->exact_div && p->h_result->range_flag != 0ULL) {
        p->exact_div = true;
        lp_enum_stats first{};
        if (stats) first = *stats;
        rc = prefix_range_once(p, begin, end, score_best, counts, stats, dense);
        if (stats) {
            stats->kernel_ms += first.kernel_ms;
            stats->launches += first.launches;
        }
    }
    return rc;
}

static int prefix_range_once(lp_enum_problem* p, uint64_t begin, uint64_t end, double* score_best,
                             uint64_t counts[3], lp_enum_stats* stats, bool dense) {
    lp_context* ctx = p->ctx;
    const EnumDev& d = p->dev;
    hipStream_t s = ctx->stream;
    const int m = d.m, n = d.n;
    const int shape = lp_enum_prefix_shape(p);
    const int pg = shape == 3 ? 32 : PG;
    // breadth-first to depth m-7 (m-6 for m = 6), then one lane per subset (enum_leaf.hip)
    // (the leaf kernel performs the pivot of depth m-6 itself, so the levels stop at depth m-7)
    const bool fused = m >= 7;
    const int D0 = fused ? m - 7 : m - 6;
    PrefixDev& pd = p->prefix;
    p->dense_active = false;
    if (dense && !fused) dense = false;
    // a listing pass whose list is already 16x over capacity stops early: the caller goes dense on that
    // count alone (LP_ENUM_LIST_CAP, the tests' pinned list, needs the exact count for its sub-ranges)
    pd.list_abort = (fused && getenv("LP_ENUM_LIST_CAP") == nullptr) ? 16 * pd.list_cap : ~0ULL;
    if (dense && pd.dense_cap < end - begin) {   // rank-indexed scores of the range (8 bytes per subset)
        lp_pool_release(ctx, pd.dense_scores, sizeof(double) * pd.dense_cap);
        pd.dense_scores = nullptr;
        pd.dense_cap = 0;
        size_t got = 0;
        if (lp_pool_alloc(ctx, (void**)&pd.dense_scores, sizeof(double) * (end - begin), &got) == hipSuccess) {
            pd.dense_cap = got / sizeof(double);
        } else {
            (void)hipGetLastError();
            dense = false;   // no memory for it: the list form
        }
    }
    // ---- buffers: two ping-pong level arrays sized for the widest level (depth D0) of the whole
    // problem if that fits the budget (C(32,16): 6.4 GB), otherwise for as many records as fit; a
    // range with more depth-D0 nodes than that is split by the caller (kEnumRangeTooWide)
    const uint64_t nodes_max = lp_host_binom(n - m + D0, D0);
    const uint64_t nodes_prev = D0 >= 1 ? lp_host_binom(n - m + D0 - 1, D0 - 1) : 1;
    const size_t rec_bytes = host_rec_doubles(n, D0, pg, d.rs) * sizeof(double);
    const size_t rec_prev_bytes = host_rec_doubles(n, D0 >= 1 ? D0 - 1 : 0, pg, d.rs) * sizeof(double);
    uint64_t cap_budget0 = 0, cap_budget1 = 0;   // records the budget allows at depth D0 / D0-1
    {
        if (ctx->total_mem == 0) {
            size_t free_b = 0;
            LP_HIP(ctx, hipMemGetInfo(&free_b, &ctx->total_mem));
        }
        size_t budget = std::min<size_t>(ctx->total_mem / 5 * 2, size_t(24) << 30);   // for both buffers together
        if (const char* e = getenv("LP_ENUM_LEVEL_BUDGET_KB")) budget = (size_t)strtoull(e, nullptr, 10) << 10;   // (tests)
        // level D0-1 holds at most as many records as level D0 (every record has a child or is a hole)
        const uint64_t cap0 = std::max<uint64_t>(std::min<uint64_t>(nodes_max, budget / (rec_bytes + rec_prev_bytes)), 1);
        const uint64_t cap1 = std::min<uint64_t>(nodes_prev, cap0);
        cap_budget0 = cap0;
        cap_budget1 = cap1;
        const size_t want[2] = {(size_t)cap0 * rec_bytes, std::max<size_t>((size_t)cap1 * rec_prev_bytes, 4096)};
        for (int k = 0; k < 2; ++k) {
            if (p->prefix_buf_bytes[k] >= want[k]) continue;
            lp_pool_release(ctx, p->prefix_buf[k], p->prefix_buf_bytes[k]);
            p->prefix_buf[k] = nullptr;
            p->prefix_buf_bytes[k] = 0;
            size_t got = 0;
            hipError_t e = lp_pool_alloc(ctx, (void**)&p->prefix_buf[k], want[k], &got);
            if (e != hipSuccess) {
                // give back what the context's pool holds before giving up on this path
                (void)hipGetLastError();
                for (auto& blk : ctx->pool) (void)hipFree(blk.first);
                ctx->pool.clear();
                e = lp_pool_alloc(ctx, (void**)&p->prefix_buf[k], want[k], &got);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return LP_ITER_LIMIT;   // caller falls back to the direct kernel
            }
            p->prefix_buf_bytes[k] = got;
        }
    }
    {
        // the range's own node counts (exact) against what the buffers hold
        // (a kept buffer may be larger than this problem's budget share: the budget decides, so that the
        // behaviour does not depend on what ran before)
        const uint64_t have0 = std::min<uint64_t>(p->prefix_buf_bytes[0] / rec_bytes, cap_budget0);
        const uint64_t have1 = std::min<uint64_t>(p->prefix_buf_bytes[1] / rec_prev_bytes, cap_budget1);
        const uint64_t want0 = host_level_nodes(n, m, begin, end, D0);
        const uint64_t want1 = D0 >= 1 ? host_level_nodes(n, m, begin, end, D0 - 1) : 1;
        if (want0 > have0 || want1 > have1 || want0 > 0x7FFFFFFFULL) {
            p->split_hint = std::max<uint64_t>(want0 / std::max<uint64_t>(have0, 1), want1 / std::max<uint64_t>(have1, 1)) + 1;
            return kEnumRangeTooWide;
        }
    }
    // depth-D0 records always end in buffer 0; levels alternate so that level D0 lands there
    LP_HIP(ctx, hipEventRecord(p->ev0, s));
    // dense form: only subsets under live depth-D0 records get a score from the leaf kernel; subtrees
    // pruned as singular leave their entries untouched, and the tie rule scans the whole range.  All
    // bits set = NaN, which fails its `>=` test (the buffer comes from a pool: old scores, level records)
    if (dense) LP_HIP(ctx, hipMemsetAsync(pd.dense_scores, 0xFF, sizeof(double) * (end - begin), s));
    int launches = 0;
    int cur = (D0 % 2 == 0) ? 0 : 1;  // buffer of level 0, so that level D0 is buffer 0
    // All levels and the leaf kernel are queued without a host round trip: every level's record
    // count stays on the device (level_counts[t]); grids are sized for the combinatorial upper
    // bound C(n-m+t, t) of the level (blocks beyond the actual count return at once).
    if (pg == 32)
        hipLaunchKernelGGL(k_enum_root<32>, 1, 64, 0, s, d, pd, p->prefix_buf[cur]);
    else
        hipLaunchKernelGGL(k_enum_root<PG>, 1, 64, 0, s, d, pd, p->prefix_buf[cur]);   // + all counters reset
    ++launches;
    int caps[32];
    caps[0] = 1;
    for (int t = 0; t < D0; ++t) {
        const int nxt = cur ^ 1;
        const uint64_t cap64 = p->prefix_buf_bytes[nxt] / (host_rec_doubles(n, t + 1, pg, d.rs) * sizeof(double));
        const int cap = cap64 > 0x7FFFFFFFULL ? 0x7FFFFFFF : (int)cap64;
        caps[t + 1] = cap;
        // parents of this level inside the range (exact; the level's holes are among them)
        const uint64_t bound = std::min<uint64_t>(host_level_nodes(n, m, begin, end, t), 0x7FFFFFFFULL);
        // k_enum_expand: 4 * ppw parents per block; ppw = 16 once that still leaves 32 waves per CU
        const int ppw = (int)std::max<uint64_t>(1, std::min<uint64_t>(kExpandParents / 4, bound / ((uint64_t)ctx->num_cus * 32)));
        const int groups_per_block = 4 * ppw;
        // narrow levels: one wave per (parent, child)
        const uint64_t waves = bound * (uint64_t)(n - m + 1);
        // (every child of a narrow level takes its slot with a returning atomic of its own, ~11 ns each on the one
        // counter: beyond a few thousand candidate waves the per-parent kernels, one allocation per block, are faster —
        // 16-row records: the LDS-staged kernel expands the 969 parents of C(32,16)'s level 3 in a fraction of the
        // 61 us the narrow form took for their 4845 children)
        const int narrow_mult = getenv("LP_ENUM_NARROW_MULT") ? atoi(getenv("LP_ENUM_NARROW_MULT")) : (pg == 16 ? 16 : 128);   // (env: A/B, tests)
        const bool narrow = waves <= (uint64_t)ctx->num_cus * (uint64_t)narrow_mult;
        const unsigned grid = narrow ? (unsigned)lp_ceil_div<uint64_t>(waves, 4)
                                     : (unsigned)lp_ceil_div<uint64_t>(bound, groups_per_block);
        const double* src = p->prefix_buf[cur];
        double* dst = p->prefix_buf[nxt];
        const int src_cap = t == 0 ? 1 : caps[t];
        const unsigned long long b = begin, e = end;
#define LP_EXPAND(PGT, NMXT)                                                                                   \
    do {                                                                                                       \
        if (narrow)                                                                                            \
            hipLaunchKernelGGL((k_enum_expand_narrow<PGT, NMXT>), grid, 256, 0, s, d, pd, t, src, src_cap, dst, \
                               cap, b, e);                                                                     \
        else                                                                                                   \
            hipLaunchKernelGGL((k_enum_expand<PGT, NMXT>), grid, 256, 0, s, d, pd, t, src, src_cap, dst, cap,   \
                               ppw, b, e);                                                                     \
    } while (0)
        // wide levels of 16-row records: the parent staged in LDS (LP_ENUM_EXPAND_UNSTAGED=1: the earlier kernel, A/B)
        const bool unstaged = getenv("LP_ENUM_EXPAND_UNSTAGED") != nullptr;
        // (the staged kernel of 16-row records: up to 64 parents per wave, 256 per block)
        const int ppwS = (int)std::max<uint64_t>(1, std::min<uint64_t>(kStagedParents / 4, bound / ((uint64_t)ctx->num_cus * 32)));
        const unsigned gridS = (unsigned)lp_ceil_div<uint64_t>(bound, 4 * (uint64_t)ppwS);
        if (shape == 1 && !narrow && !unstaged)
            hipLaunchKernelGGL(k_enum_expand_staged<16>, gridS, 256, 0, s, d, pd, t, src, src_cap, dst, cap, ppwS, b, e);
        else if (shape == 2 && !narrow && !unstaged)
            hipLaunchKernelGGL(k_enum_expand_staged<57>, gridS, 256, 0, s, d, pd, t, src, src_cap, dst, cap, ppwS, b, e);
        else if (shape == 3 && !narrow && !unstaged)
            hipLaunchKernelGGL(k_enum_expand_staged32<32>, grid, 256, 0, s, d, pd, t, src, src_cap, dst, cap, ppw, b, e);
        else if (shape == 1) LP_EXPAND(16, 16);
        else if (shape == 2) LP_EXPAND(16, 57);
        else LP_EXPAND(32, 32);
#undef LP_EXPAND
        ++launches;
        cur = nxt;
    }
    const uint64_t root_bound = std::min<uint64_t>(host_level_nodes(n, m, begin, end, D0), 0x7FFFFFFFULL);
    {
        const int rc = lp_enum_launch_leaves(p, p->prefix_buf[cur], (int)std::min<uint64_t>(root_bound, (uint64_t)caps[D0]),
                                             D0, fused, shape, dense, begin, end);
        if (rc) return rc;
    }
    ++launches;
    // objectives of the (few) feasible subsets by the direct solver, and the tie rule against this
    // range's own best score (what a sharded run asks next): queued behind the leaf kernels
    constexpr double kSpecTol = 1e-9;   // Solver::EPS, the tolerance dist.py / EnumerationSolver use
    p->spec_valid = false;
    if (dense)
        lp_enum_queue_dense_tail(p, kSpecTol, begin, end);
    else
        lp_enum_queue_list_tail(p, kSpecTol, fused ? p->prefix_buf[cur] : nullptr);
    LP_HIP(ctx, hipEventRecord(p->ev1, s));
    // result, list count, overflow flag and level counts: one block, one copy (enum_problem.hpp: EnumPassBlock)
    LP_HIP(ctx, hipMemcpyAsync(p->h_pass, p->d_pass, sizeof(EnumPassBlock), hipMemcpyDeviceToHost, s));
    LP_HIP(ctx, hipStreamSynchronize(s));
    LP_HIP(ctx, hipGetLastError());
    for (int t = 1; t <= D0; ++t)
        if (p->h_level_counts[t] > caps[t]) return LP_ITER_LIMIT;  // a level buffer was too small
    if (*p->h_overflow != 0) return LP_ITER_LIMIT;  // fall back
    float ms = 0.f;
    LP_HIP(ctx, hipEventElapsedTime(&ms, p->ev0, p->ev1));
    if (!dense && *p->h_list_count > pd.list_cap) {   // the caller grows the list or splits the range
        if (stats) {
            stats->kernel_ms = ms;
            stats->subsets = end - begin;
            stats->launches = launches;
        }
        return kEnumListOverflow;
    }
    const uint64_t nfeas = dense ? p->h_result->counts[0] : *p->h_list_count;
    p->dense_active = dense;
    double best = -INFINITY;
    if (nfeas) best = lp_key_f64(p->h_result->best_key);
    p->spec_valid = true;
    p->spec_star = best;
    p->spec_tol = kSpecTol;
    p->spec_first = nfeas ? p->h_result->first_rank : ~0ULL;
    p->list_valid = true;
    p->list_begin = begin;
    p->list_end = end;
    p->list_n = nfeas;
    *score_best = best;
    for (int k = 0; k < 3; ++k) counts[k] = p->h_result->counts[k];
    if (stats) {
        stats->kernel_ms = ms;
        stats->subsets = end - begin;
        stats->launches = launches;
    }
    return LP_OPTIMAL;
}
