// enum_tree.hpp — shared pieces of the shared-prefix enumeration (enum_prefix.hip: breadth-first
// levels + register-resident sweep; enum_leaf.hip: one-lane-per-subset leaf kernel).
#pragma once

#include <cfloat>

#include "enum_problem.hpp"

namespace lptree {

constexpr int PG = 16;            // lanes per group = max rows
constexpr int NMX = 16;           // max n - m on this path
constexpr int META = 8;           // doubles of metadata behind each node record
constexpr int kHole = -2;         // NodeMeta::last_col of a slot whose subtree was pruned

// PATH = 16 (records of at most 16 rows: 64 bytes) or 32 (up to 32 rows: 96 bytes)
template <int PATH>
struct NodeMetaT {  // stored behind the columns of a record
    unsigned long long rank_base;  // rank of the first subset below this node
    double minp, maxp;             // smallest / largest |pivot| so far
    int last_col;                  // last chosen column (-1 at the root)
    unsigned used_mask;            // bit i: row i already used as a pivot row
    unsigned char prow[PATH];      // prow[k], pcol[k]: pivot row and chosen column of depth k < the node's
    unsigned char pcol[PATH];      // depth (the objective of a listed subset is summed in this order)
};
using NodeMeta = NodeMetaT<16>;
static_assert(sizeof(NodeMeta) == META * 8, "NodeMeta must be 64 bytes");
static_assert(sizeof(NodeMetaT<32>) == 96, "NodeMetaT<32> must be 96 bytes");

// Record of a depth-t node: columns t .. n-1 (PGT doubles each, column-major), then the rhs
// column, then the node's metadata.  Only columns > last_col are meaningful.
__host__ __device__ inline size_t rec_doubles(int n, int t) { return (size_t)PG * (n - t + 1) + META; }
// Row stride of a record's columns: 16 for 16-row records (the tuned kernels index them with
// shifts); 32-row records store only the problem's rows, m rounded up to even (EnumDev::rs) — at
// m = 18 a full 32-double column would be 44 % padding, written and read at every level.
template <int PGT>
__host__ __device__ inline int rec_rs(int rs_runtime) { return PGT == 32 ? rs_runtime : PGT; }
template <int PGT>
__host__ __device__ inline size_t rec_doubles_g(int n, int t, int rs) {
    return (size_t)rec_rs<PGT>(rs) * (n - t + 1) + sizeof(NodeMetaT<PGT>) / 8;
}

__device__ __forceinline__ unsigned long long binom(const EnumDev& d, int nn, int kk) {
    if (kk < 0 || nn < kk || nn < 0) return 0ULL;
    return d.binom[nn * kBinomK + kk];
}

__device__ __forceinline__ unsigned long long overlap(unsigned long long rb, unsigned long long cnt,
                                                      unsigned long long begin, unsigned long long end) {
    const unsigned long long lo = rb > begin ? rb : begin;
    const unsigned long long hi = (rb + cnt) < end ? (rb + cnt) : end;
    return hi > lo ? hi - lo : 0ULL;
}

// max over the 16 lanes of a group (DPP row operations), result in every lane of the group
__device__ __forceinline__ double row_max_f64(double v) {
    double o;
#define LP_RSTEP(CTRL)                                                                   \
    {                                                                                    \
        const long long b = __double_as_longlong(v);                                     \
        int lo = (int)(b & 0xFFFFFFFFLL), hi = (int)(b >> 32);                           \
        lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);                 \
        hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);                 \
        o = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);              \
        v = fmax(v, o);                                                                  \
    }
    LP_RSTEP(0xB1)   // quad_perm [1,0,3,2]
    LP_RSTEP(0x4E)   // quad_perm [2,3,0,1]
    LP_RSTEP(0x141)  // row_half_mirror
    LP_RSTEP(0x140)  // row_mirror
#undef LP_RSTEP
    return v;
}

// Broadcast of lane (gbase + p)'s value to its group: addr = (gbase + p) << 2, computed once per pivot.
__device__ __forceinline__ double bcast16(double v, int addr) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xFFFFFFFFLL));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Partial-pivot row choice for one group: first unused row of largest |w| (strict > keeps the
// first).  Returns the group-relative lane p and big = |w_p| (big = -1 if no unused row).
__device__ __forceinline__ int pick_pivot_row(double w, bool used, int gbase, double& big) {
    const double a = used ? -1.0 : fabs(w);
    big = row_max_f64(a);
    const unsigned long long hit = (__ballot(a == big && !used) >> gbase) & 0xFFFFULL;
    return hit ? (int)__builtin_ctzll(hit) : 0;
}
// the same for groups of PGT = 16 or 32 lanes
template <int PGT>
__device__ __forceinline__ int pick_pivot_row_g(double w, bool used, int gbase, double& big) {
    const double a = used ? -1.0 : fabs(w);
    double v = row_max_f64(a);
    if constexpr (PGT == 32) v = fmax(v, __shfl_xor(v, 16, 64));
    big = v;
    const unsigned long long hit = (__ballot(a == big && !used) >> gbase) & (PGT == 32 ? 0xFFFFFFFFULL : 0xFFFFULL);
    return hit ? (int)__builtin_ctzll(hit) : 0;
}

}  // namespace lptree
