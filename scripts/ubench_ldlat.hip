// ubench_ldlat.hip — issue-to-use latency of ONE 16-byte buffer load instruction from a line another
// CU has just written (dirty in the XCD's L2), by cache policy, active lanes and address pattern; and the
// turn-around store -> load of the same wave.  Input to the hop model of the chip-resident simplex.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/_build/ubench_ldlat scripts/ubench_ldlat.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

struct Params {
    char* buf;
    int lanes;     // active lanes of the load
    int stride;    // bytes between the lanes' addresses
    int reps;
    unsigned long long* out;   // [0] sum, [1] min, [2] checksum
};

// block 0 (one wave) measures; block 8 (same XCD under round-robin dispatch) rewrites the lines
// between the measurements so that they are dirty in L2 and absent from the reader's L1.
template <int LAUX, int SAUX>
__global__ __launch_bounds__(64) void k_lat(Params p) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.buf, 0, 1 << 20, 0x00020000);
    if (blockIdx.x == 8) {   // writer: keeps rewriting until told to stop
        int it = 0;
        for (;;) {
            v4i g = {it, lane, it, it};
            __builtin_amdgcn_raw_buffer_store_b128(g, r, (unsigned)lane * (unsigned)p.stride, 0, SAUX);
            ++it;
            __builtin_amdgcn_s_sleep(20);
            if (__builtin_amdgcn_raw_buffer_load_b32(r, 65536, 0, 16) != 0) break;
            if (it > 400000) break;
        }
        return;
    }
    if (blockIdx.x != 0) return;
    unsigned long long sum = 0, mn = ~0ull;
    int chk = 0;
    const unsigned off = (unsigned)lane * (unsigned)p.stride;
    for (int i = 0; i < p.reps; ++i) {
        __builtin_amdgcn_s_sleep(40);
        v4i g = {0, 0, 0, 0};
        unsigned long long t0, t1;
        if (lane < p.lanes) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            g = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LAUX);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(g) : "memory");
            const unsigned long long dt = t1 - t0;
            sum += dt;
            if (dt < mn) mn = dt;
        }
        chk += g.x + g.y;
    }
    if (lane == 0) {
        p.out[0] = sum;
        p.out[1] = mn;
        p.out[2] = (unsigned long long)chk;
        __builtin_amdgcn_raw_buffer_store_b32(1, r, 65536, 0, 16);   // stop the writer
    }
}

// store -> load turn-around of one wave on its own line (what a poll right behind a publish pays)
template <int LAUX, int SAUX>
__global__ __launch_bounds__(64) void k_turn(Params p) {
    const int lane = threadIdx.x;
    if (blockIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.buf, 0, 1 << 20, 0x00020000);
    unsigned long long sum = 0, mn = ~0ull;
    int chk = 0, stale = 0;
    const unsigned off = 131072u + (unsigned)lane * (unsigned)p.stride;
    for (int i = 1; i <= p.reps; ++i) {
        __builtin_amdgcn_s_sleep(40);
        v4i g = {0, 0, 0, 0};
        unsigned long long t0, t1;
        if (lane < p.lanes) {
            v4i s = {i, lane, i, i};
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            __builtin_amdgcn_raw_buffer_store_b128(s, r, off, 0, SAUX);
            g = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LAUX);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(g) : "memory");
            const unsigned long long dt = t1 - t0;
            sum += dt;
            if (dt < mn) mn = dt;
            if (g.x != i) ++stale;
        }
        chk += g.x + g.y;
    }
    if (lane == 0) {
        p.out[0] = sum;
        p.out[1] = mn;
        p.out[2] = (unsigned long long)chk;
        p.out[3] = (unsigned long long)stale;
    }
}

template <int LAUX, int SAUX>
void run(const char* name, char* buf, unsigned long long* out, int lanes, int stride) {
    const int reps = 2000;
    unsigned long long h[4];
    CHECK(hipMemset(buf, 0, 1 << 20));
    CHECK(hipMemset(out, 0, 64));
    Params p{buf, lanes, stride, reps, out};
    hipLaunchKernelGGL((k_lat<LAUX, SAUX>), 9, 64, 0, 0, p);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf("%-28s lanes=%2d stride=%3d: load of a remotely written line avg %5.0f min %5llu", name, lanes, stride,
           (double)h[0] / reps, h[1]);
    CHECK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL((k_turn<LAUX, SAUX>), 1, 64, 0, 0, p);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
    printf(" | own store->load avg %5.0f min %5llu stale %llu\n", (double)h[0] / reps, h[1], h[3]);
    fflush(stdout);
}

int main() {
    char* buf;
    unsigned long long* out;
    CHECK(hipMalloc(&buf, 1 << 20));
    CHECK(hipMalloc(&out, 64));
    const int shapes[][2] = {{1, 16}, {16, 16}, {32, 16}, {64, 16}, {32, 32}, {64, 32}, {32, 128}, {64, 128}};
    for (auto& s : shapes) {
        run<16, 0>("load sc1, store plain", buf, out, s[0], s[1]);
        run<1, 0>("load sc0, store plain", buf, out, s[0], s[1]);
        run<17, 0>("load sc0 sc1, store plain", buf, out, s[0], s[1]);
        run<0, 0>("load plain, store plain", buf, out, s[0], s[1]);
        run<16, 16>("load sc1, store sc1", buf, out, s[0], s[1]);
        run<1, 1>("load sc0, store sc0", buf, out, s[0], s[1]);
        run<2, 0>("load nt, store plain", buf, out, s[0], s[1]);
    }
    return 0;
}
