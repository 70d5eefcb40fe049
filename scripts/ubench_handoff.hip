// ubench_handoff.hip — price of the one all-to-all hop per pivot that the chip-resident
// simplex (simplexmethod_amd/csrc/simplex_resident.hip) pays: G workgroups each publish a
// 64-B record and an M-row column of data-tagged 16-B granules, every workgroup polls all G
// records, picks the same "winner" and reads the winner's column.  No arithmetic: this is the
// communication floor of a pivot.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -o scripts/_build/ubench_handoff scripts/ubench_handoff.hip
// Run:    scripts/_build/ubench_handoff            (prints one line per variant)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

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
    int G;        // participating workgroups
    int M;        // rows of a column
    int epochs;
    int stride;   // participants are blocks b with b % stride == 0 (8: one XCD under round-robin)
    int sc1_store;  // 1: write-through stores (placement independent); 0: plain stores (same XCD only)
    int uneven;     // 1: odd workgroups publish ~2000 cycles late (the others poll stale lines meanwhile)
    char* rec;    // [2][G][64 B]
    char* col;    // [2][G][M * 16 B]
    unsigned long long* out;  // [0]=cycles, [1]=errors, [2]=timeouts, [3]=xcc mask
};

__device__ __forceinline__ v4i ld16(const __amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);   // sc1: bypass this CU's L1
}
template <int AUX>
__device__ __forceinline__ void st16(v4i v, const __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
}

template <int SC1>
__global__ __launch_bounds__(512) void k_handoff(Params p) {
    __shared__ int s_win;
    __shared__ int s_fail;
    const int b = blockIdx.x;
    if (b % p.stride != 0) return;
    const int k = b / p.stride;
    if (k >= p.G) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.G, M = p.M;
    const unsigned rec_bytes = 2u * G * 64u, col_bytes = 2u * G * (unsigned)M * 16u;
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc(p.rec, 0, rec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcol = __builtin_amdgcn_make_buffer_rsrc(p.col, 0, col_bytes, 0x00020000);
    if (tid == 0) {
        s_fail = 0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(&p.out[3], 1ull << (xcc & 15));
    }
    __syncthreads();
    unsigned long long errors = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    for (int ep = 1; ep <= p.epochs; ++ep) {
        const int par = ep & 1;
        if (p.uneven && (k & 1)) {
            const unsigned long long w0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - w0 < 2000) __builtin_amdgcn_s_sleep(2);
        }
        // ---- publish my column (every thread one granule pair) and my record (4 lanes)
        if (tid < M) {
            const long long val = ((long long)k << 40) | ((long long)ep << 12) | tid;
            v4i g = {ep, (int)(val & 0xFFFFFFFF), ep, (int)(val >> 32)};
            st16<SC1 ? 16 : 0>(g, rcol, ((unsigned)(par * G + k) * M + tid) * 16u);
        }
        if (tid < 4) {
            v4i g = {ep, k * 1000 + tid, ep, ep ^ 0x5555};
            st16<SC1 ? 16 : 0>(g, rrec, (unsigned)(par * G + k) * 64u + tid * 16u);
        }
        // ---- wave 0 polls the G records (lane -> record lane % G, quarter lane / G ... simple form:
        // every lane handles records lane, lane + 64, ...)
        if (wave == 0) {
            bool ok;
            unsigned spins = 0;
            int fail = 0;
            do {
                ok = true;
                for (int q = lane; q < G; q += 64) {
                    const unsigned base = (unsigned)(par * G + q) * 64u;
                    v4i a = ld16(rrec, base), b2 = ld16(rrec, base + 16), c = ld16(rrec, base + 32),
                        d = ld16(rrec, base + 48);
                    ok &= a.x == ep && a.z == ep && b2.x == ep && b2.z == ep && c.x == ep && c.z == ep &&
                          d.x == ep && d.z == ep;
                    if (ok && (a.y != q * 1000 || d.y != q * 1000 + 3)) ++errors;
                }
                ok = __all(ok);
                if (!ok && (++spins & 63) == 0 &&
                    __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull /* 0.2 s */) {
                    fail = 1;
                    break;
                }
            } while (!ok);
            if (lane == 0) {
                s_win = (ep * 7 + 3) % G;
                if (fail) s_fail = 1;
            }
        }
        __syncthreads();
        if (s_fail) break;
        const int w = s_win;
        // ---- read the winner's column
        if (tid < M) {
            const unsigned off = ((unsigned)(par * G + w) * M + tid) * 16u;
            v4i g;
            unsigned spins = 0;
            do {
                g = ld16(rcol, off);
                if (g.x == ep && g.z == ep) break;
            } while (++spins < (1u << 22));
            const long long val = ((long long)g.w << 32) | (unsigned)g.y;
            const long long want = ((long long)w << 40) | ((long long)ep << 12) | tid;
            if (val != want) ++errors;
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (errors) atomicAdd(&p.out[1], errors);
    if (tid == 0 && s_fail) atomicAdd(&p.out[2], 1ull);
    if (k == 0 && tid == 0) p.out[0] = t1 - t0;
}

int main() {
    const int M = 512, epochs = 2000;
    const int maxG = 256;
    char *rec, *col;
    unsigned long long* out;
    CHECK(hipMalloc(&rec, 2 * maxG * 64));
    CHECK(hipMalloc(&col, (size_t)2 * maxG * M * 16));
    CHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct V { int G, stride, sc1, uneven; };
    const V vs[] = {{32, 8, 1, 0}, {32, 1, 1, 0}, {64, 1, 1, 0}, {32, 8, 0, 0},
                    {32, 8, 1, 1}, {32, 8, 0, 1}, {32, 1, 1, 1}, {47, 1, 1, 1}, {64, 1, 1, 1}};
    for (const V& v : vs) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(rec, 0, 2 * maxG * 64));
            CHECK(hipMemset(col, 0, (size_t)2 * maxG * M * 16));
            CHECK(hipMemset(out, 0, 64));
            Params p{v.G, M, epochs, v.stride, v.sc1, v.uneven, rec, col, out};
            const int grid = v.G * v.stride;
            CHECK(hipEventRecord(e0));
            if (v.sc1)
                hipLaunchKernelGGL(k_handoff<1>, grid, 512, 0, 0, p);
            else
                hipLaunchKernelGGL(k_handoff<0>, grid, 512, 0, 0, p);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[4];
            CHECK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
            printf("G=%3d stride=%d stores=%s uneven=%d rep=%d: %.3f us/epoch (event), %.0f cycles/epoch, errors=%llu timeouts=%llu xcc_mask=0x%llx\n",
                   v.G, v.stride, v.sc1 ? "sc1  " : "plain", v.uneven, rep, 1e3 * ms / epochs, (double)h[0] / epochs, h[1],
                   h[2], h[3]);
            fflush(stdout);
        }
    }
    return 0;
}
