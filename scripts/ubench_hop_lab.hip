// ubench_hop_lab.hip — what ONE all-to-all exchange of small records between G co-resident workgroups
// costs on gfx950, form by form, to find the floor of the chip-resident simplex's per-pivot hop
// (simplexmethod_amd/csrc/simplex_resident.hip) and what stands between the kernel and that floor.
//
// Every epoch: each workgroup publishes its record (REC bytes of {tag, payload} granules), optionally
// every thread publishes one column granule (background traffic, COLB bytes per thread: 0, 8 or 16),
// then the workgroup polls all G records of the epoch and goes on.  cycles/epoch of workgroup 0 = the
// period of the exchange (publish -> last record seen everywhere), no arithmetic in between.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -o scripts/_build/ubench_hop_lab scripts/ubench_hop_lab.hip
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
typedef int v2i __attribute__((ext_vector_type(2)));

struct Params {
    int G, epochs, stride;
    int colb;       // background column bytes per thread and epoch: 0, 8 (plain payload), 16 (tagged granule pair)
    int poll;       // 0: wave 0 polls, lane = record, one poll in flight; 1: every wave polls (no barrier);
                    // 2: wave 0, two polls in flight; 3: wave 0 polls by scalar loads
    int recb;       // record bytes: 16 or 32
    int work;       // cycles of dependent local work between "all records seen" and the next publish
    char* rec;      // [2][G][32]
    char* col;      // [2][G][512 * 16]
    unsigned long long* out;
};

__device__ __forceinline__ v4i ld16(const __amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
}

template <int SC1>
__global__ __launch_bounds__(512) void k_lab(Params p) {
    __shared__ int s_fail;
    __shared__ volatile int s_ep;
    const int b = blockIdx.x;
    if (b % p.stride != 0) return;
    const int k = b / p.stride;
    if (k >= p.G) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = p.G;
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc(p.rec, 0, 2u * G * 32u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcol = __builtin_amdgcn_make_buffer_rsrc(p.col, 0, 2u * G * 512u * 16u, 0x00020000);
    if (tid == 0) {
        s_fail = 0;
        s_ep = 0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(&p.out[3], 1ull << (xcc & 15));
    }
    __syncthreads();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long polls = 0;
    int fail = 0;
    for (int ep = 1; ep <= p.epochs && !fail; ++ep) {
        const int par = ep & 1;
        // ---- publish
        if (wave == 0 && lane < p.recb / 16) {
            v4i g = {ep, k * 1000 + lane, ep, ep ^ 0x5555};
            __builtin_amdgcn_raw_buffer_store_b128(g, rrec, (unsigned)(par * G + k) * 32u + lane * 16u, 0, SC1 ? 16 : 0);
        }
        if (p.colb == 16) {
            v4i g = {ep, tid, ep, k};
            __builtin_amdgcn_raw_buffer_store_b128(g, rcol, ((unsigned)(par * G + k) * 512u + tid) * 16u, 0, SC1 ? 16 : 0);
        } else if (p.colb == 8) {
            v2i g = {tid, ep};
            __builtin_amdgcn_raw_buffer_store_b64(g, rcol, ((unsigned)(par * G + k) * 512u + tid) * 8u, 0, SC1 ? 16 : 0);
        }
        // ---- poll
        if (p.poll == 1 || wave == 0) {
            unsigned spins = 0;
            if (p.poll == 3) {
                // scalar loads: 64 B (two records) per s_load_dwordx16, glc = bypass the scalar cache
                const char* base = p.rec + (size_t)par * G * 32u;
                bool ok;
                do {
                    ok = true;
                    for (int q = 0; q < G * 32; q += 64) {
                        typedef int v16i __attribute__((ext_vector_type(16)));
                        v16i v;
                        asm volatile("s_load_dwordx16 %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "s"(q) : "memory");
                        ok &= v[0] == ep && v[2] == ep && v[8] == ep && v[10] == ep;
                        if (p.recb == 32) ok &= v[4] == ep && v[6] == ep && v[12] == ep && v[14] == ep;
                    }
                    ++polls;
                    if (!ok && (++spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull) {
                        fail = 1;
                        break;
                    }
                } while (!ok);
            } else if (p.poll == 2) {
                // two polls in flight: issue the next sweep before testing the previous one
                const unsigned off = (unsigned)(par * G + (lane < G ? lane : 0)) * 32u;
                v4i a0 = ld16(rrec, off), b0 = p.recb == 32 ? ld16(rrec, off + 16) : a0;
                bool ok;
                do {
                    v4i a1 = ld16(rrec, off), b1 = p.recb == 32 ? ld16(rrec, off + 16) : a1;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2) : "memory");
                    ok = lane >= G || (a0.x == ep && a0.z == ep && b0.x == ep && b0.z == ep);
                    ok = __all(ok);
                    a0 = a1;
                    b0 = b1;
                    ++polls;
                    if (!ok && (++spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull) {
                        fail = 1;
                        break;
                    }
                } while (!ok);
            } else {
                bool ok;
                do {
                    ok = true;
                    for (int q = lane; q < G; q += 64) {
                        const unsigned off = (unsigned)(par * G + q) * 32u;
                        v4i a = ld16(rrec, off);
                        ok &= a.x == ep && a.z == ep;
                        if (p.recb == 32) {
                            v4i b2 = ld16(rrec, off + 16);
                            ok &= b2.x == ep && b2.z == ep;
                        }
                    }
                    ok = __all(ok);
                    ++polls;
                    if (!ok && (++spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull) {
                        fail = 1;
                        break;
                    }
                } while (!ok);
            }
            if (p.work) {
                const unsigned long long w0 = __builtin_readcyclecounter();
                while (__builtin_readcyclecounter() - w0 < (unsigned long long)p.work) {}
            }
            if (p.poll != 1 && lane == 0) {
                if (fail) s_fail = 1;
            }
        }
        if (p.poll != 1) {
            __syncthreads();
            if (s_fail) fail = 1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (fail && tid == 0) atomicAdd(&p.out[2], 1ull);
    if (k == 0 && tid == 0) {
        p.out[0] = t1 - t0;
        p.out[1] = polls;
        p.out[4] = rt1 - rt0;
    }
}

int main() {
    const int epochs = 4000, maxG = 256;
    char *rec, *col;
    unsigned long long* out;
    CHECK(hipMalloc(&rec, 2 * maxG * 32));
    CHECK(hipMalloc(&col, (size_t)2 * maxG * 512 * 16));
    CHECK(hipMalloc(&out, 64));
    struct V { int G, stride, sc1, colb, poll, recb, work; const char* what; };
    const V vs[] = {
        {1, 8, 0, 0, 0, 32, 0, "G=1: loop + own record round trip"},
        {2, 8, 0, 0, 0, 32, 0, "G=2"},
        {32, 8, 0, 0, 0, 32, 0, "records only, plain stores"},
        {32, 8, 1, 0, 0, 32, 0, "records only, sc1 stores"},
        {32, 8, 0, 0, 0, 16, 0, "16-byte records"},
        {32, 8, 0, 0, 2, 32, 0, "two polls in flight"},
        {32, 8, 0, 0, 1, 32, 0, "every wave polls, no barrier"},
        {32, 8, 0, 0, 3, 32, 0, "scalar polls"},
        {32, 8, 0, 16, 0, 32, 0, "+ tagged column stores 16 B/thread"},
        {32, 8, 0, 8, 0, 32, 0, "+ plain column stores 8 B/thread"},
        {32, 8, 0, 16, 2, 32, 0, "+ tagged columns, two polls in flight"},
        {32, 8, 0, 16, 3, 32, 0, "+ tagged columns, scalar polls"},
        {32, 8, 1, 16, 0, 32, 0, "+ tagged columns, sc1 stores"},
        {32, 1, 1, 0, 0, 32, 0, "all XCDs, records only, sc1"},
        {32, 1, 1, 16, 0, 32, 0, "all XCDs, + tagged columns, sc1"},
        {32, 8, 0, 0, 0, 32, 2000, "records only + 2000 cycles of local work"},
        {32, 8, 0, 16, 0, 32, 2000, "+ tagged columns + 2000 cycles of local work"},
        {16, 8, 0, 0, 0, 32, 0, "G=16 records only"},
        {64, 1, 1, 0, 0, 32, 0, "G=64 all XCDs records only sc1"},
    };
    for (const V& v : vs) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(rec, 0, 2 * maxG * 32));
            CHECK(hipMemset(col, 0, (size_t)2 * maxG * 512 * 16));
            CHECK(hipMemset(out, 0, 64));
            Params p{v.G, epochs, v.stride, v.colb, v.poll, v.recb, v.work, rec, col, out};
            const int grid = v.G * v.stride;
            if (v.sc1)
                hipLaunchKernelGGL(k_lab<1>, grid, 512, 0, 0, p);
            else
                hipLaunchKernelGGL(k_lab<0>, grid, 512, 0, 0, p);
            CHECK(hipDeviceSynchronize());
            CHECK(hipGetLastError());
            unsigned long long h[8];
            CHECK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
            printf("%-52s G=%3d stride=%d rep=%d: %6.0f cycles/epoch = %.3f us (%.2f polls/epoch) timeouts=%llu xcc=0x%llx\n",
                   v.what, v.G, v.stride, rep, (double)h[0] / epochs, (double)h[4] / epochs / 100.0, (double)h[1] / epochs, h[2], h[3]);
            fflush(stdout);
        }
    }
    return 0;
}
