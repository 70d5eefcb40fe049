// ubench_hop_pure.hip — the all-to-all exchange of 32-byte records between G workgroups with a DEDICATED
// communication wave and nothing else: store my record -> poll until all G records of the epoch are
// fresh -> (W cycles of dependent work) -> next epoch.  cycles/epoch - W = the pure hop a chip-resident
// pivot pays between "my record is ready" and "I know everybody's".
//
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/_build/ubench_hop_pure scripts/ubench_hop_pure.hip
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
    int G, epochs, stride, work;
    int skew;    // > 0: every 16th epoch workgroup (ep / 16) % G publishes `skew` cycles late (the others poll stale lines meanwhile)
    char* rec;   // [2][G][32]
    unsigned long long* out;
};

// MODE 0: lane l loads granule l (16 B) of the 1 KB record block: one instruction per sweep, one sweep in flight
// MODE 1: the same, two sweeps in flight
// MODE 2: lane q < G loads both granules of record q (two instructions per sweep)
// MODE 3: MODE 0 with s_sleep 1 between sweeps
// SAUX: cache policy of the record store (0 plain, 16 sc1 write-through)
template <int MODE, int SAUX>
__global__ __launch_bounds__(576) void k_pure(Params p) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    if (b % p.stride != 0) return;
    const int k = b / p.stride;
    if (k >= p.G) return;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x >= 64) return;   // (extra waves of a bigger workgroup: only its residency matters here)
    const int G = p.G;
    const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc(p.rec, 0, 2u * G * 32u, 0x00020000);
    if (lane == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicOr(&p.out[3], 1ull << (xcc & 15));
    }
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long sweeps = 0;
    int fail = 0;
    const int ngran = 2 * G;   // granules of a parity block
    for (int ep = 1; ep <= p.epochs && !fail; ++ep) {
        const unsigned pbase = (unsigned)((ep & 1) * G) * 32u;
        if (p.skew && (ep & 15) == 0 && (ep >> 4) % G == k) {
            const unsigned long long w0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - w0 < (unsigned long long)p.skew) __builtin_amdgcn_s_sleep(4);
        }
        if (lane < 2) {
            v4i g = {ep, k * 1000 + lane, ep, ep ^ 0x5555};
            __builtin_amdgcn_raw_buffer_store_b128(g, rrec, pbase + (unsigned)k * 32u + lane * 16u, 0, SAUX);
        }
        unsigned spins = 0;
        bool ok;
        if (MODE == 2) {
            const unsigned off = pbase + (unsigned)(lane < G ? lane : 0) * 32u;
            do {
                v4i a = __builtin_amdgcn_raw_buffer_load_b128(rrec, off, 0, 16);
                v4i c = __builtin_amdgcn_raw_buffer_load_b128(rrec, off + 16, 0, 16);
                ok = lane >= G || (a.x == ep && a.z == ep && c.x == ep && c.z == ep);
                ok = __all(ok);
                ++sweeps;
                if (!ok && (++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull) fail = 1;
            } while (!ok && !fail);
        } else if (MODE == 1) {
            const unsigned off = pbase + (unsigned)(lane < ngran ? lane : 0) * 16u;
            v4i a0 = __builtin_amdgcn_raw_buffer_load_b128(rrec, off, 0, 16);
            do {
                v4i a1 = __builtin_amdgcn_raw_buffer_load_b128(rrec, off, 0, 16);
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                ok = lane >= ngran || (a0.x == ep && a0.z == ep);
                ok = __all(ok);
                a0 = a1;
                ++sweeps;
                if (!ok && (++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull) fail = 1;
            } while (!ok && !fail);
        } else {
            const unsigned off = pbase + (unsigned)(lane < ngran ? lane : 0) * 16u;
            do {
                v4i a = __builtin_amdgcn_raw_buffer_load_b128(rrec, off, 0, 16);
                ok = lane >= ngran || (a.x == ep && a.z == ep);
                ok = __all(ok);
                ++sweeps;
                if (MODE == 3 && !ok) __builtin_amdgcn_s_sleep(1);
                if (!ok && (++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - rt0 > 20000000ull) fail = 1;
            } while (!ok && !fail);
        }
        if (p.work) {
            const unsigned long long w0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - w0 < (unsigned long long)p.work) {}
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (fail && lane == 0) atomicAdd(&p.out[2], 1ull);
    if (k == 0 && lane == 0) {
        p.out[0] = t1 - t0;
        p.out[1] = sweeps;
        p.out[4] = rt1 - rt0;
    }
}

template <int MODE, int SAUX>
void run(const char* what, int G, int stride, int work, char* rec, unsigned long long* out, int skew = 0, int threads = 64) {
    const int epochs = 4000;
    const size_t shm = 84 * 1024;   // one workgroup per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pure<MODE, SAUX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(rec, 0, 2 * 256 * 32));
        CHECK(hipMemset(out, 0, 64));
        Params p{G, epochs, stride, work, skew, rec, out};
        hipLaunchKernelGGL((k_pure<MODE, SAUX>), G * stride, threads, shm, 0, p);
        CHECK(hipDeviceSynchronize());
        CHECK(hipGetLastError());
        unsigned long long h[8];
        CHECK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
        printf("%-44s G=%3d stride=%d work=%4d rep=%d: %6.0f cycles/epoch (hop %6.0f) = %.3f us, %.2f sweeps/epoch, timeouts=%llu xcc=0x%llx\n",
               what, G, stride, work, rep, (double)h[0] / epochs, (double)h[0] / epochs - work, (double)h[4] / epochs / 100.0,
               (double)h[1] / epochs, h[2], h[3]);
        fflush(stdout);
    }
}

int main() {
    char* rec;
    unsigned long long* out;
    CHECK(hipMalloc(&rec, 2 * 256 * 32));
    CHECK(hipMalloc(&out, 64));
    run<0, 0>("one 64-lane sweep, plain stores", 1, 8, 0, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 2, 8, 0, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 8, 8, 0, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 32, 8, 0, rec, out);
    run<1, 0>("two sweeps in flight, plain stores", 32, 8, 0, rec, out);
    run<2, 0>("lane = record, two loads, plain stores", 32, 8, 0, rec, out);
    run<3, 0>("one sweep + s_sleep 1, plain stores", 32, 8, 0, rec, out);
    run<0, 16>("one 64-lane sweep, sc1 stores", 32, 8, 0, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 32, 8, 1000, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 32, 8, 2000, rec, out);
    run<1, 0>("two sweeps in flight, plain stores", 32, 8, 1500, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 24, 8, 0, rec, out);
    run<0, 0>("one 64-lane sweep, plain stores", 16, 8, 0, rec, out);
    run<0, 16>("all XCDs: one sweep, sc1 stores", 32, 1, 0, rec, out);
    run<1, 16>("all XCDs: two sweeps in flight, sc1 stores", 32, 1, 0, rec, out);
    run<0, 16>("all XCDs: one sweep, sc1 stores", 32, 1, 1500, rec, out);
    // one workgroup late every 16th epoch: do the early pollers ever get stuck on a stale line?
    run<0, 0>("SKEW 20k: one sweep, plain", 32, 8, 0, rec, out, 20000);
    run<1, 0>("SKEW 20k: two sweeps in flight, plain", 32, 8, 0, rec, out, 20000);
    run<2, 0>("SKEW 20k: lane = record, plain", 32, 8, 0, rec, out, 20000);
    run<0, 16>("SKEW 20k: all XCDs one sweep, sc1", 32, 1, 0, rec, out, 20000);
    run<1, 16>("SKEW 20k: all XCDs two sweeps, sc1", 32, 1, 0, rec, out, 20000);
    run<2, 16>("SKEW 20k: all XCDs lane = record, sc1", 32, 1, 0, rec, out, 20000);
    run<1, 0>("SKEW 200k: two sweeps in flight, plain", 32, 8, 0, rec, out, 200000);
    run<1, 16>("SKEW 200k: all XCDs two sweeps, sc1", 32, 1, 0, rec, out, 200000);
    run<1, 0>("SKEW 20k, 576 threads: two sweeps, plain", 32, 8, 0, rec, out, 20000, 576);
    run<1, 16>("SKEW 20k, 576 threads: all XCDs two sweeps, sc1", 32, 1, 0, rec, out, 20000, 576);
    return 0;
}
