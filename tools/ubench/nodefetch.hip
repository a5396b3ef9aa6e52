// Micro-benchmark: what does a BVH node fetch cost the L1 path of a gfx950 CU, by access pattern?
//
// Every chain is a pseudo-random walk over an array of 64-byte nodes (next index = a function of all 64 loaded bytes), as a ray's
// walk over the tree is: dependent fetches of 64 lanes on 64 different lines.  The question: is a node step
// paid per load INSTRUCTION, per LANE of a load, or per cache LINE touched?
//
//   lane4   one chain per lane, four 16-byte loads into the lane's node           (trav4.h::t4_node today)
//   lane2   one chain per lane, 32-byte nodes, two loads
//   lane1   one chain per lane, 16-byte nodes, one load
//   quad1   one chain per QUAD: lane q of a quad loads word q of the quad's node   (one instruction = 16 whole nodes)
//   coop4   one chain per lane, fetched by its quad: in round j the four lanes load the four words of lane j's node
//           (four instructions = 64 whole nodes, each instruction touches 16 lines instead of 64; the words arrive transposed)
//   lds4    as coop4, but the loads go straight to LDS (global_load_lds_dwordx4) and each lane reads its node back with four ds_read_b128
//
// hipcc --offload-arch=gfx950 -O3 nodefetch.hip -o nodefetch.bin && ./nodefetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned quad_bcast(unsigned v, int j) {
    switch (j) {
    case 0: return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0x00, 0xf, 0xf, true);
    case 1: return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0x55, 0xf, 0xf, true);
    case 2: return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0xaa, 0xf, 0xf, true);
    default: return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0xff, 0xf, 0xf, true);
    }
}
__device__ __forceinline__ unsigned quad_xor(unsigned v) {
    v ^= (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0xb1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
    v ^= (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0x4e, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ unsigned fold(u4 a) { return a.x ^ a.y ^ a.z ^ a.w; }

// MODE: 0 lane4, 1 lane2, 2 lane1, 3 quad1, 4 coop4, 5 lds4.  `active_mod`: a lane takes part when (lane % active_mod) == 0 ... (1 = all)
template <int MODE>
__global__ void __launch_bounds__(256) k_walk(const u4 *__restrict__ nodes, unsigned n_mask, int steps, unsigned long long live, unsigned *out) {
    extern __shared__ u4 lds[];
    const int lane = threadIdx.x & 63, q = lane & 3, wave = threadIdx.x >> 6;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    if (MODE == 3) idx = (blockIdx.x * 64u + (threadIdx.x >> 2)) * 2654435761u;
    idx &= n_mask;
    unsigned acc = 0;
    const unsigned salt0 = idx | 1u;          // chains must not merge (a random map's walks all end in the same few cycles, which then sit in the L1)
    const bool on = MODE == 3 ? ((live >> (lane & ~3)) & 1ull) : ((live >> lane) & 1ull);
    if (MODE == 0) {
        if (on) for (int s = 0; s < steps; ++s) {
            const u4 *p = nodes + 4 * (size_t) idx;
            const u4 a = p[0], b = p[1], c = p[2], d = p[3];
            const unsigned h = fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
            acc += h; idx = (h + salt0 * (unsigned) (s + 1)) & n_mask;
        }
    } else if (MODE == 1) {
        if (on) for (int s = 0; s < steps; ++s) {
            const u4 *p = nodes + 4 * (size_t) idx;
            const u4 a = p[0], b = p[1];
            const unsigned h = fold(a) ^ fold(b);
            acc += h; idx = (h + salt0 * (unsigned) (s + 1)) & n_mask;
        }
    } else if (MODE == 2) {
        if (on) for (int s = 0; s < steps; ++s) {
            const u4 a = nodes[4 * (size_t) idx];
            const unsigned h = fold(a);
            acc += h; idx = (h + salt0 * (unsigned) (s + 1)) & n_mask;
        }
    } else if (MODE == 3) {
        if (on) for (int s = 0; s < steps; ++s) {
            const u4 a = nodes[4 * (size_t) idx + q];
            const unsigned h = quad_xor(fold(a));
            acc += h; idx = (h + salt0 * (unsigned) (s + 1)) & n_mask;
        }
    } else if (MODE == 4) {
        // (idle lanes still help their quad fetch: `on` only decides whose chains advance - a quad none of whose lanes is on sits out)
        const bool quad_on = ((live >> (lane & ~3)) & 15ull) != 0ull;
        if (quad_on) for (int s = 0; s < steps; ++s) {
            unsigned h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ij = quad_bcast(idx, j);
                const bool need = (live >> ((lane & ~3) + j)) & 1ull;
                u4 a = {0, 0, 0, 0};
                if (need) a = nodes[4 * (size_t) ij + q];
                h[j] = fold(a);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = quad_xor(h[j]);
            const unsigned mine = q == 0 ? h[0] : q == 1 ? h[1] : q == 2 ? h[2] : h[3];
            acc += mine; idx = (mine + salt0 * (unsigned) (s + 1)) & n_mask;
        }
    } else if (MODE == 5) {
        const bool quad_on = ((live >> (lane & ~3)) & 15ull) != 0ull;
        // per wave 4 rounds x 1 KB: round j's instruction writes lane l's 16 bytes at  base + j KB + 16 l
        u4 *mine_lds = lds + wave * 256;
        if (quad_on) for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned ij = quad_bcast(idx, j);
                const bool need = (live >> ((lane & ~3) + j)) & 1ull;
                if (need) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (nodes + 4 * (size_t) ij + q),
                                                           (__attribute__((address_space(3))) void *) (mine_lds + 64 * j), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // lane l = 4 k + j owns the node fetched in round j by quad k: words at (64 j + 4 k) .. + 3
            const u4 *r = mine_lds + 64 * q + (lane & ~3);
            const u4 a = r[0], b = r[1], c = r[2], d = r[3];
            const unsigned hh = fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (on) { acc += hh; idx = (hh + salt0 * (unsigned) (s + 1)) & n_mask; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static double run(int mode, const u4 *d_nodes, unsigned n_mask, int steps, unsigned long long live, unsigned *d_out, int blocks, int lds_bytes, unsigned *h_sum) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        switch (mode) {
        case 0: hipLaunchKernelGGL(k_walk<0>, dim3(blocks), dim3(256), lds_bytes, 0, d_nodes, n_mask, steps, live, d_out); break;
        case 1: hipLaunchKernelGGL(k_walk<1>, dim3(blocks), dim3(256), lds_bytes, 0, d_nodes, n_mask, steps, live, d_out); break;
        case 2: hipLaunchKernelGGL(k_walk<2>, dim3(blocks), dim3(256), lds_bytes, 0, d_nodes, n_mask, steps, live, d_out); break;
        case 3: hipLaunchKernelGGL(k_walk<3>, dim3(blocks), dim3(256), lds_bytes, 0, d_nodes, n_mask, steps, live, d_out); break;
        case 4: hipLaunchKernelGGL(k_walk<4>, dim3(blocks), dim3(256), lds_bytes, 0, d_nodes, n_mask, steps, live, d_out); break;
        default: hipLaunchKernelGGL(k_walk<5>, dim3(blocks), dim3(256), lds_bytes, 0, d_nodes, n_mask, steps, live, d_out); break;
        }
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(blocks * 256);
    hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned s = 0; for (unsigned v : h) s += v;
    *h_sum = s;
    return ms;
}

int main(int argc, char **argv) {
    const int steps = 2000;
    const char *names[6] = {"lane4", "lane2", "lane1", "quad1", "coop4", "lds4 "};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, %.2f GHz\n", prop.name, cus, clk * 1e-9);
    for (int log_n = 8; log_n <= 20; log_n += 3) {                 // 16 k nodes (1 MB: L2-resident, config 5's tree is 1.7 MB), 128 k (8 MB), 1 M (64 MB: past the L2s)
        const unsigned n = 1u << log_n;
        std::vector<unsigned> h_nodes((size_t) n * 16);
        unsigned long long x = 88172645463325252ull;
        for (auto &w : h_nodes) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (unsigned) (x >> 16); }
        u4 *d_nodes; unsigned *d_out;
        hipMalloc(&d_nodes, h_nodes.size() * 4);
        hipMemcpy(d_nodes, h_nodes.data(), h_nodes.size() * 4, hipMemcpyHostToDevice);
        for (int wg_per_cu = 2; wg_per_cu <= 8; wg_per_cu *= 2) {
            const int blocks = cus * wg_per_cu;
            const int lds_bytes = 160 * 1024 / wg_per_cu - 1024;        // LDS decides how many workgroups share a CU
            hipMalloc(&d_out, (size_t) blocks * 256 * 4);
            for (int li = 0; li < 3; ++li) {
                // live lanes: all; every second lane (scattered over all quads); the first two quads of every four (whole quads idle)
                const unsigned long long live = li == 0 ? ~0ull : li == 1 ? 0x5555555555555555ull : 0x00ff00ff00ff00ffull;
                for (int mode = 0; mode < 6; ++mode) {
                    hipFuncSetAttribute((const void *) k_walk<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    hipFuncSetAttribute((const void *) k_walk<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    hipFuncSetAttribute((const void *) k_walk<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    hipFuncSetAttribute((const void *) k_walk<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    hipFuncSetAttribute((const void *) k_walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    hipFuncSetAttribute((const void *) k_walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    unsigned sum = 0;
                    const double ms = run(mode, d_nodes, n - 1, steps, live, d_out, blocks, lds_bytes, &sum);
                    const int live_lanes = __builtin_popcountll(live);
                    int live_quads = 0; for (int k = 0; k < 16; ++k) live_quads += (int) ((live >> (4 * k)) & 1ull);
                    const double chains = (double) blocks * 4 * (mode == 3 ? live_quads : live_lanes);
                    const double node_steps = chains * steps;
                    printf("nodes 2^%d  wg/cu %d  live %2d/64 (%s)  %s  %8.3f ms  %7.2f Gnode/s  %6.2f clk per node and CU  sum %08x\n", log_n, wg_per_cu, live_lanes,
                           li == 0 ? "all" : li == 1 ? "alt lanes" : "alt quad pairs", names[mode], ms, node_steps / ms * 1e-6, ms * 1e-3 * clk * cus / node_steps, sum);
                }
            }
            hipFree(d_out);
        }
        hipFree(d_nodes);
    }
    return 0;
}
