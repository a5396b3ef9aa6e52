// scene_obj.h — the device-resident scene behind a psdr_hip_scene handle, shared by the two host translation units of
// libpsdr_hip.so: scene_build.hip (psdr_hip_scene_create / _update: tree build and refit, blob layout, uploads - what the
// reference does in Scene_OptiX::configure + the jit uploads of Scene::configure, src/scene/scene_optix.cpp:265-332,
// src/scene/scene.cpp:311-599) and api.hip (the render entry points and their kernels).
#pragma once
#include "../common/threads.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/psdr_hip.h"
#include "scene_dev.h"

namespace psdr {
// error plumbing of the C ABI (api.hip): stores the message psdr_hip_last_error() returns, -> 1
int api_fail(const std::string &msg);
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return psdr::api_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

namespace psdr {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;                    // size of the allocation
    ~DevBuf() { if (p) (void) hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    // (re)allocates when the size differs; *moved (optional) = the device pointer changed
    int ensure(size_t n, bool *moved = nullptr) {
        if (n == 0) n = 16;
        if (p && n == bytes) { if (moved) *moved = false; return 0; }
        if (p) { HIPCHK(hipFree(p)); p = nullptr; bytes = 0; }
        HIPCHK(hipMalloc(&p, n));
        bytes = n;
        if (moved) *moved = true;
        return 0;
    }
    int upload(const void *src, size_t n) {
        if (n == 0) n = 16;
        if (ensure(n)) return 1;
        if (src) HIPCHK(hipMemcpy(p, src, n, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(p, 0, n));
        return 0;
    }
    template <typename T> const T *as() const { return reinterpret_cast<const T *>(p); }
};

// host memory the DMA engines read directly (the scene blob's staging copy: psdr_hip_scene_update rewrites the sections a change
// touches in place and sends exactly those)
struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~PinnedBuf() { if (p) (void) hipHostFree(p); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    int ensure(size_t n) {
        if (p && n <= bytes) return 0;
        void *q = nullptr;
        HIPCHK(hipHostMalloc(&q, n, hipHostMallocDefault));
        if (p) { std::memcpy(q, p, bytes); (void) hipHostFree(p); }
        p = q; bytes = n;
        return 0;
    }
};

constexpr unsigned kQueueRing = 1024;
constexpr int kTermSlices = 3;            // slices of the traversal stack's global tail: interior, primary-edge and secondary-edge launches of one renderD run concurrently

// Guide table of a discrete distribution (shade.h::sample_reuse_guided): entry k = the index DiscreteDistribution::sample_reuse finds for the
// sample k / n_buckets, in the kernels' own float arithmetic (s * sum, then the first i < size - 1 whose running sum is not < s, else size - 1).
// n_buckets: the power of two nearest to size / 16 (at least 1); tables below 256 entries are not worth one (-> empty).
// per_bucket: entries left to the binary search (32: one or two cache lines of the large environment / guiding tables; the edge
// distributions - tens of thousands of entries, read by every edge sample - get 4)
inline void build_cdf_guide(const float *cmf, int size, float sum, std::vector<int> &guide, int per_bucket = 32) {
    guide.clear();
    if (size < 256) return;
    int nb = 1;
    while (nb * per_bucket <= size) nb <<= 1;
    guide.resize((size_t) nb + 1);
    // (32 768 binary searches over 122 885 edges for config 5's secondary-edge table, in every update that moves a vertex: on the host team)
    psdr::parallel_for((size_t) nb + 1, 2048, [&](size_t kb, size_t ke) {
        for (size_t k = kb; k < ke; ++k) {
            const float s = ((float) k / (float) nb) * sum;
            guide[k] = (int) (std::partition_point(cmf, cmf + (size - 1), [s](float c) { return c < s; }) - cmf);
        }
    });
}

} // namespace psdr

struct psdr_hip_scene {
    psdr::SceneTables T{};
    psdr::DevBuf blob;                   // capacity >= T.blob_words float4 words (grown on demand)
    psdr::PinnedBuf hblob;               // the host copy of the blob the sections are written into
    bool lds = false;                    // scene class 1: staged in LDS (scene_dev.h)
    bool lean = false;                   // scene class 2: global memory, Diffuse BSDFs + area lights + environment map only
    bool has_nmap = false;               // a NormalMap BSDF is present (material sweep even without a material table)
    bool simple_mats = true;             // every BSDF is Diffuse, Microfacet (constants or bitmaps) or a constant RoughConductor (the material sweep of adjoint_mat.h applies)
    bool lds_mat = false;                // scene class 3: staged in LDS, any BSDF / bitmap parameter, no environment map (forward kernels)
    size_t smem_bytes = 0;
    psdr::SecEdgeTables E{};
    // device arrays outside the blob, by name ("env.radiance", "tex.3.1", "sensor.0.live", ...): an update re-uses the allocation when the
    // size is the same and skips the copy when the caller vouches for the content (psdr_hip_scene_update's `same` mask)
    std::map<std::string, std::unique_ptr<psdr::DevBuf>> named;
    std::vector<psdr::SensorDev> sensors;
    psdr::DevBuf counters;
    psdr::DevBuf queues;                 // ring of work-queue heads, one per path-kernel launch
    psdr::DevBuf gstack;                 // traversal-stack entries beyond the LDS part (trav4.h): kTermSlices slices of gstack_slice ints, one per concurrently running term
    size_t gstack_slice = 0;
    mutable psdr::DevBuf adj_rec;        // per-lane records of the interior adjoint when they do not fit LDS (deep paths), grown on demand
    mutable size_t adj_rec_bytes = 0;
    mutable unsigned queue_slot = 0;
    mutable bool adj_attr_set = false;   // the adjoint kernels' dynamic-LDS limit has been raised on this scene's device
    int n_leaves = 0, max_depth = 0, grid = 0;
    long long tex_total = 0;             // floats of all bitmap parameters (psdr_grads.g_tex)
    std::vector<std::vector<unsigned>> live_host;      // per sensor: the live-pixel mask (empty = every pixel is live), kept for psdr_hip_scene_live_pixels
    psdr::DevBuf hot_map, hot_inv;       // adjoint accumulators kept in LDS: emitter triangles first, then by area (adjoint.h)
    int n_hot = 0;
    std::vector<long long> tex_layout;   // [3*n_bsdfs] offsets into g_tex, -1 = constant

    // ---- the tree and what a refit needs (scene_build.hip)
    std::vector<int32_t> order, orig2slot;             // device triangle slot -> original triangle id and back; fixed until the next tree build
    int tree_tris = -1;                                // triangles the tree was built for (-1: none yet)
    int tree_max_stack = 0, tree_ref_bits = 0;
    psdr::DevBuf refit_order;                          // node ids sorted by height above the leaves (children before parents)
    psdr::DevBuf refit_box;                            // float box of every node: lo.xyz hi.xyz, [n_nodes*6]
    psdr::DevBuf refit_cost;                           // one double: the tree's SAH cost after a refit
    std::vector<int> refit_level_begin;                // [levels + 1] ranges of refit_order, one kernel launch per level
    double cost_built = 0.0;                           // SAH cost of the tree when it was built; a refit that exceeds kRebuildFactor x this triggers a build
    std::vector<float> sensor_w2s;                     // world_to_sample of every sensor at the last live-mask build (a mask is rebuilt when it or the triangles changed)
    psdr_update_info last_info{};
    // geometry on the device (scene_build.hip::geometry_on_device): the topology version and the counts of every mesh at the last topology upload
    std::vector<uint64_t> geo_versions;
    std::vector<int> geo_counts;
    bool poisoned = false;                             // a psdr_hip_scene_update failed midway: no rendering until an update has gone through (scene_build.hip)

    // The launches of one scene share mutable device scratch - the work-queue ring, the counters, the traversal-stack overflow
    // `gstack` (indexed by workgroup and thread only; the forked terms of ONE call get a slice each) and the adjoint records `adj_rec` (re-allocated when they grow) - while the C
    // ABI takes a stream per call.  They are therefore SERIALISED ACROSS STREAMS (ScratchGuard below): a call on another stream than
    // the scene's previous one first makes its stream wait for that call's completion event; calls on one stream order themselves.
    mutable std::mutex mu;
    mutable hipEvent_t ev = nullptr;
    mutable hipStream_t last_stream = nullptr;
    mutable bool have_last = false;
    // side streams for the edge terms of a renderD (api.hip::render_impl): forked from the caller's stream, joined before the call returns
    mutable hipStream_t aux[2] = {nullptr, nullptr};
    mutable hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    int make_term_streams() const {
        if (aux[0]) return 0;
        for (int k = 0; k < 2; ++k) {
            if (hipStreamCreateWithFlags(&aux[k], hipStreamNonBlocking) != hipSuccess) return 1;
            if (hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming) != hipSuccess) return 1;
        }
        return hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess;
    }
    ~psdr_hip_scene() {
        if (ev) (void) hipEventDestroy(ev);
        for (int k = 0; k < 2; ++k) { if (aux[k]) (void) hipStreamDestroy(aux[k]); if (ev_join[k]) (void) hipEventDestroy(ev_join[k]); }
        if (ev_fork) (void) hipEventDestroy(ev_fork);
    }
    psdr::DevBuf &buf(const std::string &key) {
        auto it = named.find(key);
        if (it == named.end()) it = named.emplace(key, std::make_unique<psdr::DevBuf>()).first;
        return *it->second;
    }
};

// one in-flight user of a scene's scratch buffers per stream order (see psdr_hip_scene::mu)
struct ScratchGuard {
    const psdr_hip_scene *sc;
    hipStream_t st;
    hipError_t err = hipSuccess;
    ScratchGuard(const psdr_hip_scene *s, void *stream) : sc(s), st((hipStream_t) stream) {
        sc->mu.lock();
        if (!sc->ev) err = hipEventCreateWithFlags(&sc->ev, hipEventDisableTiming);
        if (err == hipSuccess && sc->have_last && sc->last_stream != st) err = hipStreamWaitEvent(st, sc->ev, 0);
    }
    ~ScratchGuard() {
        if (sc->ev && hipEventRecord(sc->ev, st) == hipSuccess) { sc->last_stream = st; sc->have_last = true; }
        sc->mu.unlock();
    }
};
#define SCRATCH_GUARD(sc, stream) if ((sc)->poisoned) return psdr::api_fail("the scene's last psdr_hip_scene_update failed: update it again (everything is re-sent) before using it"); \
    ScratchGuard guard_((sc), (stream)); if (guard_.err != hipSuccess) return psdr::api_fail(std::string("scene scratch serialisation: ") + hipGetErrorString(guard_.err))
