// gfx950 kernels of the wavefront path tracer and the device half of the C ABI (include/rt.h).
//
// One render pass traces S camera samples of every pixel this context owns as P = pixels * S paths that
// advance in lock step, one bounce per iteration:
//
//   k_raygen                 slot = local_pixel * S + s  ->  primary ray (record `slot` of bounce 0's ray array)
//   per bounce it = 0 .. max_depth-1
//     k_extend_tree          closest hit of every live path: persistent waves, one 1024-thread workgroup per CU,
//                            per-lane stacks and the top of the node array in LDS
//     (k_extend_list          instead, for a scene whose world is all on the lock-step list: one ray per lane, the list
//                            front to back)
//     k_shade_main           misses take the background and end; hits on materials with a plain colour: rebuild the
//                            hit record, emitted + scatter, and write the surviving path's next ray and throughput
//                            as a record of the next bounce (one atomic per workgroup); hits on computed textures
//                            (Perlin, image, checker) are queued by material type
//     k_shade<bucket, kinds> the same shading for one queue of computed-texture hits, with the code for the kinds of
//                            texture the scene holds
//   k_accumulate             per pixel, add its S sample radiances in sample order to the f64 sum buffer
//
// Path state per bounce is dense: a 64-byte ray record (with the path's slot and pixel), a 24-byte throughput record
// (none at the first bounce, where it is 1) and a 16-byte hit record per live path, double-buffered. The radiance array
// (3 doubles per slot) is written exactly once per path, by the kernel in which the path ends - every path ends
// (miss, absorption, light, or the last bounce), so no slot keeps a value from an earlier pass.
// All queue lengths live in a device counter table that the kernels read themselves, so the host never waits
// between launches inside a pass. Random numbers are Philox keyed on (pixel, sample, bounce): results do not
// depend on how pixels are split over GPUs or on scheduling.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt.h"
#include "rt_builder.h"
#include "rt_device.h"
#include "rt_tree_build.h"

// Nothing may unwind across the C ABI: the entry points are function-try-blocks that end in this.
#define RT_ABI_CATCH catch (const std::bad_alloc&) { return RT_ERR_OOM; } catch (...) { return RT_ERR_INVALID; }

#define RT_BLOCK 256
static_assert(RT_BLOCK % 64 == 0, "kernels hand data from lane to lane inside 64-lane waves");
#define RT_TREE_BLOCK 1024 /* k_extend_tree: one workgroup per CU, so that the whole CU shares one LDS copy of the top of the tree */
#define RT_COUNTER_STRIDE 16u /* per iteration: [0] paths entering it at the front of the arrays, [1+q] hits in queue q, [RT_COUNTER_BACK] paths entering it at the back */
#define RT_COUNTER_BACK 15u
#define RT_PATH_BYTES 232u /* device memory per path of a pass: two 64-byte ray records, two 24-byte throughputs, 24 bytes of radiance, a 16-byte hit record, four queue words */

// ---- device structures ------------------------------------------------------------------------------------
// Path state. The paths alive at a bounce are DENSE: path number i of the bounce owns record i of the bounce's ray,
// throughput and hit arrays, and a shade kernel writes a surviving path's next ray and throughput to the record the
// path gets in the next bounce's arrays (two sets, alternating). So every kernel streams through contiguous records
// and nothing is read through an index; what ties a record to its pixel is the slot number it carries. Only the
// radiance array is indexed by slot, and it is written once per path, when the path ends.
struct PathRay { double ox, oy, oz, dx, dy, dz, tm; uint32_t slot, pixel; }; // current ray (ray.rs:5-9) + whose it is: slot = local_pixel * S + sample, pixel = y * width + x
struct PathId { uint32_t slot, pixel; };
struct PathThr { double r, g, b; };                                       // product of the attenuations so far (main.rs:70); 1, 1, 1 at the first bounce, where it is neither stored nor read
struct PathBufs {
    PathRay* ray[2];  // [it & 1]: rays entering bounce it
    PathThr* thr[2];
    double* rad;      // 3 per slot: what the path adds to its pixel
    RtHitLite* hit;   // per path of the current bounce
};
struct PassInfo {
    const uint32_t* pixmap; // local pixel -> global pixel index y*width + x
    uint32_t n_local, S, sample_offset, width, height, compat, bg_mode, max_depth;
    double bg[3];
    uint64_t seed;
};
// The buckets that are shaded by a kernel of their own (the others are shaded where they are found, in k_shade_main)
// have a queue each: queue number of a bucket, and how many there are.
#define RT_QUEUES 4u
constexpr uint32_t rt_bucket_queue(uint32_t bucket) {
    return bucket == RT_BUCKET_LAMBERTIAN_TEX ? 0u : bucket == RT_BUCKET_LIGHT_TEX ? 1u : bucket == RT_BUCKET_ISOTROPIC_TEX ? 2u
         : bucket == RT_BUCKET_LAMBERTIAN_NOISE ? 3u : 0xFFFFFFFFu;
}
struct QueueSet {
    uint32_t* bucket;           // RT_QUEUES queues of bucket_cap path numbers each
    size_t bucket_cap;
    uint32_t* counters;         // (max_depth + 2) * RT_COUNTER_STRIDE
    uint32_t* error_flag;       // bit 0: traversal stack overflow
};

// The paths entering a bounce are written from both ends of the record arrays: paths that left a surface from the
// front, paths that scattered inside a medium from the back (they meet different parts of the scene, and waves that
// hold one kind only diverge less: measured 10 % of traversal time). Path i of n = front + back is record i, or
// counted down from the end.
struct LiveSet {
    uint32_t front, back, cap;
    __device__ uint32_t n() const { return front + back; }
    __device__ uint32_t record(uint32_t i) const { return i < front ? i : cap - 1u - (i - front); }
};
__device__ inline LiveSet live_set(const uint32_t* ctr, size_t cap) {
    LiveSet l;
    l.front = ctr[0]; l.back = ctr[RT_COUNTER_BACK]; l.cap = (uint32_t)cap;
    return l;
}

// The newest entry is held in a register (`tos`), the older ones in LDS: a pop hands out the register at once and only
// then asks the LDS for the entry below, which nothing needs before the next pop or push - so the wave does not sit out an
// LDS round trip in front of the node fetch that follows every pop. Everything is kept as LDS byte addresses (a push or
// pop is an add, the slot to write is `top` itself), and two things hold throughout: `tos` is RT_REF_NONE while nothing
// is held, and so is the word at `bottom` (the first push writes the empty stack's `tos` there) - so "pop, or
// RT_REF_NONE from an empty stack" needs no look at whether it is empty: it hands out `tos` and steps back to
// max(top - stride, bottom). Entry j (counted from the bottom) of the ones in LDS is at bottom + (j + 1) * stride.
struct LdsStack {
    typedef __attribute__((address_space(3))) uint32_t* Word;
    uint32_t bottom, top, lim, stride; // this lane's column (stride = 4 * threads per block: conflict-free); the slot the next push writes; the column's end
    uint32_t tos;
    uint32_t* error_flag;
    static __device__ Word at(uint32_t a) { return (Word)(uintptr_t)a; }
    __device__ void init(uint32_t* column, uint32_t cap, uint32_t threads, uint32_t* flag) {
        bottom = (uint32_t)(uintptr_t)(Word)column;
        top = bottom; tos = RT_REF_NONE; stride = 4u * threads; lim = bottom + cap * stride; error_flag = flag;
        *at(bottom) = RT_REF_NONE;
    }
    __device__ void clear() { top = bottom; tos = RT_REF_NONE; }
    __device__ void push(uint32_t v) {
        if (top < lim) { *at(top) = tos; tos = v; top += stride; }
        else if (error_flag) atomicOr(error_flag, 1u);
    }
    __device__ uint32_t below() const { return (uint32_t)max((int)(top - stride), (int)bottom); }
    __device__ uint32_t pop() { const uint32_t v = tos; top -= stride; tos = *at(top); return v; }
    __device__ uint32_t pop_or_none() { const uint32_t v = tos; top = below(); tos = *at(top); return v; }
    // What a pop would uncover, asked for before it is known whether one follows (rt_trav_node_step asks together with its
    // node record, so that one wait covers both). Nothing held, or one entry: the RT_REF_NONE at the bottom.
    __device__ uint32_t prefetch() const { return *at(below()); }
    // The stack's part of a node step, without branches but for the store: both boxes hit - `keep` is pushed and the lane
    // goes on with `go`; one box - on with `go`; none - the top entry, or RT_REF_NONE from an empty stack.
    // ROOM_CHECKED = false: the caller knows that the stack cannot overflow (rt_scene::stack_verified: the scene's upload
    // has walked the tree and found the header's stack_need sufficient) - no comparison, no branch around the store's
    // other outcome: -3 % of the traversal kernel's time on the Book-2 scene.
    template <bool ROOM_CHECKED = true>
    __device__ uint32_t node_step(bool both, bool any, uint32_t go, uint32_t keep, uint32_t under) {
        const bool pushing = both && (!ROOM_CHECKED || top < lim);
        if (pushing) *at(top) = tos;
        else if (both && error_flag) atomicOr(error_flag, 1u);
        const uint32_t cur = any ? go : tos;
        tos = pushing ? keep : (any ? tos : under);
        top = pushing ? top + stride : (any ? top : below());
        return cur;
    }
    __device__ bool empty() const { return top == bottom; }
};

// The first n_cached node records live in LDS, the rest in global memory. A lane fetches a 64-byte record with four
// 16-byte reads, each lane a different record: through the vector cache that is one tag lookup per lane per read,
// 256 per wave and node, which is what bounds traversal when the nodes come from global memory. The LDS serves the
// same reads at its full width. The four 16-byte pieces of the records are kept in four arrays of their own (piece p of
// record i at p * n_cached + i): the lanes of one read, each after a different record, then spread over all banks as
// their indices do, and the four addresses of a fetch cost one shift-and-add each.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* LdsU4; // typed as LDS so that reads are ds_read, not flat loads
// ALL = true: every node of the scene is in LDS. The kernel is then compiled without the global-memory branch, and that
// matters beyond the branch: where both branches exist they write the same registers, so the compiler has to make the LDS
// reads wait for every outstanding vector-memory operation of the wave (s_waitcnt vmcnt(0): on gfx9 that counts stores
// too) - each box step then waits for the hit record a lane has just published and for any leaf record on its way.
template <bool ALL>
struct CachedNodes {
    LdsU4 lds;
    uint32_t n_cached;
    const RtNode* global;
    const RtExactBox* exact; // f64 bounds of the boxes whose verdict is behaviour (RtNode::pad[1])
#ifdef RT_BOUNDS_CHECK
    uint32_t n_nodes;
#endif
    const __attribute__((address_space(3))) double* exact_lds; // ... and the workgroup's copy of the first n_exact_lds of them
    uint32_t n_exact_lds;
    __device__ RtExactBox exact_box(uint32_t k) const {
        if (k >= n_exact_lds) return exact[k];
        RtExactBox b;
        const __attribute__((address_space(3))) double* p = exact_lds + (size_t)k * 6u;
        b.lo[0] = p[0]; b.lo[1] = p[1]; b.lo[2] = p[2]; b.hi[0] = p[3]; b.hi[1] = p[4]; b.hi[2] = p[5];
        return b;
    }
    __device__ static uint32_t place(uint32_t index, uint32_t piece, uint32_t n_cached) { return piece * n_cached + index; }
    __device__ RtNode load(uint32_t index) const {
        u32x4 a, b, c, d; // (d as a full 16-byte read too: a narrower one gets merged with the other branch's into a flat load)
        if (ALL || index < n_cached) {
            a = lds[index]; b = (lds + n_cached)[index]; c = (lds + 2u * n_cached)[index]; d = (lds + 3u * n_cached)[index];
        } else {
            const u32x4* g = (const u32x4*)(global + index);
            a = g[0]; b = g[1]; c = g[2]; d = g[3];
        }
        RtNode n;
        n.lo0[0] = __uint_as_float(a.x); n.lo0[1] = __uint_as_float(a.y); n.lo0[2] = __uint_as_float(a.z);
        n.hi0[0] = __uint_as_float(a.w); n.hi0[1] = __uint_as_float(b.x); n.hi0[2] = __uint_as_float(b.y);
        n.lo1[0] = __uint_as_float(b.z); n.lo1[1] = __uint_as_float(b.w); n.lo1[2] = __uint_as_float(c.x);
        n.hi1[0] = __uint_as_float(c.y); n.hi1[1] = __uint_as_float(c.z); n.hi1[2] = __uint_as_float(c.w);
        n.c0 = d.x; n.c1 = d.y;
        n.pad[0] = d.z; n.pad[1] = d.w;
        return n;
    }
};

struct PassInfo;
template <bool ONLY_SPHERE_MEDIA>
struct PhiloxXi { // the uniform a ConstantMedium draws (constant_medium.rs:53) for path `slot` at bounce `bounce`
    static constexpr bool only_sphere_media = ONLY_SPHERE_MEDIA;
    const PassInfo* pi;
    PathId id;
    uint32_t bounce;
    __device__ double operator()(uint32_t medium_index) const;
};
struct ArrayXi { // injected uniforms for rt_trace_first_hit
    const double* row;
    __device__ double operator()(uint32_t medium_index) const { return row[medium_index]; }
};
template <bool ONLY_SPHERE_MEDIA>
struct ArrayXiOf : ArrayXi { static constexpr bool only_sphere_media = ONLY_SPHERE_MEDIA; };
// Caller-supplied rays for k_extend_tree<.., QUERY = true> (rt_trace_first_hit_wavefront): ray i is "path" i, with its own
// t_min / t_max and its own row of medium uniforms; the closest hit goes to hit[i].
struct QueryBatch {
    const rt_ray_d* rays;
    const double* xi_rows; // n * n_media uniforms, or null: media never hit
    RtHitLite* hit;
    uint32_t n;
};

// The lanes of the wave for which `p` holds, as a mask. (__ballot takes an int: the compiler first turns the condition,
// which it already holds as a lane mask, into a 0/1 per lane and compares that with 0 again - two vector instructions
// per ballot, which the loop heads of k_extend_tree feel.)
__device__ inline unsigned long long rt_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Hands every thread with `take` set a distinct position counted from *counter: ONE atomic per workgroup (a single
// counter word takes only ~90 atomics per microsecond chip-wide, so per-wave appends of millions of waves would
// serialise on it). Every thread of the block must call it.
// `which` = 0 or 1 selects one of two counters; `scratch` = 2 * (RT_BLOCK/64 + 1) words of LDS.
__device__ inline uint32_t block_reserve(bool take, uint32_t which, uint32_t* counter0, uint32_t* counter1, uint32_t* scratch) {
    constexpr uint32_t W = RT_BLOCK / 64u;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long m0 = __ballot(take && which == 0u), m1 = __ballot(take && which != 0u);
    const uint32_t rank = (uint32_t)__popcll((which == 0u ? m0 : m1) & below);
    __syncthreads();
    if (lane == 0u) { scratch[wave] = (uint32_t)__popcll(m0); scratch[W + 1u + wave] = (uint32_t)__popcll(m1); }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint32_t* sc = scratch + threadIdx.x * (W + 1u);
        uint32_t total = 0u;
        for (uint32_t w = 0; w < W; ++w) { const uint32_t c = sc[w]; sc[w] = total; total += c; }
        sc[W] = total ? atomicAdd(threadIdx.x == 0u ? counter0 : counter1, total) : 0u;
    }
    __syncthreads();
    const uint32_t* sc = scratch + (which == 0u ? 0u : W + 1u);
    return sc[W] + sc[wave] + rank;
}

// The same for two items per thread (item 0 of every thread comes before item 1 of any in the wave's share). The caller
// must have passed a barrier since the previous call.
__device__ inline void block_reserve2(bool take0, uint32_t which0, bool take1, uint32_t which1, uint32_t* counter0, uint32_t* counter1,
                                      uint32_t* scratch, uint32_t& pos0, uint32_t& pos1) {
    constexpr uint32_t W = RT_BLOCK / 64u;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long a0 = __ballot(take0 && which0 == 0u), a1 = __ballot(take0 && which0 != 0u);
    const unsigned long long b0 = __ballot(take1 && which1 == 0u), b1 = __ballot(take1 && which1 != 0u);
    const uint32_t rank0 = (uint32_t)__popcll((which0 == 0u ? a0 : a1) & below);
    const uint32_t rank1 = (uint32_t)(__popcll(which1 == 0u ? a0 : a1) + __popcll((which1 == 0u ? b0 : b1) & below));
    // (no barrier in front of the writes to `scratch`: the caller has had one since the last call read it)
    if (lane == 0u) { scratch[wave] = (uint32_t)(__popcll(a0) + __popcll(b0)); scratch[W + 1u + wave] = (uint32_t)(__popcll(a1) + __popcll(b1)); }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint32_t* sc = scratch + threadIdx.x * (W + 1u);
        uint32_t total = 0u;
        for (uint32_t w = 0; w < W; ++w) { const uint32_t c = sc[w]; sc[w] = total; total += c; }
        sc[W] = total ? atomicAdd(threadIdx.x == 0u ? counter0 : counter1, total) : 0u;
    }
    __syncthreads();
    const uint32_t* s0 = scratch + (which0 == 0u ? 0u : W + 1u);
    const uint32_t* s1 = scratch + (which1 == 0u ? 0u : W + 1u);
    pos0 = s0[W] + s0[wave] + rank0;
    pos1 = s1[W] + s1[wave] + rank1;
}

// R items per thread, each for one of Q queues (which[r] >= Q: none): one atomic per queue per call, two barriers.
// Queue q starts at qbase + q * qcap and counts in counters[q]. `lds` = Q * RT_BLOCK/64 words. Every thread of the
// block must call it; a barrier must separate two calls.
template <int R, int Q>
__device__ inline void block_append_multi(const uint32_t (&which)[R], const uint32_t (&slot)[R], uint32_t* qbase, size_t qcap,
                                          uint32_t* counters, uint32_t* lds) {
    constexpr uint32_t W = RT_BLOCK / 64u;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rank[r] = 0u;
#pragma unroll
    for (uint32_t q = 0; q < (uint32_t)Q; ++q) {
        uint32_t mine = 0u; // the wave's count for queue q so far (identical in every lane)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool take = which[r] == q;
            const unsigned long long mask = __ballot(take);
            if (take) rank[r] = mine + (uint32_t)__popcll(mask & below);
            mine += (uint32_t)__popcll(mask);
        }
        if (lane == 0u) lds[q * W + wave] = mine;
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)Q) {
        const uint32_t q = threadIdx.x;
        uint32_t total = 0u;
        for (uint32_t w = 0; w < W; ++w) { const uint32_t c = lds[q * W + w]; lds[q * W + w] = total; total += c; }
        const uint32_t base = total ? atomicAdd(counters + q, total) : 0u;
        for (uint32_t w = 0; w < W; ++w) lds[q * W + w] += base;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (which[r] < (uint32_t)Q) qbase[(size_t)which[r] * qcap + lds[which[r] * W + wave] + rank[r]] = slot[r];
}

__device__ inline Ray load_ray(const PathRay* rec, PathId& id) {
    Ray r;
    const double2* q = (const double2*)rec;
    const double2 a = q[0], b = q[1], c = q[2], d = q[3];
    r.o = v3(a.x, a.y, b.x);
    r.d = v3(b.y, c.x, c.y);
    r.tm = d.x;
    const unsigned long long w = (unsigned long long)__double_as_longlong(d.y);
    id.slot = (uint32_t)w; id.pixel = (uint32_t)(w >> 32);
    return r;
}
struct SlotRay { // the world-space ray of a path, re-read from its record on demand
    const PathRay* rec;
    __device__ Ray operator()() const { PathId id; return load_ray(rec, id); }
};
struct HeldRay { // the world-space ray, still held in registers
    const Ray* r;
    __device__ Ray operator()() const { return *r; }
};
struct QueryRay {
    const rt_ray_d* q;
    __device__ Ray operator()() const {
        Ray r;
        r.o = v3(q->ox, q->oy, q->oz); r.d = v3(q->dx, q->dy, q->dz); r.tm = q->time;
        return r;
    }
};
__device__ inline void store_ray(PathRay* rec, const Ray& r, PathId id) {
    double2* q = (double2*)rec;
    q[0] = make_double2(r.o.x, r.o.y);
    q[1] = make_double2(r.o.z, r.d.x);
    q[2] = make_double2(r.d.y, r.d.z);
    q[3] = make_double2(r.tm, __longlong_as_double((long long)(((unsigned long long)id.pixel << 32) | id.slot)));
}
__device__ inline Sampler slot_sampler(const PassInfo& pi, PathId id, uint32_t bounce) {
    return sampler_make(id.pixel, pi.sample_offset + id.slot % pi.S, bounce, pi.seed);
}
template <bool ONLY_SPHERE_MEDIA>
__device__ double PhiloxXi<ONLY_SPHERE_MEDIA>::operator()(uint32_t medium_index) const {
    return sampler_draw1(slot_sampler(*pi, id, bounce), RT_STREAM_MEDIUM0 + medium_index, 0u);
}

// ---- kernels ------------------------------------------------------------------------------------------------
// k_raygen, k_accumulate and k_extend_tree exchange data between the lanes of a wave (ballots over 64 lanes, LDS staging
// indexed by threadIdx.x >> 6 with wavefront-scope fences): they are written for gfx950's 64-lane waves only.
#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE) && __AMDGCN_WAVEFRONT_SIZE != 64
#error "rt_kernels.hip assumes 64-lane wavefronts"
#endif
__global__ void __launch_bounds__(RT_BLOCK) k_raygen(RtCameraDev cam, PassInfo pi, PathBufs pb) {
    __shared__ double2 stage[RT_BLOCK * 4];
    const uint32_t n = pi.n_local * pi.S;
    for (uint32_t slot = blockIdx.x * RT_BLOCK + threadIdx.x; slot < n; slot += gridDim.x * RT_BLOCK) {
        const uint32_t lp = slot / pi.S, s = slot - lp * pi.S;
        const uint32_t pixel = pi.pixmap[lp];
        const uint32_t y = pixel / pi.width, x = pixel - y * pi.width;
        const Sampler smp = sampler_make(pixel, pi.sample_offset + s, 0u, pi.seed);
        const Ray r = rt_primary_ray(cam, x, y, pi.width, pi.height, smp, pi.compat);
        PathId id;
        id.slot = slot; id.pixel = pixel;
        // before the first bounce path number = slot: the 64 records of a wave are 4 KB in a row. Written lane by lane each
        // store instruction would put 16 bytes into every fourth 16-byte piece of them; handed round through LDS, each
        // instruction writes 1 KB without a gap (-12 % kernel time). The same in the shading kernels, whose survivors
        // also get consecutive records, changed nothing there.
        const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
        const uint32_t slot0 = slot - lane;
        if (slot0 + 64u <= n) { // (uniform over the wave's active lanes; no barrier: a wave only reads what it wrote itself)
            double2* mine = stage + wave * 256u;
            mine[lane * 4u + 0u] = make_double2(r.o.x, r.o.y);
            mine[lane * 4u + 1u] = make_double2(r.o.z, r.d.x);
            mine[lane * 4u + 2u] = make_double2(r.d.y, r.d.z);
            mine[lane * 4u + 3u] = make_double2(r.tm, __longlong_as_double((long long)(((unsigned long long)id.pixel << 32) | id.slot)));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (lanes read other lanes' writes: keeps the compiler from reordering them)
            double2* out = (double2*)(pb.ray[0] + slot0);
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) out[k * 64u + lane] = mine[k * 64u + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            store_ray(pb.ray[0] + slot, r, id);
        }
    }
}

// world.hit(r, 0.001, INFINITY, rec) (main.rs:56) for every live path of iteration `it`, in two launches:
//
//   k_extend_tree   closest hit: the lock-step entries when a lane takes a ray, then the tree, persistent waves
//   k_shade_main    misses take the background and end; hits on plain-colour materials are shaded; the rest is queued

// Persistent waves. Once rays are incoherent the number of box tests differs by an order of magnitude from
// lane to lane, and box tests and leaf tests interleave differently in every lane. So:
//  * a lane that has finished its ray does not wait for the slowest one: when `refill_min` lanes of a wave
//    are out of work they take the next rays of the queue. The queue is dealt to the workgroups in pieces
//    (piece c belongs to workgroup c mod gridDim) and the waves of a workgroup draw from its pieces through a
//    counter in LDS: a counter in global memory takes ~90 atomics per microsecond chip-wide, which at one
//    atomic per 64 rays would cap the kernel at 6 G rays/s and dominate small launches;
//  * box tests run for all lanes that hold a node until `leaf_pct` per cent of the lanes that hold a ray wait at a
//    leaf; then those leaves are tested together.
// Which lane traces which ray, and when, has no effect on results.
#ifdef RT_PHASE_STATS
// Debug build only: how the waves of k_extend_tree spend their iterations, by bounce (0, 1, 2, 3+).
__device__ unsigned long long g_phase[4][48];
#define RT_PHASE_ADD(k, v) (ph[k] += (v))
#define RT_PHASE_CLOCK(var) const long long var = clock64()
#define RT_PHASE_SPAN(k, a, b) (ph[k] += (unsigned long long)((b) - (a)))
#else
#define RT_PHASE_ADD(k, v) ((void)0)
#define RT_PHASE_CLOCK(var) ((void)0)
#define RT_PHASE_SPAN(k, a, b) ((void)0)
#endif
// ---- node records as fourteen arrays, fetched by the ray's signs ----
// A box step picks, per axis, the bound the ray enters by and the one it leaves by: twelve selects on the 64-byte record.
// Where every node is in LDS and the count allows, the records are kept instead as fourteen arrays of ARRAYS words each
// (lo0 xyz, hi0 xyz, lo1 xyz, hi1 xyz, child 0, child 1), and a lane holds six LDS addresses - for each axis the array its
// ray enters by and the one it leaves by, fixed when it takes the ray. A step is then six address adds and six reads of two
// words (box 0's and, 6 * ARRAYS words on, box 1's: ds_read2st64_b32 where the distance allows) instead of four adds, four
// 16-byte reads and twelve selects; the arithmetic on the picked bounds is rt_node_boxes_f32's.
typedef const __attribute__((address_space(3))) float* LdsF32;
typedef const __attribute__((address_space(3))) uint32_t* LdsU32;
template <int ARRAYS>
struct SignedNodes {
    uint32_t near_at[3], far_at[3]; // LDS byte addresses of word 0 of the arrays this lane's ray enters / leaves by
    __device__ void take(uint32_t base, const RtSlab& k) { // base: LDS byte address of the first array
        const bool neg[3] = {k.nx, k.ny, k.nz};
#pragma unroll
        for (uint32_t a = 0u; a < 3u; ++a) {
            near_at[a] = base + 4u * (uint32_t)ARRAYS * (neg[a] ? 3u + a : a);
            far_at[a] = base + 4u * (uint32_t)ARRAYS * (neg[a] ? a : 3u + a);
        }
    }
};
// In this layout a node is not named by RT_MAKE_REF(RT_REF_NODE, i) but by RT_ARRAYS_NODE | 4 i (the children are rewritten
// when the arrays are filled, the root when a lane takes a ray): "at a node" is then one unsigned comparison, and with the
// array addresses a lane holds lowered by RT_ARRAYS_NODE the name itself is what is added to them (no shift, no mask -
// three vector instructions less per step). References to anything else keep their form; none has all four type bits set.
#define RT_ARRAYS_NODE 0xF0000000u
__device__ __forceinline__ uint32_t rt_arrays_name(uint32_t ref) { return RT_REF_TYPE(ref) == RT_REF_NODE ? (RT_ARRAYS_NODE | (RT_REF_INDEX(ref) << 2)) : ref; }
template <int ARRAYS>
__device__ __forceinline__ void rt_node_step_signed(const SignedNodes<ARRAYS>& at, uint32_t child_base, RtTrav& s, LdsStack& st, double t_min) {
    uint32_t under = st.prefetch();
    const uint32_t i4 = s.cur; // (RT_ARRAYS_NODE | 4 i against addresses lowered by RT_ARRAYS_NODE: the sum wraps to the word's address)
#ifdef RT_BOUNDS_CHECK
    if (!rt_in_range("nodes (array layout)", (s.cur - RT_ARRAYS_NODE) >> 2, (uint32_t)ARRAYS, s.cur) || ((s.cur & 3u) != 0u)) { s.cur = RT_REF_NONE; return; }
#endif
    constexpr uint32_t kBox1 = 6u * (uint32_t)ARRAYS;
    const LdsF32 nx = (LdsF32)(uintptr_t)(at.near_at[0] + i4), ny = (LdsF32)(uintptr_t)(at.near_at[1] + i4), nz = (LdsF32)(uintptr_t)(at.near_at[2] + i4);
    const LdsF32 fx = (LdsF32)(uintptr_t)(at.far_at[0] + i4), fy = (LdsF32)(uintptr_t)(at.far_at[1] + i4), fz = (LdsF32)(uintptr_t)(at.far_at[2] + i4);
    const LdsU32 kids = (LdsU32)(uintptr_t)(child_base + i4);
    const float n0x = nx[0], n1x = nx[kBox1], n0y = ny[0], n1y = ny[kBox1], n0z = nz[0], n1z = nz[kBox1];
    const float f0x = fx[0], f1x = fx[kBox1], f0y = fy[0], f1y = fy[kBox1], f0z = fz[0], f1z = fz[kBox1];
    const uint32_t c0 = kids[0], c1 = kids[ARRAYS];
    RT_VISIT(0u);
    const float lo = (float)t_min, hi = (float)s.best.t; // widened by an ulp each: the conversions round to nearest
    bool h0, h1;
    float e0, e1;
    rt_node_boxes_f32_picked(n0x, n0y, n0z, f0x, f0y, f0z, n1x, n1y, n1z, f1x, f1y, f1z, s.slab, lo - __builtin_fabsf(lo) * 0x1p-23f,
                             hi + __builtin_fabsf(hi) * 0x1p-23f, h0, h1, e0, e1);
    const bool swap = e1 < e0;
    h1 = h1 && c1 != RT_REF_NONE;
    const bool both = h0 && h1;
    const bool first0 = h0 & (!h1 | !swap); // on with child 0: the nearer of two, or the only one (and the other one is what is kept of two)
    asm volatile("" : "+v"(under)); // (keeps the read of the entry below the top where it was asked for, in front of the box arithmetic: left alone the compiler moves it into the branch that pops, and the wave then waits out the LDS round trip there)
    s.cur = st.node_step<false>(both, h0 || h1, first0 ? c0 : c1, first0 ? c1 : c0, under); // (this layout is only chosen for trees whose stack need has been verified)
}

// EXACT_NODES = false: the scene has no node whose box tests must reproduce the reference's f64 arithmetic.
// SPHERE_MEDIA = true: every ConstantMedium of the scene is bounded by a plain sphere.
// QUERY = true: the rays are a caller's batch (`qb`) instead of the live paths of bounce `it`: same scheduling, same
// node cache, same leaf rounds; per-ray t_min / t_max and injected medium uniforms.
// PLAIN = true: every leaf of the scene's tree is a primitive, a medium, or a primitive named together with a Translate /
// RotateY chain of at most two steps (RT_REF_ILEAF) - no list, no instance that changes the walk's frame, no linked instance
// record (rt_scene::plain_leaves, found at upload). The code for those is left out (25 registers less, -9 % kernel time on
// the Book-2 scene), a leaf's records are asked for all at once (rt_trav_visit<true>), and the lock-step list may be the
// one sphere-bounded medium of the short cut (rt_extend_flat<true>).
// ARRAYS > 0: the node records in LDS as fourteen arrays of that many words (rt_node_step_signed); needs PLAIN, ALL_CACHED,
// no exact nodes, and at most ARRAYS nodes.
// MERGE_MOVING = true (with PLAIN): for scenes with many moving spheres, which then go through the spheres' leaf test
// (rt_trav_visit<true, true>).
template <bool EXACT_NODES, bool SPHERE_MEDIA, bool QUERY, bool ALL_CACHED, bool PLAIN = false, int ARRAYS = 0, bool MERGE_MOVING = false>
__global__ void __launch_bounds__(RT_TREE_BLOCK) k_extend_tree(DScene sc, PassInfo pi, PathBufs pb, QueueSet qs, uint32_t it,
                                                               uint32_t stack_cap, uint32_t refill_min, uint32_t leaf_pct,
                                                               uint32_t n_cached, uint32_t n_exact_lds, QueryBatch qb) {
    extern __shared__ uint4 lds_tree[]; // stack_cap * blockDim.x words of stack, n_cached node records, the queue counter
    const PathRay* rays = pb.ray[it & 1u];
    LiveSet alive_paths;
    alive_paths.front = 0u; alive_paths.back = 0u; alive_paths.cap = 0u;
    if (QUERY) alive_paths.front = qb.n; else alive_paths = live_set(qs.counters + it * RT_COUNTER_STRIDE, qs.bucket_cap);
    const uint32_t n = alive_paths.n();
    if (n == 0u) return;
    typedef typename std::conditional<QUERY, ArrayXiOf<SPHERE_MEDIA>, PhiloxXi<SPHERE_MEDIA>>::type Xi;
    const bool skip_media = QUERY && qb.xi_rows == nullptr;
    double t_min = 0.001; // main.rs:56; a query ray brings its own
    auto xi_of = [&](PathId id, uint32_t path) -> Xi {
        Xi xi;
        if constexpr (QUERY) { xi.row = qb.xi_rows ? qb.xi_rows + (size_t)path * sc.n_media : nullptr; }
        else { xi.pi = &pi; xi.id = id; xi.bounce = it; }
        return xi;
    };
    // queue positions handed out in this workgroup so far: the 16 bytes behind the node records
    static_assert(!MERGE_MOVING || PLAIN, "moving spheres join the spheres' test only in the code for plain leaves");
    static_assert(ARRAYS == 0 || (PLAIN && ALL_CACHED && !EXACT_NODES && ARRAYS % 64 == 0), "the array layout is for small plain trees");
    const size_t node_words = ARRAYS > 0 ? (size_t)14u * (size_t)ARRAYS : (size_t)n_cached * 16u;
    uint32_t& blk_taken = *(uint32_t*)(lds_tree + (size_t)stack_cap * blockDim.x / 4u + node_words / 4u);
    if (threadIdx.x == 0u) blk_taken = 0u;
    LdsStack st;
    st.init((uint32_t*)lds_tree + threadIdx.x, stack_cap, blockDim.x, qs.error_flag);
    CachedNodes<ALL_CACHED> nodes;
    {
        uint4* cache = lds_tree + (size_t)stack_cap * blockDim.x / 4u;
        const uint4* src = (const uint4*)sc.nodes;
        if constexpr (ARRAYS > 0) { // word f of record i -> array f (the two words of padding are left out)
            uint32_t* arrays = (uint32_t*)cache;
            const uint32_t* words = (const uint32_t*)sc.nodes;
            for (uint32_t w = threadIdx.x; w < n_cached * 16u; w += blockDim.x)
                if ((w & 15u) < 14u) arrays[(w & 15u) * (uint32_t)ARRAYS + (w >> 4)] = (w & 15u) >= 12u ? rt_arrays_name(words[w]) : words[w];
        } else {
            for (uint32_t i = threadIdx.x; i < n_cached * 4u; i += blockDim.x) cache[CachedNodes<ALL_CACHED>::place(i >> 2, i & 3u, n_cached)] = src[i];
        }
        nodes.lds = (LdsU4)cache;
        nodes.n_cached = n_cached;
        nodes.global = sc.nodes;
        nodes.exact = sc.exact_boxes;
#ifdef RT_BOUNDS_CHECK
        nodes.n_nodes = sc.counts.nodes;
#endif
        // behind the node records and the queue counter: the f64 bounds of the exact nodes, when there is room for all of them
        double* exact_copy = (double*)(cache + (size_t)n_cached * 4u + 1u);
        for (uint32_t i = threadIdx.x; i < n_exact_lds * 6u; i += blockDim.x) exact_copy[i] = ((const double*)sc.exact_boxes)[i];
        nodes.exact_lds = (const __attribute__((address_space(3))) double*)exact_copy;
        nodes.n_exact_lds = n_exact_lds;
        __syncthreads();
    }
    const uint32_t lane = __lane_id();
    RtTrav s;
    s.cur = RT_REF_NONE;
    s.best.t = RT_INF; s.best.ref = RT_REF_NONE; s.best.aux = 0u;
    // (the array layout: where the arrays start in LDS, and the six this lane's ray reads its bounds from)
    const uint32_t arrays_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)((uint32_t*)lds_tree + (size_t)stack_cap * blockDim.x);
    SignedNodes<(ARRAYS > 0 ? ARRAYS : 64)> picked;
    for (int a = 0; a < 3; ++a) { picked.near_at[a] = arrays_at; picked.far_at[a] = arrays_at; }
    PathId id;
    id.slot = 0u; id.pixel = 0u;
    uint32_t path = 0u;
    bool active = false;    // tracing a ray
    bool exhausted = false; // wave-uniform: the queue has no more rays
    // pieces of 2^shift queue indices, about 8 per workgroup, between 256 and 4096 indices each
    uint32_t shift = 8u;
    while (shift < 12u && ((n / gridDim.x) >> (shift + 3u)) > 0u) ++shift;
    // position v in this workgroup's share of the queue -> queue index (monotonic in v)
    auto index_of = [&](uint32_t v) -> unsigned long long {
        const unsigned long long piece = (unsigned long long)(v >> shift) * gridDim.x + blockIdx.x;
        return (piece << shift) + (v & ((1u << shift) - 1u));
    };
    auto is_node = [](uint32_t ref) -> bool { return ARRAYS > 0 ? ref >= RT_ARRAYS_NODE : RT_REF_TYPE(ref) == RT_REF_NODE; };
#ifdef RT_PHASE_STATS
    unsigned long long ph[48] = {};
#endif
#pragma unroll 1
    for (;;) {
        const unsigned long long idle = rt_ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u && exhausted) break;
        RT_PHASE_ADD(0, 1); RT_PHASE_ADD(7, 64u - n_idle);
        if (n_idle >= refill_min && !exhausted) { // hand idle lanes the next rays
            RT_PHASE_CLOCK(c_round1);
            uint32_t v0 = 0u;
            if (lane == 0u) v0 = atomicAdd(&blk_taken, n_idle);
            v0 = __shfl(v0, 0);
            RT_PHASE_ADD(1, 1); RT_PHASE_ADD(2, n_idle);
            if (!active) {
                const unsigned long long i = index_of(v0 + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull)));
                if (i < n) {
                    path = alive_paths.record((uint32_t)i);
                    s.best.t = RT_INF; s.best.ref = RT_REF_NONE; s.best.aux = 0u;
                    if constexpr (QUERY) {
                        QueryRay q;
                        q.q = qb.rays + path;
                        s.r = q();
                        t_min = q.q->tmin; s.best.t = q.q->tmax;
                    } else {
                        s.r = load_ray(rays + path, id);
                    }
                    // the world entries every ray is tested against, by the lanes that just took a ray, in lock step
                    HeldRay held;
                    held.r = &s.r;
                    rt_extend_flat<PLAIN>(sc, held, t_min, s.best, xi_of(id, path), skip_media);
                    if (PLAIN || sc.world_tree != RT_REF_NONE) s.slab = rt_slab_of<EXACT_NODES>(s.r, sc.node_extent); // (a scene without a tree: none of the PLAIN instantiations is its kernel)
                    if constexpr (ARRAYS > 0) picked.take(arrays_at - RT_ARRAYS_NODE, s.slab);
                    s.cur = ARRAYS > 0 ? rt_arrays_name(sc.world_tree) : sc.world_tree;
                    st.clear();
                    active = true;
                }
            }
            if (index_of(v0 + n_idle) >= n) exhausted = true; // whoever draws next would find nothing
            RT_PHASE_CLOCK(c_round2);
            RT_PHASE_SPAN(41, c_round1, c_round2);
            if (rt_ballot(active) == 0ull && exhausted) break;
        }
        RT_PHASE_CLOCK(c_nodes0);
        // Leaves are tested once this many lanes wait at one: leaf_pct per cent of the lanes that hold a ray at all. (A fixed
        // number - 30 of 64 - is too many just before a refill, when 30 lanes are most of those left, and the few still
        // at a node then step alone; in proportion: -3.4 % kernel time on the Book-2 scene, -7 % on cornell_smoke.)
        const uint32_t n_active = (uint32_t)__popcll(rt_ballot(active));
        const uint32_t leaf_now = max(4u, n_active * leaf_pct / 100u);
        // ---- box tests: every lane that holds a node steps ----
        // (A lane without a ray holds RT_REF_NONE, so "at a node" needs no look at `active`: the comparison's own lane mask
        // is the ballot, and the lanes that wait are counted from it by scalar instructions alone.)
#pragma unroll 1
        for (;;) {
            const bool at_node = is_node(s.cur);
            const unsigned long long m_node = rt_ballot(at_node);
            if (m_node == 0ull) break;
            if (n_active - (uint32_t)__popcll(m_node) >= leaf_now) break;
            RT_PHASE_ADD(3, 1); RT_PHASE_ADD(4, __popcll(m_node));
            if constexpr (ARRAYS > 0) {
                const uint32_t kids_at = arrays_at - RT_ARRAYS_NODE + 4u * 12u * (uint32_t)ARRAYS;
                if (at_node) rt_node_step_signed<ARRAYS>(picked, kids_at, s, st, t_min);
                if (is_node(s.cur)) rt_node_step_signed<ARRAYS>(picked, kids_at, s, st, t_min);
            } else {
                if (at_node) rt_trav_node_step<EXACT_NODES, !PLAIN>(nodes, s, st, t_min); // (a PLAIN instantiation only runs on a tree whose stack need has been verified)
                // a second step before the next look at the wave: the two ballots above cost a sixth of a step (-3.6 % kernel
                // time; a third and fourth step bring nothing more, measured again with the relative leaf threshold)
                if (RT_REF_TYPE(s.cur) == RT_REF_NODE) rt_trav_node_step<EXACT_NODES, !PLAIN>(nodes, s, st, t_min);
            }
        }
        RT_PHASE_CLOCK(c_nodes1);
        RT_PHASE_SPAN(42, c_nodes0, c_nodes1);
        // ---- one non-node reference for every lane that holds one ----
#ifdef RT_PHASE_STATS
        {
            const bool v = active && s.cur != RT_REF_NONE && !is_node(s.cur);
            const unsigned long long any = rt_ballot(v);
            if (any) { ph[5] += 1; ph[6] += __popcll(any); }
            for (uint32_t ty = 0; ty < 16u; ++ty) {
                const unsigned long long m = rt_ballot(v && RT_REF_TYPE(s.cur) == ty);
                if (m) { ph[8 + ty] += 1; ph[24 + ty] += __popcll(m); }
            }
        }
#endif
        if (active && s.cur != RT_REF_NONE && !is_node(s.cur)) {
            if constexpr (QUERY) {
                QueryRay ray_of;
                ray_of.q = qb.rays + path;
                rt_trav_visit<PLAIN, MERGE_MOVING>(sc, s, st, t_min, xi_of(id, path), skip_media, ray_of);
            } else {
                SlotRay ray_of;
                ray_of.rec = rays + path;
                rt_trav_visit<PLAIN, MERGE_MOVING>(sc, s, st, t_min, xi_of(id, path), skip_media, ray_of);
            }
        }
        if (active && s.cur == RT_REF_NONE) { // finished: publish the closest hit
            RtHitLite h;
            h.t = s.best.t; h.ref = s.best.ref; h.aux = s.best.aux;
            (QUERY ? qb.hit : pb.hit)[path] = h;
            active = false;
        }
        RT_PHASE_CLOCK(c_leaf1);
        RT_PHASE_SPAN(43, c_nodes1, c_leaf1);
    }
#ifdef RT_PHASE_STATS
    if (lane == 0u)
        for (int k = 0; k < 48; ++k) if (ph[k]) atomicAdd(&g_phase[it < 3u ? it : 3u][k], ph[k]);
#endif
}

// The closest hits of a scene WITHOUT a tree: everything the world holds is on its lock-step list (a handful of objects, or a
// small reference-order tree laid out as steps). Every ray then runs the same sequence of tests, so there is nothing to
// schedule - no stacks, no node cache, no refill thresholds: one ray per lane, the list from front to back, all lanes of
// a wave in step. Same arguments as k_extend_tree's so that the two are launched alike (`steps_at`, `n_steps`: where the
// list holds a tree's steps, 0 steps if it holds none).
//
// A tree's steps are wave-uniform, and read through the scalar cache each step costs a chain of dependent loads (its two
// words, then the record they name) that nothing overlaps. So lane j of every wave fetches step j once, words and record
// (a box test's f64 bounds; a rect's, box's or sphere's own record), before the first ray, and the walk takes each step's
// operands out of lane j's registers (v_readlane: exec does not matter to it, a lane whose rays have run out still serves).
struct StepOperands { uint32_t ctrl, operand; double rec[6]; };
__device__ __forceinline__ double rt_lane_value(double v, uint32_t lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), (int)lane), __builtin_amdgcn_readlane(__double2loint(v), (int)lane));
}
template <class XiFn>
__device__ inline void rt_steps_from_lanes(const DScene& sc, const StepOperands& mine, uint32_t n_steps, Ray& r, const HeldRay& world_ray,
                                           double t_min, RtBest& best, const XiFn& xi, bool skip_media) {
    RtSlab k;
    k.inv = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    k.nx = k.inv.x < 0.0; k.ny = k.inv.y < 0.0; k.nz = k.inv.z < 0.0;
    uint32_t inside = 1u;
#pragma unroll 1
    for (uint32_t j = 0u; j < n_steps; ++j) {
        const uint32_t ctrl = (uint32_t)__builtin_amdgcn_readlane((int)mine.ctrl, (int)j), operand = (uint32_t)__builtin_amdgcn_readlane((int)mine.operand, (int)j);
        double rec[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) rec[c] = rt_lane_value(mine.rec[c], j);
        const uint32_t d = ctrl & RT_STEP_DEPTH_MASK;
        const bool here = ((inside >> (d - 1u)) & 1u) != 0u;
        if ((ctrl & RT_STEP_GATE) != 0u) {
            bool h = false;
            if (here) {
                const double lo[3] = {rec[0], rec[1], rec[2]}, hi[3] = {rec[3], rec[4], rec[5]};
                double entered;
                h = rt_aabb_hit(lo, hi, r, k, t_min, best.t, entered);
            }
            inside = (inside & ~(1u << d)) | ((h ? 1u : 0u) << d);
            continue;
        }
        const uint32_t leaf = RT_REF_TYPE(operand) == RT_REF_ILEAF ? RT_ILEAF_PRIM(operand) : operand;
        const uint32_t ty = RT_REF_TYPE(leaf);
        if (ty != RT_REF_SPHERE && ty != RT_REF_RECT && ty != RT_REF_BOX) { // no record in the lanes: the list's own way
            if (here) rt_flat_entry(sc, operand, r, world_ray, t_min, best, xi, skip_media);
            continue;
        }
        if (here) {
            const bool moved = RT_REF_TYPE(operand) == RT_REF_ILEAF;
            if (moved) r = rt_apply_chain(sc, RT_ILEAF_CHAIN(operand), r);
            double t;
            if (ty == RT_REF_SPHERE) {
                if (rt_sphere_t(v3(rec[0], rec[1], rec[2]), rec[3], r, t_min, best.t, t)) { best.t = t; best.ref = leaf; best.aux = 0u; }
            } else if (ty == RT_REF_RECT) {
                const uint32_t axis = (uint32_t)__double2loint(rec[5]);
                if (rt_rect_t(axis, rec[0], rec[1], rec[2], rec[3], rec[4], r, t_min, best.t, t)) { best.t = t; best.ref = leaf; best.aux = 0u; }
            } else {
                RtBox b;
                b.p0[0] = rec[0]; b.p0[1] = rec[1]; b.p0[2] = rec[2]; b.p1[0] = rec[3]; b.p1[1] = rec[4]; b.p1[2] = rec[5];
                uint32_t face = 0u;
                if (rt_box_t(b, r, t_min, best.t, t, face)) { best.t = t; best.ref = leaf; best.aux = face; }
            }
            if (moved) r = world_ray();
        }
    }
}
template <bool SPHERE_MEDIA, bool QUERY>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) k_extend_list(DScene sc, PassInfo pi, PathBufs pb, QueueSet qs, uint32_t it, uint32_t steps_at,
                                                          uint32_t n_steps, uint32_t, uint32_t, uint32_t, QueryBatch qb) {
    const PathRay* rays = pb.ray[it & 1u];
    LiveSet alive_paths;
    alive_paths.front = 0u; alive_paths.back = 0u; alive_paths.cap = 0u;
    if (QUERY) alive_paths.front = qb.n; else alive_paths = live_set(qs.counters + it * RT_COUNTER_STRIDE, qs.bucket_cap);
    const uint32_t n = alive_paths.n();
    typedef typename std::conditional<QUERY, ArrayXiOf<SPHERE_MEDIA>, PhiloxXi<SPHERE_MEDIA>>::type Xi;
    const bool skip_media = QUERY && qb.xi_rows == nullptr;
    StepOperands mine;
    mine.ctrl = 0u; mine.operand = 0u;
    for (int c = 0; c < 6; ++c) mine.rec[c] = 0.0;
    {
        const uint32_t j = __lane_id();
        if (j < n_steps) {
            mine.ctrl = sc.list_entries[steps_at + 1u + 2u * j]; mine.operand = sc.list_entries[steps_at + 2u + 2u * j];
            const uint32_t leaf = RT_REF_TYPE(mine.operand) == RT_REF_ILEAF ? RT_ILEAF_PRIM(mine.operand) : mine.operand;
            const double* src = nullptr;
            uint32_t n_doubles = 6u;
            if ((mine.ctrl & RT_STEP_GATE) != 0u) src = (const double*)(sc.exact_boxes + (mine.operand - 1u));
            else if (RT_REF_TYPE(leaf) == RT_REF_SPHERE) { src = (const double*)(sc.spheres + RT_REF_INDEX(leaf)); n_doubles = 4u; }
            else if (RT_REF_TYPE(leaf) == RT_REF_RECT) src = (const double*)(sc.rects + RT_REF_INDEX(leaf)); // (the sixth: axis and padding)
            else if (RT_REF_TYPE(leaf) == RT_REF_BOX) src = (const double*)(sc.boxes + RT_REF_INDEX(leaf));
            if (src != nullptr)
                for (uint32_t c = 0u; c < 6u; ++c) if (c < n_doubles) mine.rec[c] = src[c];
        }
    }
    for (uint32_t base = blockIdx.x * RT_BLOCK + (threadIdx.x & ~63u); base < n; base += gridDim.x * RT_BLOCK) { // (a wave stays together)
        const uint32_t i = base + __lane_id();
        const bool has_ray = i < n;
        const uint32_t path = alive_paths.record(has_ray ? i : 0u);
        double t_min = 0.001; // main.rs:56; a query ray brings its own
        RtBest best;
        best.t = RT_INF; best.ref = RT_REF_NONE; best.aux = 0u;
        PathId id;
        id.slot = 0u; id.pixel = 0u;
        Ray r;
        Xi xi;
        if constexpr (QUERY) {
            QueryRay q;
            q.q = qb.rays + path;
            r = q();
            t_min = q.q->tmin; best.t = q.q->tmax;
            xi.row = qb.xi_rows ? qb.xi_rows + (size_t)path * sc.n_media : nullptr;
        } else {
            r = load_ray(rays + path, id);
            xi.pi = &pi; xi.id = id; xi.bounce = it;
        }
        HeldRay held;
        held.r = &r;
        if (has_ray) {
            if (n_steps == 0u) {
                rt_extend_flat(sc, held, t_min, best, xi, skip_media);
            } else { // the list's entries in front of the steps, the steps, the entries behind them
                Ray moving = r;
                uint32_t li = RT_REF_INDEX(sc.world_flat);
#pragma unroll 1
                for (; li < steps_at; ++li) rt_flat_entry(sc, sc.list_entries[li], moving, held, t_min, best, xi, skip_media);
                rt_steps_from_lanes(sc, mine, n_steps, moving, held, t_min, best, xi, skip_media);
#pragma unroll 1
                for (li = steps_at + 1u + 2u * n_steps; sc.list_entries[li] != RT_REF_NONE; ++li)
                    rt_flat_entry(sc, sc.list_entries[li], moving, held, t_min, best, xi, skip_media);
            }
            RtHitLite h;
            h.t = best.t; h.ref = best.ref; h.aux = best.aux;
            (QUERY ? qb.hit : pb.hit)[path] = h;
        }
    }
}

// emitted + scatter (main.rs:60-70) for every path of iteration `it` that hit a material of type MAT.
// KINDS: the kinds of computed texture the scene holds (rt_tex_value), RT_TEX_KINDS_ALL when the launch does not care.
template <uint32_t BUCKET, uint32_t KINDS = RT_TEX_KINDS_ALL>
__global__ void __launch_bounds__(RT_BLOCK) k_shade(DScene sc, PassInfo pi, PathBufs pb, QueueSet qs, uint32_t it) {
    constexpr uint32_t MAT = rt_bucket_material(BUCKET);
    constexpr bool SOLID = rt_bucket_solid(BUCKET);
    __shared__ uint32_t scratch[2 * (RT_BLOCK / 64 + 1)];
    static_assert(rt_bucket_queue(BUCKET) < RT_QUEUES, "this bucket is shaded in k_shade_main");
    const uint32_t* bucket = qs.bucket + (size_t)rt_bucket_queue(BUCKET) * qs.bucket_cap;
    const uint32_t n = qs.counters[it * RT_COUNTER_STRIDE + 1u + rt_bucket_queue(BUCKET)];
    const PathRay* rays = pb.ray[it & 1u];
    const PathThr* thr = pb.thr[it & 1u];
    PathRay* next_rays = pb.ray[(it + 1u) & 1u];
    PathThr* next_thr = pb.thr[(it + 1u) & 1u];
    uint32_t* next_counter = qs.counters + (it + 1u) * RT_COUNTER_STRIDE;
    const uint32_t cap = (uint32_t)qs.bucket_cap;
    for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < n;
        PathId id;
        id.slot = 0u; id.pixel = 0u;
        bool alive = false;
        Ray scattered;
        PathThr t;
        scattered.o = v3(0.0, 0.0, 0.0); scattered.d = v3(0.0, 0.0, 0.0); scattered.tm = 0.0;
        t.r = 1.0; t.g = 1.0; t.b = 1.0;
        if (active) {
            const uint32_t path = bucket[i];
            const Ray r = load_ray(rays + path, id);
            const RtHitLite h = pb.hit[path];
            if (it != 0u) t = thr[path];
            // (the material is asked for between the primitive's table entry and the instance records above it, as in k_shade_main)
            const RtHitMeta g = rt_hit_meta(sc, h);
            const RtMaterial m = sc.materials[g.material];
            const RtHitLevels lv = rt_hit_levels(sc, g.parent);
            // (u, v) are read by an image only - where the scene has one, possibly below a checker
            const bool uv = (SOLID || MAT == RT_MAT_METAL || MAT == RT_MAT_DIELECTRIC || BUCKET == RT_BUCKET_LAMBERTIAN_NOISE || !(KINDS & (1u << RT_TEX_IMAGE)))
                                ? false : rt_material_reads_uv(sc, m);
            const HitRec rec = rt_reconstruct(sc, r, h, uv, g, lv);
            const Sampler smp = slot_sampler(pi, id, it);
            const ShadeOut o = rt_shade_as<MAT, SOLID, BUCKET == RT_BUCKET_LAMBERTIAN_NOISE, KINDS>(sc, m, r, rec, smp);
            // a path that used its last bounce adds nothing more: ray_color(depth <= 0) is black (main.rs:52-54)
            if (o.did_scatter && it + 1u < pi.max_depth) {
                alive = true;
                scattered = o.scattered;
                t.r *= o.attenuation.x; t.g *= o.attenuation.y; t.b *= o.attenuation.z;
            } else { // the path ends here: what it adds to its pixel (only DiffuseLight's emitted() is not black) - written once per path
                double* rad = pb.rad + 3u * (size_t)id.slot;
                rad[0] = t.r * o.emitted.x;
                rad[1] = t.g * o.emitted.y;
                rad[2] = t.b * o.emitted.z;
            }
        }
        const uint32_t end = MAT == RT_MAT_ISOTROPIC ? 1u : 0u; // scattered inside a medium: back of the arrays
        uint32_t j = block_reserve(alive, end, next_counter, next_counter + RT_COUNTER_BACK, scratch);
        if (alive) {
            if (end) j = cap - 1u - j;
            store_ray(next_rays + j, scattered, id);
            next_thr[j] = t;
        }
    }
}

// The body of ray_color after world.hit (main.rs:56-70) for every path of iteration `it`, straight from the hit records:
// a miss takes the background and ends; a hit on a material with a plain-colour texture (or none) is shaded here and
// now - these are nearly all hits, their code is short, and Lambertian and Isotropic share the same rejection loop, so
// mixing them in a wave costs little; a hit on a computed texture (Perlin noise, image, checker) is queued for the
// k_shade kernel of its bucket, which would otherwise drag its registers and code through every wave.
// A workgroup takes 2 x RT_BLOCK paths at a time, every thread one path of each half, and reserves room in the next
// bounce's arrays for both halves at once: at the first bounces the kernel ran at 82 reservations per microsecond on each
// of the two counters, which is as many as one word takes. The first half's survivor waits in LDS meanwhile.
__global__ void __launch_bounds__(RT_BLOCK, 5) k_shade_main(DScene sc, PassInfo pi, PathBufs pb, QueueSet qs, uint32_t it) {
    __shared__ uint32_t scratch[2 * (RT_BLOCK / 64 + 1)];
    __shared__ uint32_t lds[RT_QUEUES * (RT_BLOCK / 64)];
    __shared__ double2 held[5 * RT_BLOCK]; // piece k of thread x's first survivor at [k * RT_BLOCK + x]: its ray record, then (t.r, t.g)
    __shared__ double held_b[RT_BLOCK];    // ... and t.b
    uint32_t* ctr = qs.counters + it * RT_COUNTER_STRIDE;
    const LiveSet alive_paths = live_set(ctr, qs.bucket_cap);
    const uint32_t n = alive_paths.n();
    const uint32_t cap = alive_paths.cap;
    const PathRay* rays = pb.ray[it & 1u];
    const PathThr* thr = pb.thr[it & 1u];
    PathRay* next_rays = pb.ray[(it + 1u) & 1u];
    PathThr* next_thr = pb.thr[(it + 1u) & 1u];
    uint32_t* next_counter = qs.counters + (it + 1u) * RT_COUNTER_STRIDE;
    for (uint32_t base = blockIdx.x * (2u * RT_BLOCK); base < n; base += gridDim.x * (2u * RT_BLOCK)) {
        PathId id;
        id.slot = 0u; id.pixel = 0u;
        uint32_t end = 0u;
        bool alive = false;
        uint32_t first = 0xFFu; // what became of the first half's path: its queue (0xFF: none) | alive << 8 | end << 9 - one register across the second half
        uint32_t later[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        uint32_t path[2] = {0u, 0u};
        Ray scattered;
        PathThr t;
        scattered.o = v3(0.0, 0.0, 0.0); scattered.d = v3(0.0, 0.0, 0.0); scattered.tm = 0.0;
        t.r = 1.0; t.g = 1.0; t.b = 1.0;
#pragma unroll 1
        for (uint32_t half = 0u; half < 2u; ++half) {
            if (half == 1u) { // the first half's result steps aside
                first = (later[1] & 0xFFu) | (alive ? 0x100u : 0u) | (end << 9);
                if (alive) {
                    double2* to = held + threadIdx.x;
                    to[0] = make_double2(scattered.o.x, scattered.o.y);
                    to[RT_BLOCK] = make_double2(scattered.o.z, scattered.d.x);
                    to[2 * RT_BLOCK] = make_double2(scattered.d.y, scattered.d.z);
                    to[3 * RT_BLOCK] = make_double2(scattered.tm, __longlong_as_double((long long)(((unsigned long long)id.pixel << 32) | id.slot)));
                    to[4 * RT_BLOCK] = make_double2(t.r, t.g);
                    held_b[threadIdx.x] = t.b;
                }
            }
            // (a wave's two halves are 128 paths in a row, so that its survivors leave in the order the paths came in:
            // neighbours in the arrays are samples of one pixel, and the traversal likes them together)
            const uint32_t number = base + ((threadIdx.x >> 6) * 2u + half) * 64u + (threadIdx.x & 63u);
            const uint32_t i = alive_paths.record(number);
            alive = false; end = 0u;
            later[1] = 0xFFFFFFFFu; path[1] = i;
            t.r = 1.0; t.g = 1.0; t.b = 1.0;
            // (nothing of the first half's survivor is needed in registers any more)
            scattered.o = v3(0.0, 0.0, 0.0); scattered.d = v3(0.0, 0.0, 0.0); scattered.tm = 0.0;
            id.slot = 0u; id.pixel = 0u;
            if (number < n) {
                const RtHitLite h = pb.hit[i];
                const Ray r = load_ray(rays + i, id);
                if (it != 0u) t = thr[i];
                end = RT_REF_TYPE(h.ref) == RT_REF_MEDIUM ? 1u : 0u; // scattered inside a medium: back of the arrays
                double* rad = pb.rad + 3u * (size_t)id.slot;
                if (h.ref == RT_REF_NONE) { // main.rs:56-58: return background; the path ends here
                    const V3 bg = rt_background(pi.bg, pi.bg_mode, r);
                    rad[0] = t.r * bg.x;
                    rad[1] = t.g * bg.y;
                    rad[2] = t.b * bg.z;
                } else {
                    // the material is asked for between the primitive's table entry and the instance records above it: it
                    // and the first of those depend on the entry alone and travel together
                    const RtHitMeta g = rt_hit_meta(sc, h);
                    const RtMaterial m = sc.materials[g.material];
                    const RtHitLevels lv = rt_hit_levels(sc, g.parent);
                    const bool textured = m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_DIFFUSE_LIGHT || m.type == RT_MAT_ISOTROPIC;
                    const uint32_t bucket = rt_bucket_of(m.type, textured ? m.tex_type : (uint32_t)RT_TEX_SOLID);
                    if (!rt_bucket_solid(bucket) && bucket != RT_BUCKET_METAL && bucket != RT_BUCKET_DIELECTRIC) {
                        later[1] = rt_bucket_queue(bucket);
                    } else {
                        const HitRec rec = rt_reconstruct(sc, r, h, false, g, lv);
                        const Sampler smp = slot_sampler(pi, id, it);
                        const ShadeOut o = rt_shade_as<RT_MAT_COUNT, true>(sc, m, r, rec, smp);
                        if (m.type == RT_MAT_DIELECTRIC) end = 1u; // glass keeps company with the media (one of them fills a glass sphere)
                        // a path that used its last bounce adds nothing more: ray_color(depth <= 0) is black (main.rs:52-54)
                        if (o.did_scatter && it + 1u < pi.max_depth) {
                            alive = true;
                            scattered = o.scattered;
                            t.r *= o.attenuation.x; t.g *= o.attenuation.y; t.b *= o.attenuation.z;
                        } else { // the path ends here (only DiffuseLight's emitted() is not black)
                            rad[0] = t.r * o.emitted.x;
                            rad[1] = t.g * o.emitted.y;
                            rad[2] = t.b * o.emitted.z;
                        }
                    }
                }
            }
        }
        const bool alive0 = (first & 0x100u) != 0u;
        const uint32_t end0 = (first >> 9) & 1u;
        later[0] = (first & 0xFFu) == 0xFFu ? 0xFFFFFFFFu : (first & 0xFFu);
        path[0] = alive_paths.record(base + (threadIdx.x >> 6) * 128u + (threadIdx.x & 63u));
        if (__syncthreads_or(later[0] != 0xFFFFFFFFu || later[1] != 0xFFFFFFFFu)) {
            block_append_multi<2, RT_QUEUES>(later, path, qs.bucket, qs.bucket_cap, ctr + 1u, lds);
            __syncthreads();
        }
        uint32_t j0, j1;
        block_reserve2(alive0, end0, alive, end, next_counter, next_counter + RT_COUNTER_BACK, scratch, j0, j1);
        if (alive) {
            if (end) j1 = cap - 1u - j1;
            store_ray(next_rays + j1, scattered, id);
            next_thr[j1] = t;
        }
        if (alive0) { // (each thread reads back what it wrote itself)
            if (end0) j0 = cap - 1u - j0;
            const double2* from = held + threadIdx.x;
            double2* q = (double2*)(next_rays + j0);
            q[0] = from[0]; q[1] = from[RT_BLOCK]; q[2] = from[2 * RT_BLOCK]; q[3] = from[3 * RT_BLOCK];
            const double2 rg = from[4 * RT_BLOCK];
            PathThr t0;
            t0.r = rg.x; t0.g = rg.y; t0.b = held_b[threadIdx.x];
            next_thr[j0] = t0;
        }
    }
}

// pixel_color += ray_color(...) in sample order (main.rs:165), continuing from earlier passes.
// One wave per pixel: the pixel's S sample radiances are 24 S bytes in a row; the wave reads them 64 samples at a time
// (coalesced), lays them out in LDS, and three lanes - one per colour channel - add them up in sample order, while the
// next 64 samples are already on their way. (A lane per pixel, each walking its own 24 S bytes, kept 64 cache lines in
// flight per wave and re-fetched most of them: 2.6 TB/s.)
__global__ void __launch_bounds__(RT_BLOCK) k_accumulate(PassInfo pi, PathBufs pb, double* sum) {
    __shared__ double stg[RT_BLOCK / 64][192];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t total = 3u * pi.S; // doubles per pixel
    const uint32_t n_chunks = (total + 191u) / 192u;
    for (uint32_t lp = blockIdx.x * (RT_BLOCK / 64u) + wave; lp < pi.n_local; lp += gridDim.x * (RT_BLOCK / 64u)) {
        const double* rad = pb.rad + (size_t)lp * total;
        const size_t o = (size_t)pi.pixmap[lp] * 3u;
        double acc = lane < 3u ? sum[o + lane] : 0.0;
        double v[3];
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) { const uint32_t i = lane + 64u * k; v[k] = i < total ? rad[i] : 0.0; }
        for (uint32_t c = 0; c < n_chunks; ++c) {
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) stg[wave][lane + 64u * k] = v[k];
            // lanes read what other lanes of the wave wrote: without the fences the compiler, which reasons lane by lane,
            // may move the reads above the writes (it did)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (c + 1u < n_chunks) {
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k) { const uint32_t i = (c + 1u) * 192u + lane + 64u * k; v[k] = i < total ? rad[i] : 0.0; }
            }
            const uint32_t here = pi.S - c * 64u < 64u ? pi.S - c * 64u : 64u; // samples in this chunk
            if (lane < 3u) {
#pragma unroll 8
                for (uint32_t smp = 0; smp < here; ++smp) acc += stg[wave][smp * 3u + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (lane < 3u) sum[o + lane] = acc;
    }
}

__global__ void __launch_bounds__(RT_BLOCK) k_set_u32(uint32_t* p, uint32_t v) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *p = v;
}

// The gather of main.rs:244-253 (each job's rows copied into the image) for tile sharding: the pixels of the tiles that
// belong to `rank` (tile k of the image belongs to rank k % world, as ensure_pixmap deals them) are taken from `src`, a
// whole-image buffer that rank rendered into; every other pixel of `dst` is left alone.
__global__ void __launch_bounds__(RT_BLOCK) k_merge_tiles(double* dst, const double* src, uint32_t width, uint32_t height, uint32_t tile,
                                                          uint32_t rank, uint32_t world) {
    const uint32_t n = width * height, tx = (width + tile - 1u) / tile;
    for (uint32_t p = blockIdx.x * RT_BLOCK + threadIdx.x; p < n; p += gridDim.x * RT_BLOCK) {
        const uint32_t y = p / width, x = p - y * width;
        const uint32_t k = (y / tile) * tx + x / tile;
        if (k % world == rank) {
            dst[3u * (size_t)p] = src[3u * (size_t)p];
            dst[3u * (size_t)p + 1u] = src[3u * (size_t)p + 1u];
            dst[3u * (size_t)p + 2u] = src[3u * (size_t)p + 2u];
        }
    }
}

// What world.hit leaves in rec (hittable.rs:9-18) for closest hit `h` of ray `r`, as the record the C ABI reports.
__device__ inline rt_hit full_hit_record(const DScene& sc, const Ray& r, double t_max, const RtHitLite& h) {
    rt_hit o;
    memset(&o, 0, sizeof o);
    if (h.ref == RT_REF_NONE) {
        o.prim_id = RT_MISS;
        o.t = t_max;
    } else {
        const HitRec rec = rt_reconstruct(sc, r, h, true);
        o.t = rec.t;
        o.prim_id = rec.src_id;
        o.flags = RT_HIT_FLAG_HIT | (rec.front_face ? RT_HIT_FLAG_FRONT : 0u) | (h.aux << 4);
        o.px = rec.p.x; o.py = rec.p.y; o.pz = rec.p.z;
        o.nx = rec.normal.x; o.ny = rec.normal.y; o.nz = rec.normal.z;
        o.u = rec.u; o.v = rec.v;
        o.material_id = rec.material;
    }
    return o;
}

// Hittable::hit of the world list for caller-supplied rays, with the full HitRecord rebuilt.
__global__ void __launch_bounds__(RT_BLOCK) k_first_hit(DScene sc, const rt_ray_d* rays, uint32_t n, const double* xi_rows,
                                                        rt_hit* out, uint32_t stack_cap, uint32_t* error_flag) {
    extern __shared__ uint32_t lds_stack[];
    LdsStack st;
    st.init(lds_stack + threadIdx.x, stack_cap, RT_BLOCK, error_flag);
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        const rt_ray_d q = rays[i];
        QueryRay ray_of;
        ray_of.q = rays + i;
        const Ray r = ray_of();
        ArrayXi xi;
        xi.row = xi_rows ? xi_rows + (size_t)i * sc.n_media : nullptr;
        st.clear();
        const RtHitLite h = rt_extend(sc, ray_of, q.tmin, q.tmax, st, xi, xi_rows == nullptr);
        out[i] = full_hit_record(sc, r, q.tmax, h);
    }
}

// The second half of rt_trace_first_hit_wavefront: HitRecords from the closest hits k_extend_tree<.., QUERY> found.
__global__ void __launch_bounds__(RT_BLOCK) k_query_finish(DScene sc, QueryBatch qb, rt_hit* out) {
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < qb.n; i += gridDim.x * RT_BLOCK) {
        QueryRay ray_of;
        ray_of.q = qb.rays + i;
        out[i] = full_hit_record(sc, ray_of(), ray_of.q->tmax, qb.hit[i]);
    }
}

__global__ void __launch_bounds__(RT_BLOCK) k_camera_rays(RtCameraDev cam, uint32_t width, uint32_t height, uint32_t sample,
                                                          uint64_t seed, uint32_t compat, rt_ray_d* out) {
    const uint32_t n = width * height;
    for (uint32_t pixel = blockIdx.x * RT_BLOCK + threadIdx.x; pixel < n; pixel += gridDim.x * RT_BLOCK) {
        const uint32_t y = pixel / width, x = pixel - y * width;
        const Sampler smp = sampler_make(pixel, sample, 0u, seed);
        const Ray r = rt_primary_ray(cam, x, y, width, height, smp, compat);
        rt_ray_d o;
        o.ox = r.o.x; o.oy = r.o.y; o.oz = r.o.z; o.dx = r.d.x; o.dy = r.d.y; o.dz = r.d.z;
        o.time = r.tm; o.tmin = 0.001; o.tmax = RT_INF;
        out[pixel] = o;
    }
}

__global__ void k_vec3_kat(double one, uint32_t* failed) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *failed = rt_vec3_kat(one);
}

// ---- host side ----------------------------------------------------------------------------------------------
// rt_trace_first_hit_noalloc: everything rt_trace_first_hit gets from hipMalloc - the staging copies of the caller's rays,
// uniforms and hit records, and the word the kernel raises on a stack overflow - lives in this array instead. Every GPU
// the library runs on holds its own copy, at its own address, until the process ends; in host code the array's name is
// only a handle for hipGetSymbolAddress, never a device address.
#define RT_NOALLOC_BYTES (16u << 20)
#define RT_NOALLOC_HEAD 256u /* [0, 4): the overflow flag; staging starts behind it */
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];

struct rt_scene {
    void* d_blob = nullptr;
    size_t nbytes = 0;
    RtBlobHeader hdr;
    DScene view;
    bool has_flat = true, has_tree = true; // whether the world holds plain primitives / BVHs at all
    uint32_t bucket_mask = 0;              // shading buckets some material of the scene falls into
    bool sphere_media = true;              // every medium is bounded by a plain sphere (no instance, no other primitive)
    bool tree_built = false;               // rt_scene_build_tree_device has replaced the stand-in tree
    uint32_t tex_kinds = 0u;               // bit t: the texture table holds a texture of type t (RtTextureType)
    bool stack_verified = false;           // ... and a walk of that tree at upload found no node twice on a path and no path that needs more stack entries than hdr.stack_need
    bool plain_leaves = false;             // below the world's tree nothing changes frames or opens a list: every leaf is a primitive, a medium, or a primitive named with a chain of at most two steps
    uint32_t steps_at = 0u, n_steps = 0u;  // a small tree's steps on the lock-step list: index of their RT_REF_STEPS entry, their number (0: none)
    // [0]: hdr.world_flat, [1]: hdr.world_flat_any_time - the list's one medium and its sphere when that is all the list holds
    struct OneMedium { uint32_t ref = 0u; RtMedium medium; RtSphere sphere; } one_medium[2];
    void choose_flat(uint32_t which) { // the list the view's world_flat names, as DScene::flat_one_medium wants it
        view.flat_one_medium = one_medium[which].ref;
        view.flat_medium = one_medium[which].medium;
        view.flat_sphere = one_medium[which].sphere;
    }
};

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    int n_cu = 256;
    size_t lds_per_block = 65536; // most LDS one workgroup may use
    // path pool
    size_t cap_paths = 0;
    uint8_t* f64pool = nullptr; // 2 x cap_paths PathRay, 2 x cap_paths PathThr, 3 x cap_paths doubles of radiance
    RtHitLite* hit = nullptr;
    uint32_t* u32pool = nullptr; // RT_QUEUES queues, cap_paths each
    uint32_t* counters = nullptr;
    size_t cap_counters = 0;
    uint32_t* error_flag = nullptr;
    unsigned char* noalloc = nullptr; // this device's copy of `scratch` (hipGetSymbolAddress at rt_init)
    // pixel map cache
    uint32_t* pixmap = nullptr;
    uint32_t pm_w = 0, pm_h = 0, pm_tile = 0, pm_rank = 0, pm_world = 0, pm_n = 0;
    size_t pm_cap = 0;
    std::vector<hipEvent_t> events;
    // k_extend thresholds: refill once this many lanes of a wave are out of work, test leaves once this per cent of its ray-holding lanes wait at one
    uint32_t tune_refill = 0, tune_leaf = 0; // 0 = the defaults of extend_launch_of; RT_TUNE_REFILL / RT_TUNE_LEAF override
};

namespace {

int ctx_fail(rt_context* c, int code, const char* what, hipError_t e = hipSuccess) {
    if (c) {
        c->error = what;
        if (e != hipSuccess) { c->error += ": "; c->error += hipGetErrorString(e); }
    }
    return code;
}
#define RT_HIP(call, what)                                                   \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) {                                             \
            (void)hipGetLastError(); /* the error is reported through the return code, not left behind for the next HIP user of the process */ \
            return ctx_fail(ctx, e__ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_NO_DEVICE, what, e__); \
        }                                                                    \
    } while (0)

int parse_header(rt_context* ctx, const RtBlobHeader& h, size_t nbytes) {
    if (h.magic != RT_BLOB_MAGIC || h.version != RT_BLOB_VERSION) return ctx_fail(ctx, RT_ERR_INVALID, "blob: bad magic or version");
    if (h.total_bytes != nbytes) return ctx_fail(ctx, RT_ERR_INVALID, "blob: size mismatch");
    if (h.stack_need > RT_STACK_CAP) return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "blob: traversal stack need exceeds capacity");
    const RtSection* secs[] = {&h.nodes, &h.spheres, &h.sphere_meta, &h.mspheres, &h.msphere_meta, &h.rects, &h.rect_meta,
                               &h.boxes, &h.box_meta, &h.media, &h.insts, &h.list_entries, &h.materials, &h.textures,
                               &h.perlin_vec, &h.perlin_perm, &h.texels, &h.exact_boxes, &h.chains};
    const size_t elem[] = {sizeof(RtNode), sizeof(RtSphere), sizeof(RtMeta), sizeof(RtMSphere), sizeof(RtMeta), sizeof(RtRect),
                           sizeof(RtMeta), sizeof(RtBox), sizeof(RtMeta), sizeof(RtMedium), sizeof(RtInst), 4, sizeof(RtMaterial),
                           sizeof(RtTexture), 8, 4, 1, sizeof(RtExactBox), sizeof(RtChain)};
    for (size_t i = 0; i < sizeof(secs) / sizeof(secs[0]); ++i)
        if (secs[i]->offset > nbytes || secs[i]->count * elem[i] > nbytes - secs[i]->offset)
            return ctx_fail(ctx, RT_ERR_INVALID, "blob: section out of range");
    return RT_OK;
}

void make_view(rt_scene* s) {
    const uint8_t* b = (const uint8_t*)s->d_blob;
    const RtBlobHeader& h = s->hdr;
    DScene& v = s->view;
    v.nodes = (const RtNode*)(b + h.nodes.offset);
    v.spheres = (const RtSphere*)(b + h.spheres.offset);
    v.sphere_meta = (const RtMeta*)(b + h.sphere_meta.offset);
    v.mspheres = (const RtMSphere*)(b + h.mspheres.offset);
    v.msphere_meta = (const RtMeta*)(b + h.msphere_meta.offset);
    v.rects = (const RtRect*)(b + h.rects.offset);
    v.rect_meta = (const RtMeta*)(b + h.rect_meta.offset);
    v.boxes = (const RtBox*)(b + h.boxes.offset);
    v.box_meta = (const RtMeta*)(b + h.box_meta.offset);
    v.media = (const RtMedium*)(b + h.media.offset);
    v.insts = (const RtInst*)(b + h.insts.offset);
    v.list_entries = (const uint32_t*)(b + h.list_entries.offset);
    v.materials = (const RtMaterial*)(b + h.materials.offset);
    v.textures = (const RtTexture*)(b + h.textures.offset);
    v.perlin_vec = (const double*)(b + h.perlin_vec.offset);
    v.perlin_perm = (const int32_t*)(b + h.perlin_perm.offset);
    v.texels = (const uint8_t*)(b + h.texels.offset);
    v.exact_boxes = (const RtExactBox*)(b + h.exact_boxes.offset);
    v.chains = (const RtChain*)(b + h.chains.offset);
    v.world = RT_REF_NONE;
    v.world_flat = h.world_flat_any_time; // rays of any time (rt_trace_first_hit); a render narrows it, see rt_render_accumulate
    v.world_tree = h.world_tree;
    v.n_media = h.n_media;
    v.compat = 0;
    v.node_extent = h.node_extent;
    v.flat_one_medium = 0u; v.flat_pad = 0u;
#ifdef RT_BOUNDS_CHECK
    v.counts = RtCounts{(uint32_t)h.nodes.count, (uint32_t)h.spheres.count, (uint32_t)h.mspheres.count, (uint32_t)h.rects.count, (uint32_t)h.boxes.count,
                        (uint32_t)h.media.count, (uint32_t)h.insts.count, (uint32_t)h.list_entries.count, (uint32_t)h.materials.count,
                        (uint32_t)h.textures.count, (uint32_t)h.exact_boxes.count, (uint32_t)h.chains.count};
#endif
    memset(&v.flat_medium, 0, sizeof v.flat_medium);
    memset(&v.flat_sphere, 0, sizeof v.flat_sphere);
}

int ensure_paths(rt_context* ctx, size_t paths, uint32_t max_depth) {
    if (paths > ctx->cap_paths) {
        if (ctx->f64pool) hipFree(ctx->f64pool);
        if (ctx->hit) hipFree(ctx->hit);
        if (ctx->u32pool) hipFree(ctx->u32pool);
        ctx->f64pool = nullptr; ctx->hit = nullptr; ctx->u32pool = nullptr; ctx->cap_paths = 0;
        hipError_t e = hipMalloc((void**)&ctx->f64pool, paths * (2 * sizeof(PathRay) + 2 * sizeof(PathThr) + 3 * sizeof(double)));
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->hit, paths * sizeof(RtHitLite));
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->u32pool, paths * RT_QUEUES * sizeof(uint32_t));
        if (e != hipSuccess) { // all or nothing: a caller that retries with a smaller pass must find the memory free
            (void)hipGetLastError();
            if (ctx->f64pool) hipFree(ctx->f64pool);
            if (ctx->hit) hipFree(ctx->hit);
            if (ctx->u32pool) hipFree(ctx->u32pool);
            ctx->f64pool = nullptr; ctx->hit = nullptr; ctx->u32pool = nullptr;
            return ctx_fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_NO_DEVICE, "hipMalloc path state", e);
        }
        ctx->cap_paths = paths;
    }
    const size_t need = (size_t)(max_depth + 2) * RT_COUNTER_STRIDE;
    if (need > ctx->cap_counters) {
        if (ctx->counters) hipFree(ctx->counters);
        ctx->counters = nullptr; ctx->cap_counters = 0;
        RT_HIP(hipMalloc((void**)&ctx->counters, need * sizeof(uint32_t)), "hipMalloc counters");
        ctx->cap_counters = need;
    }
    return RT_OK;
}

// Pixels of the tiles this rank owns (tile k belongs to rank k % world), tile by tile, row-major inside a tile.
int ensure_pixmap(rt_context* ctx, const rt_params* p) {
    const uint32_t tile = p->tile ? p->tile : 32u;
    const uint32_t world = p->tile_world ? p->tile_world : 1u;
    const uint32_t rank = p->tile_rank;
    if (rank >= world) return ctx_fail(ctx, RT_ERR_INVALID, "tile_rank >= tile_world");
    if (ctx->pixmap && ctx->pm_w == p->width && ctx->pm_h == p->height && ctx->pm_tile == tile && ctx->pm_rank == rank &&
        ctx->pm_world == world)
        return RT_OK;
    std::vector<uint32_t> map;
    const uint32_t tx = (p->width + tile - 1) / tile, ty = (p->height + tile - 1) / tile;
    for (uint32_t k = rank; k < tx * ty; k += world) {
        const uint32_t x0 = (k % tx) * tile, y0 = (k / tx) * tile;
        for (uint32_t y = y0; y < y0 + tile && y < p->height; ++y)
            for (uint32_t x = x0; x < x0 + tile && x < p->width; ++x) map.push_back(y * p->width + x);
    }
    if (map.size() > ctx->pm_cap) {
        if (ctx->pixmap) hipFree(ctx->pixmap);
        ctx->pixmap = nullptr; ctx->pm_cap = 0;
        RT_HIP(hipMalloc((void**)&ctx->pixmap, std::max<size_t>(map.size(), 1) * sizeof(uint32_t)), "hipMalloc pixel map");
        ctx->pm_cap = map.size();
    }
    if (!map.empty())
        RT_HIP(hipMemcpy(ctx->pixmap, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "upload pixel map");
    ctx->pm_w = p->width; ctx->pm_h = p->height; ctx->pm_tile = tile; ctx->pm_rank = rank; ctx->pm_world = world;
    ctx->pm_n = (uint32_t)map.size();
    return RT_OK;
}

uint32_t grid_for(const rt_context* ctx, size_t items, uint32_t blocks_per_cu) {
    const size_t want = (items + RT_BLOCK - 1) / RT_BLOCK;
    const size_t cap = (size_t)ctx->n_cu * blocks_per_cu;
    return (uint32_t)std::max<size_t>(1, std::min(want, cap));
}

struct EventTimer { // brackets launches with events when profiling; sums are read after the pass has drained
    rt_context* ctx;
    bool on;
    size_t used = 0;
    struct Span { size_t a, b; int kind; };
    std::vector<Span> spans;
    hipEvent_t get() {
        if (used == ctx->events.size()) {
            hipEvent_t e;
            hipEventCreate(&e);
            ctx->events.push_back(e);
        }
        return ctx->events[used++];
    }
    size_t begin() {
        if (!on) return 0;
        hipEventRecord(get(), ctx->stream);
        return used - 1;
    }
    void end(size_t a, int kind) {
        if (!on) return;
        hipEventRecord(get(), ctx->stream);
        spans.push_back(Span{a, used - 1, kind});
    }
};

// How k_extend_tree is launched on a scene: which instantiation, and what its one workgroup per CU keeps in LDS.
typedef void (*ExtendKernel)(DScene, PassInfo, PathBufs, QueueSet, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, QueryBatch);
struct ExtendLaunch {
    ExtendKernel kernel;
    size_t lds;
    uint32_t stack_cap, n_cached, n_exact_lds, refill_min, leaf_pct;
    bool list_only; // the scene has no tree: k_extend_list, launched over the rays instead of one persistent workgroup per CU
    uint32_t block;
    uint32_t stack_cap_arg = 0u, refill_arg = 0u; // the kernel's fifth and sixth arguments: stack_cap and refill_min for k_extend_tree
    uint32_t grid(const rt_context* ctx, size_t rays) const {
        const size_t want = (rays + block - 1) / block;
        return (uint32_t)std::max<size_t>(1, std::min<size_t>(want, (size_t)ctx->n_cu * (list_only ? 32u : 1u)));
    }
};
// `force`: bit 0 = the instantiation with exact f64 box tests even if the scene has no node that needs them, bit 1 = the
// one with the general medium-boundary code even if every medium is bounded by a sphere (both are valid for any scene),
// bit 2 = the tree kernel even for a scene without a tree, bit 3 = not the instantiation for trees with plain leaves, bit 4 = not the
// array layout of the node records in LDS, bit 5 = moving spheres keep a leaf test of their own however many there are.
int extend_launch_of(rt_context* ctx, const rt_scene* sc, bool query, uint32_t force, ExtendLaunch* out) {
    const uint32_t stack_cap = sc->hdr.stack_need;
    // Persistent waves, one workgroup per CU: its LDS holds the traversal stacks and, in what is left, the first
    // records of the node array (the compiler numbers nodes breadth-first, so these are the ones nearest the root).
    const size_t lds_stack_bytes = (size_t)stack_cap * RT_TREE_BLOCK * sizeof(uint32_t);
    if (lds_stack_bytes + 16 > ctx->lds_per_block) return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "render: traversal stacks exceed the LDS of a workgroup");
    const uint32_t n_cached = (uint32_t)std::min<size_t>(sc->hdr.nodes.count, (ctx->lds_per_block - lds_stack_bytes - 16) / sizeof(RtNode));
    size_t lds = lds_stack_bytes + (size_t)n_cached * sizeof(RtNode) + 16;
    // the f64 bounds of the exact nodes (replay trees: a handful) ride along when all of them fit
    const size_t exact_bytes = (size_t)sc->hdr.exact_boxes.count * sizeof(RtExactBox);
    const uint32_t n_exact_lds = lds + exact_bytes <= ctx->lds_per_block ? (uint32_t)sc->hdr.exact_boxes.count : 0u;
    lds += (size_t)n_exact_lds * sizeof(RtExactBox);
    const bool exact_nodes = sc->hdr.n_exact_nodes != 0u || (force & 1u);
    const bool sphere_media = sc->sphere_media && !(force & 2u);
    // Scheduling thresholds of the traversal kernel, from sweeps (tools/tune.sh): idle lanes of a wave before it draws new
    // rays (trees that replay the reference's order - long forced descents - like later refills), and the share of its
    // ray-holding lanes that must wait at a leaf before the leaves are tested.
    out->refill_min = ctx->tune_refill ? ctx->tune_refill : (exact_nodes ? 48u : 42u);
    out->leaf_pct = ctx->tune_leaf ? ctx->tune_leaf : 80u;
#define RT_EXT_ROW(E, S) {{k_extend_tree<E, S, false, false>, k_extend_tree<E, S, false, true>}, {k_extend_tree<E, S, true, false>, k_extend_tree<E, S, true, true>}}
    static const ExtendKernel table[2][2][2][2] = {{RT_EXT_ROW(false, false), RT_EXT_ROW(false, true)}, {RT_EXT_ROW(true, false), RT_EXT_ROW(true, true)}};
#undef RT_EXT_ROW
    const bool all_cached = n_cached == sc->hdr.nodes.count; // every node of the scene has a place in LDS
    out->kernel = table[exact_nodes][sphere_media][query][all_cached];
    // a tree with plain leaves: the instantiation without the code for lists and frame changes (not for a scene without a
    // tree: its lock-step list may hold a small tree's steps, which that instantiation's list walk leaves out; and only for a
    // tree whose stack need has been verified: its box steps push without asking for room)
    if (sc->plain_leaves && sc->stack_verified && !exact_nodes && sphere_media && sc->hdr.world_tree != RT_REF_NONE && !(force & 8u)) {
        // (many moving spheres - one primitive in eight or more: the instantiations whose sphere test takes them along)
        const size_t prims = sc->hdr.spheres.count + sc->hdr.mspheres.count + sc->hdr.rects.count + sc->hdr.boxes.count;
        const bool mm = sc->hdr.mspheres.count * 8u >= prims && sc->hdr.mspheres.count != 0u && !(force & 32u);
#define RT_PLAIN_ROW(Q, M) {k_extend_tree<false, true, Q, false, true, 0, M>, k_extend_tree<false, true, Q, true, true, 0, M>}
        static const ExtendKernel plain_tree[2][2][2] = {{RT_PLAIN_ROW(false, false), RT_PLAIN_ROW(true, false)}, {RT_PLAIN_ROW(false, true), RT_PLAIN_ROW(true, true)}};
#undef RT_PLAIN_ROW
        out->kernel = plain_tree[mm][query][all_cached];
        // every node in LDS and few enough of them: the records as fourteen arrays (rt_node_step_signed), if that fits too
#define RT_ARRAYS_ROW(N, M) {k_extend_tree<false, true, false, true, true, N, M>, k_extend_tree<false, true, true, true, true, N, M>}
        static const ExtendKernel arrays512[2][2] = {RT_ARRAYS_ROW(512, false), RT_ARRAYS_ROW(512, true)};
        static const ExtendKernel arrays1536[2][2] = {RT_ARRAYS_ROW(1536, false), RT_ARRAYS_ROW(1536, true)};
#undef RT_ARRAYS_ROW
        const size_t count = sc->hdr.nodes.count;
        const uint32_t arrays = count <= 512u ? 512u : count <= 1536u ? 1536u : 0u;
        if (all_cached && arrays != 0u && !(force & 16u) && lds_stack_bytes + (size_t)arrays * 14u * sizeof(uint32_t) + 16 <= ctx->lds_per_block) {
            out->kernel = arrays == 512u ? arrays512[mm][query] : arrays1536[mm][query];
            lds = lds_stack_bytes + (size_t)arrays * 14u * sizeof(uint32_t) + 16;
        }
    }
    out->list_only = sc->hdr.world_tree == RT_REF_NONE && !(force & 4u);
    if (out->list_only) { // nothing to walk per lane
        static const ExtendKernel lists[2][2] = {{k_extend_list<false, false>, k_extend_list<false, true>}, {k_extend_list<true, false>, k_extend_list<true, true>}};
        out->kernel = lists[sphere_media][query];
        lds = 0;
        out->stack_cap_arg = sc->steps_at; out->refill_arg = sc->n_steps; // (what k_extend_list takes in those two places)
#ifdef RT_BOUNDS_CHECK
        out->refill_arg = 0u; // the checked build walks the steps through the list's own code, which checks every reference
#endif
    }
    if (lds > 65536)
        RT_HIP(hipFuncSetAttribute((const void*)out->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "LDS size of k_extend_tree");
    if (getenv("RT_DEBUG"))
        fprintf(stderr, "rt: lds_per_block %zu, stack_need %u, nodes %llu, cached %u, lds %zu\n", ctx->lds_per_block, stack_cap,
                (unsigned long long)sc->hdr.nodes.count, n_cached, lds);
    out->block = out->list_only ? (uint32_t)RT_BLOCK : (uint32_t)RT_TREE_BLOCK;
    if (!out->list_only) { out->stack_cap_arg = stack_cap; out->refill_arg = out->refill_min; }
    out->lds = lds; out->stack_cap = stack_cap; out->n_cached = n_cached; out->n_exact_lds = n_exact_lds;
    return RT_OK;
}

template <uint32_t BUCKET>
void launch_shade(rt_context* ctx, const rt_scene* sc, const PassInfo& pi, const PathBufs& pb, const QueueSet& qs, uint32_t it,
                  uint32_t grid, EventTimer& tm) {
    if (!(sc->bucket_mask >> BUCKET & 1u)) return; // no material of the scene is shaded by this kernel
    size_t a = tm.begin();
    constexpr uint32_t kChecker = 1u << RT_TEX_CHECKER, kImage = 1u << RT_TEX_IMAGE;
    // A scene whose only computed textures are checkers (a ground of two colours) or images (a globe, emissive pictures) - for
    // textured Lambertian also beside noise, which has a bucket of its own there - runs the kernel without the code for the
    // other kinds.
    bool launched = false;
    if constexpr (BUCKET != RT_BUCKET_LAMBERTIAN_NOISE) {
        if (sc->tex_kinds == kChecker) {
            hipLaunchKernelGGL((k_shade<BUCKET, kChecker>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, sc->view, pi, pb, qs, it);
            launched = true;
        } else if (sc->tex_kinds == kImage || ((sc->tex_kinds & ~(1u << RT_TEX_NOISE)) == kImage && BUCKET == RT_BUCKET_LAMBERTIAN_TEX)) {
            hipLaunchKernelGGL((k_shade<BUCKET, kImage>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, sc->view, pi, pb, qs, it);
            launched = true;
        }
    }
    if (!launched) hipLaunchKernelGGL(k_shade<BUCKET>, dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, sc->view, pi, pb, qs, it);
    tm.end(a, 2 + (int)rt_bucket_material(BUCKET));
}

} // namespace

extern "C" {

int rt_init(rt_context** out, int device) try {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RT_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return RT_ERR_INVALID;
    rt_context* ctx = new (std::nothrow) rt_context();
    if (!ctx) return RT_ERR_OOM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return RT_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) ctx->lds_per_block = (size_t)v;
    }
    if (hipMalloc((void**)&ctx->error_flag, sizeof(uint32_t)) != hipSuccess) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return RT_ERR_OOM;
    }
    if (hipMemsetAsync(ctx->error_flag, 0, sizeof(uint32_t), ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(ctx->error_flag);
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return RT_ERR_NO_DEVICE;
    }
    {   // this device's address of the scratch array (the device is current: hipSetDevice above)
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) {
            (void)hipGetLastError();
            hipFree(ctx->error_flag);
            hipStreamDestroy(ctx->stream);
            delete ctx;
            return RT_ERR_NO_DEVICE;
        }
        ctx->noalloc = (unsigned char*)p;
    }
    if (const char* e = getenv("RT_TUNE_REFILL")) ctx->tune_refill = (uint32_t)std::max(1, std::min(64, atoi(e)));
    if (const char* e = getenv("RT_TUNE_LEAF")) ctx->tune_leaf = (uint32_t)std::max(1, std::min(100, atoi(e))); // per cent
    *out = ctx;
    return RT_OK;
} RT_ABI_CATCH

void rt_shutdown(rt_context* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->events) hipEventDestroy(e);
    if (ctx->f64pool) hipFree(ctx->f64pool);
    if (ctx->hit) hipFree(ctx->hit);
    if (ctx->u32pool) hipFree(ctx->u32pool);
    if (ctx->counters) hipFree(ctx->counters);
    if (ctx->pixmap) hipFree(ctx->pixmap);
    if (ctx->error_flag) hipFree(ctx->error_flag);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

static int upload_common(rt_context* ctx, const void* src, size_t nbytes, bool src_on_device, rt_scene** out) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!src || !out || nbytes < sizeof(RtBlobHeader)) return ctx_fail(ctx, RT_ERR_INVALID, "scene_upload: bad argument");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return ctx_fail(ctx, RT_ERR_OOM, "out of memory");
    hipError_t e = hipMemcpy(&s->hdr, src, sizeof(RtBlobHeader), src_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
    if (e != hipSuccess) { delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "read blob header", e); }
    int rc = parse_header(ctx, s->hdr, nbytes);
    if (rc != RT_OK) { delete s; return rc; }
    e = hipMalloc(&s->d_blob, nbytes);
    if (e != hipSuccess) { delete s; return ctx_fail(ctx, RT_ERR_OOM, "hipMalloc scene", e); }
    e = hipMemcpy(s->d_blob, src, nbytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "copy scene", e); }
    s->nbytes = nbytes;
    make_view(s);
    s->has_flat = s->hdr.world_flat_any_time != RT_REF_NONE;
    s->has_tree = s->hdr.world_tree != RT_REF_NONE;
    {
        std::vector<RtMaterial> mats(s->hdr.materials.count);
        std::vector<RtTexture> texs(s->hdr.textures.count);
        const uint8_t* b = (const uint8_t*)src;
        const hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
        if (!mats.empty()) e = hipMemcpy(mats.data(), b + s->hdr.materials.offset, mats.size() * sizeof(RtMaterial), kind);
        if (e == hipSuccess && !texs.empty()) e = hipMemcpy(texs.data(), b + s->hdr.textures.offset, texs.size() * sizeof(RtTexture), kind);
        if (e != hipSuccess) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "read material table", e); }
        for (const RtTexture& t : texs) if (t.type <= RT_TEX_IMAGE && t.type != RT_TEX_SOLID) s->tex_kinds |= 1u << t.type;
        for (const RtMaterial& m : mats) {
            if (m.type >= RT_MAT_COUNT) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_INVALID, "blob: unknown material type"); }
            const bool textured = m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_DIFFUSE_LIGHT || m.type == RT_MAT_ISOTROPIC;
            if (textured && m.tex >= texs.size()) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_INVALID, "blob: texture index out of range"); }
            s->bucket_mask |= 1u << rt_bucket_of(m.type, textured ? texs[m.tex].type : (uint32_t)RT_TEX_SOLID);
        }
    }
    {
        std::vector<RtMedium> media(s->hdr.media.count);
        if (!media.empty())
            e = hipMemcpy(media.data(), (const uint8_t*)src + s->hdr.media.offset, media.size() * sizeof(RtMedium),
                          src_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
        if (e != hipSuccess) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "read media", e); }
        for (const RtMedium& m : media) s->sphere_media = s->sphere_media && RT_REF_TYPE(m.boundary) == RT_REF_SPHERE;
        // a lock-step list that is one medium bounded by a plain sphere, alone: its two records go into the scene view
        std::vector<uint32_t> entries(s->hdr.list_entries.count);
        const hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
        if (!entries.empty()) e = hipMemcpy(entries.data(), (const uint8_t*)src + s->hdr.list_entries.offset, entries.size() * sizeof(uint32_t), kind);
        if (e != hipSuccess) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "read list entries", e); }
        const uint32_t lists[2] = {s->hdr.world_flat, s->hdr.world_flat_any_time};
        for (int w = 0; w < 2; ++w) {
            memset(&s->one_medium[w].medium, 0, sizeof(RtMedium));
            memset(&s->one_medium[w].sphere, 0, sizeof(RtSphere));
            if (lists[w] == RT_REF_NONE || RT_REF_TYPE(lists[w]) != RT_REF_LIST) continue;
            const size_t li = RT_REF_INDEX(lists[w]);
            if (li + 1 >= entries.size() || entries[li + 1] != RT_REF_NONE) continue; // not exactly one entry
            const uint32_t ref = entries[li];
            if (RT_REF_TYPE(ref) != RT_REF_MEDIUM || RT_REF_INDEX(ref) >= media.size()) continue;
            const RtMedium& m = media[RT_REF_INDEX(ref)];
            if (RT_REF_TYPE(m.boundary) != RT_REF_SPHERE || RT_REF_INDEX(m.boundary) >= s->hdr.spheres.count) continue;
            RtSphere sp;
            e = hipMemcpy(&sp, (const uint8_t*)src + s->hdr.spheres.offset + (size_t)RT_REF_INDEX(m.boundary) * sizeof(RtSphere), sizeof sp, kind);
            if (e != hipSuccess) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "read a medium's sphere", e); }
            s->one_medium[w].ref = ref;
            s->one_medium[w].medium = m;
            s->one_medium[w].sphere = sp;
        }
        // a small tree's steps on the lock-step list (the compiler puts them on a list that serves rays of any time)
        s->steps_at = 0u; s->n_steps = 0u;
        if (s->hdr.world_flat != RT_REF_NONE && RT_REF_TYPE(s->hdr.world_flat) == RT_REF_LIST && s->hdr.world_flat == s->hdr.world_flat_any_time) {
            for (size_t li = RT_REF_INDEX(s->hdr.world_flat); li < entries.size() && entries[li] != RT_REF_NONE; ++li) {
                if (RT_REF_TYPE(entries[li]) != RT_REF_STEPS) continue;
                const uint32_t n_steps = RT_REF_INDEX(entries[li]);
                if (n_steps > 64u || li + 1u + 2u * (size_t)n_steps >= entries.size()) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_INVALID, "blob: a tree's steps run past the list"); }
                for (uint32_t j = 0; j < n_steps; ++j) { // a lane's "inside" bits are indexed by these depths
                    const uint32_t ctrl = entries[li + 1u + 2u * j], operand = entries[li + 2u + 2u * j], d = ctrl & RT_STEP_DEPTH_MASK;
                    const bool gate = (ctrl & RT_STEP_GATE) != 0u;
                    if (d < 1u || d > RT_STEP_MAX_DEPTH || (gate && (operand < 1u || operand > s->hdr.exact_boxes.count))) {
                        hipFree(s->d_blob); delete s;
                        return ctx_fail(ctx, RT_ERR_INVALID, "blob: a tree's step names a depth or a box that does not exist");
                    }
                }
                s->steps_at = (uint32_t)li; s->n_steps = n_steps;
                break;
            }
        }
        // What the leaves of the world's tree are: if none of them is a list, a frame-changing instance or a linked instance
        // record, the traversal kernel runs an instantiation without the code for those (they cost it 25 registers).
        // (a tree that the device builds later keeps its leaves, rt_tree_build.hip only rearranges them)
        if (RT_REF_TYPE(s->hdr.world_tree) == RT_REF_NODE && s->hdr.max_chain <= 2u && s->hdr.nodes.count <= (256u << 20) / sizeof(RtNode)) {
            std::vector<RtNode> nodes(s->hdr.nodes.count);
            e = hipMemcpy(nodes.data(), (const uint8_t*)src + s->hdr.nodes.offset, nodes.size() * sizeof(RtNode), kind);
            if (e != hipSuccess) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_NO_DEVICE, "read nodes", e); }
            bool plain = true;
            for (const RtNode& n : nodes)
                for (uint32_t c : {n.c0, n.c1}) {
                    const uint32_t ty = RT_REF_TYPE(c);
                    plain = plain && (ty == RT_REF_NONE || ty == RT_REF_NODE || ty == RT_REF_SPHERE || ty == RT_REF_MSPHERE || ty == RT_REF_RECT ||
                                      ty == RT_REF_BOX || ty == RT_REF_MEDIUM || ty == RT_REF_ILEAF);
                }
            s->plain_leaves = plain;
            // The stack such a tree needs, worked out here and not taken from the header: a node both of whose boxes are hit
            // keeps one child and goes on with the other, so need(node) = 1 + the larger of its children's needs (a leaf: 0;
            // one child: that child's). A node that is its own descendant would keep any traversal going for ever: refused.
            if (plain && !(s->hdr.flags & RT_BLOB_FLAG_DEVICE_TREE)) {
                const size_t count = nodes.size();
                const uint32_t root = RT_REF_INDEX(s->hdr.world_tree);
                std::vector<uint8_t> state(count, 0); // 1: on the current path, 2: done
                std::vector<uint32_t> need(count, 0u);
                std::vector<std::pair<uint32_t, uint32_t>> path; // (node, children looked at)
                bool ok = root < count, cycle = false;
                if (ok) { path.emplace_back(root, 0u); state[root] = 1; }
                while (ok && !path.empty()) {
                    const uint32_t i = path.back().first, k = path.back().second;
                    const RtNode& n = nodes[i];
                    if (k < 2u) {
                        path.back().second = k + 1u;
                        const uint32_t c = k == 0u ? n.c0 : n.c1;
                        if (RT_REF_TYPE(c) != RT_REF_NODE) continue;
                        const uint32_t j = RT_REF_INDEX(c);
                        if (j >= count) { ok = false; break; }
                        if (state[j] == 1) { ok = false; cycle = true; break; }
                        if (state[j] == 0) { state[j] = 1; path.emplace_back(j, 0u); }
                        continue;
                    }
                    const uint32_t n0 = RT_REF_TYPE(n.c0) == RT_REF_NODE ? need[RT_REF_INDEX(n.c0)] : 0u;
                    const uint32_t n1 = RT_REF_TYPE(n.c1) == RT_REF_NODE ? need[RT_REF_INDEX(n.c1)] : 0u;
                    need[i] = std::max(n0, n1) + ((n.c0 != RT_REF_NONE && n.c1 != RT_REF_NONE) ? 1u : 0u);
                    state[i] = 2;
                    path.pop_back();
                }
                if (cycle) { hipFree(s->d_blob); delete s; return ctx_fail(ctx, RT_ERR_INVALID, "blob: a node of the tree is its own descendant"); }
                s->stack_verified = ok && need[root] <= s->hdr.stack_need;
            }
        }
        s->choose_flat(1u); // (make_view names the any-time list)
    }
    *out = s;
    return RT_OK;
} RT_ABI_CATCH
int rt_scene_upload(rt_context* ctx, const void* blob, size_t nbytes, rt_scene** out) try { return upload_common(ctx, blob, nbytes, false, out); } RT_ABI_CATCH
int rt_scene_upload_from_device(rt_context* ctx, const void* d_blob, size_t nbytes, rt_scene** out) try { return upload_common(ctx, d_blob, nbytes, true, out); } RT_ABI_CATCH
void rt_scene_free(rt_context* ctx, rt_scene* s) {
    if (!s) return;
    if (ctx) hipSetDevice(ctx->device);
    if (s->d_blob) hipFree(s->d_blob);
    delete s;
}

int rt_scene_build_tree_device(rt_context* ctx, rt_scene* sc, rt_tree_stats* stats) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!sc) return ctx_fail(ctx, RT_ERR_INVALID, "build_tree_device: null scene");
    if (stats) memset(stats, 0, sizeof *stats);
    const RtBlobHeader& h = sc->hdr;
    if (!(h.flags & RT_BLOB_FLAG_DEVICE_TREE) || RT_REF_TYPE(h.device_tree_root) != RT_REF_NODE || h.device_tree_leaves < 1u)
        return ctx_fail(ctx, RT_ERR_INVALID, "build_tree_device: the scene was not compiled with a tree left to the device (builder flag 8)");
    if (sc->tree_built) return ctx_fail(ctx, RT_ERR_INVALID, "build_tree_device: the scene's tree has been built already");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const auto t0 = std::chrono::steady_clock::now();
    // stack entries: one per node on the way down, what the deepest leaf needs below it, one per node chained in front
    const uint32_t fixed = h.device_tree_leaf_need + h.device_tree_need_add + 1u;
    if (fixed >= RT_STACK_CAP) return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "build_tree_device: no stack left for a tree");
    RtTreeBuildResult res;
    std::string err;
    const int rc = rt_tree_build_device(ctx->stream, (RtNode*)((uint8_t*)sc->d_blob + h.nodes.offset), (size_t)h.nodes.count,
                                        RT_REF_INDEX(h.device_tree_root), h.device_tree_leaves, RT_STACK_CAP - fixed, &res, err);
    if (rc != RT_OK) return ctx_fail(ctx, rc, err.c_str());
    sc->tree_built = true;
    if (h.device_tree_leaves >= 2u) {
        const uint32_t need = std::max(h.device_tree_need_other, res.height + h.device_tree_leaf_need) + h.device_tree_need_add;
        sc->hdr.stack_need = std::max(1u, need);
    }
    // (a tree with plain leaves keeps at most one entry per node on the way down: the height the builder measured bounds its need)
    sc->stack_verified = sc->plain_leaves && sc->hdr.stack_need >= res.height;
    if (stats) {
        stats->leaves = h.device_tree_leaves; stats->height = res.height; stats->key_bits = res.key_bits; stats->stack_need = sc->hdr.stack_need;
        stats->ms_collect = res.ms_collect; stats->ms_sort = res.ms_sort; stats->ms_hierarchy = res.ms_hierarchy;
        stats->ms_refit = res.ms_refit; stats->ms_commit = res.ms_commit;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
} RT_ABI_CATCH

static int check_error_flag(rt_context* ctx) {
    uint32_t flag = 0;
    RT_HIP(hipMemcpy(&flag, ctx->error_flag, sizeof flag, hipMemcpyDeviceToHost), "read error flag");
    if (flag) {
        hipMemsetAsync(ctx->error_flag, 0, sizeof(uint32_t), ctx->stream); // (the stream the kernels that raise it run on)
        return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "traversal stack overflow on device");
    }
    return RT_OK;
}

int rt_trace_first_hit(rt_context* ctx, const rt_scene* sc, const rt_ray_d* rays, size_t n, const double* medium_xi, rt_hit* out) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!sc || (!rays && n) || (!out && n) || n > 0x7FFFFFFFu) return ctx_fail(ctx, RT_ERR_INVALID, "trace_first_hit: bad argument");
    if (n == 0) return RT_OK;
    for (size_t i = 0; i < n; ++i)
        if (rays[i].time < sc->hdr.valid0 || rays[i].time > sc->hdr.valid1)
            return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "trace_first_hit: a ray's time lies outside the interval the scene's BvhNodes over moving spheres were built for (compile with builder flag 4 for any time)");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    rt_ray_d* d_rays = nullptr;
    rt_hit* d_out = nullptr;
    double* d_xi = nullptr;
    int rc = RT_OK;
    const size_t xi_bytes = medium_xi ? n * (size_t)sc->hdr.n_media * sizeof(double) : 0;
    hipError_t e = hipMalloc((void**)&d_rays, n * sizeof(rt_ray_d));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, n * sizeof(rt_hit));
    if (e == hipSuccess && xi_bytes) e = hipMalloc((void**)&d_xi, xi_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray_d), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && xi_bytes) e = hipMemcpyAsync(d_xi, medium_xi, xi_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const uint32_t cap = sc->hdr.stack_need;
        const size_t lds = (size_t)cap * RT_BLOCK * sizeof(uint32_t);
        hipLaunchKernelGGL(k_first_hit, dim3(grid_for(ctx, n, 4)), dim3(RT_BLOCK), lds, ctx->stream, sc->view, d_rays, (uint32_t)n,
                           (const double*)d_xi, d_out, cap, ctx->error_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = ctx_fail(ctx, RT_ERR_NO_DEVICE, "trace_first_hit", e);
    if (d_rays) hipFree(d_rays);
    if (d_out) hipFree(d_out);
    if (d_xi) hipFree(d_xi);
    if (rc == RT_OK) rc = check_error_flag(ctx);
    return rc;
} RT_ABI_CATCH

int rt_trace_first_hit_wavefront(rt_context* ctx, const rt_scene* sc, const rt_ray_d* rays, size_t n, const double* medium_xi,
                                 rt_hit* out, uint32_t variant) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!sc || (!rays && n) || (!out && n) || n > 0x7FFFFFFFu || variant > 63u) return ctx_fail(ctx, RT_ERR_INVALID, "trace_first_hit_wavefront: bad argument");
    if (n == 0) return RT_OK;
    for (size_t i = 0; i < n; ++i)
        if (rays[i].time < sc->hdr.valid0 || rays[i].time > sc->hdr.valid1)
            return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "trace_first_hit_wavefront: a ray's time lies outside the interval the scene's BvhNodes over moving spheres were built for (compile with builder flag 4 for any time)");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    ExtendLaunch xl;
    int rc = extend_launch_of(ctx, sc, true, variant, &xl);
    if (rc != RT_OK) return rc;
    rt_ray_d* d_rays = nullptr;
    rt_hit* d_out = nullptr;
    double* d_xi = nullptr;
    RtHitLite* d_lite = nullptr;
    const size_t xi_bytes = medium_xi ? n * (size_t)sc->hdr.n_media * sizeof(double) : 0;
    hipError_t e = hipMalloc((void**)&d_rays, n * sizeof(rt_ray_d));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, n * sizeof(rt_hit));
    if (e == hipSuccess) e = hipMalloc((void**)&d_lite, n * sizeof(RtHitLite));
    if (e == hipSuccess && xi_bytes) e = hipMalloc((void**)&d_xi, xi_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray_d), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && xi_bytes) e = hipMemcpyAsync(d_xi, medium_xi, xi_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        QueryBatch qb;
        qb.rays = d_rays; qb.xi_rows = d_xi; qb.hit = d_lite; qb.n = (uint32_t)n;
        PassInfo pi;
        memset(&pi, 0, sizeof pi);
        PathBufs pb;
        memset(&pb, 0, sizeof pb);
        QueueSet qs;
        memset(&qs, 0, sizeof qs);
        qs.error_flag = ctx->error_flag;
        hipLaunchKernelGGL(xl.kernel, dim3(xl.grid(ctx, n)), dim3(xl.block), xl.lds, ctx->stream, sc->view, pi, pb, qs, 0u, xl.stack_cap_arg,
                           xl.refill_arg, xl.leaf_pct, xl.n_cached, xl.n_exact_lds, qb);
        hipLaunchKernelGGL(k_query_finish, dim3(grid_for(ctx, n, 8)), dim3(RT_BLOCK), 0, ctx->stream, sc->view, qb, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); rc = ctx_fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_NO_DEVICE, "trace_first_hit_wavefront", e); }
    if (d_rays) hipFree(d_rays);
    if (d_out) hipFree(d_out);
    if (d_lite) hipFree(d_lite);
    if (d_xi) hipFree(d_xi);
    if (rc == RT_OK) rc = check_error_flag(ctx);
    return rc;
} RT_ABI_CATCH

int rt_trace_first_hit_noalloc(rt_context* ctx, const rt_scene* sc, const rt_ray_d* rays, size_t n, const double* medium_xi, rt_hit* out) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!sc || (!rays && n) || (!out && n)) return ctx_fail(ctx, RT_ERR_INVALID, "trace_first_hit_noalloc: bad argument");
    if (n == 0) return RT_OK;
    const size_t n_media = medium_xi ? (size_t)sc->hdr.n_media : 0;
    const size_t per_ray = sizeof(rt_ray_d) + sizeof(rt_hit) + n_media * sizeof(double);
    const size_t room = RT_NOALLOC_BYTES - RT_NOALLOC_HEAD - 2u * 256u; // (the three staging areas start on 256-byte boundaries)
    if (n > room / per_ray) {
        char msg[200];
        snprintf(msg, sizeof msg, "trace_first_hit_noalloc: %zu rays exceed the %zu this scene's rays fit in the fixed %u-byte work area (use rt_trace_first_hit)",
                 n, room / per_ray, (unsigned)RT_NOALLOC_BYTES);
        return ctx_fail(ctx, RT_ERR_INVALID, msg);
    }
    for (size_t i = 0; i < n; ++i)
        if (rays[i].time < sc->hdr.valid0 || rays[i].time > sc->hdr.valid1)
            return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "trace_first_hit_noalloc: a ray's time lies outside the interval the scene's BvhNodes over moving spheres were built for (compile with builder flag 4 for any time)");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    auto up256 = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    unsigned char* base = ctx->noalloc; // the device address, never the array's name
    uint32_t* d_flag = (uint32_t*)base;
    rt_ray_d* d_rays = (rt_ray_d*)(base + RT_NOALLOC_HEAD);
    rt_hit* d_out = (rt_hit*)((unsigned char*)d_rays + up256(n * sizeof(rt_ray_d)));
    double* d_xi = n_media ? (double*)((unsigned char*)d_out + up256(n * sizeof(rt_hit))) : nullptr;
    uint32_t flag = 0;
    // the array keeps what the last call left in it: the flag must start at zero; the staging areas are overwritten
    hipError_t e = hipMemsetAsync(d_flag, 0, RT_NOALLOC_HEAD, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray_d), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && d_xi) e = hipMemcpyAsync(d_xi, medium_xi, n * n_media * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const uint32_t cap = sc->hdr.stack_need;
        const size_t lds = (size_t)cap * RT_BLOCK * sizeof(uint32_t);
        hipLaunchKernelGGL(k_first_hit, dim3(grid_for(ctx, n, 4)), dim3(RT_BLOCK), lds, ctx->stream, sc->view, (const rt_ray_d*)d_rays, (uint32_t)n,
                           (const double*)d_xi, d_out, cap, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx_fail(ctx, RT_ERR_NO_DEVICE, "trace_first_hit_noalloc", e); }
    if (flag) return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "traversal stack overflow on device");
    return RT_OK;
} RT_ABI_CATCH

int rt_generate_rays(rt_context* ctx, const rt_camera* cam, const rt_params* p, uint32_t sample, rt_ray_d* out) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!cam || !p || !out || p->width < 2 || p->height < 2) return ctx_fail(ctx, RT_ERR_INVALID, "generate_rays: bad argument");
    RtCameraDev cd;
    int rc = rt_camera_derive(cam, &cd);
    if (rc != RT_OK) return ctx_fail(ctx, rc, "camera: the reference would panic on these arguments");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t n = (size_t)p->width * p->height;
    rt_ray_d* d = nullptr;
    RT_HIP(hipMalloc((void**)&d, n * sizeof(rt_ray_d)), "hipMalloc rays");
    hipLaunchKernelGGL(k_camera_rays, dim3(grid_for(ctx, n, 8)), dim3(RT_BLOCK), 0, ctx->stream, cd, p->width, p->height, sample,
                       p->seed, p->compat, d);
    hipError_t e = hipMemcpyAsync(out, d, n * sizeof(rt_ray_d), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e != hipSuccess) return ctx_fail(ctx, RT_ERR_NO_DEVICE, "generate_rays", e);
    return RT_OK;
} RT_ABI_CATCH

int rt_render_accumulate(rt_context* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, uint32_t sample_offset,
                         uint32_t spp, double* d_rgb_sum, rt_stats* stats) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!sc || !cam || !p || !d_rgb_sum) return ctx_fail(ctx, RT_ERR_INVALID, "render: null argument");
    if (p->width < 2 || p->height < 2 || p->max_depth == 0 || p->max_depth > 65535u)
        return ctx_fail(ctx, RT_ERR_INVALID, "render: width/height must be >= 2 and 1 <= max_depth <= 65535");
    if (sc->hdr.n_media > 65535u - RT_STREAM_MEDIUM0) return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "render: too many media");
    RtCameraDev cd;
    int rc = rt_camera_derive(cam, &cd);
    if (rc != RT_OK) return ctx_fail(ctx, rc, "camera: the reference would panic on these arguments");
    if (cam->time0 < sc->hdr.valid0 || cam->time1 > sc->hdr.valid1)
        return ctx_fail(ctx, RT_ERR_UNSUPPORTED, "render: the camera's shutter leaves the interval the scene's BvhNodes over moving spheres were built for (compile with builder flag 4 for any time)");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    rc = ensure_pixmap(ctx, p);
    if (rc != RT_OK) return rc;
    const uint32_t n_local = ctx->pm_n;
    if (n_local == 0 || spp == 0) return RT_OK;
    // Samples per pixel traced together. The bounces of a pass are dependent rounds of launches, each with a latency floor and a
    // tail, so a pass runs faster the larger it is (Book-2: 4.3 G rays/s at 64 spp per pass, 7.8 at 1536; the 1 M-sphere scene
    // 5.2 at 16, 6.6 at 64): left to the library, a pass takes what the job asks for, up to seven eighths of the device memory
    // that is free (232 bytes per path) - and half of that, and half again, while the allocation does not succeed.
    size_t budget = (size_t)1 << 23;
    if (!p->spp_per_pass) {
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) {
            const size_t held = ctx->cap_paths * RT_PATH_BYTES; // (this context's own pool is free for the purpose)
            budget = std::max(budget, (free_bytes + held - (free_bytes + held) / 8) / RT_PATH_BYTES);
        } else (void)hipGetLastError();
    }
    uint32_t S = p->spp_per_pass ? p->spp_per_pass : (uint32_t)std::max<size_t>(1, std::min<size_t>(budget / n_local, 0x7FFFFFFFu / n_local));
    S = std::min(S, spp);
    if (!p->spp_per_pass && S < spp) S = (spp + (spp + S - 1u) / S - 1u) / ((spp + S - 1u) / S); // passes of equal size
    if ((size_t)n_local * S > 0x7FFFFFFFu) return ctx_fail(ctx, RT_ERR_INVALID, "render: pixels * spp_per_pass exceeds 2^31");
    for (;;) {
        rc = ensure_paths(ctx, (size_t)n_local * S, p->max_depth);
        if (rc != RT_ERR_OOM || p->spp_per_pass || S <= 1u) break;
        S = (S + 1u) / 2u; // (someone else took the memory in between: a smaller pass)
    }
    if (rc != RT_OK) return rc;

    PathBufs pb;
    pb.ray[0] = (PathRay*)ctx->f64pool;
    pb.ray[1] = pb.ray[0] + ctx->cap_paths;
    pb.thr[0] = (PathThr*)(pb.ray[1] + ctx->cap_paths);
    pb.thr[1] = pb.thr[0] + ctx->cap_paths;
    pb.rad = (double*)(pb.thr[1] + ctx->cap_paths);
    pb.hit = ctx->hit;
    QueueSet qs;
    qs.bucket = ctx->u32pool;
    qs.bucket_cap = ctx->cap_paths;
    qs.counters = ctx->counters;
    qs.error_flag = ctx->error_flag;

    rt_scene local = *sc; // per-call compat bits ride in the by-value scene view
    local.view.compat = p->compat;
    // every ray of this render has its time in the camera's shutter interval (camera.rs:62): when the bounds the
    // compiler gave moving spheres in the tree hold for all of it, those spheres need no lock-step test
    if (cam->time0 >= sc->hdr.shutter0 && cam->time1 <= sc->hdr.shutter1 && cam->time0 <= cam->time1) { local.view.world_flat = sc->hdr.world_flat; local.choose_flat(0u); }

    ExtendLaunch xl;
    rc = extend_launch_of(ctx, sc, false, 0u, &xl);
    if (rc != RT_OK) return rc;
    const uint32_t leaf_pct = xl.leaf_pct, n_cached = xl.n_cached, n_exact_lds = xl.n_exact_lds;
    const size_t lds = xl.lds;
    const ExtendKernel k_extend = xl.kernel;
    QueryBatch no_query;
    no_query.rays = nullptr; no_query.xi_rows = nullptr; no_query.hit = nullptr; no_query.n = 0u;
    const size_t n_counters = (size_t)(p->max_depth + 2) * RT_COUNTER_STRIDE;
    std::vector<uint32_t> host_counters(n_counters);

    const auto t_begin = std::chrono::steady_clock::now();
    uint32_t done = 0;
    while (done < spp) {
        const uint32_t s_now = std::min(S, spp - done);
        PassInfo pi;
        pi.pixmap = ctx->pixmap;
        pi.n_local = n_local; pi.S = s_now; pi.sample_offset = sample_offset + done;
        pi.width = p->width; pi.height = p->height; pi.compat = p->compat; pi.bg_mode = p->background_mode;
        pi.max_depth = p->max_depth;
        pi.bg[0] = p->background[0]; pi.bg[1] = p->background[1]; pi.bg[2] = p->background[2];
        pi.seed = p->seed;
        const size_t paths = (size_t)n_local * s_now;
        EventTimer tm{ctx, p->profile != 0};

        RT_HIP(hipMemsetAsync(ctx->counters, 0, n_counters * sizeof(uint32_t), ctx->stream), "clear counters");
        hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(RT_BLOCK), 0, ctx->stream, ctx->counters, (uint32_t)paths);
        size_t a = tm.begin();
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(ctx, paths, 128)), dim3(RT_BLOCK), 0, ctx->stream, cd, pi, pb);
        tm.end(a, 0);
        const uint32_t g_ext = xl.grid(ctx, paths);
        const uint32_t g_sh = grid_for(ctx, paths, 128); // (many more workgroups than fit at once: 8 per CU, each looping longer, cost k_shade_main 7 %)
        for (uint32_t it = 0; it < p->max_depth; ++it) {
            a = tm.begin();
            hipLaunchKernelGGL(k_extend, dim3(g_ext), dim3(xl.block), lds, ctx->stream, local.view, pi, pb, qs, it, xl.stack_cap_arg,
                               xl.refill_arg, leaf_pct, n_cached, n_exact_lds, no_query);
            tm.end(a, 1);
            a = tm.begin();
            hipLaunchKernelGGL(k_shade_main, dim3(g_sh), dim3(RT_BLOCK), 0, ctx->stream, local.view, pi, pb, qs, it);
            tm.end(a, 9);
            launch_shade<RT_BUCKET_LAMBERTIAN_TEX>(ctx, &local, pi, pb, qs, it, g_sh, tm);
            launch_shade<RT_BUCKET_LAMBERTIAN_NOISE>(ctx, &local, pi, pb, qs, it, g_sh, tm);
            launch_shade<RT_BUCKET_LIGHT_TEX>(ctx, &local, pi, pb, qs, it, g_sh, tm);
            launch_shade<RT_BUCKET_ISOTROPIC_TEX>(ctx, &local, pi, pb, qs, it, g_sh, tm);
        }
        a = tm.begin();
        hipLaunchKernelGGL(k_accumulate, dim3(grid_for(ctx, (size_t)n_local * 64u, 8)), dim3(RT_BLOCK), 0, ctx->stream, pi, pb, d_rgb_sum);
        tm.end(a, 7);
        RT_HIP(hipGetLastError(), "kernel launch");
        RT_HIP(hipMemcpyAsync(host_counters.data(), ctx->counters, n_counters * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream),
               "read counters");
        RT_HIP(hipStreamSynchronize(ctx->stream), "render pass");
#ifdef RT_PHASE_STATS
        {
            unsigned long long h[4][48];
            RT_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h), "phase stats");
            for (int g = 0; g < 4; ++g) {
                fprintf(stderr, "phase[bounce %d%s]:", g, g == 3 ? "+" : "");
                for (int k = 0; k < 48; ++k) fprintf(stderr, " %llu", h[g][k]);
                fprintf(stderr, "\n");
            }
        }
#endif
        if (stats) {
            stats->passes += 1;
            stats->camera_samples += paths;
            uint64_t seg = 0;
            for (uint32_t it = 0; it < p->max_depth; ++it)
                seg += (uint64_t)host_counters[(size_t)it * RT_COUNTER_STRIDE] + host_counters[(size_t)it * RT_COUNTER_STRIDE + RT_COUNTER_BACK];
            stats->ray_segments += seg;
            if (getenv("RT_DEBUG")) {
                fprintf(stderr, "rt: live paths per bounce:");
                for (uint32_t it = 0; it < p->max_depth; ++it) fprintf(stderr, " %u+%u", host_counters[(size_t)it * RT_COUNTER_STRIDE], host_counters[(size_t)it * RT_COUNTER_STRIDE + RT_COUNTER_BACK]);
                fprintf(stderr, "\n");
            }
            stats->medium_tests += seg * sc->hdr.n_media;
            for (const EventTimer::Span& sp : tm.spans) {
                float ms = 0.f;
                hipEventElapsedTime(&ms, ctx->events[sp.a], ctx->events[sp.b]);
                if (sp.kind == 0) { stats->raygen_ms += ms; stats->raygen_launches++; }
                else if (sp.kind == 1) { stats->extend_ms += ms; stats->extend_launches++; }
                else if (sp.kind == 7) { stats->accumulate_ms += ms; stats->accumulate_launches++; }
                else if (sp.kind == 9) { stats->shade_main_ms += ms; stats->shade_main_launches++; }
                else { stats->shade_ms[sp.kind - 2] += ms; stats->shade_launches++; }
            }
            if (p->profile) stats->extend_segments += seg;
        }
        done += s_now;
    }
    if (stats) stats->seconds_render += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return check_error_flag(ctx);
} RT_ABI_CATCH

void rt_tonemap(const double* rgb_sum, size_t n_pixels, double spp, uint8_t* rgb8_out) {
    if (!rgb_sum || !rgb8_out) return;
    for (size_t i = 0; i < n_pixels * 3; ++i) rgb8_out[i] = rt_tonemap_channel(rgb_sum[i], spp);
}

int rt_render(rt_context* ctx, const rt_scene* sc, const rt_camera* cam, const rt_params* p, uint32_t spp, double* rgb_sum_out,
              uint8_t* rgb8_out, rt_stats* stats) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!p || !rgb_sum_out) return ctx_fail(ctx, RT_ERR_INVALID, "render: null argument");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t n = (size_t)p->width * p->height * 3;
    double* d_sum = nullptr;
    RT_HIP(hipMalloc((void**)&d_sum, n * sizeof(double)), "hipMalloc framebuffer");
    hipError_t e = hipMemsetAsync(d_sum, 0, n * sizeof(double), ctx->stream); // ordered before the render's kernels: same stream
    int rc = e == hipSuccess ? rt_render_accumulate(ctx, sc, cam, p, 0, spp, d_sum, stats) : ctx_fail(ctx, RT_ERR_NO_DEVICE, "memset", e);
    if (rc == RT_OK) {
        e = hipMemcpy(rgb_sum_out, d_sum, n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = ctx_fail(ctx, RT_ERR_NO_DEVICE, "download framebuffer", e);
    }
    hipFree(d_sum);
    if (rc == RT_OK && rgb8_out) rt_tonemap(rgb_sum_out, (size_t)p->width * p->height, (double)spp, rgb8_out);
    return rc;
} RT_ABI_CATCH

int rt_selftest_vec3(rt_context* ctx, uint32_t* failed) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!failed) return ctx_fail(ctx, RT_ERR_INVALID, "selftest_vec3: null argument");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    uint32_t* d = nullptr;
    RT_HIP(hipMalloc((void**)&d, sizeof(uint32_t)), "hipMalloc");
    volatile double one = 1.0; // (read at run time: the kernel must do the arithmetic, not the compiler)
    hipLaunchKernelGGL(k_vec3_kat, dim3(1), dim3(64), 0, ctx->stream, (double)one, d);
    hipError_t e = hipGetLastError();
    *failed = 0xFFFFFFFFu;
    if (e == hipSuccess) e = hipMemcpyAsync(failed, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx_fail(ctx, RT_ERR_NO_DEVICE, "selftest_vec3", e); }
    return RT_OK;
} RT_ABI_CATCH

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n < 0 ? 0 : n;
}

int rt_render_multi(rt_context* const* ctxs, uint32_t n_ctx, const void* blob, size_t nbytes, const rt_camera* cam, const rt_params* p,
                    uint32_t spp, double* rgb_sum_out, uint8_t* rgb8_out, rt_stats* stats) try {
    if (!ctxs || n_ctx == 0u || !ctxs[0]) return RT_ERR_INVALID;
    rt_context* ctx = ctxs[0]; // errors are reported on the first context
    for (uint32_t i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return ctx_fail(ctx, RT_ERR_INVALID, "render_multi: null context");
        for (uint32_t j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return ctx_fail(ctx, RT_ERR_INVALID, "render_multi: the same context twice (a context serves one host thread at a time)");
    }
    if (!blob || !cam || !p || !rgb_sum_out) return ctx_fail(ctx, RT_ERR_INVALID, "render_multi: null argument");
    if (p->width < 2 || p->height < 2) return ctx_fail(ctx, RT_ERR_INVALID, "render_multi: width/height must be >= 2");
    const size_t n_doubles = (size_t)p->width * p->height * 3;
    struct Rank {
        rt_scene* scene = nullptr;
        void* d_blob = nullptr; // the blob as it arrived on this rank's device (ranks > 0)
        double* d_sum = nullptr;
        int rc = RT_OK;
        rt_stats st;
    };
    std::vector<Rank> ranks(n_ctx);
    auto cleanup = [&]() {
        for (uint32_t i = 0; i < n_ctx; ++i) {
            hipSetDevice(ctxs[i]->device);
            if (ranks[i].scene) rt_scene_free(ctxs[i], ranks[i].scene);
            if (ranks[i].d_blob) hipFree(ranks[i].d_blob);
            if (ranks[i].d_sum) hipFree(ranks[i].d_sum);
        }
    };
    // The scene goes to the first context's device from the host once, and from there to every other device directly
    // (the role of Arc::clone(&world) per job, main.rs:145): GPU to GPU over xGMI, a one-hop fan-out on a fully connected
    // node - hipMemcpyPeer; for two contexts on one device a plain device copy.
    int rc = rt_scene_upload(ctxs[0], blob, nbytes, &ranks[0].scene);
    if (rc != RT_OK) { cleanup(); return rc; }
    for (uint32_t i = 1; i < n_ctx && rc == RT_OK; ++i) {
        hipError_t e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipMalloc(&ranks[i].d_blob, nbytes);
        if (e == hipSuccess) e = hipMemcpyPeer(ranks[i].d_blob, ctxs[i]->device, ranks[0].scene->d_blob, ctxs[0]->device, nbytes);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = ctx_fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_NO_DEVICE, "render_multi: scene broadcast", e); break; }
        rc = rt_scene_upload_from_device(ctxs[i], ranks[i].d_blob, nbytes, &ranks[i].scene);
        if (rc != RT_OK) ctx_fail(ctx, rc, rt_last_error(ctxs[i]));
    }
    for (uint32_t i = 0; i < n_ctx && rc == RT_OK; ++i) {
        hipError_t e = hipSetDevice(ctxs[i]->device);
        if (e == hipSuccess) e = hipMalloc((void**)&ranks[i].d_sum, n_doubles * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(ranks[i].d_sum, 0, n_doubles * sizeof(double), ctxs[i]->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = ctx_fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_NO_DEVICE, "render_multi: framebuffer", e); }
    }
    if (rc != RT_OK) { cleanup(); return rc; }
    // One host thread per context, as the reference has one worker per job (main.rs:143-148): rank i renders every sample of
    // the tiles k with k % n_ctx == i. Random numbers are keyed on the pixel, so the split changes no bit of the image.
    {
        std::vector<std::thread> workers;
        for (uint32_t i = 0; i < n_ctx; ++i) {
            workers.emplace_back([&, i]() {
                rt_params mine = *p;
                mine.tile_rank = i; mine.tile_world = n_ctx;
                memset(&ranks[i].st, 0, sizeof(rt_stats));
                ranks[i].rc = rt_render_accumulate(ctxs[i], ranks[i].scene, cam, &mine, 0u, spp, ranks[i].d_sum, &ranks[i].st);
            });
        }
        for (std::thread& w : workers) w.join();
    }
    for (uint32_t i = 0; i < n_ctx; ++i) {
        if (ranks[i].rc != RT_OK && rc == RT_OK) { rc = ranks[i].rc; if (i) ctx_fail(ctx, rc, rt_last_error(ctxs[i])); }
        if (stats) stats[i] = ranks[i].st;
    }
    // Gather (main.rs:242-253): every other rank's buffer travels to the first device and its tiles are taken from it.
    if (rc == RT_OK && n_ctx > 1u) {
        hipError_t e = hipSetDevice(ctxs[0]->device);
        double* d_stage = nullptr;
        if (e == hipSuccess) e = hipMalloc((void**)&d_stage, n_doubles * sizeof(double));
        const uint32_t tile = p->tile ? p->tile : 32u;
        for (uint32_t i = 1; i < n_ctx && e == hipSuccess; ++i) {
            e = hipMemcpyPeerAsync(d_stage, ctxs[0]->device, ranks[i].d_sum, ctxs[i]->device, n_doubles * sizeof(double), ctxs[0]->stream);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(k_merge_tiles, dim3(grid_for(ctxs[0], (size_t)p->width * p->height, 8)), dim3(RT_BLOCK), 0, ctxs[0]->stream,
                               ranks[0].d_sum, (const double*)d_stage, p->width, p->height, tile, i, n_ctx);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctxs[0]->stream);
        if (d_stage) hipFree(d_stage);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = ctx_fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_NO_DEVICE, "render_multi: gather", e); }
    }
    if (rc == RT_OK) {
        hipError_t e = hipSetDevice(ctxs[0]->device);
        if (e == hipSuccess) e = hipMemcpy(rgb_sum_out, ranks[0].d_sum, n_doubles * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = ctx_fail(ctx, RT_ERR_NO_DEVICE, "render_multi: download framebuffer", e); }
    }
    cleanup();
    if (rc == RT_OK && rgb8_out) rt_tonemap(rgb_sum_out, (size_t)p->width * p->height, (double)spp, rgb8_out);
    return rc;
} RT_ABI_CATCH

int rt_device_alloc(rt_context* ctx, size_t nbytes, void** d_ptr) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!d_ptr) return ctx_fail(ctx, RT_ERR_INVALID, "device_alloc: null argument");
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    RT_HIP(hipMalloc(d_ptr, nbytes), "hipMalloc");
    return RT_OK;
} RT_ABI_CATCH
int rt_device_free(rt_context* ctx, void* d_ptr) try {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    RT_HIP(hipFree(d_ptr), "hipFree");
    return RT_OK;
} RT_ABI_CATCH
int rt_device_memset(rt_context* ctx, void* d_ptr, int value, size_t nbytes) try {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    RT_HIP(hipMemsetAsync(d_ptr, value, nbytes, ctx->stream), "hipMemset"); // on the stream the kernels run on, and finished on return
    RT_HIP(hipStreamSynchronize(ctx->stream), "hipMemset");
    return RT_OK;
} RT_ABI_CATCH
int rt_device_download(rt_context* ctx, void* host_dst, const void* d_src, size_t nbytes) try {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    RT_HIP(hipMemcpy(host_dst, d_src, nbytes, hipMemcpyDeviceToHost), "hipMemcpy");
    return RT_OK;
} RT_ABI_CATCH
int rt_device_upload(rt_context* ctx, void* d_dst, const void* host_src, size_t nbytes) try {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    RT_HIP(hipMemcpy(d_dst, host_src, nbytes, hipMemcpyHostToDevice), "hipMemcpy");
    return RT_OK;
} RT_ABI_CATCH

} // extern "C"
