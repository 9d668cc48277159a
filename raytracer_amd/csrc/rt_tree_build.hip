// BvhNode::new (raytracer/src/bvh.rs:15-66) for the world's tree, done on the device.
//
// The reference builds its tree by sorting the object list again at every level; the host compiler (rt_builder.cpp)
// replaces that with a surface-area-heuristic build, which for a million objects takes seconds of one host core. Here the
// same leaves are arranged by a linear BVH build instead: 63-bit Morton codes of the leaf centroids, one radix sort, the
// hierarchy of T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (HPG 2012) -
// internal node i of the sorted sequence finds its own range and split by binary searches over common prefixes - and
// bottom-up passes that unite the boxes and rotate subtrees where that shrinks them. What a BVH returns (the closest hit) does not depend on how its leaves are
// grouped as long as every box bounds what is below it, so the result is the tree the traversal kernels already walk:
// 64-byte nodes holding both children's boxes, nearest child first.
//
// The leaves come from the blob: the compiler puts them under a balanced stand-in tree of known shape and the build
// writes its nodes over the stand-in's records, the nodes nearest the root into the lowest-numbered records (the
// traversal kernel keeps a prefix of the node array in LDS).
#include "rt_tree_build.h"

#include <hipcub/hipcub.hpp>

#include "../../include/rt.h"

namespace {

constexpr uint32_t TB_BLOCK = 256;
constexpr uint32_t TB_LEAF = 0x80000000u; // a child id with this bit names a leaf (its position in the sorted order), without it an internal node
constexpr uint32_t TB_TOP = 4096;         // this many nodes nearest the root are numbered breadth-first
constexpr uint32_t TB_MAX_LEAVES = 1u << 26;

struct TbBox { float lo[3], hi[3]; };

// status words shared by the kernels and read back by the host
enum { ST_ERROR = 0, ST_HEIGHT = 1, ST_CMIN = 2, ST_CMAX = 5, ST_WORDS = 8 };

// floats mapped to unsigned integers of the same order, for atomicMin / atomicMax
__device__ inline uint32_t tb_ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float tb_unord(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// Leaf k of the stand-in: walk from the root, halving [s, e) the way the compiler did. On the way down the node whose
// split point is k is recorded as record k-1 of the tree (every k in 1..n-1 is the split point of exactly one node).
__global__ void __launch_bounds__(TB_BLOCK) k_tb_collect(const RtNode* __restrict__ nodes, uint32_t n_nodes, uint32_t root, uint32_t n,
                                                         uint32_t* __restrict__ leaf_ref, TbBox* __restrict__ leaf_box,
                                                         uint32_t* __restrict__ slot, uint32_t* status) {
    const uint32_t k = blockIdx.x * TB_BLOCK + threadIdx.x;
    if (k >= n) return;
    uint32_t node = root, s = 0, e = n;
    for (int step = 0; step < 40; ++step) {
        if (node >= n_nodes) break;
        const uint32_t mid = s + (e - s) / 2u;
        if (k == mid) slot[k - 1u] = node;
        const bool right = k >= mid;
        const RtNode* nd = nodes + node;
        const uint32_t c = right ? nd->c1 : nd->c0;
        if (right) s = mid; else e = mid;
        if (e - s == 1u) {
            if (RT_REF_TYPE(c) == RT_REF_NONE) break;
            TbBox b;
            for (int a = 0; a < 3; ++a) { b.lo[a] = right ? nd->lo1[a] : nd->lo0[a]; b.hi[a] = right ? nd->hi1[a] : nd->hi0[a]; }
            leaf_ref[k] = c;
            leaf_box[k] = b;
            return;
        }
        if (RT_REF_TYPE(c) != RT_REF_NODE) break;
        node = RT_REF_INDEX(c);
    }
    atomicOr(&status[ST_ERROR], 1u); // not the shape the compiler emits
}

// Bounds of the leaf centroids: reduced over each wave, then one atomic per wave and value.
__global__ void __launch_bounds__(TB_BLOCK) k_tb_bounds(const TbBox* __restrict__ leaf_box, uint32_t n, uint32_t* status) {
    const uint32_t k = blockIdx.x * TB_BLOCK + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (k < n) {
        const TbBox b = leaf_box[k];
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
            if (isfinite(c)) { lo[a] = c; hi[a] = c; }
        }
    }
    for (int m = 32; m >= 1; m >>= 1)
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], m));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m));
        }
    if ((threadIdx.x & 63u) == 0u)
        for (int a = 0; a < 3; ++a) {
            atomicMin(&status[ST_CMIN + a], tb_ord(lo[a]));
            atomicMax(&status[ST_CMAX + a], tb_ord(hi[a]));
        }
}

__device__ inline uint64_t tb_spread21(uint64_t v) { // bit i of the low 21 -> bit 3i
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x001F00000000FFFFull;
    v = (v | v << 16) & 0x001F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// `bits` per axis of the 21 are kept: fewer bits put more leaves into one cell, and leaves of one cell are split by
// position, evenly - a shallower tree of poorer quality.
__global__ void __launch_bounds__(TB_BLOCK) k_tb_morton(const TbBox* __restrict__ leaf_box, uint32_t n, const uint32_t* __restrict__ status,
                                                        uint32_t bits, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t k = blockIdx.x * TB_BLOCK + threadIdx.x;
    if (k >= n) return;
    const TbBox b = leaf_box[k];
    // one cell size for all three axes (the largest extent over 2^21): a flat scene is then not cut along its thin axis
    // until the cells are as small as it is thin
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float e = tb_unord(status[ST_CMAX + a]) - tb_unord(status[ST_CMIN + a]);
        if (isfinite(e) && e > ext) ext = e;
    }
    uint64_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float cmin = tb_unord(status[ST_CMIN + a]);
        const float c = 0.5f * b.lo[a] + 0.5f * b.hi[a];
        float x = 0.0f;
        if (ext > 0.0f && isfinite(c) && isfinite(cmin)) x = (c - cmin) / ext * 2097152.0f;
        x = fminf(fmaxf(x, 0.0f), 2097151.0f);
        q[a] = (uint64_t)((uint32_t)x >> (21u - bits));
    }
    keys[k] = (tb_spread21(q[0]) << 2) | (tb_spread21(q[1]) << 1) | tb_spread21(q[2]);
    vals[k] = k;
}

// Length of the common prefix of the keys at sorted positions i and j (equal keys: continued with the positions
// themselves, which makes all keys distinct); -1 outside the sequence.
__device__ inline int tb_delta(const uint64_t* __restrict__ keys, uint32_t n, uint32_t i, uint64_t ki, int64_t j) {
    if (j < 0 || j >= (int64_t)n) return -1;
    const uint64_t kj = keys[j];
    return ki == kj ? 64 + __clz((int)(i ^ (uint32_t)j)) : __clzll((long long)(ki ^ kj));
}

// Internal node i (Karras 2012, section 3): the direction of its range, the range's other end, the split.
__global__ void __launch_bounds__(TB_BLOCK) k_tb_hierarchy(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ kids,
                                                           uint32_t* __restrict__ iparent, uint32_t* __restrict__ lparent,
                                                           uint32_t* __restrict__ newpos) {
    const uint32_t i = blockIdx.x * TB_BLOCK + threadIdx.x;
    if (i >= n - 1u) return;
    const uint64_t ki = keys[i];
    const int64_t ii = (int64_t)i;
    const int64_t d = tb_delta(keys, n, i, ki, ii + 1) > tb_delta(keys, n, i, ki, ii - 1) ? 1 : -1;
    const int dmin = tb_delta(keys, n, i, ki, ii - d);
    int64_t lmax = 2;
    while (lmax < 2 * (int64_t)n && tb_delta(keys, n, i, ki, ii + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (tb_delta(keys, n, i, ki, ii + (l + t) * d) > dmin) l += t;
    const int64_t j = ii + l * d;
    const int dnode = tb_delta(keys, n, i, ki, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (tb_delta(keys, n, i, ki, ii + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int64_t gamma = ii + s * d + (d < 0 ? -1 : 0);
    const int64_t first = ii < j ? ii : j, last = ii < j ? j : ii;
    uint32_t left, right;
    if (first == gamma) { left = (uint32_t)gamma | TB_LEAF; lparent[gamma] = i; } else { left = (uint32_t)gamma; iparent[gamma] = i; }
    if (last == gamma + 1) { right = (uint32_t)(gamma + 1) | TB_LEAF; lparent[gamma + 1] = i; } else { right = (uint32_t)(gamma + 1); iparent[gamma + 1] = i; }
    kids[2u * i] = left;
    kids[2u * i + 1u] = right;
    newpos[i] = i;
    if (i == 0u) iparent[0] = 0xFFFFFFFFu;
}

__device__ inline uint32_t tb_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Child `c` of a node whose children are both complete: its box and height and, for an internal node, what it holds.
struct TbKid {
    TbBox box;
    uint32_t height;
    bool internal;
    TbBox b[2];      // boxes of its own two children
    uint32_t c[2];   // their ids
    uint32_t h[2];   // and the heights of their subtrees
};
__device__ inline TbBox tb_union(const TbBox& x, const TbBox& y) {
    TbBox o;
    for (int a = 0; a < 3; ++a) { o.lo[a] = fminf(x.lo[a], y.lo[a]); o.hi[a] = fmaxf(x.hi[a], y.hi[a]); }
    return o;
}
__device__ inline float tb_area(const TbBox& x) {
    const float dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ inline uint32_t tb_max(uint32_t x, uint32_t y) { return x > y ? x : y; }
__device__ inline TbKid tb_child(uint32_t c, const uint32_t* __restrict__ vals, const TbBox* __restrict__ leaf_box, const RtNode* tmp) {
    TbKid k;
    if (c & TB_LEAF) {
        k.box = leaf_box[vals[c & ~TB_LEAF]];
        k.height = 0u;
        k.internal = false;
        k.b[0] = k.b[1] = k.box; k.c[0] = k.c[1] = c; k.h[0] = k.h[1] = 0u;
        return k;
    }
    const uint32_t* w = (const uint32_t*)(tmp + c); // written by another workgroup: read past the vector cache
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = __uint_as_float(tb_ld(w + i));
    for (int a = 0; a < 3; ++a) { k.b[0].lo[a] = v[a]; k.b[0].hi[a] = v[3 + a]; k.b[1].lo[a] = v[6 + a]; k.b[1].hi[a] = v[9 + a]; }
    k.c[0] = tb_ld(w + 12); k.c[1] = tb_ld(w + 13);
    k.h[0] = tb_ld(w + 14); k.h[1] = tb_ld(w + 15);
    k.box = tb_union(k.b[0], k.b[1]);
    k.height = 1u + tb_max(k.h[0], k.h[1]);
    k.internal = true;
    return k;
}
__device__ inline void tb_write(RtNode* dst, const TbBox& b0, const TbBox& b1, uint32_t c0, uint32_t c1, uint32_t h0, uint32_t h1) {
    RtNode nd;
    for (int a = 0; a < 3; ++a) { nd.lo0[a] = b0.lo[a]; nd.hi0[a] = b0.hi[a]; nd.lo1[a] = b1.lo[a]; nd.hi1[a] = b1.hi[a]; }
    nd.c0 = c0; nd.c1 = c1;
    nd.pad[0] = h0; nd.pad[1] = h1; // while the tree is being built: heights of the two subtrees
    *dst = nd;
}

// One lane per leaf climbs towards the root; at every node the first of the two arrivals stops and the second, which
// knows that both children are complete, writes the node (children still as Karras ids, subtree heights in pad[]).
//
// Before it does, it may improve on what the Morton order dictated: a node with children L = {a, b} and R may hand
// one grandchild up and take R in its place (L' = {R, b} beside a, or L' = {a, R} beside b; the same with the sides
// exchanged) when that makes the rebuilt child's box smaller in surface area - the tree rotations of
// A. Kensler, "Tree Rotations for Improving Bounding Volume Hierarchies" (RT 2008), applied once, bottom-up. Everything
// such a step touches lies below the node, and below a node only the lane that writes it is still at work.
__global__ void __launch_bounds__(TB_BLOCK) k_tb_refit(uint32_t n, const uint32_t* __restrict__ vals, const TbBox* __restrict__ leaf_box,
                                                       const uint32_t* __restrict__ kids, const uint32_t* __restrict__ iparent,
                                                       const uint32_t* __restrict__ lparent, uint32_t* visit, RtNode* tmp, uint32_t rotate,
                                                       uint32_t first_pass, uint32_t* status) {
    const uint32_t p = blockIdx.x * TB_BLOCK + threadIdx.x;
    if (p >= n) return;
    uint32_t cur = lparent[p];
    for (int step = 0; step < 160; ++step) { // a Karras tree over 64 + 32 key bits is at most 96 nodes tall, a rotation adds at most one level
        if (step > 0) __threadfence(); // publish the node written in the step before (a lane that starts at its leaf has written nothing yet)
        const uint32_t earlier = atomicAdd(&visit[cur], 1u);
        if (earlier == 0u) return;
        __threadfence();
        // the node's children: from the hierarchy on the first pass, from the node itself (as the pass before left it) later
        const uint32_t id[2] = {first_pass ? kids[2u * cur] : tmp[cur].c0, first_pass ? kids[2u * cur + 1u] : tmp[cur].c1};
        TbKid k[2] = {tb_child(id[0], vals, leaf_box, tmp), tb_child(id[1], vals, leaf_box, tmp)};
        // choice = 2 * side + which: the child on `side` keeps its grandchild `1 - which`, hands `which` up, takes the other child in
        // (the surface-area cost of a tree is the sum of its nodes' areas: a rotation changes one term of it, the rebuilt child's)
        float best = 0.0f;
        int choice = -1;
        if (rotate)
            for (int side = 0; side < 2; ++side)
                if (k[side].internal) {
                    const float before = tb_area(k[side].box);
                    for (int which = 0; which < 2; ++which) {
                        const float gain = before - tb_area(tb_union(k[1 - side].box, k[side].b[1 - which]));
                        if (gain > best && gain > 1e-3f * before) { best = gain; choice = 2 * side + which; }
                    }
                }
        if (choice < 0) {
            tb_write(tmp + cur, k[0].box, k[1].box, id[0], id[1], k[0].height, k[1].height);
        } else {
            const int side = choice >> 1, which = choice & 1;
            const TbKid& in = k[side];      // the child that is rebuilt
            const TbKid& other = k[1 - side]; // the child that moves down into it
            const TbBox merged = tb_union(other.box, in.b[1 - which]);
            const uint32_t merged_h = 1u + tb_max(other.height, in.h[1 - which]);
            // the rebuilt child: the moved-in subtree takes the place of the grandchild that went up
            if (which == 0) tb_write(tmp + id[side], other.box, in.b[1], id[1 - side], in.c[1], other.height, in.h[1]);
            else tb_write(tmp + id[side], in.b[0], other.box, in.c[0], id[1 - side], in.h[0], other.height);
            // this node: the rebuilt child stays on its side, the grandchild takes the other side
            if (side == 0) tb_write(tmp + cur, merged, in.b[which], id[0], in.c[which], merged_h, in.h[which]);
            else tb_write(tmp + cur, in.b[which], merged, in.c[which], id[1], in.h[which], merged_h);
        }
        if (cur == 0u) {
            const uint32_t* w = (const uint32_t*)(tmp + cur);
            status[ST_HEIGHT] = 1u + tb_max(w[14], w[15]);
            return;
        }
        cur = iparent[cur];
    }
    atomicOr(&status[ST_ERROR], 4u);
}

// Parents as the rotations left them, for another climb.
__global__ void __launch_bounds__(TB_BLOCK) k_tb_parents(uint32_t n_internal, const RtNode* __restrict__ tmp, uint32_t* __restrict__ iparent,
                                                         uint32_t* __restrict__ lparent) {
    const uint32_t i = blockIdx.x * TB_BLOCK + threadIdx.x;
    if (i >= n_internal) return;
    const uint32_t c[2] = {tmp[i].c0, tmp[i].c1};
    for (int k = 0; k < 2; ++k) {
        if (c[k] & TB_LEAF) lparent[c[k] & ~TB_LEAF] = i;
        else iparent[c[k]] = i;
    }
}

// One workgroup: the first `top` nodes in breadth-first order from the root get positions 0..top-1; the nodes that sat
// at those positions and are not among them take the positions the others left.
__global__ void __launch_bounds__(1024) k_tb_top(uint32_t n_internal, uint32_t top, const RtNode* __restrict__ tmp, uint32_t* order,
                                                 uint32_t* clist, uint32_t* newpos, const uint32_t* __restrict__ slot_sorted, uint32_t root,
                                                 uint32_t n_nodes, uint32_t* status) {
    __shared__ uint32_t cnt, ccnt;
    __shared__ uint8_t low_in_top[TB_TOP];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < TB_TOP; i += 1024u) low_in_top[i] = 0;
    if (tid == 0u) {
        order[0] = 0u;
        cnt = 1u;
        ccnt = 0u;
        if (slot_sorted[0] != root) atomicOr(&status[ST_ERROR], 2u); // the root must stay where the rest of the scene points
        if (slot_sorted[n_internal - 1u] >= n_nodes) atomicOr(&status[ST_ERROR], 8u); // a record no leaf's walk came past
    }
    __syncthreads();
    uint32_t a = 0u, b = 1u;
    while (b < top && a < b) {
        for (uint32_t idx = a + tid; idx < b; idx += 1024u) {
            const uint32_t node = order[idx];
            const uint32_t c[2] = {tmp[node].c0, tmp[node].c1};
            for (int k = 0; k < 2; ++k)
                if (!(c[k] & TB_LEAF)) {
                    const uint32_t pos = atomicAdd(&cnt, 1u);
                    if (pos < top) order[pos] = c[k];
                }
        }
        __syncthreads();
        a = b;
        b = cnt < top ? cnt : top;
        __syncthreads();
    }
    const uint32_t tc = b;
    for (uint32_t q = tid; q < tc; q += 1024u) {
        const uint32_t node = order[q];
        newpos[node] = q;
        if (node < tc) low_in_top[node] = 1;
        else clist[atomicAdd(&ccnt, 1u)] = node;
    }
    __syncthreads();
    if (tid == 0u) {
        uint32_t r = 0u;
        for (uint32_t i = 0u; i < tc && i < n_internal; ++i)
            if (!low_in_top[i] && r < ccnt) newpos[i] = clist[r++];
    }
}

__global__ void __launch_bounds__(TB_BLOCK) k_tb_commit(uint32_t n_internal, const RtNode* __restrict__ tmp, const uint32_t* __restrict__ newpos,
                                                        const uint32_t* __restrict__ slot_sorted, const uint32_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ leaf_ref, RtNode* __restrict__ nodes, uint32_t n_nodes) {
    const uint32_t i = blockIdx.x * TB_BLOCK + threadIdx.x;
    if (i >= n_internal) return;
    RtNode nd = tmp[i];
    const uint32_t c[2] = {nd.c0, nd.c1};
    uint32_t out[2];
    for (int k = 0; k < 2; ++k)
        out[k] = (c[k] & TB_LEAF) ? leaf_ref[vals[c[k] & ~TB_LEAF]] : RT_MAKE_REF(RT_REF_NODE, slot_sorted[newpos[c[k]]]);
    nd.c0 = out[0]; nd.c1 = out[1];
    nd.pad[0] = 0u; nd.pad[1] = 0u;
    const uint32_t dst = slot_sorted[newpos[i]];
    if (dst < n_nodes) nodes[dst] = nd;
}

struct Carve { // hands out 256-byte aligned pieces of one allocation
    size_t total = 0;
    size_t take(size_t bytes) { const size_t at = total; total = (total + bytes + 255u) & ~(size_t)255u; return at; }
};

} // namespace

int rt_tree_build_device(hipStream_t stream, RtNode* d_nodes, size_t n_nodes, uint32_t root, uint32_t n, uint32_t max_height,
                         RtTreeBuildResult* result, std::string& err) {
    if (result) *result = RtTreeBuildResult();
    if (!d_nodes || root >= n_nodes) { err = "tree build: root outside the node array"; return RT_ERR_INVALID; }
    if (n < 2u) return RT_OK; // a single leaf hangs directly under its node
    if (n > TB_MAX_LEAVES || (size_t)n - 1u > n_nodes) { err = "tree build: leaf count out of range"; return RT_ERR_INVALID; }
    const uint32_t ni = n - 1u;
    const uint32_t top = ni < TB_TOP ? ni : TB_TOP;
    const uint32_t rotate = getenv("RT_TREE_NO_ROTATIONS") ? 0u : 1u; // (for A/B measurements)
    uint32_t rotation_passes = 2u;
    if (const char* e = getenv("RT_TREE_ROTATION_PASSES")) rotation_passes = (uint32_t)std::max(1, std::min(8, atoi(e)));

    size_t cub_pairs = 0, cub_keys = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, cub_pairs, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                                      (uint32_t*)nullptr, (int)n, 0, 64, stream);
    if (e == hipSuccess)
        e = hipcub::DeviceRadixSort::SortKeys(nullptr, cub_keys, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)ni, 0, 32, stream);
    if (e != hipSuccess) { (void)hipGetLastError(); err = std::string("tree build: sort set-up: ") + hipGetErrorString(e); return RT_ERR_NO_DEVICE; }

    Carve cv;
    const size_t o_status = cv.take(ST_WORDS * 4), o_leaf_ref = cv.take((size_t)n * 4), o_leaf_box = cv.take((size_t)n * sizeof(TbBox)),
                 o_slot = cv.take((size_t)ni * 4), o_slot_sorted = cv.take((size_t)ni * 4), o_keys = cv.take((size_t)n * 8),
                 o_keys_sorted = cv.take((size_t)n * 8), o_vals = cv.take((size_t)n * 4), o_vals_sorted = cv.take((size_t)n * 4),
                 o_kids = cv.take((size_t)ni * 8), o_iparent = cv.take((size_t)ni * 4), o_lparent = cv.take((size_t)n * 4),
                 o_visit = cv.take((size_t)ni * 4), o_tmp = cv.take((size_t)ni * sizeof(RtNode)), o_newpos = cv.take((size_t)ni * 4),
                 o_order = cv.take((size_t)TB_TOP * 4), o_clist = cv.take((size_t)TB_TOP * 4),
                 o_cub = cv.take(cub_pairs > cub_keys ? cub_pairs : cub_keys);
    uint8_t* base = nullptr;
    e = hipMalloc((void**)&base, cv.total);
    if (e != hipSuccess) { (void)hipGetLastError(); err = std::string("tree build: hipMalloc: ") + hipGetErrorString(e); return RT_ERR_OOM; }
    uint32_t* status = (uint32_t*)(base + o_status);
    uint32_t* leaf_ref = (uint32_t*)(base + o_leaf_ref);
    TbBox* leaf_box = (TbBox*)(base + o_leaf_box);
    uint32_t* slot = (uint32_t*)(base + o_slot);
    uint32_t* slot_sorted = (uint32_t*)(base + o_slot_sorted);
    uint64_t* keys = (uint64_t*)(base + o_keys);
    uint64_t* keys_sorted = (uint64_t*)(base + o_keys_sorted);
    uint32_t* vals = (uint32_t*)(base + o_vals);
    uint32_t* vals_sorted = (uint32_t*)(base + o_vals_sorted);
    uint32_t* kids = (uint32_t*)(base + o_kids);
    uint32_t* iparent = (uint32_t*)(base + o_iparent);
    uint32_t* lparent = (uint32_t*)(base + o_lparent);
    uint32_t* visit = (uint32_t*)(base + o_visit);
    RtNode* tmp = (RtNode*)(base + o_tmp);
    uint32_t* newpos = (uint32_t*)(base + o_newpos);
    uint32_t* order = (uint32_t*)(base + o_order);
    uint32_t* clist = (uint32_t*)(base + o_clist);
    void* cub_tmp = base + o_cub;

    hipEvent_t ev[6];
    for (hipEvent_t& x : ev) hipEventCreate(&x);
    const uint32_t grid_n = (n + TB_BLOCK - 1u) / TB_BLOCK, grid_i = (ni + TB_BLOCK - 1u) / TB_BLOCK;
    int rc = RT_OK;
    uint32_t h_status[ST_WORDS] = {0};
    auto fail = [&](int code, const char* what, hipError_t he) {
        (void)hipGetLastError();
        err = std::string("tree build: ") + what;
        if (he != hipSuccess) { err += ": "; err += hipGetErrorString(he); }
        rc = code;
    };
    do {
        e = hipMemsetAsync(status, 0, ST_WORDS * 4, stream);
        if (e == hipSuccess) e = hipMemsetAsync(status + ST_CMIN, 0xFF, 3 * 4, stream);
        if (e == hipSuccess) e = hipMemsetAsync(visit, 0, (size_t)ni * 4, stream);
        if (e == hipSuccess) e = hipMemsetAsync(slot, 0xFF, (size_t)ni * 4, stream);
        if (e != hipSuccess) { fail(RT_ERR_NO_DEVICE, "memset", e); break; }
        hipEventRecord(ev[0], stream);
        hipLaunchKernelGGL(k_tb_collect, dim3(grid_n), dim3(TB_BLOCK), 0, stream, (const RtNode*)d_nodes, (uint32_t)n_nodes, root, n, leaf_ref,
                           leaf_box, slot, status);
        hipLaunchKernelGGL(k_tb_bounds, dim3(grid_n), dim3(TB_BLOCK), 0, stream, (const TbBox*)leaf_box, n, status);
        hipEventRecord(ev[1], stream);
        // a malformed stand-in must not reach the sorts with unwritten records
        e = hipMemcpyAsync(h_status, status, sizeof h_status, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { fail(RT_ERR_NO_DEVICE, "collect", e); break; }
        if (h_status[ST_ERROR]) { fail(RT_ERR_INVALID, "the blob's stand-in tree does not have the shape the compiler emits", hipSuccess); break; }
        size_t bytes = cub_keys;
        e = hipcub::DeviceRadixSort::SortKeys(cub_tmp, bytes, (const uint32_t*)slot, slot_sorted, (int)ni, 0, 32, stream);
        if (e != hipSuccess) { fail(RT_ERR_NO_DEVICE, "radix sort", e); break; }
        // Full-resolution codes first; if that tree is taller than the traversal stack allows, once more with 10 bits per axis.
        const uint32_t attempts[2] = {21u, 10u};
        for (int at = 0; at < 2 && rc == RT_OK; ++at) {
            hipLaunchKernelGGL(k_tb_morton, dim3(grid_n), dim3(TB_BLOCK), 0, stream, (const TbBox*)leaf_box, n, (const uint32_t*)status,
                               attempts[at], keys, vals);
            bytes = cub_pairs;
            e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, bytes, (const uint64_t*)keys, keys_sorted, (const uint32_t*)vals, vals_sorted, (int)n, 0, 64, stream);
            if (e == hipSuccess) e = hipMemsetAsync(visit, 0, (size_t)ni * 4, stream);
            if (e != hipSuccess) { fail(RT_ERR_NO_DEVICE, "radix sort", e); break; }
            hipEventRecord(ev[2], stream);
            hipLaunchKernelGGL(k_tb_hierarchy, dim3(grid_i), dim3(TB_BLOCK), 0, stream, (const uint64_t*)keys_sorted, n, kids, iparent, lparent, newpos);
            hipEventRecord(ev[3], stream);
            for (uint32_t pass = 0; pass < (rotate ? rotation_passes : 1u); ++pass) {
                if (pass > 0u) {
                    hipLaunchKernelGGL(k_tb_parents, dim3(grid_i), dim3(TB_BLOCK), 0, stream, ni, (const RtNode*)tmp, iparent, lparent);
                    (void)hipMemsetAsync(visit, 0, (size_t)ni * 4, stream);
                }
                hipLaunchKernelGGL(k_tb_refit, dim3(grid_n), dim3(TB_BLOCK), 0, stream, n, (const uint32_t*)vals_sorted, (const TbBox*)leaf_box,
                                   (const uint32_t*)kids, (const uint32_t*)iparent, (const uint32_t*)lparent, visit, tmp, rotate,
                                   pass == 0u ? 1u : 0u, status);
            }
            hipLaunchKernelGGL(k_tb_top, dim3(1), dim3(1024), 0, stream, ni, top, (const RtNode*)tmp, order, clist, newpos,
                               (const uint32_t*)slot_sorted, root, (uint32_t)n_nodes, status);
            hipEventRecord(ev[4], stream);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(h_status, status, sizeof h_status, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { fail(RT_ERR_NO_DEVICE, "hierarchy", e); break; }
            if (h_status[ST_ERROR]) { fail(RT_ERR_INVALID, "inconsistent hierarchy (the stand-in's root is not its lowest record, a record was never visited, or a climb did not end)", hipSuccess); break; }
            if (result) { result->leaves = n; result->height = h_status[ST_HEIGHT]; result->key_bits = 3u * attempts[at]; }
            if (h_status[ST_HEIGHT] <= max_height) break;
            if (at == 1) {
                char msg[160];
                snprintf(msg, sizeof msg, "the tree is %u nodes tall, the traversal stack allows %u: the stand-in tree is kept", h_status[ST_HEIGHT], max_height);
                fail(RT_ERR_UNSUPPORTED, msg, hipSuccess);
            }
        }
        if (rc != RT_OK) break;
        hipLaunchKernelGGL(k_tb_commit, dim3(grid_i), dim3(TB_BLOCK), 0, stream, ni, (const RtNode*)tmp, (const uint32_t*)newpos,
                           (const uint32_t*)slot_sorted, (const uint32_t*)vals_sorted, (const uint32_t*)leaf_ref, d_nodes, (uint32_t)n_nodes);
        hipEventRecord(ev[5], stream);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { fail(RT_ERR_NO_DEVICE, "commit", e); break; }
        if (result) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) result->ms_collect = ms;
            if (hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) result->ms_sort = ms;
            if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) result->ms_hierarchy = ms;
            if (hipEventElapsedTime(&ms, ev[3], ev[4]) == hipSuccess) result->ms_refit = ms;
            if (hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess) result->ms_commit = ms;
        }
    } while (false);
    if (rc != RT_OK) hipStreamSynchronize(stream);
    for (hipEvent_t& x : ev) hipEventDestroy(x);
    hipFree(base);
    return rc;
}
