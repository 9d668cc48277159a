// Flat scene blob: the plain-old-data records the scene compiler emits and the HIP kernels read.
//
// The blob is one relocatable byte range: an RtBlobHeader followed by 256-byte-aligned sections.
// Every cross reference is a 32-bit tagged "ref" (type in the top 4 bits, index into that type's
// table in the low 28), never a pointer, so the same bytes can be uploaded to any device address.
//
// Geometry is IEEE f64 like the reference (raytracer/src/vec3.rs:5-9); only BVH node bounds are
// stored as f32, rounded outward so that the f32 box always contains the f64 box it stands for.
#pragma once
#include <stdint.h>

#define RT_BLOB_MAGIC 0x31425452u /* "RTB1" little endian */
#define RT_BLOB_VERSION 15u

// ---- tagged references -------------------------------------------------------------------------
enum RtRefType : uint32_t {
    RT_REF_NONE = 0,     // list terminator / absent child
    RT_REF_NODE = 1,     // BvhNode            (raytracer/src/bvh.rs:8-12)
    RT_REF_SPHERE = 2,   // Sphere             (raytracer/src/sphere.rs:9-13)
    RT_REF_MSPHERE = 3,  // MovingSphere       (raytracer/src/moving_sphere.rs:9-16)
    RT_REF_RECT = 4,     // XYRect/XZRect/YZRect (raytracer/src/arrect.rs)
    RT_REF_BOX = 5,      // _Box               (raytracer/src/_box.rs:8-12), six faces resolved arithmetically
    RT_REF_MEDIUM = 6,   // ConstantMedium     (raytracer/src/constant_medium.rs:8-12)
    RT_REF_INST = 7,     // Translate / RotateY (raytracer/src/hittable.rs:54-57,95-101)
    RT_REF_LIST = 8,     // HittableList: index of its first entry in the NONE-terminated entry array
    RT_REF_POP = 9,      // traversal-internal marker: leave instance `index`
    RT_REF_ILEAF = 10    // one Sphere / MovingSphere / rect / _Box seen through Translate / RotateY objects, everything in the
                         // ref itself: aux = the primitive's type - RT_REF_SPHERE, index = chain[25:20] | primitive[19:0]
                         // (RtChain record, record of the primitive's own table). A traversal needs no dependent reads to
                         // test it. The compiler emits it wherever chain and primitive numbers fit, RT_REF_INST otherwise.
    ,
    RT_REF_STEPS = 11    // on a lock-step list only: index = n, and the 2 n words behind it are a small reference-order tree
                         // (bvh.rs:112-122) laid out as the steps of its depth-first walk, (control, operand) each: control =
                         // depth | RT_STEP_GATE for a box test (operand = 1 + index of its f64 bounds in exact_boxes; lanes that
                         // are inside at depth - 1 and pass are inside at depth), control = depth for an object (operand = its
                         // reference; tested by the lanes inside at depth - 1). The walk starts with every lane inside at depth 0.
};
// layout: type[31:28] | aux[27:26] (zero but in RT_REF_ILEAF) | index[25:0]
#define RT_REF_SHIFT 28u
#define RT_REF_INDEX_MASK 0x03FFFFFFu
#define RT_REF_AUX_SHIFT 26u
#define RT_MAKE_REF(type, index) ((((uint32_t)(type)) << RT_REF_SHIFT) | ((uint32_t)(index)&RT_REF_INDEX_MASK))
#define RT_REF_TYPE(ref) ((uint32_t)(ref) >> RT_REF_SHIFT)
#define RT_REF_INDEX(ref) ((uint32_t)(ref)&RT_REF_INDEX_MASK)
#define RT_REF_AUX(ref) (((uint32_t)(ref) >> RT_REF_AUX_SHIFT) & 3u)
#define RT_REF_STRIP_AUX(ref) ((uint32_t)(ref) & ~(3u << RT_REF_AUX_SHIFT))

#define RT_STEP_GATE 0x100u
#define RT_STEP_DEPTH_MASK 63u
#define RT_STEP_MAX_DEPTH 31u
#define RT_ILEAF_CHAIN_SHIFT 20u
#define RT_ILEAF_PRIM_MASK 0xFFFFFu
#define RT_ILEAF_MAX_CHAINS 64u
#define RT_MAKE_ILEAF(chain, prim_ref) \
    ((((uint32_t)RT_REF_ILEAF) << RT_REF_SHIFT) | ((RT_REF_TYPE(prim_ref) - (uint32_t)RT_REF_SPHERE) << RT_REF_AUX_SHIFT) | ((uint32_t)(chain) << RT_ILEAF_CHAIN_SHIFT) | RT_REF_INDEX(prim_ref))
#define RT_ILEAF_PRIM(ref) RT_MAKE_REF((uint32_t)RT_REF_SPHERE + RT_REF_AUX(ref), (uint32_t)(ref)&RT_ILEAF_PRIM_MASK)
#define RT_ILEAF_CHAIN(ref) (RT_REF_INDEX(ref) >> RT_ILEAF_CHAIN_SHIFT)

#define RT_NO_PARENT (-1)
#define RT_MAX_CHAIN 4   /* deepest Translate/RotateY nesting above any primitive */
#define RT_STACK_CAP 39  /* per-lane traversal stack entries: 39 x 1024 lanes x 4 B + the queue counter fill the 160 KB of LDS a workgroup can have */

// ---- records -------------------------------------------------------------------------------------
// 64-byte BVH node: the boxes of BOTH children and their refs, so that one fetch decides both descents and the
// nearer child can be entered first. A child ref is another node or the object the reference would store in that
// Arc (bvh.rs:8-12). An absent child has ref NONE and the empty box (lo = +inf, hi = -inf), which no ray hits; a
// child that must be entered without a box test (an object directly under a reference-order node, whose hit() has
// no box of its own) has the infinite box. The nearer child is entered first unless the node is flagged ordered. Nodes of one tree are stored in depth-first pre-order.
struct RtNode {
    float lo0[3], hi0[3]; // child 0
    float lo1[3], hi1[3]; // child 1
    uint32_t c0, c1;
    uint32_t pad[2];      // pad[0]: RT_NODE_* flags; pad[1]: for an RT_NODE_EXACT node whose child-0 box is finite, 1 + the index of that box's f64 bounds in the exact_boxes table (0: none)
};
// The f64 bounds of the one real box an RT_NODE_EXACT node carries (child 0's; its child 1 is absent or has the infinite
// box): the verdict of these box tests is part of what the reference computes, so they run on the bounds the reference
// holds (aabb.rs:6-9), not on the outward-rounded f32 copy in the node, which a grazing ray could enter and the f64 box not.
struct RtExactBox { double lo[3], hi[3]; };
#define RT_NODE_ORDERED 1u /* child 0 is entered (and finished) before child 1 whatever their distances */
#define RT_NODE_EXACT 2u   /* what the reference finds depends on the verdict of these box tests: they are evaluated in the reference's f64 arithmetic (aabb.rs:35-75); boxes of other nodes only cull and are tested conservatively in f32 */

// Common per-primitive bookkeeping, one per geometry record (same index).
struct RtMeta {
    uint32_t material;   // index into materials
    int32_t parent_inst; // nearest enclosing Translate/RotateY record, or RT_NO_PARENT
    uint32_t src_id;     // handle the host API returned for this object == prim_id reported in hits
    uint32_t pad;
};

struct RtSphere { double cx, cy, cz, r; };
struct RtMSphere { double c0[3]; double c1[3]; double t0, t1, r, pad; };
// axis: 0 = XYRect (a=x, b=y, k on z); 1 = XZRect (a=x, b=z, k on y); 2 = YZRect (a=y, b=z, k on x)
struct RtRect { double a0, a1, b0, b1, k; uint32_t axis; uint32_t pad; };
struct RtBox { double p0[3]; double p1[3]; };
struct RtMedium {
    double neg_inv_density; // constant_medium.rs:18
    uint32_t boundary;      // ref: primitive, box, or instance chain ending in one
    uint32_t material;      // the Isotropic phase function
    int32_t parent_inst;
    uint32_t src_id;
    uint32_t medium_index;  // selects the Philox stream of this medium's free-flight draw
    uint32_t pad;
};
enum RtInstOp : uint32_t { RT_OP_TRANSLATE = 0, RT_OP_ROTATE_Y = 1 };
#define RT_OP_MASK 0xFFu    /* RtInst::op: the operation ... */
#define RT_INST_LEAF 0x100u /* ... and this flag: below the record are only further Translate/RotateY records and then one primitive or medium */
struct RtInst {
    double a, b, c;      // translate: offset xyz; rotate_y: a = sin(theta), b = cos(theta)
    uint32_t op;
    uint32_t child;      // ref
    int32_t parent_inst;
    uint32_t src_id;
};

// The Translate / RotateY objects above an RT_REF_ILEAF primitive, outermost first, by value: what RtInst records hold,
// without their links, so that every primitive under the same objects shares one record.
struct RtChainOp { double a, b, c; uint32_t op; uint32_t pad; };
struct RtChain { uint32_t n; uint32_t pad; RtChainOp op[RT_MAX_CHAIN]; };

enum RtMaterialType : uint32_t {
    RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4, RT_MAT_COUNT = 5
};
// Shading work is queued by what it has to run: material type, and whether the material's texture is a plain colour
// (one table read) or anything else (Perlin noise, image, checker: an order of magnitude more code and registers).
enum RtBucket : uint32_t {
    RT_BUCKET_LAMBERTIAN_SOLID = 0, RT_BUCKET_LAMBERTIAN_TEX = 1, RT_BUCKET_METAL = 2, RT_BUCKET_DIELECTRIC = 3,
    RT_BUCKET_LIGHT_SOLID = 4, RT_BUCKET_LIGHT_TEX = 5, RT_BUCKET_ISOTROPIC_SOLID = 6, RT_BUCKET_ISOTROPIC_TEX = 7,
    RT_BUCKET_LAMBERTIAN_NOISE = 8, // Lambertian over a NoiseTexture: needs no u, v - the kernel that computes them (acos, atan2) needs 80 more registers
    RT_BUCKET_COUNT = 9
};
struct RtMaterial {
    uint32_t type;
    uint32_t tex;      // Lambertian/Isotropic albedo, DiffuseLight emit
    uint32_t tex_type; // type of that texture (RT_TEX_SOLID for materials without one): shading decides where a hit goes without reading the texture
    uint32_t pad;
    double p[4];       // Metal: albedo rgb + fuzz; Dielectric: p[0] = ir; DiffuseLight: p[0] = k; Lambertian/Isotropic/DiffuseLight over a SolidColor: p[1..3] = its rgb
};

enum RtTextureType : uint32_t { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_NOISE = 2, RT_TEX_IMAGE = 3 };
struct RtTexture {
    uint32_t type;
    uint32_t a;       // checker: even texture; noise: perlin table index; image: texel byte offset
    uint32_t b;       // checker: odd texture;  image: width
    uint32_t c;       // image: height
    double p[3];      // solid: rgb; noise: p[0] = scale
};

#define RT_PERLIN_POINTS 256
// one Perlin table = 256 gradient vectors (3 doubles each) + perm_x, perm_y, perm_z (256 int32 each)

struct RtSection { uint64_t offset; uint64_t count; };

struct RtBlobHeader {
    uint32_t magic, version;
    uint64_t total_bytes;
    uint32_t world_flat_any_time; // LIST ref (or NONE): world_flat plus the moving spheres that sit in the tree under bounds valid for [shutter0, shutter1] only; used for rays whose time may lie outside that interval
    uint32_t world_flat;     // LIST ref (or NONE): the world entries (primitives or media, possibly under Translate/RotateY) that every lane tests in lock step: unbounded or huge ones
    uint32_t world_tree;     // NODE ref (or NONE): everything else of the world under one tree; walked per lane with a stack
    uint32_t n_media;        // number of ConstantMedium objects == free-flight draws per segment
    uint32_t stack_need;     // worst-case traversal stack depth for this scene
    uint32_t max_chain;      // deepest instance nesting
    uint32_t flags;          // RT_BLOB_FLAG_*
    uint32_t n_objects;      // host handles issued (prim_id range)
    double shutter0, shutter1; // ray times for which world_flat suffices: the moving spheres listed only in world_flat_any_time keep to their tree bounds (-inf, +inf when there are none)
    double valid0, valid1;     // ray times the blob may be traced with at all: those the BvhNodes holding moving spheres were built for (bvh.rs:15, time0..time1); (-inf, +inf) without such nodes or with RT_BLOB_FLAG_ANY_TIME
    double node_extent;        // largest finite |coordinate| of any node box (0 without nodes): scales the error bound of the f32 box test
    uint64_t n_exact_nodes;    // nodes flagged RT_NODE_EXACT
    RtSection nodes, spheres, sphere_meta, mspheres, msphere_meta, rects, rect_meta, boxes, box_meta,
        media, insts, list_entries, materials, textures, perlin_vec, perlin_perm, texels, exact_boxes, chains;
    // A world tree left for the device to build (RT_BLOB_FLAG_DEVICE_TREE). The compiler has put its leaves, in the order
    // it met them, under a balanced stand-in tree: a node over leaves [s, e) has child 0 over [s, m) and child 1 over
    // [m, e), m = s + (e - s) / 2, down to single leaves. The stand-in is valid (every box bounds what is below it) but
    // groups leaves that have nothing to do with each other; the device build re-arranges the same leaves in the same
    // node records.
    uint32_t device_tree_root;   // NODE ref of the stand-in's root, RT_REF_NONE when there is nothing to build
    uint32_t device_tree_leaves; // number of leaves below it
    uint32_t device_tree_leaf_need; // most stack entries anything below one of its leaves needs
    uint32_t device_tree_need_other; // stack entries the world's entries outside the tree need
    uint32_t device_tree_need_add;   // nodes chained in front of the tree, one stack entry each
    uint32_t pad_tail;
};
#define RT_BLOB_FLAG_BOOK_BBOX 1u      /* RotateY stores its rotated box (book behaviour) instead of the child's */
#define RT_BLOB_FLAG_REFERENCE_TREE 2u /* every BvhNode keeps the reference's topology (random-axis median split) */
#define RT_BLOB_FLAG_DEVICE_TREE 8u    /* the world's tree is a stand-in, to be rebuilt on the device after upload (rt_scene_build_tree_device) */
#define RT_BLOB_FLAG_ANY_TIME 4u       /* rays may carry any time: a BvhNode over moving spheres is replayed in the reference's form instead of being rebuilt */

// ---- render-time records -------------------------------------------------------------------------
// Camera in the derived form the reference's Camera struct holds (camera.rs:7-18).
struct RtCameraDev {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius, time0, time1;
};

struct RtHitLite {   // what the extend kernel writes per path: 16 bytes
    double t;
    uint32_t ref;    // RT_REF_NONE = miss; else SPHERE/MSPHERE/RECT/BOX/MEDIUM ref
    uint32_t aux;    // box: face 0..5 in _box.rs:22-71 order
};

// Philox stream ids ("consumer" field of the counter)
enum RtStream : uint32_t {
    RT_STREAM_JITTER = 0, RT_STREAM_LENS = 1, RT_STREAM_TIME = 2, RT_STREAM_SCATTER = 3,
    RT_STREAM_DIELECTRIC = 4, RT_STREAM_MEDIUM0 = 16
};
#define RT_REJECTION_CAP 4096u /* attempts before a rejection sampler gives up (probability < 1e-1000) */
