// Internal declarations of the host scene builder (see rt_builder.cpp).
#pragma once
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "../../include/rt.h"
#include "rt_math.h"
#include "rt_types.h"

struct AABB { // aabb.rs:7-10
    double lo[3], hi[3];
};

// Sequential uniform stream for scene construction (what the reference takes from thread_rng()).
struct SeqRng {
    uint64_t seed = 0, counter = 0;
    uint32_t stream = 0;
    bool have_spare = false;
    double spare = 0.0;
    double next();                      // [0,1)
    double range(double lo, double hi); // random_double, rtweekend.rs:19-22
    int irange(int lo, int hi);         // random_int,    rtweekend.rs:24-27
};

enum ObjKind { OBJ_SPHERE, OBJ_MSPHERE, OBJ_XYRECT, OBJ_XZRECT, OBJ_YZRECT, OBJ_BOX, OBJ_TRANSLATE, OBJ_ROTATE_Y,
               OBJ_MEDIUM, OBJ_LIST, OBJ_BVH };

struct BNode { // BvhNode (bvh.rs:8-12): each side is another node or an object handle
    AABB box;
    int left_node = -1, right_node = -1, left_obj = -1, right_obj = -1;
    int axis = 0; // the axis drawn at bvh.rs:22: children are ordered by their boxes' minimum on it
};

struct ObjRec {
    ObjKind kind = OBJ_SPHERE;
    int mat = -1;
    int child = -1;            // Translate/RotateY child, ConstantMedium boundary, BvhNode source list
    double d[10] = {0};        // constructor arguments (layout per kind, see rt_builder.cpp)
    std::vector<int> members;  // HittableList::objects
    std::vector<BNode> tree;   // BvhNode tree, [0] = root
    AABB box = {{0, 0, 0}, {0, 0, 0}}; // RotateY::bbox / BvhNode::_box
    bool hasbox = true;        // RotateY::hasbox
    int medium_index = -1;
    // bounding_box() really contains the object. False for RotateY (keeps the child's unrotated box, hittable.rs:115-143)
    // and HittableList (last member's box only, hittable_list.rs:55-75) and whatever is built on them.
    bool exact_bounds = true;
    // hit(r, t_min, t_max) reports its nearest hit below t_max whatever t_max is. False for what can report a t that is
    // NaN or lies outside (t_min, t_max): the answer of a list over it depends on the order of the list.
    bool monotone = true;
    bool nan_hits = false;     // reports hits at t = NaN: a list compares everything it tests afterwards against NaN
};
struct MatRec {
    uint32_t type = 0;
    int tex = -1;
    double p[4] = {0, 0, 0, 0};
};
struct TexRec {
    uint32_t type = 0;
    int a = -1, b = -1;
    double p[3] = {0, 0, 0};
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> texels;
    std::vector<double> perlin_vec;
    std::vector<int32_t> perlin_perm;
};

// Camera::new (camera.rs:30-53): fills the derived frame from the constructor arguments.
// RT_ERR_PANIC where the reference would panic (zero-length view or up vector, time0 >= time1).
int rt_camera_derive(const rt_camera* c, RtCameraDev* out);

struct rt_builder {
    struct Image { uint32_t width = 0, height = 0; std::vector<uint8_t> rgb; };
    struct Flat;

    SeqRng rng;
    uint32_t flags = 0;
    int n_media = 0;
    std::vector<ObjRec> objs;
    std::vector<MatRec> mats;
    std::vector<TexRec> texs;
    std::map<std::string, Image> images;
    std::vector<uint8_t> log; // constructor log as compact records (rt_builder.cpp: log_fmt / log_text)
    std::string axis_log; // split axes of the BVH being built, one digit per BvhNode::new call
    std::string error;
    bool seen_nan_hits = false; // an object that reports hits at t = NaN has been constructed
    bool literal = false; // set by compile(): every list and BvhNode keeps the reference's order of tests
    double valid_lo = -std::numeric_limits<double>::infinity(), valid_hi = std::numeric_limits<double>::infinity(); // set by compile()
    bool has_moving(int h) const; // a moving sphere somewhere below h
    bool has_medium(int h) const; // a ConstantMedium somewhere below h, or h itself
    void limit_time(const ObjRec& bvh);

    int fail(int code, const char* fmt, ...);
    bool valid_obj(int h) const;
    bool valid_mat(int h) const;
    bool valid_tex(int h) const;
    int add_obj(ObjRec o);
    bool bounding_box(int h, double t0, double t1, AABB& out) const;
    struct TimeSpan { double lo, hi; };
    bool world_bound(int h, AABB& out, TimeSpan* span = nullptr) const;
    bool world_bound_any(int h, AABB& out, TimeSpan* span) const; // may hold NaN
    int build_bvh_node(std::vector<BNode>& tree, std::vector<int>& objects, const std::vector<double>& keys, size_t start,
                       size_t end, double t0, double t1);
    uint32_t emit(Flat& f, int h, int parent, uint32_t depth, uint32_t& need);
    uint32_t emit_bvh(Flat& f, const ObjRec& o, int parent, uint32_t depth, uint32_t& need);
    uint32_t emit_replay(Flat& f, const ObjRec& o, int node, int parent, uint32_t depth, uint32_t& need);
    int compile(int world, void** blob, size_t* nbytes);
};
