// Device-side construction of the world's tree (see rt_tree_build.hip). Internal to librt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "rt_types.h"

struct RtTreeBuildResult {
    uint32_t leaves = 0;
    uint32_t height = 0;   // longest root-to-leaf chain of nodes in the tree that was built
    uint32_t key_bits = 0; // bits of the Morton codes it was built from (63, or 30 when 63 gave too tall a tree)
    double ms_collect = 0, ms_sort = 0, ms_hierarchy = 0, ms_refit = 0, ms_commit = 0;
};

// Rebuilds, in place, the balanced stand-in tree whose root is node `root` of `d_nodes` and which holds `n_leaves` leaves.
// `max_height`: the tallest tree the caller can walk; a taller result is not written (the stand-in stays) and
// RT_ERR_UNSUPPORTED is returned. Returns RT_OK or an RT_ERR_* code with `err` set. Blocks until the nodes are written.
int rt_tree_build_device(hipStream_t stream, RtNode* d_nodes, size_t n_nodes, uint32_t root, uint32_t n_leaves, uint32_t max_height,
                         RtTreeBuildResult* result, std::string& err);
