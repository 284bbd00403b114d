// kmg_octree.h -- Algorithm::Octree, the reference's CPU colour quantiser (core/src/octree.rs:1-242,
// core/src/operations.rs:90-97).  It is a host algorithm in the reference as well (it runs on at
// most 128x128 pixels, core/src/lib.rs:288-331); this is a C++ restatement with the same merge order:
// leaves ordered by (child_count, pixel_count >> level, node id), smallest merged into its parent
// until at most `color_count` remain; output = per-leaf mean colour, sorted and deduplicated.
#pragma once

#include <stdint.h>

#include <array>
#include <set>
#include <tuple>
#include <vector>

namespace kmg {

inline std::vector<std::array<uint8_t, 4>> octree_palette(const uint8_t *rgba, uint64_t n_pixels, uint32_t color_count)
{
    struct Node {
        uint32_t level;
        int32_t parent;          // -1 = none
        uint32_t color_index;
        int32_t children[8];
        uint32_t child_count;
        uint64_t count, r, g, b;
    };
    std::vector<std::array<uint8_t, 4>> out;
    if (color_count == 0) return out;                                       // octree.rs:67-69
    std::vector<Node> nodes;
    auto make = [](uint32_t level, int32_t parent, uint32_t ci) {
        Node nd;
        nd.level = level; nd.parent = parent; nd.color_index = ci; nd.child_count = 0;
        nd.count = nd.r = nd.g = nd.b = 0;
        for (int i = 0; i < 8; ++i) nd.children[i] = -1;
        return nd;
    };
    nodes.push_back(make(0, -1, 0));                                        // root, octree.rs:33-39
    for (uint64_t i = 0; i < n_pixels; ++i) {                               // add_color, octree.rs:41-64
        const uint8_t *px = rgba + 4 * i;
        int32_t cur = 0;
        for (uint32_t level = 0; level < 8; ++level) {
            const uint8_t mask = (uint8_t)(0x80u >> level);                 // get_color_index, :12-26
            const uint32_t ci = ((px[0] & mask) ? 4u : 0u) | ((px[1] & mask) ? 2u : 0u) | ((px[2] & mask) ? 1u : 0u);
            if (nodes[cur].children[ci] < 0) {
                const int32_t id = (int32_t)nodes.size();
                nodes[cur].children[ci] = id;
                nodes[cur].child_count += 1;
                nodes.push_back(make(level, cur, ci));                      // Node::with_parent keeps the loop level
            }
            cur = nodes[cur].children[ci];
        }
        nodes[cur].r += px[0]; nodes[cur].g += px[1]; nodes[cur].b += px[2]; nodes[cur].count += 1;
    }
    // Node::partial_cmp, octree.rs:214-233
    using Key = std::tuple<uint32_t, uint64_t, int32_t>;
    auto key = [&](int32_t id) { return Key(nodes[id].child_count, nodes[id].count >> nodes[id].level, id); };
    std::set<Key> leaves;
    for (int32_t id = 0; id < (int32_t)nodes.size(); ++id)
        if (nodes[id].count > 0) leaves.insert(key(id));                    // :71-78
    while (leaves.size() > color_count) {                                    // :80-103
        const int32_t id = std::get<2>(*leaves.begin());                     // pop_back of the descending deque
        leaves.erase(leaves.begin());
        const int32_t pid = nodes[id].parent;
        if (pid >= 0) {
            leaves.erase(key(pid));                                          // remove the parent if present (old key)
            nodes[pid].r += nodes[id].r; nodes[pid].g += nodes[id].g; nodes[pid].b += nodes[id].b;
            nodes[pid].count += nodes[id].count;
            nodes[pid].child_count -= 1;
            nodes[pid].children[nodes[id].color_index] = -1;
            nodes[id].parent = -1;
            leaves.insert(key(pid));
        }
    }
    for (auto it = leaves.rbegin(); it != leaves.rend(); ++it) {             // :106-109 output_color
        const Node &nd = nodes[std::get<2>(*it)];
        out.push_back({(uint8_t)(nd.r / nd.count), (uint8_t)(nd.g / nd.count), (uint8_t)(nd.b / nd.count), 255});
    }
    std::sort(out.begin(), out.end());                                       // :110-111 sort + dedup
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

}  // namespace kmg
